"""VidHRFormer blocks of VPTR on the MI355X HIP kernels.

Module tree, attribute names, constructor signatures and state_dict keys mirror the reference
(model/VidHRFormer_modules.py, model/VidHRFormer.py, model/MultiHeadAttentionRPE.py) so checkpoints and
training scripts drop in; leaf modules (nn.Linear, nn.LayerNorm, nn.Conv2d, nn.BatchNorm2d,
nn.MultiheadAttention) are used as *parameter containers* only -- every forward below runs through
vptr_amd.ops (C-ABI HIP kernels) on token-major [rows = (n,t,h,w), C] activations.  The reference's window
partition, (T, N*HW, C) permutes and NCHW<->NHWC permutes are index arithmetic inside the kernels.
"""
import copy
from collections import namedtuple

import torch
import torch.nn as nn

from .. import ops
from .position_encoding import relative_position_index

Geom = namedtuple("Geom", "N T H W")


def _get_clones(module, n):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(n)])


class _DropPathPlan:
    """All stochastic-depth scale vectors of one model forward in ONE launch (vptr_droppath_scales) instead of four ATen ops per call
    site (~40 call sites in the K64 NAR model): the sequence of (p, count) requests of a forward is recorded, and the next forward
    with the same sequence evaluates floor(keep + U) / keep for every request at once.  U comes from the counter-based hash of the
    step's dropout seed (ops.seed_tensor), not from torch's generator: eager steps and hipGraph replays draw identical vectors, and
    request i always hashes site DROPPATH_SITE0 + i whether it was planned or not."""

    def __init__(self):
        self.plan, self.rec, self.idx, self.scales = None, [], 0, None
        self._keep = None  # (plan, device) -> device tensor of keep probabilities: built once, not per forward (a host->device copy)

    def begin(self, device):
        self.rec, self.idx, self.scales = [], 0, None
        if self.plan:
            if self._keep is None or self._keep[0] != self.plan or self._keep[1] != device:
                self._keep = (list(self.plan), device, torch.tensor([1.0 - p for p, _ in self.plan], device=device, dtype=torch.float32))
            self.scales = ops.droppath_scales(self._keep[2], max(c for _, c in self.plan), device)

    def get(self, p, count, device):
        self.rec.append((p, count))
        i = self.idx
        self.idx += 1
        if self.scales is not None and i < len(self.plan) and self.plan[i] == (p, count):
            return self.scales[i, :count]
        self.scales = None  # the sequence changed: single requests for the rest of this forward, re-plan at its end
        return ops.droppath_scales(torch.full((1,), 1.0 - p, device=device, dtype=torch.float32), count, device, site_offset=i)[0]

    def end(self):
        self.plan = list(self.rec)


_dp_plan = [None]  # the plan of the model forward in flight (set by VidHRFormerNAR / VidHRFormerFAR)
_dp_loose = [0]    # running request index of blocks called outside a model forward (stand-alone use, tests)


def _droppath_scale(p, training, count, device):
    """Per-index stochastic-depth scale floor(keep + U)/keep (VidHRFormer_modules.py:563-575); None when inactive."""
    if p == 0.0 or not training:
        return None
    if _dp_plan[0] is not None:
        return _dp_plan[0].get(p, count, device)
    _dp_loose[0] = (_dp_loose[0] + 1) % 0x4000
    return ops.droppath_scales(torch.full((1,), 1.0 - p, device=device, dtype=torch.float32), count, device,
                               site_offset=0x4000 + _dp_loose[0])[0]


class DropPath(nn.Module):
    """Parameter-free marker kept for module-tree parity (the scale is fused into the producing kernel's epilogue)."""

    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob

    def extra_repr(self):
        return "drop_prob={}".format(self.drop_prob)


class MultiheadAttentionRPE(nn.Module):
    """Container of the window-attention parameters (MultiHeadAttentionRPE.py:22-53, 359-388): separate q/k/v/out
    Linears plus the Swin-style relative-position bias table and its index buffer."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, rpe=True, window_size=7):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.dropout = embed_dim, num_heads, dropout
        self.head_dim = embed_dim // num_heads
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.rpe = rpe
        if rpe:
            self.window_size = [window_size] * 2
            self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * window_size - 1) ** 2, num_heads))
            self.register_buffer("relative_position_index", relative_position_index(window_size))
            nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class SpatialLocalMultiheadAttention(nn.Module):
    """Local-window multi-head self-attention (VidHRFormer_modules.py:287-357)."""

    def __init__(self, embed_dim, num_heads, window_size=7, dropout=0.0, rpe=False):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.dropout, self.rpe = embed_dim, num_heads, window_size, dropout, rpe
        if rpe:
            self.attn = MultiheadAttentionRPE(embed_dim, num_heads, dropout=dropout, rpe=True, window_size=window_size)
        else:
            self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self._lw_tab = {}

    def _window_pos_table(self, lw_pos, H, W):
        key = (H, W, lw_pos.device)
        if key not in self._lw_tab:
            ws = self.window_size
            hh = torch.arange(H, device=lw_pos.device) % ws
            ww = torch.arange(W, device=lw_pos.device) % ws
            self._lw_tab[key] = lw_pos[hh[:, None], ww[None, :]].reshape(H * W, -1).contiguous()
        return self._lw_tab[key]

    def forward_tokens(self, xqk, xv, residual, g, lw_pos, site, rowscale=None, rs_div=1, rs_mod=1, x_p16=False):
        """x_p16: xqk and xv are P16 tensors (written by the LayerNorm in front; rpe=True only).  Padding / cropping move whole
        token rows, which is format-agnostic."""
        C, nh, ws = self.dim, self.num_heads, self.window_size
        P = ops.p16_ok(C)
        p = self.dropout if self.training else 0.0
        a = self.attn
        frames, H, W = g.N * g.T, g.H, g.W
        padded = bool(H % ws or W % ws)
        if padded:  # PadBlock: zero centre padding BEFORE the positional add and the projections (VidHRFormer_modules.py:332-346)
            same = xv is xqk
            xqk, H, W = ops.pad_tokens(xqk, frames, g.H, g.W, ws)
            xv = xqk if same else ops.pad_tokens(xv, frames, g.H, g.W, ws)[0]
        if self.rpe:
            Wq, Wk, Wv = a.q_proj.weight, a.k_proj.weight, a.v_proj.weight
            bq, bk, bv = a.q_proj.bias, a.k_proj.bias, a.v_proj.bias
            xin = xqk
            table, index = a.relative_position_bias_table, a.relative_position_index
        else:
            xin = ops.add_rowtab(xqk, self._window_pos_table(lw_pos, H, W), 1, H * W)
            Wq, Wk, Wv = a.in_proj_weight[:C], a.in_proj_weight[C:2 * C], a.in_proj_weight[2 * C:]
            bq, bk, bv = a.in_proj_bias[:C], a.in_proj_bias[C:2 * C], a.in_proj_bias[2 * C:]
            table, index = None, None
        # q/k/v projections (one batched launch) + attention core; q is scaled by head_dim^-0.5 in the GEMM epilogue
        if x_p16 and not self.rpe:
            raise RuntimeError("the positional add of the rpe=False window attention needs fp32 inputs")
        o = ops.proj_window_attention(xin, xv, Wq, bq, Wk, bk, Wv, bv, table, index, frames, H, W, nh, ws, p, site, x_p16=x_p16, o_p16=P)
        if padded:  # the out-projection is per token, so cropping first is equivalent to depad_if_needed after it (:347-351)
            o = ops.crop_tokens(o, frames, H, W, g.H, g.W)
        return ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=residual, rowscale=rowscale, rs_div=rs_div,
                          rs_mod=rs_mod, x_p16=P)

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"


class TemporalSpatialLocalMultiheadAttention(nn.Module):
    """Encoder-decoder attention over (time x window) tokens (VidHRFormer_modules.py:219-284): for every ws x ws window the
    T2*ws*ws query tokens attend to the T1*ws*ws memory tokens of the same window with a stock packed-weight
    nn.MultiheadAttention; TS_local_pos_embed[:T1] is added to the keys, [T1:T1+T2] to the queries, after PadBlock's zero
    centre padding.  On the HIP path pad / permute / reverse permute are index arithmetic of vptr_tsattn_fwd/bwd."""

    def __init__(self, embed_dim, num_heads, window_size=7, dropout=0.0):
        super().__init__()
        self.dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.attn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self._tabs = {}

    def _pos_table(self, Tlw, t0, T, H, W):
        key = (t0, T, H, W, Tlw.device)
        if key not in self._tabs:
            ws = self.window_size
            hh = torch.arange(H, device=Tlw.device) % ws
            ww = torch.arange(W, device=Tlw.device) % ws
            self._tabs[key] = Tlw[t0:t0 + T][:, hh[:, None], ww[None, :]].reshape(T * H * W, -1).contiguous()
        return self._tabs[key]

    def forward_tokens(self, mem, query, residual, g, T1, Tlw, site, rowscale=None, rs_div=1, rs_mod=1):
        """mem [N*T1*HW, C], query [N*T2*HW, C] (= LN(tgt) + query_pos), residual [N*T2*HW, C] -> residual + rowscale * attn"""
        C, nh, ws = self.dim, self.num_heads, self.window_size
        p = self.dropout if self.training else 0.0
        T2, H, W = g.T, g.H, g.W
        padded = bool(H % ws or W % ws)
        if padded:
            mem, H, W = ops.pad_tokens(mem, g.N * T1, g.H, g.W, ws)
            query = ops.pad_tokens(query, g.N * T2, g.H, g.W, ws)[0]
        a = self.attn
        Wq, Wk, Wv = a.in_proj_weight[:C], a.in_proj_weight[C:2 * C], a.in_proj_weight[2 * C:]
        bq, bk, bv = a.in_proj_bias[:C], a.in_proj_bias[C:2 * C], a.in_proj_bias[2 * C:]
        q_in = ops.add_rowtab(query, self._pos_table(Tlw, T1, T2, H, W), 1, T2 * H * W)
        k_in = ops.add_rowtab(mem, self._pos_table(Tlw, 0, T1, H, W), 1, T1 * H * W)
        q = ops.linear(q_in, Wq, bq, alpha=float(C // nh) ** -0.5)
        k = ops.linear(k_in, Wk, bk)
        v = ops.linear(mem, Wv, bv)
        o = ops.temporal_spatial_window_attention(q, k, v, g.N, T2, T1, H, W, ws, nh, p, site)
        if padded:
            o = ops.crop_tokens(o, g.N * T2, H, W, g.H, g.W)
        return ops.linear(o, a.out_proj.weight, a.out_proj.bias, residual=residual, rowscale=rowscale, rs_div=rs_div,
                          rs_mod=rs_mod)


class MlpDWBN(nn.Module):
    """Conv feed-forward 1x1 -> DW3x3 -> 1x1, each followed by norm + GELU (VidHRFormer_modules.py:376-442).
    AR_model=True (the constructor default, used by FAR and by every NAR *decoder* block) normalises with
    LayerNorm((ch,H,W)); AR_model=False (NAR encoder blocks) with BatchNorm2d."""

    def __init__(self, encH, encW, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 dw_act_layer=nn.GELU, drop=0.0, AR_model=True):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Conv2d(in_features, hidden_features, kernel_size=1)
        self.act1 = act_layer()
        self.norm1 = nn.LayerNorm((hidden_features, encH, encW)) if AR_model else nn.BatchNorm2d(hidden_features)
        self.dw3x3 = nn.Conv2d(hidden_features, hidden_features, kernel_size=3, stride=1, groups=hidden_features, padding=1)
        self.act2 = dw_act_layer()
        self.norm2 = nn.LayerNorm((hidden_features, encH, encW)) if AR_model else nn.BatchNorm2d(hidden_features)
        self.fc2 = nn.Conv2d(hidden_features, out_features, kernel_size=1)
        self.act3 = act_layer()
        self.norm3 = nn.LayerNorm((out_features, encH, encW)) if AR_model else nn.BatchNorm2d(out_features)
        self.drop = nn.Dropout(drop)
        self.out_features = out_features
        self.layer_norm = AR_model
        self.drop_p = drop

    def _norm_act(self, h, norm, g, **kw):
        HW = g.H * g.W
        if self.layer_norm:
            if tuple(norm.normalized_shape[1:]) != (g.H, g.W):
                raise RuntimeError("LayerNorm((C,H,W)) was built for %s but the feature map is %dx%d"
                                   % (tuple(norm.normalized_shape), g.H, g.W))
            F = norm.weight.shape[0]
            w = norm.weight.reshape(F, HW).t().contiguous()  # channel-last affine [HW, F]
            b = norm.bias.reshape(F, HW).t().contiguous()
            return ops.norm_act(h, w, b, "ln", HW, self.training, eps=norm.eps, **kw)
        nbt = norm.num_batches_tracked if (self.training and norm.track_running_stats) else None   # incremented by the statistics launch
        return ops.norm_act(h, norm.weight, norm.bias, "bn", HW, self.training, norm.running_mean, norm.running_var,
                            eps=norm.eps, momentum=norm.momentum, num_batches_tracked=nbt, **kw)

    def p16_in_ok(self):
        """can forward_tokens take its input as a P16 tensor? (its own widths decide, not the block's: hidden_features is a
        constructor argument of its own)"""
        return ops.p16_ok(self.fc1.weight.shape[1], self.fc1.weight.shape[0], self.out_features)

    def forward_tokens(self, u, residual, g, site, rowscale=None, rs_div=1, rs_mod=1, x_p16=False):
        """x_p16: u is a P16 tensor.  With P16-eligible widths the tensors that only connect a normalisation to a 1x1 convolution
        never exist as fp32: norm2 writes fc2's input as P16, and the backward passes of norm3 / norm1 hand fc2 / fc1 their
        incoming gradient as P16 (each of those tensors has exactly one consumer)."""
        p = self.drop_p if self.training else 0.0
        F, C = self.fc1.weight.shape[0], self.fc1.weight.shape[1]
        P = ops.p16_ok(C, F, self.out_features)
        # LayerNorm((ch,H,W)) statistics: where the producing kernel can deliver them (P16 GEMM epilogue halves / depthwise waves inside one
        # frame), each normalisation reads per-frame sums its producer accumulated instead of running a statistics pass of its own
        HW, frames, rows = g.H * g.W, g.N * g.T, u.shape[0]
        S = self.layer_norm and P and ops.frame_stats_ok(rows, HW, F) and ops.frame_stats_ok(rows, HW, self.out_features)
        Sd = S and ops.frame_stats_ok(rows, HW, F, g.W)
        buf = ops.frame_stats_buffer(3 * frames, u.device).view(3, frames, ops.FRAME_STATS_STRIDE) if S else None   # one fill for the three normalisations
        st = [buf[i] if k else None for i, k in enumerate((S, Sd, S))]
        h = ops.linear(u, self.fc1.weight.view(F, C), self.fc1.bias, x_p16=x_p16, dy_p16=P, frame_stats=st[0], frame_rows=HW)
        if st[0] is not None and ops.norm_dwconv_ok(rows, HW, F, g.H, g.W) and tuple(self.norm1.normalized_shape[1:]) == (g.H, g.W):
            # norm1 + act1 in the depthwise kernel's load path: the activated hidden tensor never exists in fp32 (ops.norm_dwconv3x3)
            w1 = self.norm1.weight.reshape(F, HW).t().contiguous()
            b1 = self.norm1.bias.reshape(F, HW).t().contiguous()
            h = ops.norm_dwconv3x3(h, w1, b1, self.dw3x3.weight, self.dw3x3.bias, frames, g.H, g.W, st[0], frame_stats=st[1],
                                   eps=self.norm1.eps, dx_p16=P)
        else:
            h = self._norm_act(h, self.norm1, g, dx_p16=P, raw_stats=st[0])
            h = ops.dwconv3x3(h, self.dw3x3.weight, self.dw3x3.bias, frames, g.H, g.W, frame_stats=st[1])
        h = self._norm_act(h, self.norm2, g, dropout_p=p, site=site, out_p16=P, raw_stats=st[1])
        h = ops.linear(h, self.fc2.weight.view(self.out_features, F), self.fc2.bias, x_p16=P, dy_p16=P, frame_stats=st[2], frame_rows=HW)
        return self._norm_act(h, self.norm3, g, dropout_p=p, site=site + 1, rowscale=rowscale, rs_div=rs_div, rs_mod=rs_mod,
                              residual=residual, dx_p16=P, raw_stats=st[2])


def _mha_tokens(mha, q_in, k_in, v_in, residual, Nb, Tq, Tk, HW, causal, p_attn, site, out_dropout=0.0, out_site=0,
                rowscale=None, rs_div=1, rs_mod=1, merge_v_grad=False, x_p16=False, kv_acc=None):
    """Stock nn.MultiheadAttention (packed in_proj) over time on token-major inputs (VidHRFormer_modules.py:79-84).
    merge_v_grad: q_in is k_in = v_in + a table that needs no gradient, so the three input gradients may be returned as one."""
    C, nh = mha.embed_dim, mha.num_heads
    w, b = mha.in_proj_weight, mha.in_proj_bias
    P = ops.p16_ok(C)
    o = ops.proj_temporal_attention(q_in, k_in, v_in, w[:C], b[:C], w[C:2 * C], b[C:2 * C], w[2 * C:], b[2 * C:], Nb, Tq, Tk, HW, nh,
                                    causal, p_attn, site, merge_v_grad=merge_v_grad, x_p16=x_p16, o_p16=P, kv_acc=kv_acc)
    return ops.linear(o, mha.out_proj.weight, mha.out_proj.bias, residual=residual, dropout_p=out_dropout, site=out_site,
                      rowscale=rowscale, rs_div=rs_div, rs_mod=rs_mod, x_p16=P)


class VidHRFormerBlockEnc(nn.Module):
    """Encoder block (VidHRFormer_modules.py:30-93): window MHSA -> conv-FFN -> temporal MHSA -> MLP, pre-norm."""

    def __init__(self, encH, encW, embed_dim, num_heads, window_size=7, dropout=0.0, drop_path=0.0,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024, far=False, rpe=True):
        super().__init__()
        self.embed_dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.Spatial_FFN_hidden_ratio = Spatial_FFN_hidden_ratio
        self.SLMHSA = SpatialLocalMultiheadAttention(embed_dim, num_heads, window_size, dropout, rpe)
        self.SpatialFFN = MlpDWBN(encH, encW, embed_dim, hidden_features=int(Spatial_FFN_hidden_ratio * embed_dim),
                                  out_features=embed_dim, drop=dropout, AR_model=bool(far))
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm3 = nn.LayerNorm(embed_dim)
        self.temporal_MHSA = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.linear1 = nn.Linear(embed_dim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, embed_dim)
        self.activation = nn.GELU()
        self.drop1 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.drop2 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.drop3 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.norm4 = nn.LayerNorm(embed_dim)
        self.far = far
        self.drop_path_p = drop_path
        self._site = 0

    def forward_tokens(self, x, g, lw_pos, tpos):
        """x [N*T*H*W, C]; tpos (T, C)."""
        HW = g.H * g.W
        p = self.dropout if self.training else 0.0
        s = self._site
        dp = _droppath_scale(self.drop_path_p, self.training, g.N, x.device)
        per_n = g.T * HW
        # every pre-norm LayerNorm also returns its input as a pass-through output xr, used as the sub-layer's residual: the
        # residual gradient then comes back through the LayerNorm node and is added inside its backward kernel
        # P: the LayerNorm outputs (which only feed GEMMs) are written in the P16 operand format, see ops.p16_ok / include/vptr_hip.h
        P = ops.p16_ok(self.embed_dim, self.linear1.weight.shape[0])
        Pw = P and self.SLMHSA.rpe
        u, xr = ops.layernorm(x, self.norm1.weight, self.norm1.bias, eps=self.norm1.eps, passthrough=True, out_p16=Pw)
        x = self.SLMHSA.forward_tokens(u, u, xr, g, lw_pos, s + 0, rowscale=dp, rs_div=per_n, rs_mod=g.N, x_p16=Pw)
        dp = _droppath_scale(self.drop_path_p, self.training, g.N, x.device)
        u, xr = ops.layernorm(x, self.norm2.weight, self.norm2.bias, eps=self.norm2.eps, passthrough=True,
                              out_p16=P and self.SpatialFFN.p16_in_ok())
        x = self.SpatialFFN.forward_tokens(u, xr, g, s + 1, rowscale=dp, rs_div=per_n, rs_mod=g.N, x_p16=P and self.SpatialFFN.p16_in_ok())
        u, uq, xr = ops.layernorm(x, self.norm3.weight, self.norm3.bias, tab=tpos, tab_div=HW, tab_mod=g.T, eps=self.norm3.eps,
                                  passthrough=True, out_p16=P)
        x = _mha_tokens(self.temporal_MHSA, uq, uq, u, xr, g.N, g.T, g.T, HW, self.far, p, s + 3, out_dropout=p, out_site=s + 4,
                        merge_v_grad=not tpos.requires_grad, x_p16=P)
        u, xr = ops.layernorm(x, self.norm4.weight, self.norm4.bias, eps=self.norm4.eps, passthrough=True, out_p16=P)
        if P and ops.config.fused_mlp:   # one autograd node; linear2's input gradient applies GELU' and the dropout mask in its epilogue
            return ops.mlp(u, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, residual=xr, dropout_p=p,
                           site1=s + 5, site2=s + 6, x_p16=True)
        h = ops.linear(u, self.linear1.weight, self.linear1.bias, act=ops.ACT_GELU, dropout_p=p, site=s + 5, x_p16=P, out_p16=P)
        return ops.linear(h, self.linear2.weight, self.linear2.bias, residual=xr, dropout_p=p, site=s + 6, x_p16=P)


class VidHRFormerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers, norm=None):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm

    def forward_tokens(self, x, g, lw_pos, tpos):
        for layer in self.layers:
            x = layer.forward_tokens(x, g, lw_pos, tpos)
        if self.norm is not None:
            x = ops.layernorm(x, self.norm.weight, self.norm.bias, eps=self.norm.eps)
        return x


class VidHRFormerBlockDecNAR(nn.Module):
    """NAR decoder block (VidHRFormer_modules.py:125-211): query window MHSA, conv-FFN, temporal self-attention, MLP,
    encoder-decoder temporal attention, second conv-FFN."""

    def __init__(self, encH, encW, embed_dim, num_heads, window_size=7, dropout=0.0, drop_path=0.0,
                 Spatial_FFN_hidden_ratio=4, dim_feedforward=1024, TSLMA_flag=False, rpe=True):
        super().__init__()
        self.embed_dim, self.num_heads, self.window_size, self.dropout = embed_dim, num_heads, window_size, dropout
        self.Spatial_FFN_hidden_ratio = Spatial_FFN_hidden_ratio
        hidden = int(Spatial_FFN_hidden_ratio * embed_dim)
        self.SLMHSA = SpatialLocalMultiheadAttention(embed_dim, num_heads, window_size, dropout, rpe)
        self.SpatialFFN = MlpDWBN(encH, encW, embed_dim, hidden_features=hidden, out_features=embed_dim, drop=dropout)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm3 = nn.LayerNorm(embed_dim)
        self.temporal_MHSA = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.drop1 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.linear1 = nn.Linear(embed_dim, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, embed_dim)
        self.activation = nn.GELU()
        self.drop2 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.drop3 = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.norm4 = nn.LayerNorm(embed_dim)
        self.TSLMA_flag = TSLMA_flag
        if TSLMA_flag:
            self.TSLMA = TemporalSpatialLocalMultiheadAttention(embed_dim, num_heads, window_size, dropout)
        else:
            self.EncDecAttn = nn.MultiheadAttention(embed_dim, num_heads, dropout=dropout)
        self.SpatialFFN1 = MlpDWBN(encH, encW, embed_dim, hidden_features=hidden, out_features=embed_dim, drop=dropout)
        self.norm5 = nn.LayerNorm(embed_dim)
        self.norm6 = nn.LayerNorm(embed_dim)
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.drop_path_p = drop_path
        self._site = 0

    def forward_tokens(self, tgt, g, qpos_tab, qpos_tpos_tab, mem, mem_k, T1, lw_pos, tpos_f, Tlw_pos=None, kv_acc=None):
        """tgt [N*T2*HW, C]; qpos_tab = frame_queries as [T2*HW, C]; qpos_tpos_tab = frame_queries + tpos_f per (t, pixel);
        mem, mem_k = memory and memory + past temporal pos, [N*T1*HW, C]; tpos_f (T2, C); Tlw_pos (T1+T2, ws, ws, C)."""
        HW = g.H * g.W
        T2 = g.T
        p = self.dropout if self.training else 0.0
        s = self._site
        per_n = T2 * HW
        dp = _droppath_scale(self.drop_path_p, self.training, g.N, tgt.device)
        # (pass-through outputs xr: see VidHRFormerBlockEnc.forward_tokens)
        P = ops.p16_ok(self.embed_dim, self.linear1.weight.shape[0])   # as in VidHRFormerBlockEnc.forward_tokens
        Pw = P and self.SLMHSA.rpe
        t, tq, xr = ops.layernorm(tgt, self.norm1.weight, self.norm1.bias, tab=qpos_tab, tab_div=1, tab_mod=per_n, eps=self.norm1.eps,
                                  passthrough=True, out_p16=Pw)
        x = self.SLMHSA.forward_tokens(tq, t, xr, g, lw_pos, s + 0, rowscale=dp, rs_div=per_n, rs_mod=g.N, x_p16=Pw)
        dp = _droppath_scale(self.drop_path_p, self.training, g.N, tgt.device)
        u, xr = ops.layernorm(x, self.norm2.weight, self.norm2.bias, eps=self.norm2.eps, passthrough=True,
                              out_p16=P and self.SpatialFFN.p16_in_ok())
        x = self.SpatialFFN.forward_tokens(u, xr, g, s + 1, rowscale=dp, rs_div=per_n, rs_mod=g.N, x_p16=P and self.SpatialFFN.p16_in_ok())
        u, uq, xr = ops.layernorm(x, self.norm3.weight, self.norm3.bias, tab=tpos_f, tab_div=HW, tab_mod=T2, eps=self.norm3.eps,
                                  passthrough=True, out_p16=P)
        x = _mha_tokens(self.temporal_MHSA, uq, uq, u, xr, g.N, T2, T2, HW, False, p, s + 3, out_dropout=p, out_site=s + 4,
                        merge_v_grad=not tpos_f.requires_grad, x_p16=P)
        u, xr = ops.layernorm(x, self.norm4.weight, self.norm4.bias, eps=self.norm4.eps, passthrough=True, out_p16=P)
        if P and ops.config.fused_mlp:   # as in VidHRFormerBlockEnc.forward_tokens
            x = ops.mlp(u, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias, residual=xr, dropout_p=p,
                        site1=s + 5, site2=s + 6, x_p16=True)
        else:
            h = ops.linear(u, self.linear1.weight, self.linear1.bias, act=ops.ACT_GELU, dropout_p=p, site=s + 5, x_p16=P, out_p16=P)
            x = ops.linear(h, self.linear2.weight, self.linear2.bias, residual=xr, dropout_p=p, site=s + 6, x_p16=P)
        # encoder-decoder attention; the reference applies drop_path1 to a (T2, N*HW, C) tensor, i.e. along TIME
        # (VidHRFormer_modules.py:204) -- reproduced: scale indexed by t = (row // HW) % T2
        if self.TSLMA_flag:
            # temporal-spatial window cross-attention (VidHRFormer_modules.py:195-199); here drop_path1 sees (N,T2,H,W,C): per sample
            dp = _droppath_scale(self.drop_path_p, self.training, g.N, tgt.device)
            _, uq, xr = ops.layernorm(x, self.norm5.weight, self.norm5.bias, tab=qpos_tab, tab_div=1, tab_mod=per_n, eps=self.norm5.eps,
                                      passthrough=True)
            x = self.TSLMA.forward_tokens(mem, uq, xr, g, T1, Tlw_pos, s + 7, rowscale=dp, rs_div=per_n, rs_mod=g.N)
        else:
            dpt = _droppath_scale(self.drop_path_p, self.training, T2, tgt.device)
            # mem / mem_k arrive as P16 tensors when P (converted once per forward by VidHRformerDecoderNAR.forward_tokens)
            # qpos_tpos_tab = frame_queries + temporal positions (a buffer): its gradient is frame_queries' gradient
            _, uq, xr = ops.layernorm(x, self.norm5.weight, self.norm5.bias, tab=qpos_tpos_tab, tab_div=1, tab_mod=per_n,
                                      eps=self.norm5.eps, passthrough=True, out_p16=P,
                                      tab_grad_to=None if tpos_f.requires_grad else qpos_tab)
            x = _mha_tokens(self.EncDecAttn, uq, mem_k, mem, xr, g.N, T2, T1, HW, False, p, s + 7, rowscale=dpt, rs_div=HW, rs_mod=T2,
                            x_p16=P, kv_acc=kv_acc)
        dp = _droppath_scale(self.drop_path_p, self.training, g.N, tgt.device)
        P1 = P and self.SpatialFFN1.p16_in_ok()
        u, xr = ops.layernorm(x, self.norm6.weight, self.norm6.bias, eps=self.norm6.eps, passthrough=True, out_p16=P1)
        return self.SpatialFFN1.forward_tokens(u, xr, g, s + 8, rowscale=dp, rs_div=per_n, rs_mod=g.N, x_p16=P1)


class VidHRformerDecoderNAR(nn.Module):
    def __init__(self, decoder_layer, num_layers, norm=None, return_intermediate=False):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.num_layers = num_layers
        self.norm = norm
        self.return_intermediate = return_intermediate

    def forward_tokens(self, tgt, g, frame_queries, mem, T1, lw_pos, tpos_f, tpos_p, Tlw_pos=None):
        HW, C = g.H * g.W, tgt.shape[1]
        qpos_tab = frame_queries.reshape(g.T * HW, C)
        qpos_tpos_tab = (frame_queries.reshape(g.T, HW, C) + tpos_f[:, None, :]).reshape(g.T * HW, C)
        mem_k = ops.add_rowtab(mem, tpos_p, HW, T1)
        layer0 = self.layers[0]
        if not layer0.TSLMA_flag and ops.p16_ok(C, layer0.linear1.weight.shape[0]):
            # the encoder-decoder attentions of all layers project the same memory: its two P16 images are made once
            mem, mem_k = ops.as_p16(mem), ops.as_p16(mem_k)
        x = tgt
        # every layer's encoder-decoder attention reads the same (mem_k, mem): their input gradients are summed inside the GEMMs
        kv_acc = ops.KVGradAccum((mem_k, mem)) if (torch.is_grad_enabled() and mem.requires_grad and len(self.layers) > 1) else None
        for layer in self.layers:
            x = layer.forward_tokens(x, g, qpos_tab, qpos_tpos_tab, mem, mem_k, T1, lw_pos, tpos_f, Tlw_pos, kv_acc=kv_acc)
        if self.norm is not None:
            x = ops.layernorm(x, self.norm.weight, self.norm.bias, eps=self.norm.eps)
        return x


def _ln_ffn_stats_floats(stack, frames):
    """floats of frame-statistics accumulators one forward of `stack` requests (3 x [frames, FRAME_STATS_STRIDE] per LayerNorm conv-FFN)"""
    n = getattr(stack, "_n_ln_ffn", None)
    if n is None:
        n = sum(1 for m in stack.modules() if isinstance(m, MlpDWBN) and m.layer_norm)
        stack.__dict__["_n_ln_ffn"] = n
    return n * 3 * frames * ops.FRAME_STATS_STRIDE


def _assign_sites(module, base=0):
    for m in module.modules():
        if isinstance(m, (VidHRFormerBlockEnc, VidHRFormerBlockDecNAR)):
            m._site = base
            base += 16
    return base


class VidHRFormerNAR(nn.Module):
    """Encoder(L_e) -> memory; decoder(L_d) from a zero target and learned frame queries (VidHRFormer.py:9-53)."""

    def __init__(self, in_feat_shape, num_encoder_layer, num_decoder_layer, num_past_frames, num_future_frames, embed_dim,
                 num_heads, window_size=7, dropout=0.0, drop_path=0.0, Spatial_FFN_hidden_ratio=4, dim_feedforward=512,
                 TSLMA_flag=False, rpe=True):
        super().__init__()
        self.in_C, self.H, self.W = in_feat_shape
        self.embed_dim = embed_dim
        self.num_encoder_layer, self.num_decoder_layer, self.num_heads = num_encoder_layer, num_decoder_layer, num_heads
        self.encoder = VidHRFormerEncoder(
            VidHRFormerBlockEnc(self.H, self.W, embed_dim, num_heads, window_size, dropout, drop_path, Spatial_FFN_hidden_ratio,
                                dim_feedforward, rpe=rpe), num_encoder_layer, nn.LayerNorm(embed_dim))
        self.decoder = VidHRformerDecoderNAR(
            VidHRFormerBlockDecNAR(self.H, self.W, embed_dim, num_heads, window_size, dropout, drop_path,
                                   Spatial_FFN_hidden_ratio, dim_feedforward, TSLMA_flag, rpe=rpe),
            num_decoder_layer, nn.LayerNorm(embed_dim), return_intermediate=False)
        _assign_sites(self)

    def forward(self, src, local_window_pos_embed, temporal_pos_embed, TS_local_pos_embed, query_pos, init_tgt=None):
        """src (N,Tp,C,H,W) -> (out (N,Tf,C,H,W) post-ReLU, memory (N,Tp,H,W,C))."""
        N, Tp, C, H, W = src.shape
        Tf = query_pos.shape[0]
        plan = self.__dict__.setdefault("_dp", {}).setdefault((self.training, N, Tp, Tf), _DropPathPlan())
        _dp_plan[0] = plan
        plan.begin(src.device)
        arena = ops.zero_arena(_ln_ffn_stats_floats(self.encoder, N * Tp) + _ln_ffn_stats_floats(self.decoder, N * Tf), src.device)
        arena.__enter__()
        try:
            x = ops.nchw_to_tokens(src.reshape(N * Tp, C, H, W))
            mem = self.encoder.forward_tokens(x, Geom(N, Tp, H, W), local_window_pos_embed, temporal_pos_embed[:Tp])
            tgt = torch.zeros((N * Tf * H * W, C), device=src.device, dtype=torch.float32)
            out = self.decoder.forward_tokens(tgt, Geom(N, Tf, H, W), query_pos, mem, Tp, local_window_pos_embed,
                                              temporal_pos_embed[Tp:], temporal_pos_embed[:Tp], TS_local_pos_embed)
            out = ops.tokens_to_nchw(out, N * Tf, C, H, W, relu=True).reshape(N, Tf, C, H, W)
        finally:
            arena.__exit__(None, None, None)
            plan.end()
            _dp_plan[0] = None
        return out, mem.reshape(N, Tp, H, W, C)


class VidHRFormerFAR(nn.Module):
    """Encoder-only stack with causal temporal attention and LayerNorm conv-FFNs (VidHRFormer.py:56-88)."""

    def __init__(self, in_feat_shape, num_encoder_layer, num_past_frames, num_future_frames, embed_dim, num_heads,
                 window_size=7, dropout=0.0, drop_path=0.0, Spatial_FFN_hidden_ratio=4, dim_feedforward=512, rpe=True):
        super().__init__()
        self.in_C, self.H, self.W = in_feat_shape
        self.embed_dim = embed_dim
        self.num_encoder_layer, self.num_heads = num_encoder_layer, num_heads
        self.encoder = VidHRFormerEncoder(
            VidHRFormerBlockEnc(self.H, self.W, embed_dim, num_heads, window_size, dropout, drop_path, Spatial_FFN_hidden_ratio,
                                dim_feedforward, far=True, rpe=rpe), num_encoder_layer, nn.LayerNorm(embed_dim))
        _assign_sites(self)

    def forward(self, input_feat, local_window_pos_embed, temporal_pos_embed):
        N, T, C, H, W = input_feat.shape
        plan = self.__dict__.setdefault("_dp", {}).setdefault((self.training, N, T), _DropPathPlan())
        _dp_plan[0] = plan
        plan.begin(input_feat.device)
        arena = ops.zero_arena(_ln_ffn_stats_floats(self.encoder, N * T), input_feat.device)
        arena.__enter__()
        try:
            x = ops.nchw_to_tokens(input_feat.reshape(N * T, C, H, W))
            x = self.encoder.forward_tokens(x, Geom(N, T, H, W), local_window_pos_embed, temporal_pos_embed[:T])
            out = ops.tokens_to_nchw(x, N * T, C, H, W, relu=True).reshape(N, T, C, H, W)
        finally:
            arena.__exit__(None, None, None)
            plan.end()
            _dp_plan[0] = None
        return out
