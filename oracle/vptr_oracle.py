"""TEST INFRASTRUCTURE -- CPU oracle for the VPTR hot path (not product code).

A functional, state_dict-driven restatement of the reference algorithm
(XiYe20/VPTR) written with plain torch CPU ops.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this file; the product path (`vptr_amd/`) never does, and fails loudly when the
HIP library is missing.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4).  This restatement is pinned against the *imported
reference itself* (run in the build container, see `oracle/make_golden.py`)
and against the fixtures under `tests/golden/` that the same script generated.
`tests/test_cpu.py::test_oracle_*_golden` re-check the pin on every CPU test run.

Every function cites the reference file:line it follows (paths relative to
the reference repo root).  Dropout / DropPath are identity by default: the
reference RNG stream is not reproducible across implementations, so the golden
fixtures are defined at dropout = 0.  For the
trained configuration (dropout 0.1) the masks can be INJECTED: inside a
`with dropout_masks({site: scale tensor})` scope every nn.Dropout / attention
dropout / DropPath site of the reference multiplies by the given tensor
(values 0 or 1/keep), which lets a test run this oracle with exactly the masks
another implementation drew (tests/test_03_dropout_parity_gpu.py).
"""
import math

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------
# injected dropout / DropPath masks (test infrastructure)
# ----------------------------------------------------------------------------
_MASKS = [None]


class dropout_masks:
    """scope in which `_drop(site, x)` multiplies by masks[site]; a site that is reached without a mask raises KeyError"""

    def __init__(self, masks):
        self.masks = masks

    def __enter__(self):
        self.prev = _MASKS[0]
        _MASKS[0] = self.masks
        return self

    def __exit__(self, *exc):
        _MASKS[0] = self.prev
        return False


def _drop(site, x, dims=None):
    """nn.Dropout / F.dropout / drop_path at `site`: identity outside a dropout_masks scope.  dims: for DropPath, the leading
    dimensions the scale vector indexes (VidHRFormer_modules.py:563-575 draws one number per index of dim 0)."""
    m = _MASKS[0]
    if m is None:
        return x
    k = m[site].to(x.dtype)
    if dims is not None:
        k = k.reshape(tuple(k.shape) + (1,) * (x.dim() - k.dim()))
    elif k.shape != x.shape:
        raise ValueError("dropout mask of %s has shape %s, the tensor %s" % (site, tuple(k.shape), tuple(x.shape)))
    return x * k

# ----------------------------------------------------------------------------
# position tables  (utils/position_encoding.py:29-49, 67-93, 117-161)
# ----------------------------------------------------------------------------


def _sincos(pos, E):
    """pos: (...,) float positions starting at 1; returns (..., E) interleaved sin/cos."""
    i = torch.arange(E, dtype=torch.float32)
    dim_t = 10000.0 ** (2 * torch.div(i, 2, rounding_mode="floor") / E)
    ang = pos[..., None] / dim_t
    out = torch.empty_like(ang)
    out[..., 0::2] = ang[..., 0::2].sin()
    out[..., 1::2] = ang[..., 1::2].cos()
    return out


def pos1d(L, E):
    """temporal_pos (L, E): utils/position_encoding.py:29-49, used VPTR_modules.py:118-121."""
    return _sincos(torch.arange(1, L + 1, dtype=torch.float32), E)


def pos2d(E, H, W):
    """lw_pos (H, W, E): cat(y-half, x-half); utils/position_encoding.py:67-93, VPTR_modules.py:123-125."""
    y = torch.arange(1, H + 1, dtype=torch.float32)[:, None].expand(H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, :].expand(H, W)
    return torch.cat([_sincos(y, E // 2), _sincos(x, E // 2)], dim=-1)


def pos3d(E, T, H, W):
    """Tlw_pos (T, H, W, E): cat(t, y, x thirds); utils/position_encoding.py:117-161, VPTR_modules.py:127-129."""
    t = torch.arange(1, T + 1, dtype=torch.float32)[:, None, None].expand(T, H, W)
    y = torch.arange(1, H + 1, dtype=torch.float32)[None, :, None].expand(T, H, W)
    x = torch.arange(1, W + 1, dtype=torch.float32)[None, None, :].expand(T, H, W)
    return torch.cat([_sincos(t, E // 3), _sincos(y, E // 3), _sincos(x, E // 3)], dim=-1)


def rpe_index(ws):
    """relative_position_index (ws*ws, ws*ws) int64: MultiHeadAttentionRPE.py:373-387."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    dy = ys[:, None] - ys[None, :] + ws - 1
    dx = xs[:, None] - xs[None, :] + ws - 1
    return dy * (2 * ws - 1) + dx


# ----------------------------------------------------------------------------
# window partition (VidHRFormer_modules.py:497-525, pad 538-561)
# ----------------------------------------------------------------------------


def _pad_amounts(H, W, ws):
    ph = math.ceil(H / ws) * ws - H
    pw = math.ceil(W / ws) * ws - W
    return ph, pw


def win_partition(x, ws):
    """x (B,H,W,C) -> (B*nqh*nqw, ws*ws, C); centre zero pad if needed."""
    B, H, W, C = x.shape
    ph, pw = _pad_amounts(H, W, ws)
    if ph or pw:
        x = F.pad(x, (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    Hp, Wp = H + ph, W + pw
    x = x.reshape(B, Hp // ws, ws, Wp // ws, ws, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B * (Hp // ws) * (Wp // ws), ws * ws, C)


def win_reverse(xw, B, H, W, ws):
    C = xw.shape[-1]
    ph, pw = _pad_amounts(H, W, ws)
    Hp, Wp = H + ph, W + pw
    x = xw.reshape(B, Hp // ws, Wp // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hp, Wp, C)
    if ph or pw:
        x = x[:, ph // 2: ph // 2 + H, pw // 2: pw // 2 + W, :]
    return x


# ----------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------


def _heads(x, nh):
    """(B, L, C) -> (B, nh, L, hd); channel c -> head c // hd  (MultiHeadAttentionRPE.py:586-590)."""
    B, L, C = x.shape
    return x.reshape(B, L, nh, C // nh).transpose(1, 2)


def _attend(q, k, v, bias=None, causal=False, drop_site=None):
    """q already scaled. q (B,nh,Lq,hd), k/v (B,nh,Lk,hd).  drop_site: dropout on the attention probabilities
    (MultiHeadAttentionRPE.py:677-680; F.multi_head_attention_forward's dropout_p)."""
    s = q @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    if causal:
        Lq, Lk = s.shape[-2:]
        m = torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), diagonal=1)
        s = s.masked_fill(m, float("-inf"))
    p = s.softmax(dim=-1)
    if drop_site is not None:
        p = _drop(drop_site, p)
    o = p @ v
    B, nh, L, hd = o.shape
    return o.transpose(1, 2).reshape(B, L, nh * hd)


def win_attn(P, pre, xqk, xv, ws, nh, rpe, lw_pos=None):
    """SpatialLocalMultiheadAttention.forward (VidHRFormer_modules.py:321-357).

    xqk, xv: (N,T,H,W,C).  rpe=True -> MultiheadAttentionRPE (separate q/k/v
    Linears + bias table, MultiHeadAttentionRPE.py:543-545, 629-650);
    rpe=False -> stock packed-weight MHA with q=k=x+lw_pos.
    """
    N, T, H, W, C = xqk.shape
    hd = C // nh
    Xq = win_partition(xqk.reshape(N * T, H, W, C), ws)
    Xv = win_partition(xv.reshape(N * T, H, W, C), ws)
    a = pre + "attn."
    if rpe:
        q = F.linear(Xq, P[a + "q_proj.weight"], P[a + "q_proj.bias"]) * (hd ** -0.5)
        k = F.linear(Xq, P[a + "k_proj.weight"], P[a + "k_proj.bias"])
        v = F.linear(Xv, P[a + "v_proj.weight"], P[a + "v_proj.bias"])
        table = P[a + "relative_position_bias_table"]
        idx = P[a + "relative_position_index"].reshape(-1).long()
        bias = table[idx].reshape(ws * ws, ws * ws, nh).permute(2, 0, 1)
    else:
        Xq = Xq + lw_pos.reshape(ws * ws, C)
        Wq, Wk, Wv = P[a + "in_proj_weight"].chunk(3)
        bq, bk, bv = P[a + "in_proj_bias"].chunk(3)
        q = F.linear(Xq, Wq, bq) * (hd ** -0.5)
        k = F.linear(Xq, Wk, bk)
        v = F.linear(Xv, Wv, bv)
        bias = None
    o = _attend(_heads(q, nh), _heads(k, nh), _heads(v, nh), bias, drop_site=a + "probs")
    o = F.linear(o, P[a + "out_proj.weight"], P[a + "out_proj.bias"])
    return win_reverse(o, N * T, H, W, ws).reshape(N, T, H, W, C)


def mha(P, pre, qi, ki, vi, nh, causal=False):
    """stock nn.MultiheadAttention, seq-first (L, B, C) (VidHRFormer_modules.py:49,79-84,185-187,204-205)."""
    C = qi.shape[-1]
    hd = C // nh
    Wq, Wk, Wv = P[pre + "in_proj_weight"].chunk(3)
    bq, bk, bv = P[pre + "in_proj_bias"].chunk(3)
    q = F.linear(qi, Wq, bq).transpose(0, 1) * (hd ** -0.5)
    k = F.linear(ki, Wk, bk).transpose(0, 1)
    v = F.linear(vi, Wv, bv).transpose(0, 1)
    o = _attend(_heads(q, nh), _heads(k, nh), _heads(v, nh), None, causal, drop_site=pre + "probs")
    o = F.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])
    return o.transpose(0, 1)


# ----------------------------------------------------------------------------
# conv-FFN  (MlpDWBN.forward, VidHRFormer_modules.py:424-442)
# ----------------------------------------------------------------------------


def _ffn_norm(P, pre, y, norm, training, eps=1e-5):
    if norm == "bn":
        return F.batch_norm(y, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"], P[pre + "bias"],
                            training, 0.1, eps)
    w = P[pre + "weight"]
    return F.layer_norm(y, tuple(w.shape), w, P[pre + "bias"], eps)


def conv_ffn(P, pre, x, norm, training):
    """x (N,T,H,W,C).  norm 'bn' (NAR encoder blocks) or 'ln' (LayerNorm((ch,H,W)): FAR + all NAR decoder blocks)."""
    N, T, H, W, C = x.shape
    y = x.reshape(N * T, H, W, C).permute(0, 3, 1, 2)
    y = F.conv2d(y, P[pre + "fc1.weight"], P[pre + "fc1.bias"])
    y = F.gelu(_ffn_norm(P, pre + "norm1.", y, norm, training))
    y = F.conv2d(y, P[pre + "dw3x3.weight"], P[pre + "dw3x3.bias"], padding=1, groups=y.shape[1])
    y = _drop(pre + "drop.0", F.gelu(_ffn_norm(P, pre + "norm2.", y, norm, training)))      # self.drop, :431
    y = F.conv2d(y, P[pre + "fc2.weight"], P[pre + "fc2.bias"])
    y = _drop(pre + "drop.1", F.gelu(_ffn_norm(P, pre + "norm3.", y, norm, training)))      # self.drop again, :435
    return y.permute(0, 2, 3, 1).reshape(N, T, H, W, -1)


def _ln(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + "weight"], P[pre + "bias"], 1e-5)


# ----------------------------------------------------------------------------
# blocks  (VidHRFormer_modules.py:60-93, 164-211)
# ----------------------------------------------------------------------------


def enc_block(P, pre, x, lw_pos, tpos, cfg, far, training):
    N, T, H, W, C = x.shape
    nh, ws, rpe = cfg["nhead"], cfg["window_size"], cfg["rpe"]
    u = _ln(P, pre + "norm1.", x)
    x = x + _drop(pre + "drop_path.0", win_attn(P, pre + "SLMHSA.", u, u, ws, nh, rpe, lw_pos), dims=1)          # :68, per sample
    x = x + _drop(pre + "drop_path.1", conv_ffn(P, pre + "SpatialFFN.", _ln(P, pre + "norm2.", x), "ln" if far else "bn", training),
                  dims=1)                                                                                           # :71
    x = x.permute(1, 0, 2, 3, 4).reshape(T, N * H * W, C)
    u = _ln(P, pre + "norm3.", x)
    qk = u + tpos[:, None, :]
    x = x + _drop(pre + "drop1", mha(P, pre + "temporal_MHSA.", qk, qk, u, nh, causal=far))                       # :79-84
    u = _ln(P, pre + "norm4.", x)
    h = _drop(pre + "drop2", F.gelu(F.linear(u, P[pre + "linear1.weight"], P[pre + "linear1.bias"])))             # :88
    x = x + _drop(pre + "drop3", F.linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"]))                 # :89
    return x.reshape(T, N, H, W, C).permute(1, 0, 2, 3, 4)


def _ts_permute(x, ws):
    """TemporalLocalPermuteModule.permute (VidHRFormer_modules.py:444-470) after PadBlock centre padding:
    (N,T,H,W,C) -> ((T*ws*ws), N*(Hp/ws)*(Wp/ws), C) plus the padded size."""
    N, T, H, W, C = x.shape
    Hp, Wp = -(-H // ws) * ws, -(-W // ws) * ws
    ph, pw = Hp - H, Wp - W
    x = F.pad(x, (0, 0, pw // 2, pw - pw // 2, ph // 2, ph - ph // 2))
    x = x.reshape(N, T, Hp // ws, ws, Wp // ws, ws, C).permute(1, 3, 5, 0, 2, 4, 6)   # t ph pw n qh qw c
    return x.reshape(T * ws * ws, N * (Hp // ws) * (Wp // ws), C), (Hp, Wp)


def tslma(P, pre, mem, query, Tlw, ws, nh):
    """TemporalSpatialLocalMultiheadAttention.forward (VidHRFormer_modules.py:247-284): cross attention of the T2*ws*ws
    query tokens of a window to the T1*ws*ws memory tokens of the same window, stock packed-weight MHA."""
    N, T1, H, W, C = mem.shape
    T2 = query.shape[1]
    mp, (Hp, Wp) = _ts_permute(mem, ws)
    qp, _ = _ts_permute(query, ws)
    out = mha(P, pre + "attn.", qp + Tlw[T1:T1 + T2].flatten(0, 2)[:, None, :], mp + Tlw[:T1].flatten(0, 2)[:, None, :], mp, nh)
    out = out.reshape(T2, ws, ws, N, Hp // ws, Wp // ws, C).permute(3, 0, 4, 1, 5, 2, 6).reshape(N, T2, Hp, Wp, C)
    ph, pw = Hp - H, Wp - W
    return out[:, :, ph // 2:ph // 2 + H, pw // 2:pw // 2 + W, :]


def dec_block(P, pre, tgt, qpos, mem, lw_pos, tpos_f, tpos_p, cfg, training):
    N, T2, H, W, C = tgt.shape
    T1 = mem.shape[1]
    nh, ws, rpe = cfg["nhead"], cfg["window_size"], cfg["rpe"]
    t = _ln(P, pre + "norm1.", tgt)
    x = tgt + _drop(pre + "drop_path.0", win_attn(P, pre + "SLMHSA.", t + qpos, t, ws, nh, rpe, lw_pos), dims=1)   # :177
    x = x + _drop(pre + "drop_path.1", conv_ffn(P, pre + "SpatialFFN.", _ln(P, pre + "norm2.", x), "ln", training), dims=1)  # :179
    x = x.permute(1, 0, 2, 3, 4).reshape(T2, N * H * W, C)
    u = _ln(P, pre + "norm3.", x)
    qk = u + tpos_f[:, None, :]
    x = x + _drop(pre + "drop1", mha(P, pre + "temporal_MHSA.", qk, qk, u, nh))                                     # :184-187
    u = _ln(P, pre + "norm4.", x)
    h = _drop(pre + "drop2", F.gelu(F.linear(u, P[pre + "linear1.weight"], P[pre + "linear1.bias"])))               # :191
    x = x + _drop(pre + "drop3", F.linear(h, P[pre + "linear2.weight"], P[pre + "linear2.bias"]))                   # :192
    if cfg.get("TSLMA", False):       # VidHRFormer_modules.py:195-199
        x = x.reshape(T2, N, H, W, C).permute(1, 0, 2, 3, 4)
        u = _ln(P, pre + "norm5.", x)
        x = x + _drop(pre + "drop_path1.0", tslma(P, pre + "TSLMA.", mem, u + qpos, P["Tlw_pos"], ws, nh), dims=1)  # per sample
    else:
        u = _ln(P, pre + "norm5.", x)
        mem_s = mem.permute(1, 0, 2, 3, 4).reshape(T1, N * H * W, C)
        qpos_s = qpos.permute(1, 0, 2, 3, 4).reshape(T2, N * H * W, C)
        # :204 applies drop_path1 to a (T2, N*H*W, C) tensor: ONE draw per TIME STEP, not per sample
        x = x + _drop(pre + "drop_path1.0", mha(P, pre + "EncDecAttn.", u + qpos_s + tpos_f[:, None, :], mem_s + tpos_p[:, None, :],
                                                mem_s, nh), dims=1)
        x = x.reshape(T2, N, H, W, C).permute(1, 0, 2, 3, 4)
    x = x + _drop(pre + "drop_path1.1", conv_ffn(P, pre + "SpatialFFN1.", _ln(P, pre + "norm6.", x), "ln", training), dims=1)  # :209
    return x


def nar_forward(P, feat, cfg, training=False, return_pre=False):
    """VPTRFormerNAR.forward (VPTR_modules.py:140-147) -> VidHRFormerNAR.forward (VidHRFormer.py:28-53).

    feat (N,Tp,C,H,W) -> (N,Tf,C,H,W).  cfg keys: Tp, Tf, nhead, window_size,
    num_encoder_layers, num_decoder_layers, rpe.  return_pre: also return the pre-ReLU tensor (fixture generation
    uses it to keep gradient cotangents away from the ReLU kink).
    """
    Tp = feat.shape[1]
    x = feat.permute(0, 1, 3, 4, 2)
    tpos, lw = P["temporal_pos"], P["lw_pos"]
    for i in range(cfg["num_encoder_layers"]):
        x = enc_block(P, f"transformer.encoder.layers.{i}.", x, lw, tpos[:Tp], cfg, False, training)
    mem = _ln(P, "transformer.encoder.norm.", x)
    N = feat.shape[0]
    qpos = P["frame_queries"][None].expand(N, -1, -1, -1, -1)
    out = torch.zeros_like(qpos)
    for i in range(cfg["num_decoder_layers"]):
        out = dec_block(P, f"transformer.decoder.layers.{i}.", out, qpos, mem, lw, tpos[Tp:], tpos[:Tp], cfg, training)
    out = _ln(P, "transformer.decoder.norm.", out).permute(0, 1, 4, 2, 3)
    return (F.relu(out), out) if return_pre else F.relu(out)


def far_forward(P, feat, cfg, training=False, return_pre=False):
    """VPTRFormerFAR.forward (VPTR_modules.py:186-192) -> VidHRFormerFAR.forward (VidHRFormer.py:71-88)."""
    T = feat.shape[1]
    x = feat.permute(0, 1, 3, 4, 2)
    for i in range(cfg["num_encoder_layers"]):
        x = enc_block(P, f"transformer.encoder.layers.{i}.", x, P["lw_pos"], P["temporal_pos"][:T], cfg, True, training)
    x = _ln(P, "transformer.encoder.norm.", x).permute(0, 1, 4, 2, 3)
    return (F.relu(x), x) if return_pre else F.relu(x)


def nce_projector(P, feat):
    """NCE_projector on channel-last feats (VPTR_modules.py:135-137; train_NAR.py:81-82). feat (N,T,C,H,W)."""
    x = feat.permute(0, 1, 3, 4, 2)
    x = F.linear(F.relu(F.linear(x, P["NCE_projector.0.weight"], P["NCE_projector.0.bias"])),
                 P["NCE_projector.2.weight"], P["NCE_projector.2.bias"])
    return x.permute(0, 1, 4, 2, 3)


# ----------------------------------------------------------------------------
# ResNet auto-encoder (ResNetAutoEncoder.py:8-51, 53-101, 104-158)
# ----------------------------------------------------------------------------


def _pad(x, p, padding_type):
    if padding_type == "reflect":
        return F.pad(x, (p, p, p, p), mode="reflect"), 0
    if padding_type == "replicate":
        return F.pad(x, (p, p, p, p), mode="replicate"), 0
    if padding_type == "zero":
        return x, p
    raise NotImplementedError("padding [%s] is not implemented" % padding_type)


def _bn(P, pre, x, training):
    return F.batch_norm(x, P[pre + "running_mean"], P[pre + "running_var"], P[pre + "weight"], P[pre + "bias"],
                        training, 0.1, 1e-5)


def enc_forward(P, x, n_down=3, padding_type="reflect", training=False):
    """VPTREnc.forward (VPTR_modules.py:16-29). x (N,T,Cimg,H,W) -> (N,T,feat,H/8,W/8)."""
    N, T = x.shape[:2]
    y = x.flatten(0, 1)
    m = "encoder.model."
    y = F.conv2d(F.pad(y, (3, 3, 3, 3), mode="reflect"), P[m + "1.weight"])
    y = F.relu(_bn(P, m + "2.", y, training))
    idx = 4
    for _ in range(n_down):
        y = F.conv2d(y, P[m + f"{idx}.weight"], stride=2, padding=1)
        y = F.relu(_bn(P, m + f"{idx + 1}.", y, training))
        idx += 3
    for _ in range(9):
        b = m + f"{idx}.conv_block."
        if padding_type == "zero":
            c1, n1, c2, n2 = 0, 1, 3, 4
        else:
            c1, n1, c2, n2 = 1, 2, 5, 6
        h, p = _pad(y, 1, padding_type)
        h = F.relu(_bn(P, b + f"{n1}.", F.conv2d(h, P[b + f"{c1}.weight"], padding=p), training))
        h, p = _pad(h, 1, padding_type)
        h = _bn(P, b + f"{n2}.", F.conv2d(h, P[b + f"{c2}.weight"], padding=p), training)
        y = y + h
        idx += 1
    y = F.relu(y)
    return y.reshape(N, T, *y.shape[1:])


def dec_forward(P, feat, n_down=3, out_layer="Tanh", training=False):
    """VPTRDec.forward (VPTR_modules.py:36-47). feat (N,T,C,h,w) -> (N,T,Cimg,8h,8w)."""
    N, T = feat.shape[:2]
    y = feat.flatten(0, 1)
    m = "decoder.model."
    idx = 0
    for _ in range(n_down):
        y = F.conv_transpose2d(y, P[m + f"{idx}.weight"], stride=2, padding=1, output_padding=1)
        y = F.relu(_bn(P, m + f"{idx + 1}.", y, training))
        idx += 3
    y = F.conv2d(F.pad(y, (3, 3, 3, 3), mode="reflect"), P[m + f"{idx + 1}.weight"], P[m + f"{idx + 1}.bias"])
    if out_layer == "Tanh":
        y = torch.tanh(y)
    elif out_layer == "Sigmoid":
        y = torch.sigmoid(y)
    else:
        raise ValueError("Unsupported output layer")
    return y.reshape(N, T, *y.shape[1:])


# ----------------------------------------------------------------------------
# losses (criterion.py:115-132, 145-204, 227-259)
# ----------------------------------------------------------------------------


def mse_loss(gt, pred):
    return ((pred - gt) ** 2).mean()


def gdl_loss(gt, pred):
    """alpha = 1, no temporal weight (train_NAR.py:215)."""
    g, p = gt.flatten(0, -4), pred.flatten(0, -4)
    t1 = (g[:, :, 1:, :] - g[:, :, :-1, :]).abs()
    t2 = (p[:, :, 1:, :] - p[:, :, :-1, :]).abs()
    t3 = (g[:, :, :, :-1] - g[:, :, :, 1:]).abs()
    t4 = (p[:, :, :, :-1] - p[:, :, :, 1:]).abs()
    return (t1 - t2).abs().mean() + (t3 - t4).abs().mean()


def bipatch_nce(gt_f, pred_f, temperature=1.0):
    """gt_f/pred_f (N,T,C,h,w), stop-grad on negatives (criterion.py:227-259)."""
    N, T, C, h, w = gt_f.shape
    g = gt_f.permute(0, 1, 3, 4, 2).reshape(N * T, h * w, C)
    p = pred_f.permute(0, 1, 3, 4, 2).reshape(N * T, h * w, C)
    eye = torch.eye(h * w, dtype=g.dtype)
    s1 = (g @ p.transpose(1, 2)) * eye + (g @ p.detach().transpose(1, 2)) * (1 - eye)
    s2 = (p @ g.transpose(1, 2)) * eye + (p @ g.detach().transpose(1, 2)) * (1 - eye)
    tgt = torch.arange(h * w).repeat(N * T)
    l1 = F.cross_entropy(s1.flatten(0, 1) / temperature, tgt)
    l2 = F.cross_entropy(s2.flatten(0, 1) / temperature, tgt)
    return 0.5 * (l1 + l2)


def nar_losses(P_T, pred_frames, future, pred_feats, future_feats, lam_pc=0.1):
    """cal_lossT without GAN (train_NAR.py:33-47) incl. the NCE projector calls (:81-82)."""
    pf = nce_projector(P_T, pred_feats)
    gf = nce_projector(P_T, future_feats)
    l_mse = mse_loss(pred_frames, future)
    l_gdl = gdl_loss(future, pred_frames)
    l_pc = bipatch_nce(F.normalize(gf, p=2.0, dim=2), F.normalize(pf, p=2.0, dim=2))
    return l_gdl + l_mse + lam_pc * l_pc, l_gdl, l_mse, l_pc


# ----------------------------------------------------------------------------
# the measured step: single_iter (train_NAR.py:49-107), dropout 0
# ----------------------------------------------------------------------------


class NARStep:
    """Functional NAR train step on plain tensors: 2x Enc (no grad), NAR, Dec, MSE+GDL+0.1*BiPatchNCE,
    backward, clip_grad_norm_(1.0) on the transformer params, AdamW(1e-4).  State lives in dicts of leaf tensors."""

    def __init__(self, P_enc, P_dec, P_T, cfg, padding_type="reflect", out_layer="Tanh", lr=1e-4, max_grad_norm=1.0,
                 lam_pc=0.1, P_disc=None, lam_gan=None):
        self.cfg, self.padding_type, self.out_layer = cfg, padding_type, out_layer
        self.lam_gan = lam_gan
        self.P_disc = None
        if P_disc is not None:   # optional adversarial branch (train_NAR.py:22-30,37-41,66-79), Adam(lr, betas (0.5, 0.999)) :204
            self.P_disc = {k: v.detach().clone() for k, v in P_disc.items()}
            for k, v in self.P_disc.items():
                if v.is_floating_point() and not (k.endswith("running_mean") or k.endswith("running_var")):
                    v.requires_grad_(True)
            self.params_D = [v for v in self.P_disc.values() if v.requires_grad]
            self.opt_D = torch.optim.Adam(self.params_D, lr=lr, betas=(0.5, 0.999))
        self.P_enc = {k: v.detach().clone() for k, v in P_enc.items()}
        self.P_dec = {k: v.detach().clone() for k, v in P_dec.items()}
        self.P_T = {k: v.detach().clone() for k, v in P_T.items()}
        self.buffers_T = {"temporal_pos", "lw_pos", "Tlw_pos"}
        for k, v in self.P_T.items():
            if v.is_floating_point() and not self._is_buffer(k):
                v.requires_grad_(True)
        for k, v in self.P_dec.items():
            if v.is_floating_point() and not self._is_buffer(k):
                v.requires_grad_(True)  # reference leaves Dec params trainable (train_NAR.py:190-191)
        self.params_T = [v for k, v in self.P_T.items() if v.requires_grad]
        self.opt = torch.optim.AdamW(self.params_T, lr=lr)
        self.max_grad_norm, self.lam_pc = max_grad_norm, lam_pc

    def _is_buffer(self, k):
        return (k in ("temporal_pos", "lw_pos", "Tlw_pos") or k.endswith("running_mean") or k.endswith("running_var")
                or k.endswith("num_batches_tracked") or k.endswith("relative_position_index"))

    def forward_losses(self, past, future):
        with torch.no_grad():
            pf = enc_forward(self.P_enc, past, padding_type=self.padding_type)
            ff = enc_forward(self.P_enc, future, padding_type=self.padding_type)
        pred_feats = nar_forward(self.P_T, pf, self.cfg, training=True)
        pred_frames = dec_forward(self.P_dec, pred_feats, out_layer=self.out_layer)
        return nar_losses(self.P_T, pred_frames, future, pred_feats, ff, self.lam_pc) + (pred_frames,)

    def step(self, past, future):
        for p in self.params_T:
            p.grad = None
        for v in self.P_dec.values():
            if v.requires_grad:
                v.grad = None
        loss, l_gdl, l_mse, l_pc, pred_frames = self.forward_losses(past, future)
        extra = {}
        if self.P_disc is not None:
            for p in self.params_D:
                p.requires_grad_(True)
                p.grad = None
            l_fake = gan_loss_vanilla(disc_forward(self.P_disc, pred_frames.detach().flatten(0, 1), training=True), False)
            l_real = gan_loss_vanilla(disc_forward(self.P_disc, future.flatten(0, 1), training=True), True)
            loss_D = (l_fake + l_real) * 0.5 * self.lam_gan
            loss_D.backward()
            self.opt_D.step()
            for p in self.params_D:
                p.requires_grad_(False)
            t_gan = gan_loss_vanilla(disc_forward(self.P_disc, pred_frames.flatten(0, 1), training=True), True)
            loss = loss + self.lam_gan * t_gan
            extra = {"Dtotal": loss_D.item(), "Dfake": l_fake.item(), "Dreal": l_real.item(), "T_gan": t_gan.item()}
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.params_T, self.max_grad_norm)
        self.opt.step()
        return dict({"T_total": loss.item(), "T_GDL": l_gdl.item(), "T_MSE": l_mse.item(), "T_bpc": l_pc.item(),
                     "grad_norm": float(gn)}, **extra)


class FARStep:
    """Functional FAR train step without the GAN branch (train_FAR.py:48-101 with VPTR_Disc = None, its default :186-192):
    Enc(cat(past, future[:, :-1])) under no_grad, FAR (causal), Dec, MSE + GDL against cat(past[:, 1:], future), backward,
    clip_grad_norm_(1.0) on the transformer params, AdamW(1e-4)."""

    def __init__(self, P_enc, P_dec, P_T, cfg, padding_type="reflect", out_layer="Tanh", lr=1e-4, max_grad_norm=1.0):
        self.cfg, self.padding_type, self.out_layer = cfg, padding_type, out_layer
        self.P_enc = {k: v.detach().clone() for k, v in P_enc.items()}
        self.P_dec = {k: v.detach().clone() for k, v in P_dec.items()}
        self.P_T = {k: v.detach().clone() for k, v in P_T.items()}
        isbuf = NARStep._is_buffer
        for d in (self.P_T, self.P_dec):  # the reference leaves Dec params trainable here too (train_FAR.py:181-182)
            for k, v in d.items():
                if v.is_floating_point() and not isbuf(self, k):
                    v.requires_grad_(True)
        self.params_T = [v for v in self.P_T.values() if v.requires_grad]
        self.opt = torch.optim.AdamW(self.params_T, lr=lr)
        self.max_grad_norm = max_grad_norm

    def forward_losses(self, past, future):
        x = torch.cat([past, future[:, :-1]], dim=1)                     # train_FAR.py:54
        with torch.no_grad():
            gt_feats = enc_forward(self.P_enc, x, padding_type=self.padding_type)
        pred_feats = far_forward(self.P_T, gt_feats, self.cfg, training=True)
        pred_frames = dec_forward(self.P_dec, pred_feats, out_layer=self.out_layer)
        target = torch.cat([past[:, 1:], future], dim=1)                 # :80
        l_mse = mse_loss(pred_frames, target)                            # cal_lossT :32-46
        l_gdl = gdl_loss(target, pred_frames)
        return l_gdl + l_mse, l_gdl, l_mse, pred_frames

    def step(self, past, future):
        for p in self.params_T:
            p.grad = None
        for v in self.P_dec.values():
            if v.requires_grad:
                v.grad = None
        loss, l_gdl, l_mse, _ = self.forward_losses(past, future)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(self.params_T, self.max_grad_norm)
        self.opt.step()
        return {"T_total": loss.item(), "T_GDL": l_gdl.item(), "T_MSE": l_mse.item(), "grad_norm": float(gn)}


# ---------------------------------------------------------------------------------------------------------------------
# stage-1 auto-encoder training with the PatchGAN discriminator (train_AutoEncoder.py:20-86)
# ---------------------------------------------------------------------------------------------------------------------
def disc_forward(P, x, training=False, n_layers=3):
    """VPTRDisc.forward (VPTR_modules.py:49-95): x (N,Cimg,H,W) -> patch logits (N,1,h,w); keys `model.{i}.*`."""
    y = F.leaky_relu(F.conv2d(x, P["model.0.weight"], P["model.0.bias"], stride=2, padding=1), 0.2)
    i = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        y = F.conv2d(y, P[f"model.{i}.weight"], None, stride=stride, padding=1)
        y = F.leaky_relu(_bn(P, f"model.{i + 1}.", y, training), 0.2)
        i += 3
    return F.conv2d(y, P[f"model.{i}.weight"], P[f"model.{i}.bias"], stride=1, padding=1)


def gan_loss_vanilla(pred, target_is_real):
    """GANLoss('vanilla') (criterion.py:15-74): BCEWithLogits against a constant label map."""
    return F.binary_cross_entropy_with_logits(pred, torch.full_like(pred, 1.0 if target_is_real else 0.0))


class AEStep:
    """Functional stage-1 step (train_AutoEncoder.py:44-78): rec = Dec(Enc(x)) with train-mode BatchNorm; discriminator
    update on (rec.detach(), x) with Adam(2e-4, betas (0.5, 0.999)); generator update with
    lam_gan * GAN(D(rec), real) + MSE + GDL and the same Adam on Enc + Dec parameters."""

    def __init__(self, P_enc, P_dec, P_disc, padding_type="reflect", out_layer="Tanh", lr=2e-4, lam_gan=0.01):
        self.padding_type, self.out_layer, self.lam_gan = padding_type, out_layer, lam_gan
        self.P_enc = {k: v.detach().clone() for k, v in P_enc.items()}
        self.P_dec = {k: v.detach().clone() for k, v in P_dec.items()}
        self.P_disc = {k: v.detach().clone() for k, v in P_disc.items()}

        def isbuf(k):
            return k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
        for d in (self.P_enc, self.P_dec, self.P_disc):
            for k, v in d.items():
                if v.is_floating_point() and not isbuf(k):
                    v.requires_grad_(True)
        self.params_G = [v for d in (self.P_enc, self.P_dec) for v in d.values() if v.requires_grad]
        self.params_D = [v for v in self.P_disc.values() if v.requires_grad]
        self.opt_G = torch.optim.Adam(self.params_G, lr=lr, betas=(0.5, 0.999))
        self.opt_D = torch.optim.Adam(self.params_D, lr=lr, betas=(0.5, 0.999))

    def step(self, past, future):
        x = torch.cat([past, future], dim=1)
        for p in self.params_G + self.params_D:
            p.grad = None
        feat = enc_forward(self.P_enc, x, padding_type=self.padding_type, training=True)
        rec = dec_forward(self.P_dec, feat, out_layer=self.out_layer, training=True)
        for p in self.params_D:
            p.requires_grad_(True)
        pred_fake = disc_forward(self.P_disc, rec.detach().flatten(0, 1), training=True)
        pred_real = disc_forward(self.P_disc, x.flatten(0, 1), training=True)
        l_fake, l_real = gan_loss_vanilla(pred_fake, False), gan_loss_vanilla(pred_real, True)
        loss_D = (l_fake + l_real) * 0.5 * self.lam_gan
        loss_D.backward()
        self.opt_D.step()
        for p in self.params_D:
            p.requires_grad_(False)
        l_gan = gan_loss_vanilla(disc_forward(self.P_disc, rec.flatten(0, 1), training=True), True)
        l_mse, l_gdl = mse_loss(rec, x), gdl_loss(x, rec)
        loss_G = self.lam_gan * l_gan + l_mse + l_gdl
        loss_G.backward()
        self.opt_G.step()
        return {"AEgan": l_gan.item(), "AE_MSE": l_mse.item(), "AE_GDL": l_gdl.item(), "AE_total": loss_G.item(),
                "Dtotal": loss_D.item(), "Dfake": l_fake.item(), "Dreal": l_real.item()}
