"""Public module API of VPTR -- the drop-in boundary (reference: model/VPTR_modules.py).

Same class names, constructor/forward signatures, attribute names and state_dict keys as the reference's
`model.VPTREnc / VPTRDec / VPTRFormerNAR / VPTRFormerFAR`, so train_NAR.py / train_FAR.py keep
working and released checkpoints load.  Underneath every forward/backward is HIP kernels via vptr_amd.ops.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .autoencoder import ResnetDecoder, ResnetEncoder
from .position_encoding import temporal_table, temporal_window_table, window_table
from .vidhrformer import VidHRFormerFAR, VidHRFormerNAR


class VPTREnc(nn.Module):
    """frames (N,T,Cimg,H,W) -> features (N,T,feat_dim,H/8,W/8)   (VPTR_modules.py:10-29)."""

    def __init__(self, img_channels, feat_dim=528, n_downsampling=3, padding_type="reflect"):
        super().__init__()
        self.feat_dim = feat_dim
        self.encoder = ResnetEncoder(input_nc=img_channels, out_dim=feat_dim, n_downsampling=n_downsampling,
                                     padding_type=padding_type)

    def forward(self, x):
        N, T = x.shape[:2]
        feat = self.encoder(x.flatten(0, 1))
        return feat.reshape(N, T, *feat.shape[1:])


class VPTRDec(nn.Module):
    """features (N,T,feat_dim,h,w) -> frames (N,T,Cimg,8h,8w)   (VPTR_modules.py:31-47)."""

    def __init__(self, img_channels, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect"):
        super().__init__()
        self.decoder = ResnetDecoder(output_nc=img_channels, feat_dim=feat_dim, n_downsampling=n_downsampling,
                                     out_layer=out_layer, padding_type=padding_type)

    def forward(self, feat):
        N, T = feat.shape[:2]
        out = self.decoder(feat.flatten(0, 1))
        return out.reshape(N, T, *out.shape[1:])


class VPTRDisc(nn.Module):
    """PatchGAN discriminator (VPTR_modules.py:49-95): Conv4x4(s2)+LeakyReLU, n_layers-1 x [Conv4x4(s2)+BN+LeakyReLU],
    Conv4x4(s1)+BN+LeakyReLU, Conv4x4(s1) -> 1-channel patch logits.  On the HIP path every conv is an MFMA implicit GEMM
    with dgrad / wgrad (ops.conv2d_nhwc); the 1- or 3-channel input and the 1-channel output are zero-padded to 4 channels
    (the GEMM stages 4 channels per load)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        kw, padw = 4, 1
        seq = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf = 1
        for n in range(1, n_layers):
            nf_prev, nf = nf, min(2 ** n, 8)
            seq += [nn.Conv2d(ndf * nf_prev, ndf * nf, kernel_size=kw, stride=2, padding=padw, bias=False), norm_layer(ndf * nf),
                    nn.LeakyReLU(0.2, True)]
        nf_prev, nf = nf, min(2 ** n_layers, 8)
        seq += [nn.Conv2d(ndf * nf_prev, ndf * nf, kernel_size=kw, stride=1, padding=padw, bias=False), norm_layer(ndf * nf),
                nn.LeakyReLU(0.2, True)]
        seq += [nn.Conv2d(ndf * nf, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*seq)

    def forward(self, input):
        """input (N, Cimg, H, W) -> patch logits (N, 1, h, w)"""
        from ..ops import ACT_LRELU, ACT_NONE, conv2d_nhwc, nchw_to_tokens, tokens_to_nchw
        from .autoencoder import _bn_act
        N, Cimg, H, W = input.shape
        mods = list(self.model)
        cpad = (-Cimg) % 4
        x = nchw_to_tokens(F.pad(input.float(), (0, 0, 0, 0, 0, cpad)).contiguous())           # [N*H*W, Cimg+pad]
        conv = mods[0]
        y, h, w = conv2d_nhwc(x, F.pad(conv.weight, (0, 0, 0, 0, 0, cpad)), conv.bias, N, H, W, conv.stride[0], conv.padding[0],
                              "zero", act=ACT_LRELU)
        i = 2
        while i < len(mods) - 1:                                                               # [conv, norm, LeakyReLU] groups
            conv, bn = mods[i], mods[i + 1]
            if not isinstance(bn, nn.BatchNorm2d):
                raise NotImplementedError("VPTRDisc on the HIP path supports norm_layer=nn.BatchNorm2d only")
            y, h, w = conv2d_nhwc(y, conv.weight, conv.bias, N, h, w, conv.stride[0], conv.padding[0], "zero")
            y = _bn_act(y, bn, h * w, ACT_LRELU)
            i += 3
        conv = mods[-1]
        y, h, w = conv2d_nhwc(y, F.pad(conv.weight, (0, 0, 0, 0, 0, 0, 0, 3)), F.pad(conv.bias, (0, 3)), N, h, w, conv.stride[0],
                              conv.padding[0], "zero", act=ACT_NONE)
        return tokens_to_nchw(y[:, :1].contiguous(), N, 1, h, w)


class _NCEProjector(nn.Sequential):
    """Linear-ReLU-Linear on channel-last features, called by the train scripts as
    `T.NCE_projector(feats.permute(0,1,3,4,2)).permute(0,1,4,2,3)` (train_NAR.py:81-82)."""

    def forward(self, x):
        C = x.shape[-1]
        lead = x.shape[:-1]
        nchw = x.permute(0, 1, 4, 2, 3) if x.dim() == 5 else None
        if nchw is not None and nchw.is_contiguous():
            N, T, _, H, W = nchw.shape
            tok = ops.nchw_to_tokens(nchw.reshape(N * T, C, H, W))  # transposes in one HIP pass
        else:
            tok = x.reshape(-1, C)
        h = ops.linear(tok, self[0].weight, self[0].bias, act=ops.ACT_RELU)
        y = ops.linear(h, self[2].weight, self[2].bias)
        return y.reshape(*lead, self[2].weight.shape[0])


class VPTRFormerNAR(nn.Module):
    """Non-autoregressive VPTR transformer (VPTR_modules.py:98-152)."""

    def __init__(self, num_past_frames, num_future_frames, encH=8, encW=8, d_model=528, nhead=8, num_encoder_layers=6,
                 num_decoder_layers=6, dropout=0.1, window_size=4, Spatial_FFN_hidden_ratio=4, TSLMA_flag=False, rpe=True):
        super().__init__()
        self.num_past_frames, self.num_future_frames = num_past_frames, num_future_frames
        self.nhead, self.d_model = nhead, d_model
        self.num_encoder_layers, self.num_decoder_layers = num_encoder_layers, num_decoder_layers
        self.dropout, self.window_size, self.Spatial_FFN_hidden_ratio = dropout, window_size, Spatial_FFN_hidden_ratio
        self.transformer = VidHRFormerNAR((d_model, encH, encW), num_encoder_layers, num_decoder_layers, num_past_frames,
                                          num_future_frames, d_model, nhead, window_size=window_size, dropout=dropout,
                                          drop_path=dropout, Spatial_FFN_hidden_ratio=Spatial_FFN_hidden_ratio,
                                          dim_feedforward=d_model * Spatial_FFN_hidden_ratio, TSLMA_flag=TSLMA_flag, rpe=rpe)
        T = num_past_frames + num_future_frames
        self.register_buffer("temporal_pos", temporal_table(T, d_model))
        self.register_buffer("lw_pos", window_table(d_model, window_size))
        self.register_buffer("Tlw_pos", temporal_window_table(d_model, T, window_size))
        self.frame_queries = nn.Parameter(torch.randn(num_future_frames, encH, encW, d_model), requires_grad=True)
        self.NCE_projector = _NCEProjector(nn.Linear(d_model, d_model), nn.ReLU(inplace=True), nn.Linear(d_model, d_model))
        self._reset_parameters()

    def forward(self, past_gt_feat):
        """past_gt_feat (N,Tp,C,H,W) -> predicted future features (N,Tf,C,H,W), post-ReLU."""
        ops.new_seed_scope(past_gt_feat.device)
        ops.ensure_module_planes(self)
        pred, _ = self.transformer(past_gt_feat, self.lw_pos, self.temporal_pos, self.Tlw_pos, self.frame_queries,
                                   init_tgt=None)
        return pred

    def _reset_parameters(self):
        # xavier on EVERY parameter with dim > 1 -- including the RPE tables, frame_queries and the 3-D LayerNorm
        # affines of the conv-FFNs, exactly as the reference does (VPTR_modules.py:149-152)
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)


class VPTRFormerFAR(nn.Module):
    """Fully autoregressive VPTR transformer: encoder-only, causal temporal attention (VPTR_modules.py:154-197)."""

    def __init__(self, num_past_frames, num_future_frames, encH=8, encW=8, d_model=528, nhead=8, num_encoder_layers=6,
                 dropout=0.1, window_size=4, Spatial_FFN_hidden_ratio=4, rpe=True):
        super().__init__()
        self.num_past_frames, self.num_future_frames = num_past_frames, num_future_frames
        self.nhead, self.d_model, self.num_encoder_layers = nhead, d_model, num_encoder_layers
        self.dropout, self.window_size, self.Spatial_FFN_hidden_ratio = dropout, window_size, Spatial_FFN_hidden_ratio
        self.transformer = VidHRFormerFAR((d_model, encH, encW), num_encoder_layers, num_past_frames, num_future_frames,
                                          d_model, nhead, window_size=window_size, dropout=dropout, drop_path=dropout,
                                          Spatial_FFN_hidden_ratio=Spatial_FFN_hidden_ratio,
                                          dim_feedforward=d_model * Spatial_FFN_hidden_ratio, rpe=rpe)
        T = num_past_frames + num_future_frames
        self.register_buffer("temporal_pos", temporal_table(T, d_model))
        self.register_buffer("lw_pos", window_table(d_model, window_size))
        self._reset_parameters()

    def forward(self, input_feats):
        ops.new_seed_scope(input_feats.device)
        ops.ensure_module_planes(self)
        return self.transformer(input_feats, self.lw_pos, self.temporal_pos)

    def _reset_parameters(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
