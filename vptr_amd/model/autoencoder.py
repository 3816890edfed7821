"""ResNet auto-encoder of VPTR (stage 1) on the MI355X HIP kernels.

Mirrors the module tree / state_dict keys of the reference (model/ResNetAutoEncoder.py:8-51 ResnetEncoder,
:53-101 ResnetDecoder, :104-158 ResnetBlock, :160-189 init_weights): `self.model` is an nn.Sequential whose
indices hold the same parameterised leaves (nn.Conv2d / nn.BatchNorm2d / nn.ConvTranspose2d), used here as
parameter containers.  The forward pass runs NHWC implicit-GEMM convolutions on the matrix cores with BatchNorm
folded into the GEMM epilogue (eval mode), ReLU / residual fused, and direct 7x7 kernels at the image ends.

Two modes on the HIP path.  Eval-mode BatchNorm (stage-2 training and inference, train_NAR.py:190-191): BatchNorm folded into
the GEMM epilogues; encoder forward (the frozen encoder's ResnetBlock convolutions on plane operands), decoder forward and
decoder backward w.r.t. its input and -- as the reference computes them -- its weights.  Train-mode BatchNorm (stage 1,
train_AutoEncoder.py:44-86): raw convolution -> batch statistics -> normalise + ReLU with full autograd through encoder and
decoder (`ops.conv2d_nhwc`, `ops.norm_act`), used by `vptr_amd.train.AETrainer`.
"""
import functools

import os

import torch
import torch.nn as nn
from torch.nn import init

from .. import ops
from .._lib import check, lib, ptr, stream


def _use_bias(norm_layer):
    if type(norm_layer) == functools.partial:
        return norm_layer.func == nn.InstanceNorm2d
    return norm_layer == nn.InstanceNorm2d


def _pad_layers(padding_type):
    if padding_type == "reflect":
        return [nn.ReflectionPad2d(1)], 0
    if padding_type == "replicate":
        return [nn.ReplicationPad2d(1)], 0
    if padding_type == "zero":
        return [], 1
    raise NotImplementedError("padding [%s] is not implemented" % padding_type)


class ResnetBlock(nn.Module):
    """x + BN(conv3x3(pad(ReLU(BN(conv3x3(pad(x)))))))  (ResNetAutoEncoder.py:104-158)."""

    def __init__(self, dim, padding_type, norm_layer, use_dropout, use_bias):
        super().__init__()
        if use_dropout:
            raise NotImplementedError("ResnetBlock dropout is never enabled by the reference scripts")
        block = []
        pads, p = _pad_layers(padding_type)
        block += pads + [nn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim), nn.ReLU(True)]
        pads, p = _pad_layers(padding_type)
        block += pads + [nn.Conv2d(dim, dim, kernel_size=3, padding=p, bias=use_bias), norm_layer(dim)]
        self.conv_block = nn.Sequential(*block)
        self.padding_type = padding_type

    def _layers(self):
        convs = [m for m in self.conv_block if isinstance(m, nn.Conv2d)]
        bns = [m for m in self.conv_block if isinstance(m, nn.BatchNorm2d)]
        return convs, bns


def _bn_eval(bn):
    if bn.training:
        raise NotImplementedError("HIP auto-encoder path supports eval-mode BatchNorm only (stage-2 / inference); "
                                  "stage-1 AE training is a 'next' row (SURVEY.md section 8f)")
    # the fold (4 tiny launches) is cached per module and recomputed only when one of its tensors was modified in place
    # or re-pointed: stage 2 runs ~100 eval-BN folds per step on parameters that never change
    if not ops.config.weights_frozen:
        return ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    key = tuple(t._version for t in ts) + tuple(t.data_ptr() for t in ts)
    hit = getattr(bn, "_vptr_fold", None)
    if hit is not None and hit[0] == key:
        return hit[1]
    with torch.no_grad():
        val = ops.bn_fold(bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)
    bn._vptr_fold = (key, val)
    return val


def _bn_act(y, bn, HW, act, residual=None):
    """BatchNorm2d (+ ReLU / LeakyReLU, + residual) on NHWC tokens [pixels, C]: batch statistics + running-stat update in
    train mode, running statistics in eval mode -- both with autograd (vptr_colstats + vptr_norm_act_fwd/bwd)."""
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats) else None   # incremented by the statistics launch
    return ops.norm_act(y, bn.weight, bn.bias, "bn", HW, bn.training, running_mean=bn.running_mean, running_var=bn.running_var,
                        act=act, eps=bn.eps, momentum=bn.momentum if bn.momentum is not None else 0.1, residual=residual,
                        num_batches_tracked=nbt)


class ResnetEncoder(nn.Module):
    def __init__(self, input_nc, ngf=64, out_dim=528, n_downsampling=2, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 padding_type="reflect"):
        super().__init__()
        use_bias = _use_bias(norm_layer)
        model = [nn.ReflectionPad2d(3), nn.Conv2d(input_nc, ngf, kernel_size=7, padding=0, bias=use_bias), norm_layer(ngf),
                 nn.ReLU(True)]
        for i in range(n_downsampling - 1):
            mult = 2 ** i
            model += [nn.Conv2d(ngf * mult, ngf * mult * 2, kernel_size=3, stride=2, padding=1, bias=use_bias),
                      norm_layer(ngf * mult * 2), nn.ReLU(True)]
        mult = 2 ** (n_downsampling - 1)
        model += [nn.Conv2d(ngf * mult, out_dim, kernel_size=3, stride=2, padding=1, bias=use_bias), norm_layer(out_dim),
                  nn.ReLU(True)]
        for _ in range(9):
            model += [ResnetBlock(out_dim, padding_type=padding_type, norm_layer=norm_layer, use_dropout=use_dropout,
                                  use_bias=use_bias)]
        model += [nn.ReLU()]
        self.model = nn.Sequential(*model)
        self.padding_type = padding_type
        self.n_downsampling = n_downsampling

    def forward(self, x):
        """x (B, Cimg, H, W) NCHW -> (B, out_dim, H/2^n, W/2^n)."""
        with ops.frozen_weights(ops.weights_cacheable(self)):
            return self._forward(x)

    def _forward(self, x):
        if self.training:
            # stage-1 training (train_AutoEncoder.py:52-56): train-mode BatchNorm, gradients for every parameter
            return self._forward_train(x)
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("gradients w.r.t. the input frames of an eval-mode VPTREnc are not on the HIP path")
        with torch.no_grad():  # stage 2 / inference: the encoder runs under no_grad (train_NAR.py:54-56)
            return self._forward_impl(x)

    def _forward_train(self, x):
        """Differentiable forward with whatever mode each BatchNorm2d is in: conv (MFMA implicit GEMM, raw output) ->
        batch statistics -> normalise + ReLU, ResnetBlocks with the skip add fused into the second normalise pass."""
        B, Cimg, H, W = x.shape
        m = self.model
        y = ops.conv7_in(x.contiguous().float(), m[1].weight)
        if m[1].bias is not None:
            y = y + m[1].bias
        y = _bn_act(y, m[2], H * W, ops.ACT_RELU)
        h, w, cin = H, W, m[1].weight.shape[0]
        idx = 4
        for _ in range(self.n_downsampling):
            conv, bn = m[idx], m[idx + 1]
            y, h, w = ops.conv2d_nhwc(y, conv.weight, conv.bias, B, h, w, 2, 1, "zero")
            y = _bn_act(y, bn, h * w, ops.ACT_RELU)
            cin = conv.weight.shape[0]
            idx += 3
        for bi in range(9):
            convs, bns = m[idx + bi]._layers()
            t, _, _ = ops.conv2d_nhwc(y, convs[0].weight, convs[0].bias, B, h, w, 1, 1, self.padding_type)
            t = _bn_act(t, bns[0], h * w, ops.ACT_RELU)
            t, _, _ = ops.conv2d_nhwc(t, convs[1].weight, convs[1].bias, B, h, w, 1, 1, self.padding_type)
            y = _bn_act(t, bns[1], h * w, ops.ACT_NONE, residual=y)                  # out = x + conv_block(x)
        return ops.tokens_to_nchw(y, B, cin, h, w, relu=True)                        # the trailing nn.ReLU()

    def _block_plane_bufs(self, B, h, w, cin, device):
        """the two persistent, zero-initialised plane buffers the ResnetBlock convolutions alternate between"""
        key = (B * h * w, cin, device)
        if getattr(self, "_plane_bufs", (None,))[0] != key:
            shape = (B * h * w + 1, (cin + 31) // 32, 64)
            self._plane_bufs = (key, torch.zeros(shape, device=device, dtype=torch.bfloat16),
                                torch.zeros(shape, device=device, dtype=torch.bfloat16))
        return self._plane_bufs[1], self._plane_bufs[2]

    def _forward_impl(self, x):
        B, Cimg, H, W = x.shape
        m = self.model
        x = x.contiguous().float()
        scale, shift = _bn_eval(m[2])
        ngf = m[1].weight.shape[0]
        # frozen encoder (stage 2), 1 - 4 image channels: the whole down-sampling chain in plane form as well -- the 7x7 kernel writes
        # its output as bf16 hi / lo planes, every strided 3x3 convolution reads planes and writes planes (persistent, zero-initialised
        # buffers: pad channels and the all-zero row are never touched), the last one straight into the ResnetBlocks' first buffer
        # (n_downsampling >= 1: the loop below is what produces the fp32 residual `y` of the first ResnetBlock.  The persistent plane /
        # head buffers are per-module scratch keyed on shape: one forward at a time per module, i.e. one stream -- as every caller here)
        chain = (ops.config.weights_frozen and self.n_downsampling >= 1 and (Cimg == 1 or (Cimg <= 4 and W <= 512)) and W % 4 == 0 and ngf == 64 and os.environ.get("VPTR_ENC_PLANES", "1") != "0"
                 and os.environ.get("VPTR_ENC_PLANES_HEAD", "1") != "0"
                 and all(m[4 + 3 * i].weight.shape[1] % 32 == 0 and m[4 + 3 * i].weight.shape[0] % 4 == 0 for i in range(self.n_downsampling)))
        y = None
        chain_out = None
        if chain:
            geo = [(H, W, ngf)]
            for i in range(self.n_downsampling):
                hh, ww, _ = geo[-1]
                geo.append(((hh + 2 - 3) // 2 + 1, (ww + 2 - 3) // 2 + 1, m[4 + 3 * i].weight.shape[0]))
            key = ("head", B, H, W, x.device, tuple(geo))
            if getattr(self, "_head_bufs", (None,))[0] != key:
                self._head_bufs = (key, [torch.zeros((B * hh * ww + 1, (cc + 31) // 32, 64), device=x.device, dtype=torch.bfloat16)
                                         for (hh, ww, cc) in geo[:-1]])
            bufs = list(self._head_bufs[1]) + [self._block_plane_bufs(B, geo[-1][0], geo[-1][1], geo[-1][2], x.device)[0]]
            check(lib.vptr_conv7_in_fwd_planes(ptr(x), ptr(m[1].weight.contiguous()), ptr(scale.contiguous()), ptr(shift.contiguous()),
                                               ptr(bufs[0]), B, Cimg, H, W, ngf, stream()), "vptr_conv7_in_fwd_planes")
            idx = 4
            for i in range(self.n_downsampling):
                conv, bn = m[idx], m[idx + 1]
                (h, w, cin), (oh, ow, cout) = geo[i], geo[i + 1]
                scale, shift = _bn_eval(bn)
                last = i == self.n_downsampling - 1   # the ResnetBlocks' residual path needs the last layer as fp32 too
                y = ops.conv_nhwc_planes(bufs[i], ops.conv_weight_as_planes(conv.weight), B, h, w, cin, oh, ow, 3, 3, 2, 1, "zero", cout,
                                         colscale=scale, bias=shift, act=ops.ACT_RELU, planes_out=bufs[i + 1], fp32_out=last)
                idx += 3
            h, w, cin = geo[-1]
            chain_out = bufs[-1]
        else:
            y = torch.empty((B * H * W, ngf), device=x.device, dtype=torch.float32)
            check(lib.vptr_conv7_in_fwd(ptr(x), ptr(m[1].weight.contiguous()), ptr(scale.contiguous()), ptr(shift.contiguous()),
                                        ptr(y), B, Cimg, H, W, ngf, stream()), "vptr_conv7_in_fwd")
            h, w, cin = H, W, ngf
            idx = 4
            for _ in range(self.n_downsampling):
                conv, bn = m[idx], m[idx + 1]
                cout = conv.weight.shape[0]
                oh, ow = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
                scale, shift = _bn_eval(bn)
                y = ops.conv_nhwc(y, ops.conv_weight_as_gemm_b(conv.weight, False), B, h, w, cin, oh, ow, 3, 3, 2, 1, "zero", False,
                                  cout, colscale=scale, bias=shift, act=ops.ACT_RELU)
                h, w, cin = oh, ow, cout
                idx += 3
        pad_mode = self.padding_type
        # frozen encoder (stage 2): activations and weights of the 18 ResnetBlock convolutions as bf16 hi / lo planes, staged
        # by global_load_lds ("convert once"); anywhere else the register-staged fp32 path
        planes = ops.config.weights_frozen and cin % 4 == 0 and os.environ.get("VPTR_ENC_PLANES", "1") != "0"
        # frozen encoder, maps of whole 4 x 4 tiles: the 18 ResnetBlock convolutions as Winograd F(4x4, 3x3) -- 36 multiplies per tile instead
        # of 144 (ops.wino_conv3x3: input transform, one 36-member strided-batch P16 GEMM, output transform with BN / ReLU / skip fused)
        wino = planes and ops.wino_ok(h, w, cin, cin) and all(c.bias is None for c in m[idx]._layers()[0])
        wbufs = ops.wino_buffers(B, h, w, cin, x.device, self) if wino else None
        if wino:
            blocks = []
            for bi in range(9):
                convs, bns = m[idx + bi]._layers()
                blocks.append((ops.wino_filter(convs[0].weight),) + tuple(_bn_eval(bns[0])) + (ops.wino_filter(convs[1].weight),) + tuple(_bn_eval(bns[1])))
            y = ops.wino_resnet_blocks(y, blocks, B, h, w, wbufs, pad_mode, last_relu=True)
        for bi in range(0 if not wino else 9, 9):
            blk = m[idx + bi]
            convs, bns = blk._layers()
            s1, b1 = _bn_eval(bns[0])
            s2, b2 = _bn_eval(bns[1])
            if planes:
                # two persistent, zero-initialised plane buffers: the convs write their result in plane form themselves
                # (pad channels and the all-zero row are never touched), only the very first input needs a split pass
                if bi == 0:
                    pa, pb = self._block_plane_bufs(B, h, w, cin, y.device)
                    if chain_out is not pa:
                        ops.split_planes(y, out=pa)
                ops.conv_nhwc_planes(pa, ops.conv_weight_as_planes(convs[0].weight), B, h, w, cin, h, w, 3, 3, 1, 1, pad_mode, cin,
                                     colscale=s1, bias=b1, act=ops.ACT_RELU, planes_out=pb, fp32_out=False)
                y = ops.conv_nhwc_planes(pb, ops.conv_weight_as_planes(convs[1].weight), B, h, w, cin, h, w, 3, 3, 1, 1, pad_mode, cin,
                                         colscale=s2, bias=b2, residual=y, act_after=(bi == 8), planes_out=(pa if bi < 8 else None))
                continue
            t = ops.conv_nhwc(y, ops.conv_weight_as_gemm_b(convs[0].weight, False), B, h, w, cin, h, w, 3, 3, 1, 1, pad_mode,
                              False, cin, colscale=s1, bias=b1, act=ops.ACT_RELU)
            y = ops.conv_nhwc(t, ops.conv_weight_as_gemm_b(convs[1].weight, False), B, h, w, cin, h, w, 3, 3, 1, 1, pad_mode,
                              False, cin, colscale=s2, bias=b2, residual=y, act_after=(bi == 8))
        out = torch.empty((B, cin, h, w), device=x.device, dtype=torch.float32)
        check(lib.vptr_tokens_to_nchw(ptr(y), ptr(out), B, cin, h * w, 0, stream()), "vptr_tokens_to_nchw")
        return out


class _DecoderFn(torch.autograd.Function):
    """Whole ResnetDecoder as one autograd node: forward keeps the NHWC activations, backward walks the layers in
    reverse (7x7 data-gradient kernel, then folded-BN/ReLU mask + strided-conv dgrad GEMM per up-sampling layer)."""

    @staticmethod
    def forward(ctx, feat, dec, *params):
        with ops.frozen_weights(ops.weights_cacheable(dec)):
            return _DecoderFn._forward(ctx, feat, dec, *params)

    @staticmethod
    def backward(ctx, dout):
        with ops.frozen_weights(ops.weights_cacheable(ctx.dec)):
            return _DecoderFn._backward(ctx, dout)

    @staticmethod
    def _forward(ctx, feat, dec, *params):
        B, C, h, w = feat.shape
        m = dec.model
        n_up = dec.n_upsampling
        x = torch.empty((B * h * w, C), device=feat.device, dtype=torch.float32)
        check(lib.vptr_nchw_to_tokens(ptr(feat.contiguous()), ptr(x), B, C, h * w, stream()), "vptr_nchw_to_tokens")
        acts, scales, geoms = [], [], []
        cin = C
        x0 = x
        for i in range(n_up):
            convt, bn = m[3 * i], m[3 * i + 1]
            cout = convt.weight.shape[1]
            scale, shift = _bn_eval(bn)
            oh, ow = 2 * h, 2 * w
            y = ops.conv_nhwc(x, ops.conv_weight_as_gemm_b(convt.weight, True), B, h, w, cin, oh, ow, 3, 3, 2, 1, "zero", True,
                              cout, colscale=scale, bias=shift, act=ops.ACT_RELU)
            acts.append(y)
            scales.append(scale)
            geoms.append((h, w, cin, oh, ow, cout))
            x, h, w, cin = y, oh, ow, cout
        conv = m[3 * n_up + 1]
        cimg = conv.weight.shape[0]
        out = torch.empty((B, cimg, h, w), device=feat.device, dtype=torch.float32)
        check(lib.vptr_conv7_out_fwd(ptr(x), ptr(conv.weight.contiguous()), ptr(conv.bias.contiguous()), ptr(out), B, cin, h, w,
                                     cimg, dec.out_act, stream()), "vptr_conv7_out_fwd")
        ctx.dec, ctx.acts, ctx.scales, ctx.geoms, ctx.B = dec, acts, scales, geoms, B
        ctx.x0 = x0 if any(p.requires_grad for p in dec.parameters()) else None
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def _backward(ctx, dout):
        (out,) = ctx.saved_tensors
        dec, acts, scales, geoms, B = ctx.dec, ctx.acts, ctx.scales, ctx.geoms, ctx.B
        m = dec.model
        n_up = dec.n_upsampling
        conv = m[3 * n_up + 1]
        dout = dout.contiguous()
        h, w, cin = geoms[-1][3], geoms[-1][4], geoms[-1][5]
        cimg = conv.weight.shape[0]
        g = torch.empty((B * h * w, cin), device=dout.device, dtype=torch.float32)
        check(lib.vptr_conv7_out_bwd_data(ptr(dout), ptr(out), ptr(conv.weight.contiguous()), ptr(g), B, cin, h, w, cimg,
                                          dec.out_act, stream()), "vptr_conv7_out_bwd_data")
        # weight gradients (the reference leaves the decoder trainable in stage 2 and computes them every step although
        # nothing steps them, train_NAR.py:190-191,205); produced only for parameters that require grad
        wgrads = {}
        want_w = ctx.x0 is not None
        if want_w and conv.weight.requires_grad:
            dw7 = torch.zeros_like(conv.weight)
            db7 = torch.zeros_like(conv.bias)
            wsp = torch.empty((lib.vptr_conv7_out_bwd_weight_workspace(B, cimg),), device=dout.device, dtype=torch.float32)
            check(lib.vptr_conv7_out_bwd_weight_ws(ptr(dout), ptr(out), ptr(acts[-1]), ptr(dw7), ptr(db7), B, cin, h, w, cimg,
                                                   dec.out_act, ptr(wsp), wsp.numel(), stream()), "vptr_conv7_out_bwd_weight_ws")
            wgrads[id(conv.weight)], wgrads[id(conv.bias)] = dw7, db7
        convt_jobs = []
        for i in reversed(range(n_up)):
            ih, iw, ic, oh, ow, oc = geoms[i]
            gm = torch.empty_like(g)
            bn, convt = m[3 * i + 1], m[3 * i]
            if want_w and bn.weight.requires_grad and oc % 4 == 0:
                # masked gradient and the affine gradients in one pass over (g, activation)
                dbw, dbb = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
                check(lib.vptr_bnrelu_bwd_fused(ptr(g), ptr(acts[i]), ptr(scales[i].contiguous()), ptr(bn.weight.contiguous()),
                                                ptr(bn.bias.contiguous()), ptr(gm), ptr(dbw), ptr(dbb), B * oh * ow, oc, stream()),
                      "vptr_bnrelu_bwd_fused")
                wgrads[id(bn.weight)], wgrads[id(bn.bias)] = dbw, dbb
            else:
                check(lib.vptr_bnrelu_bwd(ptr(g), ptr(acts[i]), ptr(scales[i].contiguous()), ptr(gm), B * oh * ow, oc, stream()),
                      "vptr_bnrelu_bwd")
                if want_w and bn.weight.requires_grad:
                    dbw, dbb = torch.zeros_like(bn.weight), torch.zeros_like(bn.bias)
                    check(lib.vptr_bnrelu_bwd_params(ptr(g), ptr(acts[i]), ptr(bn.weight.contiguous()), ptr(bn.bias.contiguous()),
                                                     ptr(dbw), ptr(dbb), B * oh * ow, oc, stream()), "vptr_bnrelu_bwd_params")
                    wgrads[id(bn.weight)], wgrads[id(bn.bias)] = dbw, dbb
            if want_w and convt.weight.requires_grad:
                # dW[ci][co][ky][kx] = sum_pix x[pix][ci] * gm[(iy*2-1+ky, ix*2-1+kx)][co]: im2col of gm (3x3, s2, p1) and
                # D[ci][(ky,kx,co)] = x^T . P
                xin = ctx.x0 if i == 0 else acts[i - 1]
                if ops.p16_ok(ic, oc) and os.environ.get("VPTR_DEC_WGRAD_P16", "1") != "0":
                    # all layers together at the end: token-range sub-problems in one launch of the token-major P16 kernel
                    convt_jobs.append((convt, (xin, gm, B, ih, iw, ic, oh, ow, oc)))
                else:
                    P = torch.empty((B * ih * iw, 9 * oc), device=dout.device, dtype=torch.float32)
                    check(lib.vptr_im2col_nhwc(ptr(gm), ptr(P), B, oh, ow, oc, ih, iw, 3, 3, 2, 1, 0, stream()), "vptr_im2col_nhwc")
                    D = torch.zeros((ic, 9 * oc), device=dout.device, dtype=torch.float32)
                    tiles = ((ic + 127) // 128) * ((9 * oc + 175) // 176)
                    ops.gemm_raw(xin, P, D, ic, 9 * oc, B * ih * iw, 1, 1, atomic=True, split_k=ops._split_k_for(tiles, B * ih * iw))
                    wgrads[id(convt.weight)] = D.view(ic, 3, 3, oc).permute(0, 3, 1, 2).contiguous()
            # dgrad of ConvTranspose2d(3x3, s2, p1, op1) = Conv2d(3x3, s2, p1) of the output gradient with
            # B[ci][(ky,kx,co)] = W[ci][co][ky][kx]
            wt = m[3 * i].weight
            Bm = ops.conv_weight_as_gemm_b(wt, False)  # = wt.permute(0, 2, 3, 1).reshape(Cin, -1), cached
            g = ops.conv_nhwc(gm, Bm, B, oh, ow, oc, ih, iw, 3, 3, 2, 1, "zero", False, ic)
        if convt_jobs:
            for (convt, job), D in zip(convt_jobs, ops.convt_weight_grads([j for _, j in convt_jobs])):
                ic, oc = job[5], job[8]
                wgrads[id(convt.weight)] = D.view(ic, 3, 3, oc).permute(0, 3, 1, 2).contiguous()
        h0, w0, c0 = geoms[0][0], geoms[0][1], geoms[0][2]
        dfeat = torch.empty((B, c0, h0, w0), device=dout.device, dtype=torch.float32)
        check(lib.vptr_tokens_to_nchw(ptr(g), ptr(dfeat), B, c0, h0 * w0, 0, stream()), "vptr_tokens_to_nchw")
        pgrads = tuple(wgrads.get(id(p)) for p in dec.parameters())
        return (dfeat, None) + pgrads


class ResnetDecoder(nn.Module):
    def __init__(self, output_nc, ngf=64, feat_dim=528, n_downsampling=2, norm_layer=nn.BatchNorm2d, use_dropout=False,
                 padding_type="reflect", out_layer="Tanh"):
        super().__init__()
        use_bias = _use_bias(norm_layer)
        model = []
        mult = 2 ** n_downsampling
        model += [nn.ConvTranspose2d(feat_dim, int(ngf * mult / 2), kernel_size=3, stride=2, padding=1, output_padding=1,
                                     bias=use_bias), norm_layer(int(ngf * mult / 2)), nn.ReLU(True)]
        for i in range(1, n_downsampling):
            mult = 2 ** (n_downsampling - i)
            model += [nn.ConvTranspose2d(ngf * mult, int(ngf * mult / 2), kernel_size=3, stride=2, padding=1,
                                         output_padding=1, bias=use_bias), norm_layer(int(ngf * mult / 2)), nn.ReLU(True)]
        model += [nn.ReflectionPad2d(3)]
        model += [nn.Conv2d(ngf, output_nc, kernel_size=7, padding=0)]
        if out_layer == "Tanh":
            model += [nn.Tanh()]
            self.out_act = 1
        elif out_layer == "Sigmoid":
            model += [nn.Sigmoid()]
            self.out_act = 2
        else:
            raise ValueError("Unsupported output layer")
        self.model = nn.Sequential(*model)
        self.n_upsampling = n_downsampling

    def forward(self, x):
        """x (B, feat_dim, h, w) -> (B, output_nc, h*2^n, w*2^n); differentiable w.r.t. x (decoder weights are frozen
        in stage 2: the reference never steps them, train_NAR.py:205)."""
        if self.training:
            return self._forward_train(x)
        return _DecoderFn.apply(x, self, *[p for p in self.parameters()])

    def _forward_train(self, x):
        """stage-1 training (train_AutoEncoder.py:52-56): ConvTranspose2d (gather-form MFMA GEMM) -> train-mode BatchNorm +
        ReLU, then the direct 7x7 output convolution + Tanh / Sigmoid, all with parameter gradients."""
        B, C, h, w = x.shape
        m = self.model
        y = ops.nchw_to_tokens(x.contiguous().float())
        for i in range(self.n_upsampling):
            convt, bn = m[3 * i], m[3 * i + 1]
            y, h, w = ops.conv2d_nhwc(y, convt.weight, convt.bias, B, h, w, 2, 1, "zero", transposed=True, output_padding=1)
            y = _bn_act(y, bn, h * w, ops.ACT_RELU)
        conv = m[3 * self.n_upsampling + 1]
        return ops.conv7_out(y, conv.weight, conv.bias, B, h, w, self.out_act)


def init_weights(net, init_type="normal", init_gain=0.02):
    """N(0, gain) on every Conv*/Linear* weight (bias 0), N(1, gain) on BatchNorm2d weights -- matched by class-name
    substring exactly like the reference (ResNetAutoEncoder.py:160-189)."""

    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    print("initialize network with %s" % init_type)
    net.apply(init_func)
