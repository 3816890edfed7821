"""Diagnostic (GPU box): gradient w.r.t. each block boundary for the 2-layer tiny FAR config."""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from helpers import build_transformer, rel
from oracle import fill, vptr_oracle as O
import vptr_amd.model as pkg
import vptr_amd.ops as ops
from vptr_amd.model.vidhrformer import Geom
import torch.nn.functional as F

cfg = dict(Tp=3, Tf=3, Tin=5, H=8, W=8, C=48, nhead=8, window_size=4, num_encoder_layers=2, rpe=True)
N, seed = 2, 13
dev = torch.device("cuda:0")
m = build_transformer(pkg, cfg, True); fill.apply_fill(m, seed)
sd = {k: v.clone() for k, v in m.state_dict().items()}
T = 5
x = fill.rand_normal((N, T, 48, 8, 8), seed + 1).abs()
g = fill.rand_normal((N, T, 48, 8, 8), seed + 2)
# oracle, block by block
P = {k: v.clone() for k, v in sd.items()}
xo = x.permute(0, 1, 3, 4, 2).clone().requires_grad_(True)
acts_o = [xo]
h = xo
for i in range(2):
    h = O.enc_block(P, f"transformer.encoder.layers.{i}.", h, P["lw_pos"], P["temporal_pos"][:T], cfg, True, True)
    h.retain_grad(); acts_o.append(h)
y = O._ln(P, "transformer.encoder.norm.", h); y.retain_grad()
out_o = F.relu(y.permute(0, 1, 4, 2, 3))
(out_o * g).sum().backward()
# HIP, block by block
m = m.to(dev).train()
enc = m.transformer.encoder
geom = Geom(N, T, 8, 8)
xt = x.permute(0, 1, 3, 4, 2).reshape(-1, 48).contiguous().to(dev).requires_grad_(True)
acts_d = [xt]
h = xt
for layer in enc.layers:
    h = layer.forward_tokens(h, geom, m.lw_pos, m.temporal_pos[:T])
    h.retain_grad(); acts_d.append(h)
yd = ops.layernorm(h, enc.norm.weight, enc.norm.bias); yd.retain_grad()
out_d = ops.tokens_to_nchw(yd, N * T, 48, 8, 8, relu=True).reshape(N, T, 48, 8, 8)
(out_d * g.to(dev)).sum().backward()
print("out", rel(out_d, out_o))
print("d(LN out)", rel(yd.grad, y.grad.reshape(-1, 48)))
for i in (2, 1, 0):
    print("act", i, "fwd", rel(acts_d[i], acts_o[i].reshape(-1, 48)), "grad", rel(acts_d[i].grad, acts_o[i].grad.reshape(-1, 48)))
# run HIP backward a second time to check determinism
g1 = acts_d[0].grad.clone()
for a in acts_d: a.grad = None
yd.grad = None
h = xt
for layer in enc.layers:
    h = layer.forward_tokens(h, geom, m.lw_pos, m.temporal_pos[:T])
yd2 = ops.layernorm(h, enc.norm.weight, enc.norm.bias)
out_d2 = ops.tokens_to_nchw(yd2, N * T, 48, 8, 8, relu=True).reshape(N, T, 48, 8, 8)
(out_d2 * g.to(dev)).sum().backward()
print("repeat dx diff", rel(xt.grad, g1))
