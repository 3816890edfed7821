"""Losses of the VPTR training steps (reference: model/criterion.py).

Element-wise / reduction work that is < 0.1 % of a train step: these stay ordinary torch
device ops on the MI355X (no custom kernels), with the reference's call signatures: `loss(gt, pred)`.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def temporal_weight_func(T):
    """exp-increasing per-frame weights w_t = T^(t/(T-1)) (criterion.py:8-13)."""
    return torch.exp(np.log(T) / (T - 1) * torch.linspace(0, T - 1, T))


def _apply_temporal_weight(e, w):
    if w is None:
        return e
    w = w.to(e.device)
    return e * w.reshape((1, -1) + (1,) * (e.dim() - 2))


class _NormedPointwise(nn.Module):
    def __init__(self, temporal_weight=None, norm_dim=None):
        super().__init__()
        self.temporal_weight, self.norm_dim = temporal_weight, norm_dim

    def _prep(self, gt, pred):
        if self.norm_dim is not None:
            gt, pred = F.normalize(gt, p=2, dim=self.norm_dim), F.normalize(pred, p=2, dim=self.norm_dim)
        return gt, pred


class MSELoss(_NormedPointwise):
    """mean((pred - gt)^2), optional temporal weights over dim 1 (criterion.py:105-132)."""

    def __call__(self, gt, pred):
        gt, pred = self._prep(gt, pred)
        return _apply_temporal_weight(torch.square(pred - gt), self.temporal_weight).mean()


class L1Loss(_NormedPointwise):
    """mean(|pred - gt|) (criterion.py:76-103)."""

    def __call__(self, gt, pred):
        gt, pred = self._prep(gt, pred)
        return _apply_temporal_weight(torch.abs(pred - gt), self.temporal_weight).mean()


class GDL(nn.Module):
    """Gradient-difference loss: mean| |d_h gt| - |d_h pred| |^alpha + the same along w (criterion.py:134-204)."""

    def __init__(self, alpha=1, temporal_weight=None):
        super().__init__()
        self.alpha, self.temporal_weight = alpha, temporal_weight

    def __call__(self, gt, pred):
        lead = gt.shape[:-3]
        g, p = gt.flatten(0, -4), pred.flatten(0, -4)
        dh = (torch.abs(g[:, :, 1:, :] - g[:, :, :-1, :]) - torch.abs(p[:, :, 1:, :] - p[:, :, :-1, :])).abs()
        dw = (torch.abs(g[:, :, :, :-1] - g[:, :, :, 1:]) - torch.abs(p[:, :, :, :-1] - p[:, :, :, 1:])).abs()
        if self.alpha != 1:
            dh, dw = dh.pow(self.alpha), dw.pow(self.alpha)
        if self.temporal_weight is not None:
            assert self.temporal_weight.shape[0] == lead[1], "Mismatch between temporal_weight and predicted sequence length"
            dh = _apply_temporal_weight(dh.reshape(*lead, *dh.shape[1:]), self.temporal_weight)
            dw = _apply_temporal_weight(dw.reshape(*lead, *dw.shape[1:]), self.temporal_weight)
        return dh.mean() + dw.mean()


class BiPatchNCE(nn.Module):
    """Bidirectional patch-wise contrastive loss with stop-gradient on the negatives (criterion.py:206-259).
    Constructed for a fixed (N, T, h, w) like the reference (the positive-pair mask is a registered buffer)."""

    def __init__(self, N, T, h, w, temperature=0.07):
        super().__init__()
        mask = torch.eye(h * w).long().unsqueeze(0).repeat(N * T, 1, 1).requires_grad_(False)
        self.register_buffer("mask", mask)
        self.temperature = temperature

    def forward(self, gt_f, pred_f):
        N, T, C, h, w = gt_f.shape
        g = gt_f.permute(0, 1, 3, 4, 2).reshape(N * T, h * w, C)
        p = pred_f.permute(0, 1, 3, 4, 2).reshape(N * T, h * w, C)
        pos = self.mask.to(g.dtype)
        neg = 1.0 - pos
        s1 = (torch.matmul(g, p.transpose(1, 2)) * pos + torch.matmul(g, p.detach().transpose(1, 2)) * neg) / self.temperature
        s2 = (torch.matmul(p, g.transpose(1, 2)) * pos + torch.matmul(p, g.detach().transpose(1, 2)) * neg) / self.temperature
        target = torch.arange(h * w, device=g.device).repeat(N * T)
        return 0.5 * (F.cross_entropy(s1.flatten(0, 1), target) + F.cross_entropy(s2.flatten(0, 1), target))


class GANLoss(nn.Module):
    """vanilla / lsgan / wgangp objectives with label tensors expanded to the prediction (criterion.py:15-74)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode == "lsgan":
            self.loss = nn.MSELoss()
        elif gan_mode == "vanilla":
            self.loss = nn.BCEWithLogitsLoss()
        elif gan_mode == "wgangp":
            self.loss = None
        else:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode == "wgangp":
            return -prediction.mean() if target_is_real else prediction.mean()
        label = self.real_label if target_is_real else self.fake_label
        return self.loss(prediction, label.expand_as(prediction))
