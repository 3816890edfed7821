"""Evaluation metrics of the reference (`utils/metrics.py:12-106`): PSNR, summed-squared-error score and SSIM with an
11x11 Gaussian window (sigma 1.5, zero padding), as device-side torch ops like the losses.

The Gaussian filtering is written as two banded matrix products, blur(X) = G_H . X . G_W^T with
G[i, j] = g[j - i + r] for |j - i| <= r: exactly the zero-padded separable 2-D convolution the reference computes with a
grouped `F.conv2d`, without a convolution library call.
"""
import math

import torch


def PSNR(x, y, data_range=1.0):
    """average PSNR of two image batches (N, C, H, W) -> float (utils/metrics.py:12-28)"""
    x = x / float(data_range)
    y = y / float(data_range)
    mse = torch.mean((x - y) ** 2, dim=(1, 2, 3))
    return torch.mean(-10.0 * torch.log10(mse + 1e-8)).item()


def MSEScore(x, y):
    """batch mean of the per-image SUM of squared errors (utils/metrics.py:30-39)"""
    return torch.mean(torch.sum((x - y) ** 2, dim=(1, 2, 3))).item()


def _band(n, g, device, dtype):
    r = g.numel() // 2
    idx = torch.arange(n, device=device)
    off = idx[None, :] - idx[:, None] + r                  # j - i + r
    ok = (off >= 0) & (off < g.numel())
    return torch.where(ok, g[off.clamp(0, g.numel() - 1)], torch.zeros((), device=device, dtype=dtype))


class SSIM(torch.nn.Module):
    """utils/metrics.py:42-99: window 11, sigma 1.5, C1 = 0.01^2, C2 = 0.03^2, per-channel statistics."""

    def __init__(self, window_size=11, size_average=True):
        super().__init__()
        self.window_size, self.size_average = window_size, size_average
        g = torch.tensor([math.exp(-(i - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for i in range(window_size)])
        self.register_buffer("g", g / g.sum())

    def _blur(self, x):
        H, W = x.shape[-2:]
        g = self.g.to(device=x.device, dtype=x.dtype)
        return _band(H, g, x.device, x.dtype) @ x @ _band(W, g, x.device, x.dtype).t()

    def forward(self, img1, img2):
        mu1, mu2 = self._blur(img1), self._blur(img2)
        mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
        s1 = self._blur(img1 * img1) - mu1_sq
        s2 = self._blur(img2 * img2) - mu2_sq
        s12 = self._blur(img1 * img2) - mu12
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((2 * mu12 + c1) * (2 * s12 + c2)) / ((mu1_sq + mu2_sq + c1) * (s1 + s2 + c2))
        return m.mean() if self.size_average else m.mean(dim=(1, 2, 3))
