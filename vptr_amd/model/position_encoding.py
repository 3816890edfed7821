"""Sine position tables of VPTR, computed once at construction into module buffers.

Same values as the reference's PositionEmbeddding1D/2D/3D (utils/position_encoding.py:29-49, 67-93, 117-161,
normalize=False): positions start at 1, channel i uses temperature 10000^(2*floor(i/2)/E), even channels sin,
odd channels cos; the 2-D table concatenates a y-half and an x-half, the 3-D table t/y/x thirds.
Host-side, init-time only.
"""
import torch


def _interleaved_sincos(pos, E):
    i = torch.arange(E, dtype=torch.float32)
    freq = 10000.0 ** (2 * torch.div(i, 2, rounding_mode="floor") / E)
    ang = pos[..., None] / freq
    out = torch.empty_like(ang)
    out[..., 0::2] = ang[..., 0::2].sin()
    out[..., 1::2] = ang[..., 1::2].cos()
    return out


def temporal_table(T, E):
    """(T, E) -- buffer `temporal_pos`."""
    return _interleaved_sincos(torch.arange(1, T + 1, dtype=torch.float32), E)


def window_table(E, ws):
    """(ws, ws, E) -- buffer `lw_pos`."""
    if E % 2:
        raise AssertionError("Embedding size should be even number")
    y = torch.arange(1, ws + 1, dtype=torch.float32)[:, None].expand(ws, ws)
    x = torch.arange(1, ws + 1, dtype=torch.float32)[None, :].expand(ws, ws)
    return torch.cat([_interleaved_sincos(y, E // 2), _interleaved_sincos(x, E // 2)], dim=-1)


def temporal_window_table(E, T, ws):
    """(T, ws, ws, E) -- buffer `Tlw_pos`."""
    if E % 3:
        raise AssertionError("Embedding size should be divisible by 3")
    t = torch.arange(1, T + 1, dtype=torch.float32)[:, None, None].expand(T, ws, ws)
    y = torch.arange(1, ws + 1, dtype=torch.float32)[None, :, None].expand(T, ws, ws)
    x = torch.arange(1, ws + 1, dtype=torch.float32)[None, None, :].expand(T, ws, ws)
    return torch.cat([_interleaved_sincos(t, E // 3), _interleaved_sincos(y, E // 3), _interleaved_sincos(x, E // 3)], dim=-1)


def relative_position_index(ws):
    """(ws*ws, ws*ws) int64 index into the (2ws-1)^2 bias table (MultiHeadAttentionRPE.py:373-387)."""
    ys, xs = torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing="ij")
    ys, xs = ys.flatten(), xs.flatten()
    return (ys[:, None] - ys[None, :] + ws - 1) * (2 * ws - 1) + (xs[:, None] - xs[None, :] + ws - 1)
