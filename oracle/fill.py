"""TEST INFRASTRUCTURE -- deterministic weight/input fill shared by the golden generator and the tests.

Weights are drawn from np.random.RandomState (a frozen bit-stream, stable
across numpy versions) in sorted-key order, so that full-size models never
have to be committed: a fixture stores (seed, key->shape list, outputs).
Scales are chosen to keep every code path numerically non-trivial (LN/BN
affines away from 1/0, BN running stats away from 0/1, RPE table and frame
queries large enough to move the softmax).
"""
import numpy as np
import torch


def _scale_for(key, shape):
    k = key.split(".")[-1]
    if k == "num_batches_tracked" or k == "relative_position_index":
        return None
    if k == "running_var":
        return ("absshift", 0.5, 0.5)
    if k == "running_mean":
        return ("normal", 0.0, 0.2)
    if k == "relative_position_bias_table":
        return ("normal", 0.0, 0.5)
    if key.endswith("frame_queries"):
        return ("normal", 0.0, 0.5)
    if key in ("temporal_pos", "lw_pos", "Tlw_pos"):
        return None  # deterministic sine tables: keep the module's own values
    is_norm = any(s in key for s in (".norm", "norm1", "norm2", "norm3", "norm4", "norm5", "norm6"))
    if len(shape) == 1 or (is_norm and len(shape) == 3):
        # LayerNorm / BatchNorm affine or a bias vector
        if k == "weight":
            return ("normal", 1.0, 0.1)
        return ("normal", 0.0, 0.1)
    if is_norm:
        return ("normal", 0.0, 0.1)
    # matrices / conv kernels: ~ 1/sqrt(fan_in) so activations stay O(1)
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
    if k == "in_proj_weight" or "proj" in key or "linear" in key or "fc" in key or "NCE" in key:
        return ("normal", 0.0, 1.0 / np.sqrt(max(fan_in, 1)))
    return ("normal", 0.0, 1.0 / np.sqrt(max(fan_in, 1)))


def fill_state(template, seed):
    """template: list of (key, shape, dtype_str) or a state_dict; returns {key: tensor} (float32 for float entries).

    Non-float entries and the sine tables are NOT produced (callers keep their own).
    BN encoder blocks inside an AE: BN affine keys are '<idx>.weight' with 1-D shape -> handled by the 1-D rule.
    """
    if isinstance(template, dict):
        template = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in template.items()]
    rs = np.random.RandomState(seed)
    out = {}
    for key, shape, dt in sorted(template, key=lambda t: t[0]):
        if not dt.startswith("float"):
            continue
        rule = _scale_for(key, shape)
        if rule is None:
            continue
        kind, mu, sd = rule
        n = rs.standard_normal(size=tuple(shape)).astype(np.float32)
        if kind == "absshift":
            v = mu + sd * np.abs(n)
        else:
            v = mu + sd * n
        out[key] = torch.from_numpy(np.ascontiguousarray(v.astype(np.float32)))
    return out


def apply_fill(module_or_sd, seed):
    """Loads a deterministic fill into an nn.Module (strict=False on buffers we do not touch)."""
    sd = module_or_sd.state_dict() if hasattr(module_or_sd, "state_dict") else module_or_sd
    new = fill_state(sd, seed)
    merged = {k: (new[k].to(v.dtype) if k in new else v) for k, v in sd.items()}
    if hasattr(module_or_sd, "load_state_dict"):
        module_or_sd.load_state_dict(merged)
    return merged


def rand_input(shape, seed, lo=0.0, hi=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(lo, hi, size=shape).astype(np.float32))


_BAIR_MEAN, _BAIR_STD = (0.61749697, 0.6050092, 0.52180636), (2.1824553, 2.1553133, 1.9115673)


def clip_input(shape, seed, norm):
    """synthetic clip (N, T, C, H, W) in the value range the reference's data pipeline hands the model (utils/dataset.py:19-58):
    "kth" = VidNormalize(0.6013795, 2.7570653) on [0, 1) frames, "raw" = MovingMNIST (VidToTensor only: un-normalised [0, 1)),
    "bair" = the per-channel VidNormalize of the BAIR set (:47)"""
    x = rand_input(shape, seed)
    if norm == "kth":
        return (x - 0.6013795) / 2.7570653
    if norm == "bair":
        m = torch.tensor(_BAIR_MEAN).view(1, 1, 3, 1, 1)
        sd = torch.tensor(_BAIR_STD).view(1, 1, 3, 1, 1)
        return (x - m) / sd
    if norm != "raw":
        raise ValueError(norm)
    return x


def rand_normal(shape, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((scale * rs.standard_normal(size=shape)).astype(np.float32))


def sample_index(numel, count, seed):
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.randint(0, numel, size=min(count, numel)).astype(np.int64))


def digest(t, count=2048, seed=7):
    """(l2 norm, sampled values) of a tensor, for full-size fixtures."""
    f = t.detach().reshape(-1).to(torch.float64)
    idx = sample_index(f.numel(), count, seed).to(f.device)
    return float(f.norm()), f[idx].to(torch.float32).cpu().numpy()
