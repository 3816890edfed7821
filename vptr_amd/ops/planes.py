"""P16 weight-plane stores: one image pair (W, W^T) per Linear-shaped weight, rebuilt by one launch per optimizer step."""
import bisect

import torch

from .._lib import WPlaneEntry, check, lib, ptr, stream
from .core import config
from .grads import _arm_grad_arena, _grad_arenas, _register_grad_arena, flat_grad_for


class WeightPlanes:
    """P16 images of a set of nn.Linear-shaped weights ([N, K] views with N, K multiples of 16), rebuilt by ONE launch
    (vptr_weight_planes) after every optimizer step: Wp [N, K] for the forward GEMMs and WT [K, N] for the input-gradient GEMMs."""

    def __init__(self, weights):
        self.weights = [w for w in weights]
        dev = self.weights[0].device
        total = sum(w.shape[0] * w.shape[1] for w in self.weights)
        self.wp = torch.empty(total, device=dev, dtype=torch.float32)
        self.wt = torch.empty(total, device=dev, dtype=torch.float32)
        ents = (WPlaneEntry * len(self.weights))()
        starts, off, tiles = [], 0, 0
        self.index = []   # (ptr, nbytes, offset, N, K, weight tensor)
        for i, w in enumerate(self.weights):
            N, K = w.shape
            if N % 16 or K % 16 or w.stride(1) != 1:
                raise RuntimeError("WeightPlanes: weight %d of shape %s is not P16-eligible" % (i, tuple(w.shape)))
            e = ents[i]
            e.W, e.ldw, e.N, e.K = w.data_ptr(), w.stride(0), N, K
            e.Wp = self.wp.data_ptr() + off * 4
            e.WT = self.wt.data_ptr() + off * 4
            self.index.append((w.data_ptr(), N * w.stride(0) * 4, off, N, K, w))
            starts.append(tiles)
            tiles += ((N + 31) // 32) * ((K + 31) // 32)
            off += N * K
        import struct
        self.table = torch.frombuffer(bytearray(bytes(ents)), dtype=torch.uint8).to(dev)
        self.starts = torch.frombuffer(bytearray(struct.pack("%di" % len(starts), *starts)), dtype=torch.uint8).to(dev)
        self.tiles, self.versions, self._views = tiles, None, {}
        self.index.sort(key=lambda t: t[0])
        self.bases = [t[0] for t in self.index]
        self.refresh()

    def refresh(self):
        check(lib.vptr_weight_planes(ptr(self.table), ptr(self.starts), len(self.weights), self.tiles, stream()), "vptr_weight_planes")
        self.versions = [t[5]._version for t in self.index]
        self.dirty = False

    grad_arena = None
    dirty = False   # set by invalidate_weight_planes(): a write torch's version counters cannot see (.data, slab writes, c10d collectives)

    def stale(self):
        """True when a registered weight changed through torch (load_state_dict, an external optimizer) since the last refresh, or
        when somebody declared the planes invalid (invalidate_weight_planes)"""
        return self.dirty or any(v != t[5]._version for v, t in zip(self.versions, self.index))

    def lookup(self, W):
        """(Wp, ld, WT, ld) for W = a registered weight or a whole-row slice of one, or None; stale planes (the weight changed
        through torch since the last refresh) are rebuilt first"""
        p = W.data_ptr()
        i = bisect.bisect_right(self.bases, p) - 1
        if i < 0:
            return None
        base, nbytes, off, N, K, w = self.index[i]
        if not (base <= p < base + nbytes):
            return None
        if w.stride(0) != K or W.shape[1] != K or W.stride(0) != K or (p - base) % (K * 4):
            return None
        if self.dirty or self.versions[i] != w._version:
            self.refresh()
        r0, n = (p - base) // (K * 4), W.shape[0]
        if r0 % 16 or n % 16 or r0 + n > N:
            return None
        key = (i, r0, n)
        hit = self._views.get(key)
        if hit is None:   # view construction costs ~10 us of host time per call site and step otherwise
            wp = self.wp[off + r0 * K: off + (r0 + n) * K].view(n, K)
            wt = self.wt[off: off + N * K].view(K, N)[:, r0:r0 + n]
            hit = self._views[key] = (wp, K, wt, N)
        return hit


_wplane_stores = []      # weakrefs of WeightPlanes registered by the trainers (FlatAdamW slabs)
_wplane_cache = {}       # (ptr, version, N, K, ld) -> (WeightPlanes, bytes): weights outside any store (eval / tests), LRU by bytes
_WPLANE_CACHE_BYTES = 3 << 30


def register_weight_planes(store):
    import weakref
    _wplane_stores.append(weakref.ref(store))


def ensure_module_planes(module):
    """One P16 weight store per MODEL for modules used without a trainer (the reference's scripts: plain nn.Parameters stepped by
    torch.optim.AdamW): every Linear-shaped weight of `module` gets its planes from ONE vptr_weight_planes launch per optimizer step
    (WeightPlanes.lookup rebuilds the whole store when a version counter moved) instead of one launch + two table uploads per weight
    and forward (196 per K64 forward: ~15 ms of host time, tools/dropin_prof.py).  Called at the top of VPTRFormerNAR / FAR.forward;
    a no-op when a trainer's store (FlatAdamW) already covers the module's weights or when the store is current."""
    if not config.use_p16:
        return
    st = module.__dict__.get("_vptr_planes")
    if st is not None and st.grad_arena is not None and torch.is_grad_enabled():
        _arm_grad_arena(st.grad_arena)
    first = next((p for p in module.parameters() if p.dim() == 2 and p.shape[0] % 16 == 0 and p.shape[1] % 16 == 0 and p.is_contiguous()), None)
    if first is None or not first.is_cuda:
        return
    if st is not None:
        if st.sentinel == (first.data_ptr(), first.shape):
            return
        module.__dict__["_vptr_planes"] = None      # the parameters moved (.to(), a flat slab took them over): rebuild or defer
        if st.grad_arena is not None:
            for k in [k for k, e in _grad_arenas.items() if e[1]() is st.grad_arena]:
                del _grad_arenas[k]
        _wplane_stores[:] = [r for r in _wplane_stores if r() is not None and r() is not st]
    for ref in _wplane_stores:
        other = ref()
        if other is not None and other.lookup(first.detach()) is not None:
            return                                   # a trainer's store serves these weights
    lin = linear_weights_of(module.parameters())
    if not lin:
        return
    with torch.no_grad():
        st = WeightPlanes(lin)
    st.sentinel = (first.data_ptr(), first.shape)
    module.__dict__["_vptr_planes"] = st             # not a registered buffer / submodule: never in state_dict
    register_weight_planes(st)
    st.grad_arena = None
    if flat_grad_for(first.detach()) is None and not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        st.grad_arena = _register_grad_arena(module)   # kept alive by the store; entries of dead parameters are replaced on re-registration
        if st.grad_arena is not None and torch.is_grad_enabled():
            _arm_grad_arena(st.grad_arena)


def invalidate_weight_planes():
    """Declare every cached P16 weight image stale.  The images are keyed on torch's tensor version counters, which miss writes
    through `param.data`, direct writes to an optimizer slab and the in-place c10d collectives (dist.broadcast bumps no version):
    vptr_amd.parallel.broadcast_module / _broadcast_any and FlatAdamW.load_state_dict call this; so must any other code that
    rewrites weights behind autograd's back.  The next GEMM that needs a weight's planes rebuilds them (one launch per store)."""
    for ref in list(_wplane_stores):
        st = ref()
        if st is None:
            _wplane_stores.remove(ref)
        else:
            st.dirty = True
    _wplane_cache.clear()


def weight_planes_for(W):
    """P16 planes (Wp [N,K], ld, WT [K,N] view, ld) of a Linear-shaped weight: from a trainer's store, else from a small cache"""
    for ref in list(_wplane_stores):
        st = ref()
        if st is None:
            _wplane_stores.remove(ref)
            continue
        hit = st.lookup(W)
        if hit is not None:
            return hit
    key = (W.data_ptr(), W._version, W.shape[0], W.shape[1], W.stride(0))
    hit = _wplane_cache.get(key)
    if hit is None:
        with torch.no_grad():
            st = WeightPlanes([W.detach()])
        nbytes = 8 * W.shape[0] * W.shape[1]
        tot = nbytes + sum(v[1] for v in _wplane_cache.values())
        for k in list(_wplane_cache):      # insertion order = least recently built first
            if tot <= _WPLANE_CACHE_BYTES:
                break
            tot -= _wplane_cache.pop(k)[1]
        hit = _wplane_cache[key] = (st, nbytes)
    N, K = W.shape
    return hit[0].wp.view(N, K), K, hit[0].wt.view(K, N), N


def linear_weights_of(params):
    """the nn.Linear-shaped members of a parameter list ([N, K] or 1x1-conv [N, K, 1, 1]; N, K multiples of 16) as [N, K] views"""
    out = []
    for p in params:
        if p.dim() == 4 and p.shape[2] == 1 and p.shape[3] == 1 and p.is_contiguous():
            w = p.detach().view(p.shape[0], p.shape[1])
        elif p.dim() == 2 and p.is_contiguous():
            w = p.detach()
        else:
            continue
        if w.shape[0] % 16 == 0 and w.shape[1] % 16 == 0:
            out.append(w)
    return out
