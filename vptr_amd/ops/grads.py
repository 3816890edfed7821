"""Where parameter gradients land: flat optimizer slabs, the per-model gradient arena of script-style use, and the autograd hand-off for everything else."""

import torch

from .core import _dev_key, config


_flat_slabs = []  # (param_base_ptr, nbytes, grad_slab) registered by vptr_amd.train.FlatAdamW


def register_flat_slab(param_slab, grad_slab):
    """Parameters that live inside `param_slab` have their gradient at the same offset of `grad_slab`: backward kernels
    then accumulate weight gradients straight into the slab (fp32 atomics) instead of materialising a zero-filled
    temporary that autograd adds to `.grad` (2 extra launches and 3 passes over every parameter per step)."""
    import weakref
    _flat_slabs.append((param_slab.data_ptr(), param_slab.numel() * 4, weakref.ref(param_slab), weakref.ref(grad_slab)))


def unregister_flat_slabs():
    del _flat_slabs[:]


def unregister_flat_slab(param_slab):
    """drop the registration of ONE slab (FlatAdamW.close / __del__) and of slabs that no longer exist -- by identity, not by address:
    a later optimizer's slab may have been given the address of a collected one"""
    _flat_slabs[:] = [e for e in _flat_slabs if e[2]() is not None and e[2]() is not param_slab]


def flat_grad_for(t):
    """Gradient-slab view for a parameter tensor (or a contiguous slice of one) that lives in a registered slab."""
    if t is None or not _flat_slabs or not t.is_contiguous():
        return None
    p = t.data_ptr()
    for base, nbytes, pref, gref in _flat_slabs:
        if base <= p < base + nbytes:
            pslab, gslab = pref(), gref()
            if pslab is None or gslab is None or pslab.data_ptr() != base:
                continue  # stale registration (the optimizer that owned the slab is gone)
            off = (p - base) // 4
            return gslab[off:off + t.numel()].view(t.shape)
    return None


# ---- gradient arena of a model used WITHOUT a trainer (the reference's scripts: zero_grad(set_to_none=True) every iteration) ----------
# After set_to_none every parameter's first gradient of the next backward pass needs a zero-filled `.grad` to accumulate into: as
# torch.zeros_like per parameter that is 664 allocations + 664 fill launches per K64 iteration (tools/dropin_prof.py).  A model that
# ran ensure_module_planes() owns ONE flat fp32 buffer instead: a forward pass that finds every `.grad` None zero-fills it with one
# launch, and in backward a parameter's `.grad` becomes a view of its (still zero) range -- contiguous, own shape: torch.optim and
# clip_grad_norm_ see ordinary tensors.  A range is handed out once per fill; anything else (a `.grad` set to None by hand between two
# backward passes, ...) falls back to a fresh zeros_like.  A `.grad` tensor (or a view of one) somebody KEPT from the previous iteration
# is never overwritten: the fill sees the extra reference on the buffer's storage and takes a new buffer for this iteration, the kept
# tensors keep the old one alive (stock-autograd semantics; config.loose_grad_arena = False restores per-parameter tensors).
_grad_arenas = {}     # id(param) -> (weakref(param), weakref(arena), offset); the arena itself is owned by the model's weight-plane store


def _storage_refs(t):
    """number of live tensors (views included) that share t's storage, + the temporary wrapper of this query; a very large number
    when the runtime cannot tell (the arena then always takes a fresh buffer: correct, merely slower)"""
    f = getattr(torch._C, "_storage_Use_Count", None)
    if f is None:
        return 1 << 30
    return int(f(t.untyped_storage()._cdata))


class _GradArena:
    """one flat gradient buffer per model; lives as long as the model's `_vptr_planes` store does (module.__dict__)"""

    def __init__(self, params):
        import weakref
        self.buf = torch.empty(sum(p.numel() for p in params), device=params[0].device, dtype=torch.float32)
        self.base_refs = _storage_refs(self.buf)     # the buffer alone: anything above it at arm time is a gradient somebody kept
        self.clean = False
        self.handed = set()
        self.params = [weakref.ref(p) for p in params]


def _register_grad_arena(module):
    import weakref
    params = [p for p in module.parameters() if p.requires_grad and p.dtype == torch.float32 and p.is_contiguous()]
    if not params or not config.loose_grad_arena:
        return None
    arena = _GradArena(params)
    aref = weakref.ref(arena)
    for k in [k for k, e in _grad_arenas.items() if e[0]() is None or e[1]() is None]:   # entries of models that are gone
        del _grad_arenas[k]
    off = 0
    for p in params:
        _grad_arenas[id(p)] = (weakref.ref(p), aref, off)
        off += p.numel()
    return arena


def _arm_grad_arena(arena):
    """forward pass: with every gradient None (the iteration began with zero_grad(set_to_none=True)) the arena is zero-filled"""
    for r in arena.params:
        p = r()
        if p is not None and p.grad is not None:
            return
    if _storage_refs(arena.buf) > arena.base_refs:
        # somebody KEPT a gradient of the previous iteration (a stashed `p.grad` or a view of it: per-task gradients, logging, manual
        # accumulation): stock autograd would never touch that tensor again, so it keeps the old buffer and this iteration gets a new one
        arena.buf = torch.empty_like(arena.buf)
        arena.base_refs = _storage_refs(arena.buf)
    arena.buf.zero_()
    arena.handed.clear()
    arena.clean = True


def _arena_grad_for(base):
    ent = _grad_arenas.get(id(base))
    arena = ent[1]() if ent is not None and ent[0]() is base else None
    if arena is None or not arena.clean or id(base) in arena.handed or arena.buf.device != base.device:
        return torch.zeros_like(base)
    arena.handed.add(id(base))
    return arena.buf[ent[2]:ent[2] + base.numel()].view(base.shape)


def _engine_accumulates_into(leaf):
    """True when the running backward pass is one that ACCUMULATES into `leaf.grad` (`loss.backward()`, or `backward(inputs=[...
    leaf ...])`), False under `torch.autograd.grad(...)` / `backward(inputs=<others>)`: there the engine captures or drops the
    gradient, and writing `.grad` behind its back would hand the caller None and pollute `.grad`.  The engine is
    asked through `torch._C._will_engine_execute_node` on the leaf's AccumulateGrad node: with no explicit inputs every node of the
    graph executes (True); with inputs it answers for this node, and raises for a leaf that `autograd.grad` captures."""
    will = getattr(torch._C, "_will_engine_execute_node", None)
    if will is None:      # a torch build without the query: take the conservative autograd hand-off
        return False
    # the AccumulateGrad node of a leaf is unique, but it OWNS its variable: a process-wide cache of nodes would pin every parameter
    # (with its .grad and the arena range it views) for the life of the process.  The cache therefore lives for ONE
    # backward pass: keyed by the engine's graph-task id, emptied by an end-of-backward callback (and by the next pass, should the
    # callback of a failed pass never have run).
    task = torch._C._current_graph_task_id()
    if _acc_nodes["task"] != task:
        _acc_nodes["nodes"].clear()
        _acc_nodes["task"] = task
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_drop_acc_nodes)
        except RuntimeError:   # not inside a backward pass
            pass
    acc = _acc_nodes["nodes"].get(id(leaf))
    if acc is None:
        with torch.enable_grad():
            acc = leaf.view_as(leaf).grad_fn.next_functions[0][0]
        _acc_nodes["nodes"][id(leaf)] = acc
    try:
        return bool(will(acc))
    except (RuntimeError, TypeError):
        return False


_acc_nodes = {"task": None, "nodes": {}}    # graph-task id -> {id(leaf): its AccumulateGrad node}, for the running backward pass only


def _drop_acc_nodes():
    _acc_nodes["nodes"].clear()
    _acc_nodes["task"] = None


def _loose_grad_for(t):
    """Gradient destination for a parameter OUTSIDE any flat slab (the reference's scripts: plain nn.Parameters, torch.optim.AdamW,
    zero_grad(set_to_none=True)): a view into the `.grad` of the leaf parameter that `t` is (or is a contiguous view of -- a row
    block of in_proj_weight, a 1x1 conv weight seen as [N, K]), created zero-filled if it is None.  The weight gradient can then
    join the grouped end-of-backward launch exactly like a slab-backed one, instead of running as a launch of its own (12-60 tiles
    with a 10 240-token K loop on 256 CUs: 196 of those made the script-style step 2.7x slower than NARTrainer's).  Returns None
    -- the caller then hands a fresh tensor to autograd -- outside a backward pass, under `torch.autograd.grad` / `backward(inputs=
    ...)` without this parameter / `create_graph=True` (the engine is not accumulating into `.grad` there), in a torch.distributed job
    (DDP's reducer must see gradients arrive through AccumulateGrad hooks) and for parameters with hooks."""
    if not (config.group_wgrads and config.group_loose_wgrads) or t is None or not t.is_contiguous():
        return None
    if torch._C._current_graph_task_id() < 0:
        return None
    if torch.is_grad_enabled():      # backward(create_graph=True): the gradient must stay a graph output, not an in-place sum
        return None
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return None
    base = t if t.is_leaf else t._base
    if base is None or not base.is_leaf or not base.requires_grad or not base.is_contiguous() or base.dtype != torch.float32:
        return None
    if base._backward_hooks or getattr(base, "_post_accumulate_grad_hooks", None):
        return None
    if not _engine_accumulates_into(base):
        return None
    off = (t.data_ptr() - base.data_ptr()) // 4
    if off < 0 or off + t.numel() > base.numel():
        return None
    if base.grad is None:
        base.grad = _arena_grad_for(base)
    elif not base.grad.is_contiguous() or base.grad.dtype != torch.float32:
        return None
    return base.grad.view(-1)[off:off + t.numel()].view(t.shape)


def grad_dest_for(t):
    """where a parameter's gradient is accumulated in place: its range of a registered flat gradient slab, else its own `.grad`"""
    d = flat_grad_for(t)
    return d if d is not None else _loose_grad_for(t)


_bw_blocks = {}     # (device key, stream handle) -> [graph-task id, current zeroed block, floats used, floats of the next block]
_BW_BLOCK = 16 << 20   # largest block: 64 MB
_BW_FIRST = 1 << 18    # first block of a backward pass: 1 MB (doubling from there: a small autograd.grad call does not zero 64 MB)


def _drop_bw_blocks():
    _bw_blocks.clear()    # end of the backward pass: gradients somebody keeps keep their blocks alive, nothing else does


def _bw_zeros(shape, device):
    """Zero-filled fp32 tensor for a parameter gradient that is handed to autograd (torch.distributed jobs -- DDP's reducer must see every
    gradient arrive through its AccumulateGrad hook --, torch.autograd.grad, parameters with hooks): a slice of a block zeroed by ONE fill
    instead of one allocation + fill launch per parameter (a stock-DDP K64 iteration spent 1100 launches / 3.9 ms of GPU time on those
    fills).  Blocks grow from 1 MB to 64 MB within one backward pass, are keyed by (device, STREAM) -- a slice is only ever handed to work
    on the stream its fill ran on --, are never reused or re-zeroed (a gradient somebody keeps keeps its block alive) and are dropped by
    an end-of-backward callback; the next backward pass (another graph-task id) starts new ones."""
    n = 1
    for d in shape:
        n *= int(d)
    task = torch._C._current_graph_task_id()
    if task < 0 or n == 0 or n > _BW_BLOCK:
        return torch.zeros(tuple(shape), device=device, dtype=torch.float32)
    key = (_dev_key(device), torch.cuda.current_stream(device).cuda_stream)
    st = _bw_blocks.get(key)
    if st is None or st[0] != task:
        if not any(e[0] == task for e in _bw_blocks.values()):
            _bw_blocks.clear()
            try:
                torch.autograd.Variable._execution_engine.queue_callback(_drop_bw_blocks)
            except RuntimeError:
                pass
        st = _bw_blocks[key] = [task, None, 0, _BW_FIRST]
    n_al = (n + 63) // 64 * 64     # 256-byte aligned slices
    if st[1] is None or st[2] + n_al > st[1].numel():
        size = max(st[3], n_al)
        st[1] = torch.zeros(size, device=device, dtype=torch.float32)
        st[2] = 0
        st[3] = min(2 * size, _BW_BLOCK)
    v = st[1][st[2]:st[2] + n].view(tuple(shape))
    st[2] += n_al
    return v
