"""Launch planning of the deferred, grouped weight gradients (and of the deferred partial-sum reductions): recording, the end-of-backward flush, tile-row tuning, pinned table uploads."""
import ctypes
import os

import torch

from .. import _lib
from .._lib import GemmDesc, check, lib, ptr, stream
from .core import A_P16T, B_P16T, config, profiling, to_p16


# ---- deferred, grouped weight gradients ---------------------------------------------------------------------------------
# dW = dY^T . X of one nn.Linear is 12-60 output tiles with K = all tokens: alone it cannot fill 256 CUs without ~30
# K-splits (each paying a prologue and a 90 KB atomic epilogue; measured 80 TFLOP/s).  When the weight's gradient lives
# in a registered flat slab (so nothing has to be handed back to autograd), the call is only recorded here and the whole
# backward pass's weight gradients run as ONE vptr_gemm_grouped launch, queued on the autograd engine's end-of-backward
# callback: every tile then runs the full K loop and writes once.
_wgrad_q = []


def defer_wgrad(g, x, dW, N, K, M, db=None, alpha=1.0, p16=False):
    """record dW[N,K] += g[M,N]^T . x[M,K] (dW, and db if given, must be views of a flat gradient slab); with db the bias
    gradient db[N] += column sums of g rides on the same launch (vptr_gemm_desc::a_rowsum).  p16: g and x are P16 tensors."""
    _wgrad_q.append((g, x, dW, N, K, M, config.gemm_precision, db, float(alpha), bool(p16)))
    if config.wgrad_async and not _wgrad_hold[0]:
        _wgrad_side["tiles"] += ((N + 127) // 128) * ((K + 175) // 176)
        if _wgrad_side["tiles"] >= config.wgrad_chunk_tiles:
            _flush_wgrads_side()
    # one end-of-backward callback per recorded call: flush_wgrads is idempotent, and registering every time stays correct
    # when an earlier backward died before its callbacks ran (a "callback already queued" flag would then be stale)
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_auto_flush_wgrads)
    except RuntimeError:  # not inside a backward pass: the caller flushes explicitly
        pass


def take_wgrads():
    """hand the recorded (not yet launched) weight gradients to the caller and clear the queue (NARTrainer.capture_front keeps the
    records of the captured backward pass: their operands have static addresses in the graph's pool)"""
    items = list(_wgrad_q)
    del _wgrad_q[:]
    return items


def requeue_wgrads(items):
    """put records obtained from take_wgrads() back (after a replay of the graph that produces their operands)"""
    _wgrad_q.extend(items)


def discard_wgrads():
    """drop recorded weight gradients that were never launched (a backward pass that raised); called by FlatAdamW.zero_grad"""
    del _wgrad_q[:]
    del _reduce_q[:]


# ---- deferred partial-sum reductions (parameter gradients of the LayerNorms) -----------------------------------------------
_reduce_q = []


def defer_partial_reduce(part, dst0, dst1, nparts, C):
    """record dst0[C] += sum_p part[p][0][:], dst1[C] += sum_p part[p][1][:]; every record of a backward pass is served by one
    vptr_partial_reduce launch at its end (before the grouped weight gradients: a data-parallel step sends gradient ranges out as
    soon as their weight-gradient chunk is done)."""
    _reduce_q.append((part, dst0, dst1, int(nparts), int(C)))
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_auto_flush_wgrads)
    except RuntimeError:  # not inside a backward pass: the caller flushes explicitly (flush_wgrads)
        pass


def flush_partial_reduces():
    if not _reduce_q:
        return
    items = list(_reduce_q)
    del _reduce_q[:]
    # one table, one upload; entries sorted by width and launched per width class (the grid is sized for the widest entry of a launch:
    # the 528-wide LayerNorm rows must not ride on the grid of the 135 168-wide LayerNorm((F,H,W)) rows)
    items.sort(key=lambda it: it[4])
    tab = (_lib.ReduceEntry * len(items))()
    for i, (part, d0, d1, nparts, C) in enumerate(items):
        tab[i].part, tab[i].dst0, tab[i].dst1, tab[i].nparts, tab[i].C = ptr(part), ptr(d0), ptr(d1), nparts, C
    dev = items[0][0].device
    raw = _to_device_async(bytes(tab), dev)
    esz = ctypes.sizeof(_lib.ReduceEntry)
    lo = 0
    while lo < len(items):
        hi = lo
        while hi < len(items) and items[hi][4] <= 4 * items[lo][4]:
            hi += 1
        dsts = [it[k].data_ptr() for it in items[lo:hi] for k in (1, 2)]
        unique = len(set(dsts)) == len(dsts)   # a module applied twice in one forward: atomics
        check(lib.vptr_partial_reduce(ctypes.c_void_p(raw.data_ptr() + lo * esz), hi - lo, items[hi - 1][4], int(unique), stream()),
              "vptr_partial_reduce")
        lo = hi


_wgrad_hold = [False]  # set by hold_wgrads(): the end-of-backward callback leaves the queue to an explicit chunked flush


class hold_wgrads:
    """Scope in which the end-of-backward callback does NOT launch the recorded weight gradients: the data-parallel trainer
    flushes them itself in a few chunks (flush_wgrads(chunks=..., on_chunk=...)) so that the all-reduce of one chunk's
    gradient range overlaps the GEMM launch of the next."""

    def __enter__(self):
        self.prev = _wgrad_hold[0]
        _wgrad_hold[0] = True
        return self

    def __exit__(self, *exc):
        _wgrad_hold[0] = self.prev
        return False


# ---- weight gradients on a side stream, overlapped with the rest of the backward pass -------------------------------------------
# The grouped weight-gradient launch is MFMA-bound, about half of the backward pass's other kernels are HBM-bound (normalisation,
# attention cores, LayerNorm) or leave CUs idle (240-tile GEMMs): instead of one launch at the very end, the recorded problems are
# flushed in chunks of >= config.wgrad_chunk_tiles tiles onto a second HIP stream while backward keeps running on the main one.
# g and x stay alive through record_stream (the caching allocator defers their reuse until the side stream has passed them).
_wgrad_side = {"stream": None, "tiles": 0, "dirty": False}


def _flush_wgrads_side():
    items = list(_wgrad_q)
    del _wgrad_q[:]
    _wgrad_side["tiles"] = 0
    if not items:
        return
    cur = torch.cuda.current_stream()
    if _wgrad_side["stream"] is None:
        _wgrad_side["stream"] = torch.cuda.Stream()
    side = _wgrad_side["stream"]
    side.wait_stream(cur)          # every operand recorded so far has been produced on the main stream
    with torch.cuda.stream(side):
        # never the panel-synchronous persistent kernel here: it assumes all its workgroups resident and owns the device-wide barrier
        # words (include/vptr_hip.h), and a side-stream launch runs beside the main stream's backward kernels
        _launch_wgrad_group(items, allow_sync=False)
    for it in items:
        it[0].record_stream(side)
        it[1].record_stream(side)
    _wgrad_side["dirty"] = True


def join_wgrad_stream():
    """make the current stream wait for weight-gradient chunks still running on the side stream (before the optimizer reads them)"""
    if _wgrad_side["dirty"]:
        torch.cuda.current_stream().wait_stream(_wgrad_side["stream"])
        _wgrad_side["dirty"] = False


def _auto_flush_wgrads():
    if not _wgrad_hold[0]:
        flush_partial_reduces()
        if config.wgrad_async and _wgrad_q:
            _flush_wgrads_side()
        else:
            flush_wgrads()
        join_wgrad_stream()


_pin_pool = {"slots": [], "next": 0}
_pin_pool_small = {"slots": [], "next": 0}
_wgrad_tune = {}    # problem-set signature -> {"samples": {tile rows: [ms, ...]}, "pending": (rows, e0, e1) | None, "choice": rows | None, "ms": {rows: best ms}}
_graph_keepalive = []   # pinned upload sources of captured launches (must outlive every replay)
_graph_reserve = []     # pinned buffers set aside for the next capture


_upload_stats = {"count": 0, "max_bytes": 0}   # table uploads since the last reset (NARTrainer.capture sizes its reserve from a warm-up step)


def reserve_graph_staging(count=8, nbytes=1 << 18):
    """set `count` pinned staging buffers of `nbytes` aside for the host-built tables of a whole-step graph capture (pinned memory
    cannot be allocated while capturing); buffers that are too small are replaced"""
    _graph_reserve[:] = [b for b in _graph_reserve if b.numel() >= nbytes]
    while len(_graph_reserve) < count:
        _graph_reserve.append(torch.empty(nbytes, dtype=torch.uint8).pin_memory())


def _to_device_async(host_bytes, dev):
    """bytes -> uint8 device tensor through a rotating pool of pinned staging buffers with a non-blocking copy: a pageable
    `.to(device)` would block the host until every kernel enqueued so far has finished (once per step, right where the host
    should be running ahead into the optimizer and the next forward pass)."""
    n = len(host_bytes)
    _upload_stats["count"] += 1
    _upload_stats["max_bytes"] = max(_upload_stats["max_bytes"], n)
    if os.environ.get("VPTR_SYNC_UPLOAD") == "1":
        return torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8).to(dev)
    if torch.cuda.is_current_stream_capturing():
        # the copy becomes a memcpy node that reads the HOST buffer at every replay: it gets a pinned buffer of its own that is
        # never reused (the rotating pool below is rewritten by later eager launches -- replays would upload stale descriptors)
        # (pinned memory cannot be allocated while capturing: reserve_graph_staging() set buffers aside beforehand)
        for i, cand in enumerate(_graph_reserve):
            if cand.numel() >= n:
                buf = _graph_reserve.pop(i)
                break
        else:
            raise RuntimeError("graph capture: no reserved pinned staging buffer of %d bytes (ops.reserve_graph_staging)" % n)
        buf[:n].copy_(torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8))
        _graph_keepalive.append(buf)
        return buf[:n].to(dev, non_blocking=True)
    # two rotating pools: 1024 small buffers (descriptor tables of one-layer launches: a torch.distributed job makes ~200 of those per
    # backward pass, and with 16 buffers the host had to wait for the device every 8 launches -- it could never run ahead) and 16 large ones
    small = n <= 8192
    pool = _pin_pool_small if small else _pin_pool
    i = pool["next"] % (1024 if small else 16)
    pool["next"] += 1
    while len(pool["slots"]) <= i:
        pool["slots"].append([torch.empty(8192 if small else (1 << 16), dtype=torch.uint8).pin_memory(), None])
    slot = pool["slots"][i]
    if slot[1] is not None:
        slot[1].synchronize()  # the copy that last used this staging buffer (16 transfers ago) must have completed
    if slot[0].numel() < n:
        slot[0] = torch.empty(2 * n, dtype=torch.uint8).pin_memory()
    slot[0][:n].copy_(torch.frombuffer(bytearray(host_bytes), dtype=torch.uint8))
    out = slot[0][:n].to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    slot[1] = ev
    return out


_TUNE_SAMPLES = 3   # timed flushes per tile-row setting; the first of each is cold (pinned allocations, first-launch work) and not counted


def _wgrad_tune_decide(tune, final=False):
    """fix the tile rows of a problem set once both settings have _TUNE_SAMPLES timings (`final`: now, from whatever has been measured)"""
    s = tune["samples"]
    if not final and any(len(s[r]) < _TUNE_SAMPLES for r in (128, 256)):
        return
    best = {r: min(v[1:] or v) for r, v in s.items() if v}     # a setting's first flush is cold: counted only if it is all there is
    tune["choice"] = min(best, key=best.get) if best else 128
    tune["ms"] = best


def _wgrad_tune_book(tune):
    r_, e0_, e1_ = tune["pending"]
    tune["samples"][r_].append(e0_.elapsed_time(e1_))
    tune["pending"] = None
    _wgrad_tune_decide(tune)


def wgrad_tune_open():
    """True while some problem set seen so far still alternates tile-row settings (auto mode): more eager steps will time them"""
    return any(t["choice"] is None for t in _wgrad_tune.values())


def wgrad_tune_settle():
    """book every timed weight-gradient flush that is still in flight (device synchronisation) and FIX the choice of every problem set seen
    so far; the trainers call it before the last eager warm-up step in front of a graph capture (events cannot be queried while capturing,
    and the step that sizes the capture's staging reserve must already run the geometry the capture will run)"""
    if any(t["pending"] is not None for t in _wgrad_tune.values()):
        torch.cuda.synchronize()
    for t in _wgrad_tune.values():
        if t["pending"] is not None:
            _wgrad_tune_book(t)
        if t["choice"] is None:
            _wgrad_tune_decide(t, final=True)


def plan_wgrad_launches(probs, cols, p16, atomic, allow_sync, rows_mode, split_rem=False, token_split=True):
    """Pure planning step of the grouped weight-gradient flush (no tensors, no launches: tests/test_cpu.py drives it with made-up
    addresses).  probs: (A ptr, B ptr, D ptr, rowsum ptr, lda, ldb, ldd, rows, cols, tokens, alpha, transposed) per weight, pointers as
    integers, leading dimensions in floats.  Returns [(sub-problems, vouch)]: one entry per kernel launch, every sub-problem the same
    tuple + its tile rows as a 13th element where they are not 128; `vouch` = every sub-problem of the launch walks the same number of
    tokens (the panel-synchronous persistent kernel may serve it).  Rows of a problem are cut between a 256- (or 192-) row launch and
    the 128-row launch; a small group is cut into token ranges that accumulate into one destination; problems of different token
    counts go to different persistent launches."""
    subs = []
    for (ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip) in probs:
        small_group = p16 and atomic and token_split and len(probs) <= 3   # one layer's launch: token ranges on 128-row tiles (below)
        if p16 and atomic and rows_mode == 192 and not small_group and rows_ >= 384:
            # 192 x 176 tiles (three stages, one workgroup per CU): 2112 = 11 x 192 exactly, 528 = 2.75 (three tiles, the last 3/4 full,
            # against 4.125 128-row tiles); a remainder that pads a 128-row tile less than a 192-row one joins the 128-row launch
            rem = rows_ % 192
            to128 = 0      # trailing rows handed to the 128-row launch
            if flip and rp:   # the column sums of a flipped problem need a free 16-row fragment in the tile that holds its last rows
                if rem == 0:
                    to128 = 192
                elif 192 - rem < 16:
                    to128 = rem
            elif rem and (192 - rem) > ((rem + 127) // 128) * 128 - rem:
                to128 = rem
            if to128:
                full = rows_ - to128
                subs.append((ap, bp, dp, 0 if flip else rp, lda, ldb, ldd, full, cols_, M, alpha, flip, 192))
                subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                             lda, ldb, ldd, to128, cols_, M, alpha, flip, 128))
            else:
                subs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip, 192))
            continue
        if p16 and atomic and rows_mode == 256 and not small_group and rows_ >= 1024 and (rows_ // 256) * 256 >= 0.85 * rows_:
            # tall problems on 256 x 176 tiles (1.47x the flops per staged byte, one workgroup per CU): the multiple-of-256 part
            # goes to the 256-row launch, the rest of the rows stays a 128-row problem (and keeps the bias gradient of a flipped one)
            full = (rows_ // 256) * 256
            rem256 = rows_ - full
            subs.append((ap, bp, dp, 0 if (flip and rem256) else rp, lda, ldb, ldd, full, cols_, M, alpha, flip, 256))
            if rem256:
                subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                             lda, ldb, ldd, rem256, cols_, M, alpha, flip, 128))
            continue
        rem = rows_ % 128
        if p16 and split_rem and rem and rows_ > 128:
            # the partly filled last row tile of every problem becomes a problem of its own, launched after all full tiles: full
            # tiles then all take the same time, so the tiles that share an operand panel stay in step (and in one L2), instead
            # of being scattered by the short tiles that used to finish early between them
            full = rows_ - rem
            subs.append((ap, bp, dp, 0 if flip else rp, lda, ldb, ldd, full, cols_, M, alpha, flip))
            subs.append((ap + full * 4, bp, dp + (full * 4 if flip else full * ldd * 4), (rp + (0 if flip else full * 4)) if rp else 0,
                         lda, ldb, ldd, rem, cols_, M, alpha, flip))
        else:
            subs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip))

    if p16 and split_rem:
        subs.sort(key=lambda t: (0 if t[7] >= 128 else 1, -t[7] * t[8], t[1], t[2]))   # full-tile problems first (largest first), remainders last
    def trows(sub):
        return sub[12] if len(sub) > 12 else 128

    def tiles_of(sub):
        return ((sub[7] + trows(sub) - 1) // trows(sub)) * ((sub[8] + cols - 1) // cols)
    # the panel-synchronous persistent launch needs ONE token count per launch (its barrier counts K-blocks): problems are classed by
    # token count, every class of >= 1024 tiles gets a launch of its own, the rest share a plain launch.  K64 / BAIR: one class.  KTH128
    # 10 -> 40: encoder layers (10 frames of tokens) and decoder layers (40 frames) = two persistent launches instead of one plain launch
    # that re-fetched every operand panel 5x over the fabric (80 GB per launch, profiles/r05_cfg5_kernel_stats.md).
    if p16 and atomic and token_split:   # (several adders per destination: not bit-reproducible)
        # a SMALL group (one layer's weight: the launches a torch.distributed job / torch.autograd.grad make, where every gradient must be
        # complete when its autograd node returns) is 15 - 60 tiles with a K loop over every token: 6 - 25 % of the CUs for the whole
        # launch.  Its problems are cut into token ranges that accumulate into the same (zero-initialised) destination, enough of them
        # to put ~2 workgroups on every CU -- what ops.convt_weight_grads does for the decoder (stock-DDP step: see bench.py
        # other_configs.drop_in_ddp_single_iter)
        tot = sum(tiles_of(x) for x in subs)
        if 0 < tot < 384:
            want = (512 + tot - 1) // tot      # ~2 workgroups per CU; every range >= 1024 tokens (each range pays a full atomic epilogue)
            cut = []
            for sub in subs:
                Mtok = sub[9]
                S = max(1, min(want, Mtok // 1024))
                if S == 1:
                    cut.append(sub)
                    continue
                chunk = (((Mtok + S - 1) // S + 31) // 32) * 32
                t0 = 0
                while t0 < Mtok:
                    n_t = min(chunk, Mtok - t0)
                    cut.append((sub[0] + t0 * sub[4] * 4, sub[1] + t0 * sub[5] * 4) + tuple(sub[2:9]) + (n_t,) + tuple(sub[10:]))
                    t0 += n_t
            subs = cut
    tall = [x for x in subs if trows(x) != 128]
    subs = [x for x in subs if trows(x) == 128]
    launches = [(subs, False)] if subs else []
    if allow_sync and p16 and atomic:
        classes = {}
        for sub in subs:
            classes.setdefault(sub[9], []).append(sub)
        if len(classes) == 1:
            launches = [(subs, True)]
        elif classes:
            big = [(t, c) for t, c in classes.items() if sum(tiles_of(x) for x in c) >= 1024]
            rest = [x for t, c in classes.items() if sum(tiles_of(y) for y in c) < 1024 for x in c]
            launches = [(c, True) for _, c in sorted(big, key=lambda tc: -tc[0])] + ([(rest, False)] if rest else [])
    if tall:   # one 256-row launch per token count (panel-synchronous when allowed), ahead of the 128-row launches
        tclasses = {}
        for sub in tall:
            tclasses.setdefault(sub[9], []).append(sub)
        launches = [(c, bool(allow_sync)) for _, c in sorted(tclasses.items(), key=lambda tc: -tc[0])] + launches
    return launches


def _launch_wgrad_group(its, atomic=1, allow_sync=True):
    groups = {}
    for it in its:
        p16 = it[9]
        groups.setdefault((176 if p16 else int(lib.vptr_gemm_tile_cols(it[4])), it[6], p16), []).append(it)
    for (cols, prec, p16), grp in groups.items():
        # tile rows of this flush: a fixed setting, or -- VPTR_WGRAD_ROWS=auto, the default -- whichever of 128 / 256 ran faster on THIS set
        # of problems (measured once per problem set with HIP events on the launch stream, during the eager warm-up steps every caller
        # runs before it times or captures anything: the two settings trade a better tile for a second launch with a tail of its own, and
        # which side wins depends on the model -- K64 7.06 vs 7.25 ms, KTH128 11.0 vs 12.2, BAIR FAR 16.7 vs 15.0)
        rows_mode = config.wgrad_rows
        tune = None
        if rows_mode == "auto":
            rows_mode = 128
            if p16 and atomic and len(grp) > 3:
                sig = (allow_sync,) + tuple(sorted((it[3], it[4], it[5]) for it in grp))
                tune = _wgrad_tune.setdefault(sig, {"samples": {128: [], 256: []}, "ms": {}, "pending": None, "choice": None})
                capturing = torch.cuda.is_current_stream_capturing()     # (no event queries under capture: wgrad_tune_settle ran before it)
                if not capturing and tune["pending"] is not None and tune["pending"][2].query():     # the timed flush has finished: book it
                    _wgrad_tune_book(tune)
                if tune["choice"] is not None:
                    rows_mode, tune = tune["choice"], None
                elif capturing or tune["pending"] is not None:
                    # no timing now: the best known so far (a capture never gets here undecided: wgrad_tune_settle fixed the choice)
                    known = {r: min(v[1:] or v) for r, v in tune["samples"].items() if v}
                    rows_mode, tune = (min(known, key=known.get) if known else 128), None
                else:   # alternate the two settings until each has its samples (one cold sample each decided a 3 % question)
                    rows_mode = 128 if len(tune["samples"][128]) <= len(tune["samples"][256]) else 256
        # problems: (A ptr, B ptr, D ptr, rowsum ptr, lda, ldb, ldd, rows, cols, tokens, alpha, transposed); plan_wgrad_launches cuts them
        # into the sub-problems of one or more launches
        probs = []
        flops = 0.0
        for (g, x, dW, N, K, M, _, db, alpha, _p) in grp:
            flops += 2.0 * M * N * K
            # token-major P16 problems: put the 176-wide tile side on the dimension it divides.  dW[528][2112] as 128 x 176 tiles of
            # (rows of dW) x (columns) is 5 x 12 tiles with every fifth row tile one-eighth full; computed as X^T . dY and stored
            # transposed (vptr_gemm_desc.d_transposed) it is 17 x 3 tiles.  The bias gradient then needs >= 32 tile rows beyond K.
            flip = (p16 and config.wgrad_flip and N < K and N % 176 == 0 and K % 128 != 0 and 128 - K % 128 >= 32)
            if flip:
                a, b, rows_, cols_, lda, ldb = x, g, K, N, x.stride(0), g.stride(0)
            else:
                a, b, rows_, cols_, lda, ldb = g, x, N, K, g.stride(0), x.stride(0)
            ap, bp, dp, rp, ldd = a.data_ptr(), b.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else 0), dW.stride(0)
            probs.append((ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip))
        launches = plan_wgrad_launches(probs, cols, p16, atomic, allow_sync, rows_mode, split_rem=config.wgrad_split,
                                       token_split=config.wgrad_token_split and not config.deterministic)

        def trows(sub):
            return sub[12] if len(sub) > 12 else 128

        def tiles_of(sub):
            return ((sub[7] + trows(sub) - 1) // trows(sub)) * ((sub[8] + cols - 1) // cols)
        dev = grp[0][0].device
        import struct
        if tune is not None:
            t_e0, t_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t_e0.record()
        for lsubs, vouch in launches:
            n = len(lsubs)
            descs = (GemmDesc * n)()
            starts = []
            total = 0
            lflops = 0.0
            tr = trows(lsubs[0])
            for i, sub in enumerate(lsubs):
                (ap, bp, dp, rp, lda, ldb, ldd, rows_, cols_, M, alpha, flip) = sub[:12]
                d = descs[i]
                d.precision, d.split_k, d.atomic, d.alpha = prec, 1, int(atomic), alpha
                d.A, d.B, d.D, d.a_rowsum = ap, bp, dp, (rp or None)
                d.lda, d.ldb, d.ldd = lda, ldb, ldd
                d.M, d.N, d.K = rows_, cols_, M
                d.d_transposed = int(flip)
                d.a_mode, d.b_mode = (A_P16T, B_P16T) if p16 else (1, 1)
                starts.append(total)
                total += tiles_of(lsubs[i])
                lflops += 2.0 * rows_ * cols_ * M
            if tr == 256:
                descs[0].split_k = -2 if vouch else -3   # 256-row tiles: panel-synchronous / plain (include/vptr_hip.h)
            elif tr == 192:
                descs[0].split_k = -4 if vouch else -5   # 192-row tiles, three stages
            elif vouch:
                descs[0].split_k = -1     # every problem walks the same number of tokens: the panel-synchronous launch may serve the group (VPTR_WGRAD_SYNC)
            raw = _to_device_async(bytes(descs), dev)
            st = _to_device_async(struct.pack("%di" % len(starts), *starts), dev)
            prof = profiling.gemm
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.vptr_gemm_grouped(ctypes.byref(descs[0]), ptr(raw), ptr(st), n, total, stream()), "vptr_gemm_grouped")
            if prof is not None:
                e1.record()
                # (which kernel the launcher picks for this group: the panel-synchronous persistent one needs VPTR_WGRAD_SYNC != 0 (default 16), the
                # uniform-token vouch and >= 1024 tiles -- mirrored here so that bench.py names the kernel rocprofv3 will list)
                sync = p16 and atomic and vouch and total >= (512 if tr != 128 else 1024) and os.environ.get("VPTR_WGRAD_SYNC", "16") not in ("0", "")
                prof.append(((cols // 16, prec, 6 if p16 else 1, 4 if p16 else 1, (("grouped_sync%d" % tr if tr != 128 else "grouped_sync") if sync else "grouped") if atomic else "grouped_split"),
                             lflops, e0, e1))
        if tune is not None:
            t_e1.record()
            tune["pending"] = (rows_mode, t_e0, t_e1)


def convt_weight_grads(layers, tokens_per_split=2560):
    """Weight gradients of ConvTranspose2d(3x3, s2, p1, op1) layers, D[ci][(ky, kx, co)] = sum_pix x[pix][ci] * P[pix][(ky, kx, co)] with
    P = im2col (3x3, s2, p1) of the output gradient, on the token-major P16 kernel of the grouped weight-gradient launch.  Alone such
    a problem is 4 - 70 output tiles over up to 164 k tokens, so every layer is cut into token ranges of `tokens_per_split`; all ranges
    of all layers run as ONE grouped launch (plain stores into per-range buffers), one vptr_partial_reduce launch adds them up.
    layers: (x [pix, ci] fp32, g [B * oh * ow, co] fp32, B, ih, iw, ci, oh, ow, co); returns the D tensors [ci, 9 * co]."""
    items, keep, outs, red = [], [], [], []
    for (x, g, B, ih, iw, ci, oh, ow, co) in layers:
        pix = B * ih * iw
        P = torch.empty((pix, 9 * co), device=g.device, dtype=torch.float32)      # P16
        check(lib.vptr_im2col_nhwc_p16(ptr(g), ptr(P), B, oh, ow, co, ih, iw, 3, 3, 2, 1, 0, stream()), "vptr_im2col_nhwc_p16")
        xs = to_p16(x)
        S = max(1, pix // int(tokens_per_split))
        step = (pix + S - 1) // S
        step = (step + 31) // 32 * 32                                              # whole 32-token steps per range
        S = (pix + step - 1) // step
        part = torch.empty((S, ci, 9 * co), device=g.device, dtype=torch.float32)
        for k in range(S):
            r0, r1 = k * step, min(pix, (k + 1) * step)
            items.append((xs[r0:r1], P[r0:r1], part[k], ci, 9 * co, r1 - r0, config.gemm_precision, None, 1.0, True))
        D = torch.zeros((ci, 9 * co), device=g.device, dtype=torch.float32)
        red.append((part, D, S, ci * 9 * co))
        keep += [P, xs, part]
        outs.append(D)
    _launch_wgrad_group(items, atomic=0)
    tab = (_lib.ReduceEntry * len(red))()
    for i, (part, D, S, C) in enumerate(red):
        tab[i].part, tab[i].dst0, tab[i].dst1, tab[i].nparts, tab[i].C = ptr(part), ptr(D), None, S, C
    raw = _to_device_async(bytes(tab), outs[0].device)
    esz = ctypes.sizeof(_lib.ReduceEntry)
    for i, (part, D, S, C) in enumerate(red):   # one launch per layer: the widths differ by 4x
        check(lib.vptr_partial_reduce(ctypes.c_void_p(raw.data_ptr() + i * esz), 1, C, 1, stream()), "vptr_partial_reduce")
    return outs


def flush_wgrads(chunks=1, on_chunk=None):
    """Launch every recorded weight gradient (idempotent; runs automatically at the end of a backward pass).

    chunks > 1: the records are ordered by the address of their destination and launched as `chunks` grouped GEMMs of about
    equal work; after each launch `on_chunk(first_dW_ptr)` is called with the lowest destination address of the NEXT chunk
    (None after the last): everything below it is final, so its gradient range can go out to the other ranks while the next
    chunk computes."""
    flush_partial_reduces()
    if not _wgrad_q:
        if on_chunk is not None:
            on_chunk(None)
        return
    items = list(_wgrad_q)
    del _wgrad_q[:]
    if chunks <= 1 or len(items) < 2 * chunks:
        # largest problems first (their 51-tile waves fill the chip; the 15-tile problems then pack the tail), problems that read the
        # same X next to each other: 8.55 -> 8.42 ms on the K64 step's 196 problems (tools/wgrad_ab.sh)
        items.sort(key=lambda it: (-it[3] * it[4], it[1].data_ptr(), it[2].data_ptr()))
        _launch_wgrad_group(items)
        if on_chunk is not None:
            on_chunk(None)
        return
    items.sort(key=lambda it: it[2].data_ptr())
    work = [float(it[3]) * it[4] for it in items]
    per = sum(work) / chunks
    acc, lo = 0.0, 0
    bounds = []
    for i, w in enumerate(work):
        acc += w
        if acc >= per * (len(bounds) + 1) and len(bounds) < chunks - 1 and i + 1 < len(items):
            bounds.append(i + 1)
    bounds.append(len(items))
    for hi in bounds:
        # the chunk boundaries follow slab addresses (what makes a gradient range final); INSIDE a chunk the single-launch order applies:
        # largest problems first, problems that read the same X next to each other
        # plain launch for the chunks: the persistent panel-synchronous kernel assumes that ALL its 512 workgroups are resident at once (every
        # CU's whole LDS), and a chunk runs beside the all-reduce kernels of the previous one -- a displaced workgroup would cost the others
        # a bounded-spin time-out (VPTR_WGRAD_SYNC_CHUNKS=1 allows it anyway)
        _launch_wgrad_group(sorted(items[lo:hi], key=lambda it: (-it[3] * it[4], it[1].data_ptr(), it[2].data_ptr())),
                            allow_sync=os.environ.get("VPTR_WGRAD_SYNC_CHUNKS") == "1")
        if on_chunk is not None:
            on_chunk(items[hi][2].data_ptr() if hi < len(items) else None)
        lo = hi


def _split_k_for(tiles, K):
    """Enough K-splits to put >= ~512 workgroups on the 256 CUs, each split >= 256 deep."""
    if tiles >= 384:
        return 1
    return max(1, min((512 + tiles - 1) // tiles, K // 256))
