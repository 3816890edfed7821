#!/usr/bin/env python
"""bench.py -- predicted frames/sec of the VPTR-NAR train step (KTH 10->10 @ 64x64) on N MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python bench.py --gpus N ...                      # WORLD_SIZE unset: bench.py starts its N ranks itself (one per GPU), like
                                                      # the reference's mp.spawn entry point (train_NAR_mp.py:319-326)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W        # ranks started from outside: RANK / LOCAL_RANK / WORLD_SIZE from the env
    python bench.py --config bair_far --gpus 4        # BASELINE config 4 (FARTrainer);  --config kth128 --gpus 8 = config 5

A run that ends up with fewer ranks or fewer visible devices than --gpus exits non-zero; it never prints a smaller n_gpus.

One "step" = `single_iter` of the reference's stage-2 trainer (train_NAR.py:49-107): 2x VPTREnc (no grad), VPTRFormerNAR
(4 enc + 8 dec layers, dropout 0.1), VPTRDec, MSE + GDL + 0.1*BiPatchNCE, backward, clip_grad_norm_(1.0), AdamW(1e-4);
data parallel = all-reduce (mean) of the transformer gradients over RCCL.  Inputs are synthetic KTH-shaped batches that
are already resident in HBM when the timed region starts.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PER_GPU_BATCH = 16          # train_NAR.py:165
TP, TF = 10, 10
GF_PER_SAMPLE = 554.0       # algorithmic GFLOP of one train step per sample
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 MFMA peak of MI355X (MI355X_MICROARCH.md)

# BASELINE.json's configurations as bench jobs.  `frames` = frames one clip contributes to the "predicted frames" count of a step;
# `gf` = algorithmic GFLOP of one train step per clip (encoder passes no-grad once, transformer and decoder 3x):
#   k64      2 * 62.64 + 3 * 138.04 + 3 * 4.834 + 1.5                                        = 554
#   bair_far 29 * 6.264 (Enc, 29 frames, once) + 3 * 301.0 (FAR 12 layers, T = 29) + 3 * 29 * 0.483 = 1127
#   kth128   50 * 25.06 (Enc, 10 + 40 frames) + 3 * 1740.8 + 3 * 40 * 1.93 + 24 (NCE projector, losses) = 6731
JOBS = {
    "k64": {"metric": "predicted frames/sec (train step) NAR KTH 10->10 @64x64", "batch": PER_GPU_BATCH, "frames": TF, "gf": GF_PER_SAMPLE,
            "baseline_config": 3},
    "mnist": {"metric": "predicted frames/sec (train step) NAR MovingMNIST 10->10 @64x64", "batch": PER_GPU_BATCH, "frames": TF, "gf": GF_PER_SAMPLE,
              "baseline_config": 2},
    "bair_far": {"metric": "predicted frames/sec (train step) FAR BAIR 2->28 @64x64", "batch": 16, "frames": 29, "gf": 1127.0,
                 "baseline_config": 4},
    "kth128": {"metric": "predicted frames/sec (train step) NAR KTH 10->40 @128x128", "batch": 2, "frames": 40, "gf": 6731.0,
               "baseline_config": 5},
}


def _quiet_init(*mods):
    import contextlib
    import io
    import vptr_amd.model as M
    with contextlib.redirect_stdout(io.StringIO()):
        for m in mods:
            M.init_weights(m)


def build_models(dev, dropout):
    import vptr_amd.model as M
    torch.manual_seed(3407)  # train_NAR_mp.py:278
    enc = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
    dec = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect")
    _quiet_init(enc, dec)
    T = M.VPTRFormerNAR(TP, TF, 8, 8, 528, 8, 4, 8, dropout, 4, 4, False, True)
    return enc.to(dev), dec.to(dev), T.to(dev)


def synth_batch(n, rank, dev):
    rs = np.random.RandomState(2021 + rank)
    past = (rs.uniform(0, 1, size=(n, TP, 1, 64, 64)).astype(np.float32) - 0.6013795) / 2.7570653  # utils/dataset.py:23
    fut = (rs.uniform(0, 1, size=(n, TF, 1, 64, 64)).astype(np.float32) - 0.6013795) / 2.7570653
    return torch.from_numpy(past).to(dev), torch.from_numpy(fut).to(dev)


def build_job(name, dev, dropout, batch, rank):
    """(enc, dec, transformer, trainer class, trainer kwargs, past, future, workload description) of one BASELINE configuration, modules on `dev`,
    synthetic batch of the configuration's own shapes and normalisation (utils/dataset.py:23,49) seeded per rank"""
    import vptr_amd.model as M
    from vptr_amd.train import FARTrainer, NARTrainer
    torch.manual_seed(3407)
    rs = np.random.RandomState(2021 + rank)

    def clip(shape, mean, std):
        x = rs.uniform(0, 1, size=shape).astype(np.float32)
        m = np.asarray(mean, np.float32).reshape(1, 1, -1, 1, 1)
        sd = np.asarray(std, np.float32).reshape(1, 1, -1, 1, 1)
        return torch.from_numpy((x - m) / sd).to(dev)
    if name == "k64":
        enc, dec, T = build_models(dev, dropout)
        past, fut = synth_batch(batch, rank, dev)
        return (enc, dec, T, NARTrainer, {"batch_size": batch, "lam_pc": 0.1}, past, fut,
                "K64: KTH 64x64x1 10->10, VPTREnc/Dec(528, Tanh, reflect) + VPTRFormerNAR(4 enc + 8 dec, d=528, 8 heads, ws 4, dropout %.2f), "
                "single_iter of train_NAR.py, random-init weights" % dropout)
    if name == "mnist":
        enc = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
        dec = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Sigmoid", padding_type="reflect")
        _quiet_init(enc, dec)
        T = M.VPTRFormerNAR(TP, TF, 8, 8, 528, 8, 4, 8, dropout, 4, 4, False, True)
        past, fut = clip((batch, TP, 1, 64, 64), [0.0], [1.0]), clip((batch, TF, 1, 64, 64), [0.0], [1.0])
        return (enc.to(dev), dec.to(dev), T.to(dev), NARTrainer, {"batch_size": batch, "lam_pc": 0.1}, past, fut,
                "MovingMNIST 64x64x1 10->10: the K64 transformer with a Sigmoid decoder, pixels in [0, 1), single_iter of train_NAR.py")
    if name == "bair_far":
        enc = M.VPTREnc(3, feat_dim=528, n_downsampling=3, padding_type="zero")
        dec = M.VPTRDec(3, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="zero")
        _quiet_init(enc, dec)
        T = M.VPTRFormerFAR(2, 28, 8, 8, 528, 8, 12, dropout, 4, 4, True)
        mean, std = (0.6175, 0.6050, 0.5218), (2.182, 2.155, 1.912)     # utils/dataset.py:49
        past, fut = clip((batch, 2, 3, 64, 64), mean, std), clip((batch, 28, 3, 64, 64), mean, std)
        return (enc.to(dev), dec.to(dev), T.to(dev), FARTrainer, {}, past, fut,
                "BAIR 64x64x3 2->28: VPTREnc/Dec(528, Tanh, zero pad) + VPTRFormerFAR(12 layers, causal temporal attention, T_in = 29), "
                "single_iter of train_FAR.py, per-GPU batch %d (train_FAR_mp.py:300: 64 over 4 GPUs), random-init weights" % batch)
    if name == "kth128":
        enc = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
        dec = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect")
        _quiet_init(enc, dec)
        T = M.VPTRFormerNAR(10, 40, 16, 16, 528, 8, 4, 8, dropout, 8, 4, False, True)
        past = clip((batch, 10, 1, 128, 128), [0.6013795], [2.7570653])
        fut = clip((batch, 40, 1, 128, 128), [0.6013795], [2.7570653])
        return (enc.to(dev), dec.to(dev), T.to(dev), NARTrainer, {"batch_size": batch, "lam_pc": 0.1}, past, fut,
                "KTH 128x128x1 10->40: 16x16 features, VPTRFormerNAR(4 enc + 8 dec, 8x8 windows), single_iter of train_NAR.py, per-GPU batch %d, "
                "random-init weights" % batch)
    raise SystemExit("unknown --config %r" % name)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    """distinct (physical id, core id) pairs of /proc/cpuinfo; falls back to os.cpu_count()"""
    try:
        cores, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
        if cores:
            return len(cores)
    except OSError:
        pass
    return os.cpu_count() or 1


def _numa_core_list():
    """one logical CPU per physical core, NUMA node by node (SMT siblings dropped): [[cpu, ...] of node 0, [...] of node 1, ...]"""
    def parse(txt):
        out = []
        for part in txt.strip().split(","):
            if "-" in part:
                a, b = part.split("-")
                out += list(range(int(a), int(b) + 1))
            elif part:
                out.append(int(part))
        return out
    nodes = []
    try:
        import glob
        allowed = set(os.sched_getaffinity(0))
        for nd in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda q: int(q.rsplit("node", 1)[1])):
            cpus, seen = [], set()
            for c in parse(open(nd + "/cpulist").read()):
                if c not in allowed:
                    continue
                try:
                    sib = min(parse(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read()))
                except OSError:
                    sib = c
                if sib not in seen:
                    seen.add(sib)
                    cpus.append(c)
            if cpus:
                nodes.append(cpus)
    except OSError:
        pass
    return nodes or [sorted(os.sched_getaffinity(0))]


def _cpu_probe(thread_list, pin, steps):
    """child process of `cpu_baseline` (bench.py --cpu-probe 8,16,32): the affinity mask is set BEFORE torch creates its thread pool, so every
    worker inherits it.  pin = 1: the process is bound to max(thread_list) physical cores of as few NUMA nodes as possible (numactl
    --physcpubind style; the smaller counts of the sweep run inside that set), 0: no mask."""
    cpus = None
    if pin:
        flat = [c for node in _numa_core_list() for c in node]
        cpus = flat[:max(thread_list)]
        if len(cpus) == max(thread_list):
            os.sched_setaffinity(0, set(cpus))
        else:
            cpus = None
    from oracle import vptr_oracle as O
    import vptr_amd.model as M
    torch.manual_seed(3407)
    torch.set_num_threads(max(thread_list))
    n = 4
    cfg = dict(Tp=TP, Tf=TF, H=8, W=8, C=528, nhead=8, window_size=4, num_encoder_layers=4, num_decoder_layers=8, rpe=True)
    enc = M.VPTREnc(1, 528, 3, "reflect")
    dec = M.VPTRDec(1, 528, 3, "Tanh", "reflect")
    T = M.VPTRFormerNAR(TP, TF, 8, 8, 528, 8, 4, 8, 0.0, 4, 4, False, True)
    st = O.NARStep(dict(enc.state_dict()), dict(dec.state_dict()), dict(T.state_dict()), cfg)
    past, fut = synth_batch(n, 0, "cpu")
    for k in thread_list:
        torch.set_num_threads(k)
        st.step(past, fut)   # warm-up at this thread count
        times = []
        for _ in range(steps):
            t0 = time.perf_counter()
            st.step(past, fut)
            times.append(time.perf_counter() - t0)
        print("CPU_PROBE " + json.dumps({"threads": k, "pinned": cpus is not None, "bound_to_cpus": len(cpus) if cpus else None, "times": times, "n": n}))
        sys.stdout.flush()


def cpu_baseline(seconds_budget=120.0, timed_steps=3):
    """The oracle (oracle/vptr_oracle.py: CPU restatement of the reference, parity-pinned by tests/golden) timed on the host cores of this
    box on a bounded sample of the same workload: N = 4 KTH-shaped clips.  Thread sweep over {8, 16, 32} (VERDICT r5
    item 7) in ONE child process bound -- before its thread pool exists -- to 32 physical cores of one NUMA node, 1 warm-up + `timed_steps`
    timed steps per point; the reported figure is the best mean.  All physical cores are not swept: the oracle's small fp32 GEMMs stop scaling
    near 16 threads (32 threads: 2x the time of 16 in every run of this pool; 128: 13.7 s/step, measured in round 3)."""
    import subprocess
    phys = _physical_cores()
    nodes = _numa_core_list()
    cand = [k for k in (8, 16, 32) if k <= phys] or [phys]
    t_start = time.perf_counter()
    points = []
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", ",".join(str(k) for k in cand), "--cpu-probe-pin", "1",
                              "--cpu-probe-steps", str(timed_steps)], capture_output=True, text=True, timeout=seconds_budget * 2.5,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in out.stdout.splitlines():
            if line.startswith("CPU_PROBE "):
                r = json.loads(line[len("CPU_PROBE "):])
                r["mean"] = sum(r["times"]) / len(r["times"])
                points.append(r)
    except subprocess.TimeoutExpired as e:
        for line in (e.stdout or b"").decode(errors="replace").splitlines():
            if line.startswith("CPU_PROBE "):
                r = json.loads(line[len("CPU_PROBE "):])
                r["mean"] = sum(r["times"]) / len(r["times"])
                points.append(r)
    if not points:
        raise RuntimeError("cpu_baseline: no probe finished")
    best = min(points, key=lambda r: r["mean"])
    n, mean = best["n"], best["mean"]
    return {"value": round(n * TF / mean, 3), "unit": "predicted frames/s", "cores": best["threads"], "kind": "port",
            "pinned": best["pinned"], "bound_to_cpus": best["bound_to_cpus"], "physical_cores": phys, "logical_cpus": os.cpu_count(),
            "numa_nodes": len(nodes), "cpu_model": _cpu_model(),
            "s_per_step": round(mean, 3), "s_per_step_min_max": [round(min(best["times"]), 3), round(max(best["times"]), 3)],
            "timed_steps": len(best["times"]),
            "thread_sweep": [{"threads": r["threads"], "timed_steps": len(r["times"]), "s_per_step": round(r["mean"], 3),
                              "min": round(min(r["times"]), 3)} for r in points],
            "thread_sweep_skipped": [k for k in cand if k not in [r["threads"] for r in points]] + ([phys] if phys > 32 else []),
            "wall_s": round(time.perf_counter() - t_start, 1),
            "sample": "oracle NAR train step (fp32 torch CPU), batch %d x 10->10 @64x64; one child process bound (sched_setaffinity before the "
                      "thread pool exists) to 32 physical cores of one NUMA node, 1 warm-up + %d timed steps per thread count; "
                      "best: %d threads, mean %.2f s/step" % (n, len(best["times"]), best["threads"], mean)}


def gemm_roofline(trainer, past, fut, precision):
    """Per-launch HIP-event timing of every GEMM launch of three train steps (instrumented passes, outside the timed region; per-step averages):
    returns the roofline entry of the dominant MFMA kernel instantiation and the GEMM-wide aggregate."""
    import vptr_amd.ops as ops
    NPASS = 3   # instrumented steps: the grouped tn kernel is launched once per step, one sample of it is too noisy (8.8 vs 9.7 ms seen)
    recs, opt_recs = [], []
    ops.profiling.gemm = recs
    ops.profiling.opt = opt_recs
    try:
        for _ in range(NPASS):
            trainer.step(past, fut)
        torch.cuda.synchronize()
    finally:
        ops.profiling.gemm = None
        ops.profiling.opt = None
    by = {}
    for key, flops, e0, e1 in recs:
        ms = e0.elapsed_time(e1)
        d = by.setdefault(key, [0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += ms
    for d in by.values():   # per step
        d[0] //= NPASS
        d[1] /= NPASS
        d[2] /= NPASS
    tot_f = sum(d[1] for d in by.values())
    tot_ms = sum(d[2] for d in by.values())
    dom = max(by.items(), key=lambda kv: kv[1][2])
    (cnt, fl, ms) = dom[1]
    dominant_by_time = dom[0]

    def kernel_name(key):   # the name rocprofv3 lists the launch under
        nfn, prec, am, bm = key[:4]
        if am == 5:
            return "vptr_gemm_p16_kernel<%d, %d>" % (key[5], key[6])
        if am == 6:   # the end-of-backward launch into the gradient slab / the plain-store launches of token-range sub-problems
            if len(key) > 4 and key[4] == "grouped_sync256":   # the same on 256 x 176 tiles (VPTR_WGRAD_ROWS=256)
                return "vptr_wgrad_p16_sync_kernel<16, 8, 4, 2>"
            if len(key) > 4 and key[4] == "grouped_sync192":   # ... on 192 x 176 tiles, three stages (VPTR_WGRAD_ROWS=192)
                return "vptr_wgrad_p16_sync_kernel<16, 8, 3, 3>"
            if len(key) > 4 and key[4] == "grouped_sync":   # the panel-synchronous persistent launch (default)
                return "vptr_wgrad_p16_sync_kernel<%s>" % (os.environ.get("VPTR_WGRAD_SYNC", "16") if os.environ.get("VPTR_WGRAD_SYNC", "16") in ("8", "32") else "16")
            return "vptr_wgrad_p16_kernel<2, 1>" if (len(key) > 4 and key[4] == "grouped_split") else "vptr_wgrad_p16_kernel<2, 0>"
        if am == 3:
            return "vptr_conv_planes_kernel<true>"
        base = "vptr_gemm_grouped_kernel" if (len(key) > 4 and key[4] == "grouped") else ("vptr_gemm_kernel_p" if (len(key) > 5 and key[5] == "p") else "vptr_gemm_kernel")
        return "%s<%d, %d, %d, %d>" % (base, nfn, prec, am, bm)
    # ONE fixed headline kernel across rounds: the lone-workgroup nt P16 instantiation that carries the K = 2112 products
    # (fc2 forward, fc1 input gradients; 121 launches per K64 step) -- the most-launched MFMA kernel of the step.  The entry with the
    # largest time share of THIS run is still named (`dominant_by_time`), and every instantiation is listed in `per_kernel`.
    fixed = [kv for kv in by.items() if kernel_name(kv[0]) == HEADLINE_KERNEL]
    if fixed:
        cnt = sum(kv[1][0] for kv in fixed)
        fl = sum(kv[1][1] for kv in fixed)
        ms = sum(kv[1][2] for kv in fixed)
        kname = HEADLINE_KERNEL
    else:
        kname = kernel_name(dom[0])
    fam = kname.split("<")[0] if "vptr_gemm_p16_kernel" in kname else kname   # the PMC file keeps the nt P16 instantiations as one family
    peak = MFMA_PEAK_TFLOPS
    ach = fl / (ms * 1e-3) / 1e12
    # HBM-side bytes per launch of that kernel: PMC counters cannot be read in-process, so this is the committed rocprofv3
    # measurement of this same step (tools/prof_r02.sh: separate FETCH_SIZE / WRITE_SIZE passes; KB units; FETCH_SIZE doubled for
    # 16-B/lane streaming reads on gfx950 as MI355X_MICROARCH.md prescribes).  The file names the kernels it was taken with: if the
    # dominant kernel of THIS run is not in it, the entry is stale and traffic stays null with the reason spelt out.
    traffic, traffic_source = None, None
    import glob
    import hashlib
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    src = os.path.relpath(cands[-1], ROOT) if cands else os.path.join("profiles", "r03_pmc_traffic.json")
    try:
        pm = json.load(open(os.path.join(ROOT, src)))
        # the file records a hash of the GEMM sources it was measured with (tools/prof_round.sh): a kernel edited since then makes the
        # committed counters somebody else's -- traffic stays null until the PMC passes are re-run
        here = hashlib.sha256(b"".join(open(os.path.join(ROOT, "vptr_amd", "csrc", f), "rb").read()
                                       for f in ("gemm_p16.hip", "gemm_shared.h", "gemm.hip"))).hexdigest()[:16]
        e = pm.get("kernels", {}).get(fam)
        if pm.get("gemm_source_sha16") != here:
            traffic_source = "STALE: %s was measured with GEMM sources %s, this build has %s -- re-run tools/prof_round.sh" % (
                src, pm.get("gemm_source_sha16"), here)
        elif e:
            traffic = round((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0)
            traffic_source = "committed rocprofv3 PMC passes of this step with these GEMM sources (%s; `%s`), average over %d launches; not a counter of this run" % (
                src, pm.get("command", "bench.py"), e["launches"])
        else:
            traffic_source = "STALE: %s holds no entry for %s -- re-run tools/prof_round.sh" % (src, fam)
    except Exception as ex:  # noqa
        traffic_source = "unavailable: %s" % str(ex)[:120]
    per_kernel = {}
    for key, (c_, f_, m_) in sorted(by.items(), key=lambda kv: -kv[1][2]):
        d = per_kernel.setdefault(kernel_name(key), [0, 0.0, 0.0])
        d[0] += c_
        d[1] += f_
        d[2] += m_
    return {
        "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
        "traffic_unit": "HBM-side bytes per launch", "traffic_source": traffic_source,
        "kernel": kname, "dominant_by_time": kernel_name(dominant_by_time),
        "frac_of_3pass_ceiling": round(ach / (peak / 3.0), 4) if "p16" in kname or "planes" in kname else None,
        "launches_per_step": cnt, "avg_launch_us": round(ms * 1e3 / cnt, 2), "alg_gflop_per_launch": round(fl / cnt / 1e9, 3),
        "all_gemm": {"launches_per_step": sum(d[0] for d in by.values()), "ms_per_step": round(tot_ms, 3),
                     "achieved": round(tot_f / (tot_ms * 1e-3) / 1e12, 2), "alg_gflop_per_step": round(tot_f / 1e9, 1)},
        "per_kernel": {k: {"launches": d[0], "ms_per_step": round(d[2], 3), "achieved": round(d[1] / (d[2] * 1e-3) / 1e12, 1)}
                       for k, d in per_kernel.items()},
        "operand_stream": operand_stream_roof(kname, ach),
        "hbm_side": hbm_roofline(opt_recs), "infinity_cache_side": infinity_cache_rate(),
        "note": "algorithmic FLOPs = 2*M*N*K per launch (HIP events on the launch stream around every GEMM launch of three instrumented "
                "steps, averaged); the split-bf16 kernels issue 3 bf16 MFMA passes per algorithmic FLOP (fp32-class accuracy), so their ceiling "
                "is peak/3 = 833 TFLOP/s, and ~480 TFLOP/s under the MFMA power envelope (DESIGN.md section 4)",
    }


HBM_PEAK_GBS = 8000.0
HEADLINE_KERNEL = "vptr_gemm_p16_kernel<1, 4>"
L2_TO_CU_TBPS = 9.2    # measured: the DMA-only build of the grouped weight-gradient launch stages 61.5 GB in 6.66 ms (profiles/r05_ingest_roofline.log)


def operand_stream_roof(kname, achieved_tflops):
    """The roof that binds the P16 GEMMs: algorithmic flops per byte a tile stages per K-step (TM x TN x 32:
    2 TM TN 32 flop against (TM + TN) x 128 B of bf16 hi + lo operands) x the measured bandwidth of the L2 -> CU operand path."""
    if "p16" not in kname:
        return None
    tm = 256 if ("sync_kernel<16, 8, 4" in kname or "wgrad_p16_kernel<2, 0, 8, 4>" in kname) else 128
    tn = 176
    fpb = 2.0 * tm * tn * 32 / ((tm + tn) * 128.0)
    roof = fpb * L2_TO_CU_TBPS
    return {"tile": "%d x %d x 32" % (tm, tn), "flop_per_staged_byte": round(fpb, 2), "l2_to_cu_TBps": L2_TO_CU_TBPS, "roof": round(roof, 1),
            "unit": "TFLOP/s", "frac": round(achieved_tflops / roof, 4),
            "source": "profiles/r05_ingest_roofline.log (DMA-only elimination build; a committed measurement, not a counter of this run)"}


def hbm_roofline(opt_recs, root=ROOT):
    """The HBM roofline of the step's genuinely HBM-streaming kernel, `adamw_kernel` (clip + AdamW over the flat slabs: reads p, g, m, v
    and writes p, m, v = 7 x 4 B per parameter, nothing is re-used), event-timed on the launch stream inside the instrumented steps;
    `weight_planes_kernel` (re-reads the fresh parameters, writes the two P16 images) beside it while it exists.  Counter bytes come
    from the committed PMC file (rocprofv3 cannot run in-process)."""
    if not opt_recs:
        return {"error": "no optimizer launch was recorded"}
    n = opt_recs[0][0]
    plane_elems = opt_recs[0][1]
    us_a = sum(r[2][0].elapsed_time(r[2][1]) for r in opt_recs) * 1e3 / len(opt_recs)
    us_p = sum(r[2][1].elapsed_time(r[2][2]) for r in opt_recs) * 1e3 / len(opt_recs)
    alg = 7 * 4 * n
    out = {"bound": "hbm", "kernel": "adamw_kernel", "alg_bytes": alg, "avg_launch_us": round(us_a, 2), "launches_timed": len(opt_recs),
           "achieved": round(alg / us_a / 1e3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg / us_a / 1e3 / HBM_PEAK_GBS, 4),
           "traffic": None,
           "note": "algorithmic bytes = 7 x 4 B x %d slab elements (read p, g, m, v; write p, m, v); HIP events around the launch in the "
                   "instrumented steps; 8 TB/s is the HBM3E peak of MI355X_MICROARCH.md (its measured achievable stream rate is ~6.3 TB/s)" % n}
    if plane_elems:
        algp = 3 * 4 * plane_elems   # read W fp32, write the [N, K] and the [K, N] P16 image (4 B per element each)
        out["weight_planes_kernel"] = {"alg_bytes": algp, "avg_launch_us": round(us_p, 2), "achieved": round(algp / us_p / 1e3, 1),
                                       "frac": round(algp / us_p / 1e3 / HBM_PEAK_GBS, 4),
                                       "note": "reads the parameters adamw_kernel just wrote (473 MB: they do not fit the 256 MB Infinity Cache)"}
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic.json")))
        pm = json.load(open(cands[-1]))
        e = pm.get("kernels", {}).get("adamw_kernel")
        if e:
            out["traffic"] = round((2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0)
            out["traffic_source"] = "%s (rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this step; not a counter of this run)" % os.path.relpath(cands[-1], root)
    except Exception as ex:  # noqa
        out["traffic_source"] = "unavailable: %s" % str(ex)[:100]
    return out


def infinity_cache_rate():
    """rate of the largest elementwise pass of the K64 step (LayerNorm((2112,8,8)) + GELU of a conv-FFN hidden tensor, 86.5 MB): it fits
    the 256 MB Infinity Cache, so this is an on-die figure, reported beside -- never as -- the HBM fraction"""
    import vptr_amd.ops as ops
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        rows, F, HW = PER_GPU_BATCH * TF * 64, 2112, 64
        x = torch.randn(rows, F, device=dev)
        w, b = torch.randn(HW, F, device=dev), torch.randn(HW, F, device=dev)
        with torch.no_grad():
            for _ in range(3):
                ops.norm_act(x, w, b, "ln", HW, True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.norm_act(x, w, b, "ln", HW, True)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nbytes = 3.0 * rows * F * 4      # statistics pass reads x; normalise pass reads x and writes y
        return {"kernels": "vptr_groupstats + vptr_norm_act_fwd (LayerNorm((2112,8,8)) + GELU on [10240 x 2112] fp32)", "us_per_call": round(us, 2),
                "alg_bytes": int(nbytes), "rate": round(nbytes / us / 1e3, 1), "unit": "GB/s",
                "note": "cache-resident (86.5 MB < 256 MB Infinity Cache): NOT an HBM figure"}
    except Exception as ex:  # noqa
        return {"error": str(ex)[:160]}


def _time_steps(step, warm=3, timed=5):
    """median of `timed` individually synchronised steps (a one-off allocator stall in a fresh configuration would otherwise
    dominate a 3-step mean: one default run once reported 131 ms for a 54 ms step)"""
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(timed):
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def other_configs(dev, dropout, skip=()):
    """ms/step of BASELINE.json's other configurations on this GPU (rank 0, after the timed region, 2 warm-up + 3 timed steps each;
    reported, never part of `value`): config 2 (MNIST NAR: the K64 shapes with a Sigmoid decoder), config 4 (BAIR FAR at its literal
    size: 3-channel zero-padded auto-encoder, VPTRFormerFAR(2, 28), T_in = 29, per-GPU batch 16 = train_FAR_mp.py's 64 over 4 GPUs)
    and config 5 (KTH 128x128 10 -> 40, 16x16 features, 8x8 windows, per-GPU batch 2).  Each of them is also a timed bench job of its
    own: `bench.py --config mnist|bair_far|kth128 [--gpus N]`."""
    import vptr_amd.model as M
    import vptr_amd.ops as ops
    out = {}
    for key, name in (("config2_mnist_nar_sigmoid", "mnist"), ("config4_bair_far_2to28", "bair_far"), ("config5_kth128_nar_10to40", "kth128")):
        if name in skip:
            continue
        try:
            job = JOBS[name]
            enc, dec, T, cls, kw, past, fut, _ = build_job(name, dev, dropout, job["batch"], 0)
            tr = cls(enc, dec, T, lr=1e-4, max_grad_norm=1.0, **kw)
            launch = "eager"
            try:     # the same launch mode as the headline line (an eager step of these models is host-bound on slow hosts: 1000 - 1450 launches)
                tr.capture(past, fut, warmup=2)
                launch = "hipGraph (not re-verified here: `bench.py --config %s` runs verify_graph on it)" % name
            except Exception:  # noqa
                tr._graph = None
            ms = _time_steps(lambda: tr.step(past, fut))
            out[key] = {"ms_per_step": round(ms, 2), "per_gpu_batch": job["batch"], "frames_per_s": round(job["batch"] * job["frames"] / ms * 1e3, 1),
                        "step_tflops": round(job["gf"] * job["batch"] / 1e3 / (ms * 1e-3), 1), "launch": launch}
            tr.opt.close()
            del tr, enc, dec, T, past, fut
        except Exception as e:  # noqa
            out[key] = {"error": str(e)[:160]}
        torch.cuda.empty_cache()
    ops.unregister_flat_slabs()
    # ---- the K64 step the way the reference's script itself drives the package (no NARTrainer: stock AdamW, clip_grad_norm_, criterion
    # classes; vptr_amd.train.script_style_nar_iter), on fresh modules so that no flat slab is involved
    try:
        from vptr_amd.train import script_style_nar_iter
        enc6, dec6, T6 = build_models(dev, dropout)
        enc6, dec6 = enc6.eval(), dec6.eval()
        opt6 = torch.optim.AdamW(T6.parameters(), lr=1e-4)
        mse, gdl = M.MSELoss(), M.GDL(alpha=1)
        bp = M.BiPatchNCE(PER_GPU_BATCH, TF, 8, 8, 1.0).to(dev)
        past, fut = synth_batch(PER_GPU_BATCH, 0, dev)
        ms = _time_steps(lambda: script_style_nar_iter(enc6, dec6, T6, opt6, past, fut, mse, gdl, bp, 0.1, 1.0))
        out["drop_in_single_iter"] = {"ms_per_step": round(ms, 2), "per_gpu_batch": PER_GPU_BATCH, "frames_per_s": round(PER_GPU_BATCH * TF / ms * 1e3, 1),
                                      "what": "train_NAR.py:49-107 recipe on the package's modules with torch.optim.AdamW + clip_grad_norm_ + criterion classes, eager"}
        del opt6, T6, enc6, dec6
    except Exception as e:  # noqa
        out["drop_in_single_iter"] = {"error": str(e)[:160]}
    torch.cuda.empty_cache()
    # ---- the same recipe the way the reference's DATA-PARALLEL script drives it: Enc / Dec / transformer wrapped in stock
    # DistributedDataParallel, the projector reached through `.module` (train_NAR_mp.py:94-118,132-189), here in a ONE-rank RCCL group
    # (what a one-GPU box can run: DDP's reducer, bucket copies and per-parameter autograd hooks are all live; the all-reduces are no-ops)
    # (in a process of its own: an RCCL / DDP failure there must not cost the bench its result line)
    try:
        import subprocess
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--ddp-probe", "--dropout", str(dropout)], env=env, capture_output=True,
                           text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            out["drop_in_ddp_single_iter"] = json.loads(lines[-1])
        else:
            out["drop_in_ddp_single_iter"] = {"error": "probe process exited with code %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
    except Exception as e:  # noqa
        out["drop_in_ddp_single_iter"] = {"error": str(e)[:200]}
    return out


def _ddp_script_iter(dev, dropout):
    import torch.distributed as dist
    import torch.nn.functional as F
    from torch.nn.parallel import DistributedDataParallel as DDP
    import vptr_amd.model as M
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        enc, dec, T = build_models(dev, dropout)
        Enc, Dec = DDP(enc, device_ids=[dev.index]).eval(), DDP(dec, device_ids=[dev.index]).eval()
        opt = torch.optim.AdamW(T.parameters(), lr=1e-4)
        TD = DDP(T, device_ids=[dev.index])
        proj = TD.module.NCE_projector
        mse, gdl = M.MSELoss(), M.GDL(alpha=1)
        bp = M.BiPatchNCE(PER_GPU_BATCH, TF, 8, 8, 1.0).to(dev)
        past, fut = synth_batch(PER_GPU_BATCH, 0, dev)

        def it():
            with torch.no_grad():
                pf, ff = Enc(past), Enc(fut)
            TD.train()
            TD.zero_grad(set_to_none=True)
            Dec.zero_grad(set_to_none=True)
            pred_f = TD(pf)
            pred = Dec(pred_f)
            a = proj(pred_f.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
            b = proj(ff.permute(0, 1, 3, 4, 2)).permute(0, 1, 4, 2, 3)
            loss = gdl(fut, pred) + mse(pred, fut) + 0.1 * bp(F.normalize(b, p=2.0, dim=2), F.normalize(a, p=2.0, dim=2))
            loss.backward()
            torch.nn.utils.clip_grad_norm_(TD.parameters(), max_norm=1.0, norm_type=2)
            opt.step()
        ms = _time_steps(it)
        res = {"ms_per_step": round(ms, 2), "per_gpu_batch": PER_GPU_BATCH, "frames_per_s": round(PER_GPU_BATCH * TF / ms * 1e3, 1),
               "what": "train_NAR_mp.py:132-189 recipe: DistributedDataParallel(Enc / Dec / VPTRFormerNAR) in a one-rank RCCL group, stock AdamW + "
                       "clip_grad_norm_ + criterion classes, eager"}
        del TD, Enc, Dec, opt, T, enc, dec
        return res
    finally:
        if own_group:
            dist.destroy_process_group()


# ---- rank start-up ------------------------------------------------------------------------------------------------------------------
def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _visible_gpus():
    try:
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:  # noqa
        return 0


def self_launch(n, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks here, one process per GPU, the way the
    reference's data-parallel entry point spawns its workers (train_NAR_mp.py:319-326 mp.spawn(main_worker, nprocs=world_size);
    :191-198 setup: MASTER_ADDR / MASTER_PORT + init_process_group).  Rendezvous on 127.0.0.1 and a free port; rank 0's stdout is this
    process's stdout (the one JSON line).  If any rank fails the others are stopped (by PID) and the exit code is non-zero."""
    import signal
    import subprocess
    share = os.environ.get("VPTR_BENCH_SHARE_GPU") == "1"
    have = _visible_gpus()
    need = 1 if share else n
    if have < need:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible to this process; refusing to measure fewer ranks than asked for" % (n, have))
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), VPTR_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
                    for q in sorted(live):
                        procs[q].send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    finally:
        deadline = time.time() + 10.0
        for p_ in procs:
            if p_.poll() is None:
                try:
                    p_.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p_.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)      # >= 50 steps after >= 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", choices=sorted(JOBS), default="k64",
                    help="k64 = BASELINE.json's headline (config 3; the default); mnist = config 2; bair_far = config 4 (FARTrainer); kth128 = config 5")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (weak scaling); default: the configuration's own")
    ap.add_argument("--precision", type=int, default=int(os.environ.get("VPTR_GEMM_PRECISION", "3")), choices=[1, 3],
                    help="3 = split-bf16 MFMA (meets the 1e-3 parity bar, default); 1 = single-pass bf16")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VPTR_GRAPH", "1")),
                    help="1 (default): one GPU: the whole step is one hipGraph; several ranks: forward + backward are one hipGraph, the "
                         "exchange and the optimizer stay eager (no collective is captured); 0: eager")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the ms/step lines of BASELINE configs 2 / 4 / 5")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch clips per GPU; strong: the reference's DistributedSampler semantics, a fixed "
                         "--global-batch split as global // world per rank (utils/dataset.py:72)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="one GPU only: bring up a ONE-rank RCCL process group and run the step through the multi-rank code path (chunked "
                         "weight-gradient launches + asynchronous all-reduces of the gradient slab on c10d's RCCL stream)")
    ap.add_argument("--dp-chunks", type=int, default=0, help="grouped weight-gradient launches per step on the exchange path (0 = vptr_amd.train.DP_CHUNKS, 4): "
                                                             "more chunks = earlier first all-reduce, smaller GEMM launches")
    ap.add_argument("--bucket-mb", type=int, default=64, help="largest all-reduce piece of the gradient slab in MiB (ring collectives over xGMI are "
                                                              "per-link bound: tune against the exchange_timeline of the line)")
    ap.add_argument("--global-batch", type=int, default=64, help="global batch of --scaling strong (train_FAR_mp.py:300 uses 64)")
    ap.add_argument("--ddp-probe", action="store_true", help=argparse.SUPPRESS)   # internal: other_configs' DDP-wrapped iteration, own process
    ap.add_argument("--cpu-probe", type=str, default="", help=argparse.SUPPRESS)        # internal: one thread-sweep point of cpu_baseline, own process
    ap.add_argument("--cpu-probe-pin", type=int, default=1, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-probe-steps", type=int, default=3, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.cpu_probe:
        _cpu_probe([int(k) for k in args.cpu_probe.split(",")], args.cpu_probe_pin, args.cpu_probe_steps)
        return
    if args.ddp_probe:
        import faulthandler
        faulthandler.enable()
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        print(json.dumps(_ddp_script_iter(dev, args.dropout)))
        sys.stdout.flush()
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus, sys.argv[1:])       # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VPTR_BENCH_HANG_DUMP_S"):   # diagnostics: every thread's stack to stderr after that many seconds (a hung rank says where)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VPTR_BENCH_HANG_DUMP_S"]), exit=False)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:   # in either direction: never report a job of a different size than the one asked for
        raise SystemExit("bench.py --gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # VPTR_BENCH_SHARE_GPU=1 + VPTR_BENCH_BACKEND=gloo: functional check of the multi-rank path on a ONE-GPU box (every rank
    # on cuda:0, gradient exchange through gloo); never set by the driver -- its runs use one GPU per rank over RCCL
    share = os.environ.get("VPTR_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("VPTR_BENCH_BACKEND", "nccl")
    have = _visible_gpus()
    isolated = have == 1 and world > 1 and not share and any(
        os.environ.get(v, "").strip() not in ("",) and "," not in os.environ.get(v, "") for v in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    want_index = 0 if (share or isolated) else local_rank     # a launcher that hands every rank exactly one device: that device is index 0
    if have <= want_index:
        raise SystemExit("bench.py rank %d: needs GPU index %d but %d GPU(s) are visible; refusing to run" % (rank, want_index, have))
    if world > 1 and backend == "nccl" and share:
        raise SystemExit("VPTR_BENCH_SHARE_GPU=1 needs VPTR_BENCH_BACKEND=gloo (RCCL refuses two ranks on one device)")
    dev = torch.device("cuda", want_index)
    torch.cuda.set_device(dev)
    pg = None
    dist = torch.distributed
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        pg = dist.group.WORLD
    elif args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ["VPTR_DP_FORCE_EXCHANGE"] = "1"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        pg = dist.group.WORLD
    # the collective layer must actually connect `world` ranks: an all-reduce whose result only the right number of distinct ranks can
    # produce, and an all-gather of the device every rank sits on (two ranks on one GPU is an error unless it was asked for)
    comm = None
    if pg is not None:
        probe = torch.tensor([float(rank + 1)], device=dev)
        dist.all_reduce(probe, op=dist.ReduceOp.SUM)
        got, want = float(probe.item()), world * (world + 1) / 2.0
        if dist.get_world_size() != args.gpus or got != want:
            raise SystemExit("bench.py: the process group connects %d ranks (rank-sum %g, expected %g) but --gpus is %d"
                             % (dist.get_world_size(), got, want, args.gpus))
        ids = [None] * world
        props = torch.cuda.get_device_properties(dev)
        dist.all_gather_object(ids, (-1 if isolated else dev.index, str(getattr(props, "uuid", "")), str(getattr(props, "pci_bus_id", ""))))
        informative = all(i[0] >= 0 or i[1] or i[2] for i in ids)    # an isolated rank is only identifiable through uuid / bus id
        if world > 1 and not share and informative and len(set(ids)) != world:
            raise SystemExit("bench.py: %d ranks sit on %d distinct GPU(s): %s" % (world, len(set(ids)), ids))
        comm = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "rank_sum_check": got, "devices": [list(map(str, i)) for i in ids],
                "launched_by": "bench.py self-launch (one process per GPU)" if os.environ.get("VPTR_BENCH_SELF_LAUNCHED") == "1" else "external launcher (env RANK / WORLD_SIZE)"}
        if backend == "nccl":
            try:
                comm["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:  # noqa
                pass

    import vptr_amd.ops as ops
    job = JOBS[args.config]
    if args.batch is None:
        args.batch = job["batch"]
    if args.scaling == "strong":
        from vptr_amd.parallel import shard_batch
        args.batch = shard_batch(args.global_batch, rank, world)[1]
    ops.config.gemm_precision = args.precision
    enc, dec, T, trainer_cls, tkw, past, fut, workload = build_job(args.config, dev, args.dropout, args.batch, rank)
    if world > 1:  # identical replicas: broadcast rank 0's parameters and buffers (what the DDP constructor does), one message per dtype
        from vptr_amd.parallel import broadcast_modules_flat
        broadcast_modules_flat([T, enc, dec], 0, pg)
    if args.dp_chunks > 0:
        import vptr_amd.train as _tr
        _tr.DP_CHUNKS = args.dp_chunks
    trainer = trainer_cls(enc, dec, T, lr=1e-4, max_grad_norm=1.0, process_group=pg, bucket_mb=args.bucket_mb, **tkw)

    # one GPU: the whole step is one hipGraph.  Several ranks (or --force-exchange): forward + backward are one hipGraph, the part that
    # talks to other ranks (grouped weight-gradient chunks, RCCL all-reduces, optimizer) stays eager -- no collective is captured
    dp = world > 1 or args.force_exchange
    use_graph = bool(args.graph)
    graph_note = "eager"
    graph_check = None
    if use_graph:
        # every rank must take the same branch, whatever happens on any of them: a rank whose capture or check raised still joins the
        # collectives the others issue (verify_graph steps the data-parallel trainer), and the verdict is agreed by an all-reduce(MIN)
        ok, err = False, None
        try:
            if dp:
                trainer.capture_front(past, fut, warmup=2)
            else:
                trainer.capture(past, fut, warmup=2)
        except Exception as e:  # noqa
            err = "graph capture failed: %s" % str(e).split("\n")[0][:160]
            trainer._graph = trainer._front = None
        if world > 1:
            flag = torch.tensor([0.0 if err else 1.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if flag.item() < 0.5 and err is None:
                err = "graph capture failed on another rank"
        if err is None:
            try:
                # the graph is only timed if it computes what the eager step computes: replay i vs an eager step from the same state, and a
                # run of replays vs the eager trajectory (loss terms, gradient norm, post-step parameters); the state is restored afterwards
                ok, graph_check = trainer.verify_graph(past, fut, steps=3, rtol=2e-3)
                graph_check["nodes"] = trainer.graph_nodes
                if not ok:
                    err = "hipGraph replays disagree with eager steps: %s" % (graph_check.get("worst_term"),)
            except Exception as e:  # noqa: on ONE rank this leaves the others inside verify_graph's collectives -- nothing to agree on then
                if world > 1:
                    raise
                err = "graph check failed: %s" % str(e).split("\n")[0][:160]
            if world > 1:
                flag = torch.tensor([1.0 if ok else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if flag.item() < 0.5 and err is None:
                    err = "hipGraph replays disagree with eager steps on another rank"
        if err is None:
            graph_note = "hipGraph (forward + backward) + eager exchange / optimizer" if dp else "hipGraph"
        else:
            trainer._graph = trainer._front = None
            graph_note = "eager (%s)" % err
    if args.force_exchange:
        graph_note += " [one-rank RCCL group, forced gradient exchange]"

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(past, fut)
    if dp:
        trainer.comm_stats = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(past, fut)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0          # this rank's own device-complete time
    sync()
    dt = time.perf_counter() - t0
    cs = trainer.comm_stats
    trainer.comm_stats = None
    per_rank = [dt_own]
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        allt = [torch.zeros(1, device=dev, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allt, torch.tensor([dt_own], device=dev, dtype=torch.float64))
        per_rank = [float(x.item()) for x in allt]
    if comm is not None and cs:
        exp_ms = sum(a.elapsed_time(b) for a, b in cs.get("events", ())) / max(cs.get("steps", 1), 1)
        mine = torch.tensor([exp_ms, cs.get("wait_host_s", 0.0) * 1e3 / max(cs.get("steps", 1), 1)], device=dev, dtype=torch.float64)
        if world > 1:
            allc = [torch.zeros(2, device=dev, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allc, mine)
        else:
            allc = [mine]
        comm.update({
            "allreduce_bytes_per_step": cs.get("bytes", 0) // max(cs.get("steps", 1), 1),
            "allreduce_calls_per_step": cs.get("calls", 0) / max(cs.get("steps", 1), 1),
            "bucket_mb": trainer.bucket_elems * 4 // (1 << 20),
            "exposed_comm_ms_per_step": {"min": round(min(float(c[0]) for c in allc), 3), "max": round(max(float(c[0]) for c in allc), 3),
                                         "what": "device time the launch stream idles between the last weight-gradient chunk and the end of the "
                                                 "last all-reduce (HIP events either side of Work.wait())"},
            "wait_host_ms_per_step": {"min": round(min(float(c[1]) for c in allc), 3), "max": round(max(float(c[1]) for c in allc), 3)},
            "overlap": os.environ.get("VPTR_DP_OVERLAP", "1") != "0",
        })
        import vptr_amd.train as _tr
        comm["dp_chunks"] = _tr.DP_CHUNKS
        # tuning table for the first real multi-GPU run: ONE extra, untimed step with HIP events behind every
        # weight-gradient chunk and behind every Work.wait(), per rank, in ms from the start of the exchange
        try:
            trainer.comm_stats = {"timeline": True}
            trainer.step(past, fut)
            torch.cuda.synchronize()
            tl = trainer.comm_stats.get("last_timeline")
            trainer.comm_stats = None
            mine_tl = None
            if tl is not None:
                mine_tl = {"rank": rank, "chunk_end_ms": [round(tl["t0"].elapsed_time(e), 3) for e in tl["chunk_end"]],
                           "bytes_released_by_chunk": tl["sent_bytes"],
                           "allreduce_done_ms": [round(tl["t0"].elapsed_time(e), 3) for e in tl["ar_done"]], "allreduce_bytes": tl["ar_bytes"]}
            if world > 1:
                alltl = [None] * world
                dist.all_gather_object(alltl, mine_tl)
            else:
                alltl = [mine_tl]
            comm["exchange_timeline"] = {
                "what": "one untimed step: HIP events on the launch stream, ms since the first weight-gradient chunk was enqueued; chunk_end = the chunk's "
                        "GEMM finished (its slab range goes out), allreduce_done = that piece's Work.wait() passed on the launch stream",
                "per_rank": alltl}
        except Exception as e:  # noqa
            trainer.comm_stats = None
            comm["exchange_timeline"] = {"error": str(e)[:200]}
        if world > 1 and comm["allreduce_bytes_per_step"] > 0:   # ring all-reduce moves 2 (n - 1) / n of the payload over each rank's links
            comm["bus_GBps_if_fully_exposed"] = None if comm["exposed_comm_ms_per_step"]["max"] <= 0 else round(
                comm["allreduce_bytes_per_step"] * 2.0 * (world - 1) / world / (comm["exposed_comm_ms_per_step"]["max"] * 1e-3) / 1e9, 1)
    loss = float(out["T_total"])
    terms = {k: round(float(v), 6) for k, v in out.items()}
    # the decoder ends in Tanh / Sigmoid and the targets are O(1): MSE <= 1.5, GDL <= 4, BiPatchNCE ~ ln(tokens per frame); anything else is a broken step
    loss_sane = bool(0.0 <= terms["T_MSE"] <= 1.5 and 0.0 <= terms["T_GDL"] <= 4.0 and 0.0 < terms.get("T_bpc", 1.0) < 8.0
                     and terms["grad_norm"] == terms["grad_norm"] and terms["grad_norm"] < 1e3)

    if rank == 0:
        ms = dt / args.steps * 1e3
        frames = job["frames"]
        value = world * args.batch * frames * args.steps / dt
        res = {
            "metric": job["metric"],
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16x3 (split-bf16 MFMA, fp32 accumulate/storage)" if args.precision == 3 else "bf16 (1-pass MFMA, fp32 accumulate/storage)",
            "data": "synthetic",
            "config": {"workload": workload, "name": args.config, "baseline_config": job["baseline_config"],
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch, "frames_per_clip": frames, "parallelism": "dp%d" % world,
                       "launch": graph_note, "dec_weight_grads": True,
                       # the frozen encoder's 18 ResnetBlock convolutions: Winograd F(4x4, 3x3) executes 25.7 of their 102.8 algorithmic GF each;
                       # alg_tflop_per_step_per_gpu below stays the reference's algorithmic count, roofline.all_gemm counts what is executed
                       "frozen_encoder_convs": "winograd F(4x4,3x3), 36 strided-batch P16 products per convolution" if ops.config.winograd else "direct implicit GEMM",
                       "alg_tflop_per_step_per_gpu": round(job["gf"] * args.batch / 1e3, 2),
                       "step_tflops_per_gpu": round(job["gf"] * args.batch / 1e3 / (ms * 1e-3), 1)},
            "per_rank_ms_per_step": {"min": round(min(per_rank) / args.steps * 1e3, 3), "max": round(max(per_rank) / args.steps * 1e3, 3)},
            "final_loss": round(loss, 5), "final_terms": terms, "loss_sane": loss_sane, "graph_check": graph_check,
        }
        if comm is not None:
            res["comm"] = comm
            res["rccl_ranks"] = comm["rccl_ranks"] if comm["backend"] == "nccl" else 0
        if not args.no_roofline:
            try:
                trainer._graph = trainer._front = None  # instrumented eager pass
                trainer.world = 1      # rank 0 alone runs it: no collective may be issued (the other ranks are at the final barrier)
                trainer.pg = None
                trainer._bufsync = None   # ... including the per-forward BatchNorm-buffer broadcast
                res["roofline"] = gemm_roofline(trainer, past, fut, args.precision)
                # the WHOLE step against the same roof: every algorithmic GEMM / convolution FLOP of one step over the timed step time
                # (memory-bound kernels, launch gaps and the optimizer all count against it)
                step_t = res["roofline"]["all_gemm"]["alg_gflop_per_step"] / 1e3 / (ms * 1e-3)
                ref_t = job["gf"] * args.batch / 1e3 / (ms * 1e-3)   # the reference's algorithmic FLOPs (direct convolutions) over the same time
                res["roofline"]["step"] = {"achieved": round(step_t, 1), "unit": "TFLOP/s", "frac": round(step_t / MFMA_PEAK_TFLOPS, 4),
                                           "frac_of_3pass_ceiling": round(step_t / (MFMA_PEAK_TFLOPS / 3.0), 4),
                                           "achieved_reference_flops": round(ref_t, 1), "frac_reference_flops": round(ref_t / MFMA_PEAK_TFLOPS, 4),
                                           "what": "EXECUTED GEMM + convolution FLOPs of one step (event-timed launches' 2MNK sum; the Winograd encoder executes "
                                                   "25.7 of the 102.8 algorithmic GF of each ResnetBlock convolution) / ms_per_step; *_reference_flops: the "
                                                   "reference's algorithmic count (config.alg_tflop_per_step_per_gpu) over the same time -- the figure that is "
                                                   "comparable across rounds"}
            except Exception as e:  # noqa
                res["roofline"] = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                                   "traffic": None, "error": str(e)[:200]}
        if world == 1 and not args.no_other_configs:
            trainer.opt.close()
            del trainer
            torch.cuda.empty_cache()
            res["other_configs"] = other_configs(dev, args.dropout, skip=(args.config,))
        if world == 1 and not args.no_cpu_baseline and args.config == "k64":
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
    if world > 1 or args.force_exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
