#!/usr/bin/env python
"""bench.py -- predicted frames/sec of the VPTR-NAR train step (KTH 10->10 @ 64x64) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

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


def build_models(dev, dropout):
    import vptr_amd.model as M
    torch.manual_seed(3407)  # train_NAR_mp.py:278
    enc = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
    dec = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        M.init_weights(enc)
        M.init_weights(dec)
    T = M.VPTRFormerNAR(TP, TF, 8, 8, 528, 8, 4, 8, dropout, 4, 4, False, True)
    return enc.to(dev), dec.to(dev), T.to(dev)


def synth_batch(n, rank, dev):
    rs = np.random.RandomState(2021 + rank)
    past = (rs.uniform(0, 1, size=(n, TP, 1, 64, 64)).astype(np.float32) - 0.6013795) / 2.7570653  # utils/dataset.py:23
    fut = (rs.uniform(0, 1, size=(n, TF, 1, 64, 64)).astype(np.float32) - 0.6013795) / 2.7570653
    return torch.from_numpy(past).to(dev), torch.from_numpy(fut).to(dev)


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


def cpu_baseline(seconds_budget=45.0, final_steps=5):
    """The oracle (oracle/vptr_oracle.py: CPU restatement of the reference, parity-pinned by tests/golden) timed on the host
    cores of this box on a bounded sample of the same workload: N = 4 KTH-shaped clips.  A short thread sweep
    (1 warm-up + 1 timed step per point; points that do not fit the time budget are listed as skipped) picks the thread count, then
    `final_steps` steps are timed at that count and their mean is the reported figure."""
    from oracle import vptr_oracle as O
    import vptr_amd.model as M
    torch.manual_seed(3407)
    n = 4
    cfg = dict(Tp=TP, Tf=TF, H=8, W=8, C=528, nhead=8, window_size=4, num_encoder_layers=4, num_decoder_layers=8, rpe=True)
    enc = M.VPTREnc(1, 528, 3, "reflect")
    dec = M.VPTRDec(1, 528, 3, "Tanh", "reflect")
    T = M.VPTRFormerNAR(TP, TF, 8, 8, 528, 8, 4, 8, 0.0, 4, 4, False, True)
    st = O.NARStep(dict(enc.state_dict()), dict(dec.state_dict()), dict(T.state_dict()), cfg)
    past, fut = synth_batch(n, 0, "cpu")
    phys = _physical_cores()
    sweep, t_start = [], time.perf_counter()
    before = torch.get_num_threads()
    # two-socket hosts of this pool: the oracle's small fp32 GEMMs stop scaling near one quarter of the cores (32 threads 2.6 s/step,
    # 64 threads 6.2, all 128 cores 13.7 -- measured in round 3), so the sweep starts there and never spends the budget on the full count
    cand = [max(1, phys // 4), max(1, phys // 8), max(1, phys // 2)] if phys > 32 else [phys, max(1, phys // 2), max(1, phys // 4)]
    order = [k for i, k in enumerate(cand) if k not in cand[:i]]
    skipped = []
    sweep_budget = seconds_budget * 0.55
    for k in order:
        if sweep and time.perf_counter() - t_start + 2.0 * sweep[-1][1] > sweep_budget:   # warm-up + one timed step would not fit
            skipped.append(k)
            continue
        torch.set_num_threads(k)
        st.step(past, fut)                      # warm-up at this thread count
        t0 = time.perf_counter()
        st.step(past, fut)
        sweep.append((k, time.perf_counter() - t0))
    best = min(sweep, key=lambda kv: kv[1])
    torch.set_num_threads(best[0])
    times = []
    for _ in range(final_steps):
        t0 = time.perf_counter()
        st.step(past, fut)
        times.append(time.perf_counter() - t0)
    torch.set_num_threads(before)
    mean = sum(times) / len(times)
    return {"value": round(n * TF / mean, 3), "unit": "predicted frames/s", "cores": best[0], "kind": "port",
            "physical_cores": phys, "logical_cpus": os.cpu_count(), "cpu_model": _cpu_model(),
            "s_per_step": round(mean, 3), "s_per_step_min_max": [round(min(times), 3), round(max(times), 3)], "timed_steps": len(times),
            "thread_sweep": [{"threads": k, "s_per_step": round(t, 3)} for k, t in sweep], "thread_sweep_skipped": skipped,
            "sample": "oracle NAR train step (fp32 torch CPU), batch %d x 10->10 @64x64: thread sweep with 1 warm-up + 1 timed step per "
                      "point, then %d timed steps at the best count (%d threads): mean %.2f s/step" % (n, len(times), best[0], mean)}


def gemm_roofline(trainer, past, fut, precision):
    """Per-launch HIP-event timing of every GEMM launch of three train steps (instrumented passes, outside the timed region; per-step averages):
    returns the roofline entry of the dominant MFMA kernel instantiation and the GEMM-wide aggregate."""
    import vptr_amd.ops as ops
    NPASS = 3   # instrumented steps: the grouped tn kernel is launched once per step, one sample of it is too noisy (8.8 vs 9.7 ms seen)
    recs = []
    ops._gemm_prof = recs
    for _ in range(NPASS):
        trainer.step(past, fut)
    torch.cuda.synchronize()
    ops._gemm_prof = None
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

    def kernel_name(key):   # the name rocprofv3 lists the launch under
        nfn, prec, am, bm = key[:4]
        if am == 5:
            return "vptr_gemm_p16_kernel<%d, %d>" % (key[5], key[6])
        if am == 6:   # the end-of-backward launch into the gradient slab / the plain-store launches of token-range sub-problems
            if len(key) > 4 and key[4] == "grouped_sync":   # the panel-synchronous persistent launch (default)
                return "vptr_wgrad_p16_sync_kernel<%s>" % (os.environ.get("VPTR_WGRAD_SYNC", "16") if os.environ.get("VPTR_WGRAD_SYNC", "16") in ("8", "32") else "16")
            return "vptr_wgrad_p16_kernel<2, 1>" if (len(key) > 4 and key[4] == "grouped_split") else "vptr_wgrad_p16_kernel<2, 0>"
        if am == 3:
            return "vptr_conv_planes_kernel<true>"
        base = "vptr_gemm_grouped_kernel" if (len(key) > 4 and key[4] == "grouped") else ("vptr_gemm_kernel_p" if (len(key) > 5 and key[5] == "p") else "vptr_gemm_kernel")
        return "%s<%d, %d, %d, %d>" % (base, nfn, prec, am, bm)
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
        "kernel": kname,
        "launches_per_step": cnt, "avg_launch_us": round(ms * 1e3 / cnt, 2), "alg_gflop_per_launch": round(fl / cnt / 1e9, 3),
        "all_gemm": {"launches_per_step": sum(d[0] for d in by.values()), "ms_per_step": round(tot_ms, 3),
                     "achieved": round(tot_f / (tot_ms * 1e-3) / 1e12, 2), "alg_gflop_per_step": round(tot_f / 1e9, 1)},
        "per_kernel": {k: {"launches": d[0], "ms_per_step": round(d[2], 3), "achieved": round(d[1] / (d[2] * 1e-3) / 1e12, 1)}
                       for k, d in per_kernel.items()},
        "hbm_side": hbm_roofline(),
        "note": "algorithmic FLOPs = 2*M*N*K per launch (HIP events on the launch stream around every GEMM launch of three instrumented "
                "steps, averaged); the split-bf16 kernels issue 3 bf16 MFMA passes per algorithmic FLOP (fp32-class accuracy), so their ceiling "
                "is peak/3 = 833 TFLOP/s, and ~480 TFLOP/s under the MFMA power envelope (DESIGN.md section 4)",
    }


HBM_PEAK_GBS = 8000.0


def hbm_roofline():
    """achieved GB/s of the largest HBM-bound pass of the step, timed with HIP events on its own (20 launches): the LayerNorm((F,H,W))
    + GELU normalisation of a conv-FFN hidden tensor [10 240 x 2112] fp32: algorithmic bytes = read x + write y = 2 x 86.5 MB."""
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
            # statistics pass + normalise pass per call; time the pair
            e0.record()
            for _ in range(20):
                ops.norm_act(x, w, b, "ln", HW, True)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        nbytes = 3.0 * rows * F * 4      # statistics pass reads x; normalise pass reads x and writes y
        return {"kernels": "vptr_groupstats + vptr_norm_act_fwd (LayerNorm((2112,8,8)) + GELU on [10240 x 2112] fp32)", "us_per_call": round(us, 2),
                "alg_bytes": int(nbytes), "infinity_cache_rate": round(nbytes / us / 1e3, 1), "unit": "GB/s", "hbm_peak": HBM_PEAK_GBS,
                "note": "NOT an HBM fraction: the 86.5 MB tensor fits the 256 MB Infinity Cache, its re-reads are served on-die; the figure is the "
                        "rate the largest elementwise pass of the step sustains on the L2 / MALL side"}
    except Exception as ex:  # noqa
        return {"error": str(ex)[:160]}


def _time_steps(step, warm=2, timed=3):
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


def other_configs(dev, enc, T_k64, trainer, dropout):
    """ms/step of BASELINE.json's other configurations on this GPU (rank 0, after the timed region, 2 warm-up + 3 timed steps each;
    reported, never part of `value`): config 2 (MNIST NAR: the K64 shapes with a Sigmoid decoder), config 4 (BAIR FAR at its literal
    size: 3-channel zero-padded auto-encoder, VPTRFormerFAR(2, 28), T_in = 29, per-GPU batch 16 = train_FAR_mp.py's 64 over 4 GPUs)
    and config 5 (KTH 128x128 10 -> 40, 16x16 features, 8x8 windows, per-GPU batch 2)."""
    import contextlib
    import io
    import vptr_amd.model as M
    import vptr_amd.ops as ops
    from vptr_amd.train import FARTrainer, NARTrainer
    out = {}

    def init(*mods):
        with contextlib.redirect_stdout(io.StringIO()):
            for m in mods:
                M.init_weights(m)

    def rnd(*shape):
        return torch.rand(shape, device=dev) * 0.36 - 0.22
    # ---- config 2: same transformer / slab, Sigmoid decoder
    try:
        dec2 = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Sigmoid", padding_type="reflect")
        init(dec2)
        dec2 = dec2.to(dev).eval()
        for p in dec2.parameters():
            p.requires_grad_(True)
        for mod in dec2.modules():
            mod._vptr_frozen = True
        keep = trainer.dec
        trainer.dec = dec2
        past, fut = rnd(PER_GPU_BATCH, TP, 1, 64, 64), torch.rand((PER_GPU_BATCH, TF, 1, 64, 64), device=dev)
        ms = _time_steps(lambda: trainer.step(past, fut))
        trainer.dec = keep
        out["config2_mnist_nar_sigmoid"] = {"ms_per_step": round(ms, 2), "per_gpu_batch": PER_GPU_BATCH,
                                            "frames_per_s": round(PER_GPU_BATCH * TF / ms * 1e3, 1)}
        del dec2
    except Exception as e:  # noqa
        out["config2_mnist_nar_sigmoid"] = {"error": str(e)[:160]}
    # ---- config 4: BAIR FAR 2 -> 28
    try:
        n = 16
        enc4 = M.VPTREnc(3, feat_dim=528, n_downsampling=3, padding_type="zero")
        dec4 = M.VPTRDec(3, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="zero")
        init(enc4, dec4)
        far = M.VPTRFormerFAR(2, 28, 8, 8, 528, 8, 12, dropout, 4, 4, True)
        tr4 = FARTrainer(enc4.to(dev), dec4.to(dev), far.to(dev), lr=1e-4, max_grad_norm=1.0)
        past, fut = rnd(n, 2, 3, 64, 64), rnd(n, 28, 3, 64, 64)
        ms = _time_steps(lambda: tr4.step(past, fut))
        out["config4_bair_far_2to28"] = {"ms_per_step": round(ms, 2), "per_gpu_batch": n, "T_in": 29,
                                          "frames_per_s": round(n * 29 / ms * 1e3, 1)}
        del tr4, far, enc4, dec4
    except Exception as e:  # noqa
        out["config4_bair_far_2to28"] = {"error": str(e)[:160]}
    torch.cuda.empty_cache()
    # ---- config 5: KTH 128x128 10 -> 40
    try:
        n = 2
        enc5 = M.VPTREnc(1, feat_dim=528, n_downsampling=3, padding_type="reflect")
        dec5 = M.VPTRDec(1, feat_dim=528, n_downsampling=3, out_layer="Tanh", padding_type="reflect")
        init(enc5, dec5)
        nar5 = M.VPTRFormerNAR(10, 40, 16, 16, 528, 8, 4, 8, dropout, 8, 4, False, True)
        tr5 = NARTrainer(enc5.to(dev), dec5.to(dev), nar5.to(dev), batch_size=n, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1)
        past, fut = rnd(n, 10, 1, 128, 128), rnd(n, 40, 1, 128, 128)
        ms = _time_steps(lambda: tr5.step(past, fut))
        out["config5_kth128_nar_10to40"] = {"ms_per_step": round(ms, 2), "per_gpu_batch": n,
                                            "frames_per_s": round(n * 40 / ms * 1e3, 1)}
        del tr5, nar5, enc5, dec5
    except Exception as e:  # noqa
        out["config5_kth128_nar_10to40"] = {"error": str(e)[:160]}
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
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch (weak scaling)")
    ap.add_argument("--precision", type=int, default=int(os.environ.get("VPTR_GEMM_PRECISION", "3")), choices=[1, 3],
                    help="3 = split-bf16 MFMA (meets the 1e-3 parity bar, default); 1 = single-pass bf16")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--graph", type=int, default=int(os.environ.get("VPTR_GRAPH", "1")),
                    help="1 (default): capture the whole step in one hipGraph on a single GPU (same kernels, no host launch cost: "
                         "57.2 vs 61.3 ms on one box); 0: eager.  Multi-rank runs are always eager (RCCL calls stay outside graphs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the ms/step lines of BASELINE configs 2 / 4 / 5")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch clips per GPU; strong: the reference's DistributedSampler semantics, a fixed "
                         "--global-batch split as global // world per rank (utils/dataset.py:72)")
    ap.add_argument("--force-exchange", action="store_true",
                    help="one GPU only: bring up a ONE-rank RCCL process group and run the step through the multi-rank code path (chunked "
                         "weight-gradient launches + asynchronous all-reduces of the 473.5 MB gradient slab on c10d's RCCL stream); eager")
    ap.add_argument("--global-batch", type=int, default=64, help="global batch of --scaling strong (train_FAR_mp.py:300 uses 64)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # VPTR_BENCH_SHARE_GPU=1 + VPTR_BENCH_BACKEND=gloo: functional check of the multi-rank path on a ONE-GPU box (every rank
    # on cuda:0, gradient exchange through gloo); never set by the driver -- its runs use one GPU per rank over RCCL
    share = os.environ.get("VPTR_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("VPTR_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    pg = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
        pg = torch.distributed.group.WORLD
    elif args.force_exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ["VPTR_DP_FORCE_EXCHANGE"] = "1"
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        pg = torch.distributed.group.WORLD

    import vptr_amd.ops as ops
    from vptr_amd.train import NARTrainer
    if args.scaling == "strong":
        from vptr_amd.parallel import shard_batch
        args.batch = shard_batch(args.global_batch, rank, world)[1]
    ops.config.gemm_precision = args.precision
    enc, dec, T = build_models(dev, args.dropout)
    if world > 1:  # identical replicas: broadcast rank 0's parameters and buffers (what the DDP constructor does), one message per dtype
        from vptr_amd.parallel import broadcast_modules_flat
        broadcast_modules_flat([T, enc, dec], 0, pg)
    trainer = NARTrainer(enc, dec, T, batch_size=args.batch, lr=1e-4, max_grad_norm=1.0, lam_pc=0.1, process_group=pg)
    past, fut = synth_batch(args.batch, rank, dev)

    # one GPU: the whole step is one hipGraph.  Several ranks (or --force-exchange): forward + backward are one hipGraph, the part that
    # talks to other ranks (grouped weight-gradient chunks, RCCL all-reduces, optimizer) stays eager -- no collective is captured
    dp = world > 1 or args.force_exchange
    use_graph = bool(args.graph)
    graph_note = "eager"
    graph_check = None
    if use_graph:
        try:
            if dp:
                trainer.capture_front(past, fut, warmup=2)
            else:
                trainer.capture(past, fut, warmup=2)
            # the graph is only timed if it computes what the eager step computes: replay i vs an eager step from the same state, and a
            # run of replays vs the eager trajectory (loss terms, gradient norm, post-step parameters); the state is restored afterwards
            ok, graph_check = trainer.verify_graph(past, fut, steps=3, rtol=2e-3)
            graph_check["nodes"] = trainer.graph_nodes
            if world > 1:   # every rank takes the same branch
                flag = torch.tensor([1.0 if ok else 0.0], device=dev)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                ok = bool(flag.item() > 0.5)
            if ok:
                graph_note = "hipGraph (forward + backward) + eager exchange / optimizer" if dp else "hipGraph"
            else:
                trainer._graph = trainer._front = None
                graph_note = "eager (hipGraph replays disagree with eager steps: %s)" % (graph_check.get("worst_term"),)
        except Exception as e:  # noqa: keep the bench alive, report eager numbers
            trainer._graph = trainer._front = None
            graph_note = "eager (graph capture failed: %s)" % str(e).split("\n")[0][:160]
    if args.force_exchange:
        graph_note += " [one-rank RCCL group, forced gradient exchange]"

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(past, fut)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(past, fut)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(out["T_total"])
    terms = {k: round(float(v), 6) for k, v in out.items()}
    # the decoder ends in Tanh and the targets lie in [-0.22, 0.15]: MSE <= 1.5, GDL <= 4, BiPatchNCE ~ ln 64; anything else is a broken step
    loss_sane = bool(0.0 <= terms["T_MSE"] <= 1.5 and 0.0 <= terms["T_GDL"] <= 4.0 and 0.0 < terms["T_bpc"] < 8.0
                     and terms["grad_norm"] == terms["grad_norm"] and terms["grad_norm"] < 1e3)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * args.batch * TF * args.steps / dt
        res = {
            "metric": "predicted frames/sec (train step) NAR KTH 10->10 @64x64",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16x3 (split-bf16 MFMA, fp32 accumulate/storage)" if args.precision == 3 else "bf16 (1-pass MFMA, fp32 accumulate/storage)",
            "data": "synthetic",
            "config": {"workload": "K64: KTH 64x64x1 10->10, VPTREnc/Dec(528, Tanh, reflect) + VPTRFormerNAR(4 enc + 8 dec, d=528, 8 heads, "
                                   "ws 4, dropout %.2f), single_iter of train_NAR.py, random-init weights" % args.dropout,
                       "per_gpu_batch": args.batch, "global_batch": world * args.batch, "parallelism": "dp%d" % world,
                       "launch": graph_note, "dec_weight_grads": True,
                       "alg_tflop_per_step_per_gpu": round(GF_PER_SAMPLE * args.batch / 1e3, 2),
                       "step_tflops_per_gpu": round(GF_PER_SAMPLE * args.batch / 1e3 / (ms * 1e-3), 1)},
            "final_loss": round(loss, 5), "final_terms": terms, "loss_sane": loss_sane, "graph_check": graph_check,
        }
        if not args.no_roofline:
            try:
                trainer._graph = trainer._front = None  # instrumented eager pass
                trainer.world = 1      # rank 0 alone runs it: no collective may be issued (the other ranks are at the final barrier)
                trainer.pg = None
                trainer._bufsync = None   # ... including the per-forward BatchNorm-buffer broadcast
                res["roofline"] = gemm_roofline(trainer, past, fut, args.precision)
            except Exception as e:  # noqa
                res["roofline"] = {"bound": "mfma", "achieved": None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                                   "traffic": None, "error": str(e)[:200]}
        if world == 1 and not args.no_other_configs:
            res["other_configs"] = other_configs(dev, enc, T, trainer, args.dropout)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res))
    if world > 1:
        torch.distributed.barrier()
    if world > 1 or args.force_exchange:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
