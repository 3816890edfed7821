"""Sustained-load check (GPU box): the same GEMM back-to-back for ~6 s, time per call in 0.25 s windows, with the
clocks rocm-smi reports before / during / after -- separates power / clock management from cache or context effects."""
import os, sys, subprocess, time
import torch
R_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R_)
import vptr_amd.ops as ops

dev = torch.device("cuda:0")
M, N, K = 10240, int(os.environ.get("N", 528)), int(os.environ.get("K", 528))
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); D = torch.empty(M, N, device=dev)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "mclk" in l or "Power" in l or "fclk" in l)]
        return " | ".join(keep)[:400]
    except Exception as e:  # noqa
        return "rocm-smi unavailable: %r" % e


print("idle:", smi())
n = 2000
t_end = time.time() + float(os.environ.get("SECS", 6))
w = 0
while time.time() < t_end:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.gemm_raw(A, W, D, M, N, K, 0, 0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print("window %2d: %6.1f us/call %6.1f TF/s" % (w, us, 2.0 * M * N * K / us / 1e6))
    if w in (3, 40):
        print("  load:", smi())
    w += 1
