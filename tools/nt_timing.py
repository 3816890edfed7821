"""Where a workgroup of the nt P16 kernel spends its time: wall-clock stamps (100 MHz) at entry, first tile landed, end of the K loop,
end of the epilogue, per workgroup (instrumented build: tools/build_timing.sh, VPTR_HIP_LIB=tools/_bin/libvptr_hip_timing.so)."""
import ctypes, os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import vptr_amd.ops as ops
from vptr_amd._lib import GemmDesc, lib, ptr, stream

dev = torch.device("cuda:0")
ROT = 6
for (M, N, K, flags) in [(10240, 528, 528, "b"), (10240, 528, 528, "br"), (10240, 528, 2112, "b"), (10240, 2112, 528, "b")]:
    tiles = ((M + 127) // 128) * ((N + 175) // 176)
    sets = []
    for r in range(ROT):
        A = ops.to_p16(torch.randn(M, K, device=dev)); B = ops.to_p16(torch.randn(N, K, device=dev) * 0.05)
        D = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if "r" in flags else None
        tb = torch.zeros(tiles, 16, dtype=torch.int64, device=dev)
        d = GemmDesc()
        d.A, d.B, d.D, d.Dpre = ptr(A), ptr(B), ptr(D), ptr(tb)
        d.lda, d.ldb, d.ldd = K, K, N
        d.M, d.N, d.K = M, N, K
        d.a_mode, d.b_mode, d.precision, d.split_k, d.alpha = ops.A_P16, ops.B_P16, 3, 1, 1.0
        d.bias, d.residual, d.ldr = ptr(bias), ptr(res), (N if res is not None else 0)
        d.rs_div = d.rs_mod = 1
        sets.append((d, tb, A, B, D, bias, res))
    st = stream()
    for _ in range(3):
        for s in sets:
            assert lib.vptr_gemm(ctypes.byref(s[0]), st) == 0
    torch.cuda.synchronize()
    t = sets[-1][1].cpu().double() * 0.01          # us
    t0 = t[:, 0].min()
    start, first, loop, epi = t[:, 0] - t0, t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print("M %d N %d K %d '%s' tiles %d: kernel span %.1f us | start offset med %.1f max %.1f | to first tile med %.1f max %.1f | K loop med %.1f "
          "max %.1f (%.2f us/step) | epilogue med %.1f max %.1f | workgroup total med %.1f max %.1f | epilogue stamps (us after K loop): "
          "column operands %.1f, half 0 spilled %.1f, half 0 stores issued %.1f, half 1 spilled %.1f, half 1 stores issued %.1f" % (
              M, N, K, flags, tiles, float(t[:, 3].max() - t0), start.median(), start.max(), first.median(), first.max(), loop.median(), loop.max(),
              float(loop.median()) / ((K + 31) // 32), epi.median(), epi.max(), (t[:, 3] - t[:, 0]).median(), (t[:, 3] - t[:, 0]).max(),
              *[float((t[:, 4 + i] - t[:, 2]).median()) for i in range(5)]))
    raw = sets[-1][1].cpu().double()
    epi_cyc = raw[:, 11]
    ghz = float((epi_cyc / ((t[:, 3] - t[:, 2]) * 1e3)).median())     # shader-clock cycles per ns over the epilogue
    print("    wave 0 of each workgroup, K loop: waiting for the step (waitcnt + barrier) %.1f us, issuing DMA %.1f us of %.1f us  (shader clock %.2f GHz)"
          % (float(raw[:, 9].median()) / ghz / 1e3, float(raw[:, 10].median()) / ghz / 1e3, float(loop.median()), ghz))
