#!/usr/bin/env python
"""Does a hipMemsetAsync captured into a hipGraph (a memset node) take effect, in order, at every replay?  (root-cause probe for the
whole-step graph corruption: ATen's multi-block reductions zero their semaphores with cudaMemsetAsync -- Reduce.cuh -- and the
F.normalize backward of the BiPatchNCE branch left its output unwritten from the second replay on.)"""
import ctypes
import sys

import torch
import torch.nn.functional as F

dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int


def raw_stream():
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(0))


def probe_memset(nbytes, filler_ms):
    """graph: [optional long filler kernel] dirty sem (+1) -> memset(sem, 0) -> out = sem (copy kernel)"""
    sem = torch.zeros(nbytes // 4, device=dev, dtype=torch.int32)
    out = torch.full_like(sem, -1)
    big = torch.randn(64 << 20, device=dev) if filler_ms else None
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        if big is not None:
            big.mul_(1.0001)
        sem.add_(7)
        rc = hip.hipMemsetAsync(ctypes.c_void_p(sem.data_ptr()), 0, nbytes, raw_stream())
        assert rc == 0, rc
        out.copy_(sem)
    res = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append((int(out.abs().max()), int(sem.abs().max())))
    return res


def probe_normalize(N):
    """the failing piece itself: F.normalize(b, dim=2) backward with a ZERO upstream gradient must give zeros at every replay"""
    torch.manual_seed(0)
    base = torch.randn(N, 10, 8, 8, 528, device=dev, requires_grad=True)
    up = torch.zeros(N, 10, 528, 8, 8, device=dev)
    static_grad = torch.zeros_like(base)
    junk = torch.randn(64 << 18, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            y = F.normalize(base.permute(0, 1, 4, 2, 3), p=2.0, dim=2)
            (gr,) = torch.autograd.grad(y, base, up)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        t = junk * 3.0          # leaves non-zero data in pool blocks that later allocations of the capture reuse
        del t
        y = F.normalize(base.permute(0, 1, 4, 2, 3), p=2.0, dim=2)
        (gr,) = torch.autograd.grad(y, base, up)
        static_grad.copy_(gr)
        fill = [torch.full((n,), 5.0, device=dev) for n in (128, 1024, 4096, N * 640, N * 640 * 528)]   # dirties freed blocks for the next replay
        del fill
    res = []
    for _ in range(4):
        g.replay()
        torch.cuda.synchronize()
        res.append(float(static_grad.double().norm()))
    return res


if __name__ == "__main__":
    for nbytes in (4, 64, 512, 4096, 1 << 20):
        for filler in (0, 1):
            print("memset node %8d B filler %d -> (max|out|, max|sem|) per replay:" % (nbytes, filler), probe_memset(nbytes, filler), flush=True)
    for N in (1, 2, 3, 4, 16):
        print("F.normalize backward, zero upstream grad, N=%d -> |grad| per replay:" % N, probe_normalize(N), flush=True)
