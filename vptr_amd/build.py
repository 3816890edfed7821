"""Builds libvptr_hip.so (hand-written HIP kernels, gfx950) in-tree with hipcc.

    python -m vptr_amd.build            # incremental, parallel per translation unit
    python -m vptr_amd.build --force

hipcc cross-compiles for gfx950 without a GPU present, so this also runs in the CPU-only build container.
The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libvptr_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_p16.hip", "norm.hip", "attn.hip", "attn_mfma.hip", "attn16.hip", "elementwise.hip", "conv7.hip", "losses.hip", "winograd.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize",
         "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    deps = [os.path.join(CSRC, src), os.path.join(CSRC, "common.h"), os.path.join(CSRC, "gemm_shared.h"), os.path.join(CSRC, "attn_mfma.h"),
            os.path.join(HERE, "..", "include", "vptr_hip.h")]
    dig = _digest(deps)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [_hipcc()] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout[-4000:], r.stderr[-8000:]))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
