#!/bin/bash
# instrumented build of the library (wall-clock stamps inside the nt P16 kernel): tools/_bin/libvptr_hip_timing.so, used through
# VPTR_HIP_LIB by tools/nt_timing.py.  Runs in the build container (hipcc cross-compiles gfx950).
cd "$(dirname "$0")/.." && mkdir -p tools/_bin/timing_obj
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result -DVPTR_P16_TIMING -Iinclude"
for s in api gemm gemm_p16 norm attn attn_mfma elementwise conv7; do
  /opt/rocm/bin/hipcc $FLAGS -c vptr_amd/csrc/$s.hip -o tools/_bin/timing_obj/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libvptr_hip_timing.so tools/_bin/timing_obj/*.o && echo built tools/_bin/libvptr_hip_timing.so
