#!/bin/bash
# build the library from the CURRENT sources into tools/_bin/libvptr_<name>.so (A/B runs of two source states on one box via VPTR_HIP_LIB)
cd "$(dirname "$0")/.." && mkdir -p tools/_bin/var_obj_$1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result -Iinclude $2"
for s in api gemm gemm_p16 norm attn attn_mfma attn16 elementwise conv7 losses; do
  /opt/rocm/bin/hipcc $FLAGS -c vptr_amd/csrc/$s.hip -o tools/_bin/var_obj_$1/$s.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_bin/libvptr_$1.so tools/_bin/var_obj_$1/*.o && echo built tools/_bin/libvptr_$1.so
