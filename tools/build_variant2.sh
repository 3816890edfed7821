#!/bin/bash
# A/B library with extra compile flags for gemm_p16.hip only, linked against the objects of the regular build (python -m vptr_amd.build first):
#   tools/build_variant2.sh <name> "<extra flags>"  ->  vptr_amd/_variants/libvptr_<name>.so (git-ignored; travels with gpurun; select with VPTR_HIP_LIB)
cd "$(dirname "$0")/.." && mkdir -p vptr_amd/_variants /tmp/var_obj_$1
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -Wno-unused-result $2"
/opt/rocm/bin/hipcc $FLAGS -c vptr_amd/csrc/gemm_p16.hip -o /tmp/var_obj_$1/gemm_p16.o || exit 1
OBJS=$(ls vptr_amd/csrc/_build/*.o | grep -v gemm_p16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o vptr_amd/_variants/libvptr_$1.so /tmp/var_obj_$1/gemm_p16.o $OBJS && echo built vptr_amd/_variants/libvptr_$1.so
