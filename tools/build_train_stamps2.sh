#!/bin/bash
# Diagnostic build of the register-chained training kernels (MINIBOONE geometry) with s_memtime phase stamps:
# tools/libgbnf_hip_tstamps.so.  Never shipped; read the SHARES tools/train_stamps2.py prints.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_tstamps; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_STAMPS"
hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_ARGS=0,14,3,0,0 -c variant_bwd.hip -o $OUT/b.o &
hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_TRAIN=1 -DGBNF_V_ARGS=0,14,3,1,0,0,0,1 -c variant_hx3.hip -o $OUT/t.o &
hipcc $F -c gbnf_api.hip -o $OUT/api.o &
wait
OBJS=$(ls obj/*.o | grep -v "v_hx3b_0_14_3_0_0.o\|v_hx3t_0_14_3_1_0_0_0_1.o\|gbnf_api.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_hip_tstamps.so $OBJS $OUT/b.o $OUT/t.o $OUT/api.o
echo "built tools/libgbnf_hip_tstamps.so"
