#!/bin/bash
# Diagnostic build: both flow kernels with in-kernel s_memtime phase stamps (-DGBNF_STAMPS).
# Never shipped, never timed end to end; read the SHARES it prints, not its run time.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_stamps; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -DGBNF_STAMPS"
python3 build.py > /dev/null          # obj/gbnf_train.o, gbnf_image.o and the bf16x6 repair variants of the shipped build
for nt in 1 2; do
  hipcc $F -DGBNF_V_ARGS=0,14,2,6,3,$nt,1,0,0 -c variant.hip -o $OUT/v_$nt.o &
  hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_ARGS=0,14,3,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/h_$nt.o &
done
hipcc $F -c gbnf_api.hip -o $OUT/api.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_hip_stamps.so $OUT/api.o obj/gbnf_train.o obj/gbnf_image.o $OUT/v_1.o $OUT/v_2.o $OUT/h_1.o $OUT/h_2.o \
    obj/v_hx3_0_14_3_1_0_0_1_1.o obj/v_hx3_0_14_3_2_0_0_1_1.o
echo "built tools/libgbnf_hip_stamps.so"
