#!/bin/bash
# Diagnostic build: the MINIBOONE f16x3 kernel with absolute per-stage s_memtime stamps of workgroup 0 (-DGBNF_TIMELINE).
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_timeline; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_TIMELINE"
python3 build.py > /dev/null
for nt in 1 2; do
  hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_ARGS=0,14,3,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/h_$nt.o &
done
hipcc $F -c gbnf_api.hip -o $OUT/api.o
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_hip_timeline.so $OUT/api.o obj/gbnf_train.o obj/gbnf_image.o $OUT/h_1.o $OUT/h_2.o \
    obj/v_hx3_0_14_3_1_0_0_1_1.o obj/v_hx3_0_14_3_2_0_0_1_1.o
echo "built tools/libgbnf_hip_timeline.so"
