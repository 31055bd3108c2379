#!/bin/bash
# Experimental builds of the split kernel that showed the round-2 tile corruption (glow HT=8 OT=4, f16x3), one library per
# GBNF_HX3_TAIL_MODE, for tools/tail_repro.py:   tools/ablate/libgbnf_hip_tail<mode>.so
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_tail; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1"
MODES="${@:-0 1 2 3}"
for m in $MODES; do
  for nt in 1 2; do
    hipcc $F -DGBNF_HX3_TAIL_MODE=$m -DGBNF_V_ARGS=0,8,4,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/tail${m}_$nt.o &
  done
done
wait
for m in $MODES; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_tail$m.so obj/gbnf_api.o obj/gbnf_train.o obj/gbnf_image.o \
      $OUT/tail${m}_1.o $OUT/tail${m}_2.o obj/v_hx3_0_8_4_1_0_0_1_1.o obj/v_hx3_0_8_4_2_0_0_1_1.o obj/v_0_8_4_8_4_1_1_0_0.o obj/v_0_8_4_8_4_2_1_0_0.o
done
echo "built tail modes: $MODES"
