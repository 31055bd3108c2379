#!/bin/bash
# Diagnostic builds of the hx3 kernel (MINIBOONE geometry only) with one cost removed -- outputs are WRONG, timing only:
#   tools/ablate/libgbnf_hip_<what>.so   what = base | act | split | mfma | dma | barrier | frag | bias | combinations (a+b)
# usage: tools/build_ablate.sh [what ...]      then on the GPU box: python tools/ablate_bench.py
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_ablate; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1"
WHATS="${@:-base waves4 mfma dma barrier frag bias dma+barrier+frag+bias waves4+dma+barrier+frag+bias}"
python3 build.py > /dev/null          # obj/gbnf_api.o, gbnf_train.o, gbnf_image.o of the shipped build
for what in $WHATS; do
  defs=""
  if [ "$what" != base ]; then
    for w in ${what//+/ }; do
      if [ "$w" = waves4 ]; then defs="$defs -DGBNF_HX3_FORCE_WAVES=4"; else defs="$defs -DGBNF_ABLATE_$(echo $w | tr a-z A-Z)"; fi
    done
  fi
  for nt in 1 2; do
    hipcc $F $defs -DGBNF_V_ARGS=0,14,3,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/${what}_$nt.o &
  done
done
wait
for what in $WHATS; do
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_$what.so obj/gbnf_api.o obj/gbnf_train.o obj/gbnf_image.o \
      $OUT/${what}_1.o $OUT/${what}_2.o obj/v_hx3_0_14_3_1_0_0_1_1.o obj/v_hx3_0_14_3_2_0_0_1_1.o
done
echo "built: $WHATS"
