#!/bin/bash
# Diagnostic builds of the traced forward kernel (MINIBOONE geometry, 16-sample waves), timing only:
# tools/ablate/libgbnf_hip_tfwd_<name>.so with the weight DMA, the stage barrier, or both removed (results wrong).
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_tfwd; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_TRAIN=1 -DGBNF_V_ARGS=0,14,3,1,0,0,0,1"
hipcc $F -DGBNF_ABLATE_DMA -c variant_hx3.hip -o $OUT/dma.o &
hipcc $F -DGBNF_ABLATE_BARRIER -c variant_hx3.hip -o $OUT/bar.o &
hipcc $F -DGBNF_ABLATE_DMA -DGBNF_ABLATE_BARRIER -c variant_hx3.hip -o $OUT/both.o &
hipcc $F -DGBNF_ABLATE_DMA -DGBNF_ABLATE_BARRIER -DGBNF_ABLATE_FRAG -c variant_hx3.hip -o $OUT/all3.o &
wait
OBJS=$(ls obj/*.o | grep -v "v_hx3t_0_14_3_1_0_0_0_1.o")
for n in dma bar both all3; do hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_tfwd_$n.so $OBJS $OUT/$n.o; done
echo built
