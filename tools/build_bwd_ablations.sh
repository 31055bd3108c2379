#!/bin/bash
# Diagnostic builds of the training path's backward kernel (MINIBOONE geometry), timing only: tools/ablate/libgbnf_hip_bwd<mask>.so
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_bwd; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1"
for m in "$@"; do hipcc $F -DGBNF_BWD_ABLATE=$m -DGBNF_V_ARGS=0,14,3,0,0 -c variant_bwd.hip -o $OUT/b$m.o & done; wait
OBJS=$(ls obj/*.o | grep -v "v_hx3b_0_14_3_0_0.o")
for m in "$@"; do hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_bwd$m.so $OBJS $OUT/b$m.o; done
echo built $@
