#!/bin/bash
# Diagnostic builds of the hx3 kernel with one cost removed (outputs are WRONG; timing only):
#   tools/libgbnf_hip_ablate_{act,split,mfma}.so
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_ablate; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1"
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c gbnf_api.hip -o $OUT/api.o &
for what in ACT SPLIT MFMA; do
  for nt in 1 2; do
    hipcc $F -DGBNF_ABLATE_$what -DGBNF_V_ARGS=0,14,3,$nt,0,0 -c variant_hx3.hip -o $OUT/${what}_$nt.o &
  done
done
wait
for what in ACT SPLIT MFMA; do
  lw=$(echo $what | tr A-Z a-z)
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_hip_ablate_$lw.so $OUT/api.o $OUT/${what}_1.o $OUT/${what}_2.o
done
echo built
