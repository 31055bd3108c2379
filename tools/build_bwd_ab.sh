#!/bin/bash
# A/B builds of the backward kernel of given geometries: tools/build_bwd_ab.sh name "flags" [V_ARGS ...]  ->  tools/ablate/libgbnf_hip_<name>.so
# (every other object is the shipped build's).  Default geometries: MINIBOONE Glow (0,14,3,0,0,1) and HEPMASS RealNVP (1,7,1,0,0,1).
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
name="$1"; defs="$2"; shift 2
vargs=("$@"); [ ${#vargs[@]} -eq 0 ] && vargs=("0,14,3,0,0,1" "1,7,1,0,0,1")
OUT=/tmp/gbnf_bwd_ab; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1"
excl=""
objs_new=""
for va in "${vargs[@]}"; do
  tag=$(echo $va | tr ',' '_')
  hipcc $F $defs -DGBNF_V_ARGS=$va -c variant_bwd.hip -o $OUT/${name}_$tag.o &
  excl="$excl|v_hx3b_${tag}.o"
  objs_new="$objs_new $OUT/${name}_$tag.o"
done
wait
python3 ../../tools/isa_hazard_lint.py $objs_new | tail -3
objs=$(ls obj/*.o | grep -v -E "${excl#|}")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_$name.so $objs_new $objs -ldl
echo "built tools/ablate/libgbnf_hip_$name.so"
