#!/bin/bash
# A/B builds of the MINIBOONE-geometry f16x3 kernel (hx3 0 14 3): one library per "name=flags" argument, everything else is the
# shipped build.  Results stay CORRECT (these are candidate kernels, not ablations); time them with tools/ab_bench.py on ONE box.
#   tools/build_ab.sh base= mixlo=-DGBNF_HX3_MIXLO=1 "all=-DGBNF_HX3_MIXLO=1 -DGBNF_HX3_EDGE=1"
# A name that starts with tl_ is also compiled with -DGBNF_TIMELINE (tools/timeline.py reads its stage stamps).
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_ab; mkdir -p $OUT ../../tools/ablate
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -mllvm -amdgpu-mfma-vgpr-form=1"
python3 build.py > /dev/null
names=()
for arg in "$@"; do
  name="${arg%%=*}"; defs="${arg#*=}"; names+=("$name")
  tl=""; api=obj/gbnf_api.o
  if [[ "$name" == tl_* ]]; then tl="-DGBNF_TIMELINE"; fi
  for nt in 1 2; do
    hipcc $F $defs $tl -DGBNF_V_ARGS=0,14,3,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/${name}_$nt.o &
  done
done
if [ ! -f $OUT/api_tl.o ] || [ gbnf_api.hip -nt $OUT/api_tl.o ]; then
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_TIMELINE -c gbnf_api.hip -o $OUT/api_tl.o &
fi
wait
for name in "${names[@]}"; do
  api=obj/gbnf_api.o; [[ "$name" == tl_* ]] && api=$OUT/api_tl.o
  python3 ../../tools/isa_hazard_lint.py $OUT/${name}_1.o $OUT/${name}_2.o > $OUT/${name}.lint 2>&1 || { echo "LINT FAILED for $name"; tail -5 $OUT/${name}.lint; }
  objs=$(ls obj/*.o | grep -v "v_hx3_0_14_3_[12]_0_0_0_1.o" | grep -v "obj/gbnf_api.o")      # every other object of the shipped build
  hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_$name.so $api $OUT/${name}_1.o $OUT/${name}_2.o $objs -ldl
done
echo "built: ${names[*]}"
