#!/bin/bash
# Diagnostic build of the latency-form kernel (MINIBOONE geometry) with in-kernel s_memtime phase stamps (-DGBNF_STAMPS) -> tools/ablate/libgbnf_hip_coop_stamps.so
# (never shipped; read the shares tools/coop_stamps.py prints).  usage: tools/build_coop_stamps.sh ["extra flags" [name [stamp flag, "" = an A/B build without stamps]]]
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
NAME=${2:-coop_stamps}
OUT=/tmp/gbnf_$NAME; mkdir -p $OUT ../../tools/ablate
STAMPS=${3--DGBNF_STAMPS}
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm $STAMPS $1"
python3 build.py > /dev/null
for nt in 1 2 3 4; do
  hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_ARGS=0,14,3,$nt,0,0 -c variant_coop.hip -o $OUT/c_$nt.o &
done
hipcc $F -c gbnf_api.hip -o $OUT/api.o &
wait
objs=$(ls obj/*.o | grep -v "v_coop_0_14_3_[1234]_0_0.o" | grep -v "obj/gbnf_api.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libgbnf_hip_$NAME.so $OUT/api.o $OUT/c_1.o $OUT/c_2.o $OUT/c_3.o $OUT/c_4.o $objs -ldl
echo "built tools/ablate/libgbnf_hip_$NAME.so"
