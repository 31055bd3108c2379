#!/bin/bash
# Diagnostic build: the MINIBOONE f16x3 kernel with one s_memtime / s_memrealtime pair around every workgroup (-DGBNF_CLOCK):
# the clock the chip holds under this kernel (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Read by tools/clock_probe.py.
set -e
cd "$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc"
OUT=/tmp/gbnf_clock; mkdir -p $OUT
F="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-value -Wno-inline-asm -DGBNF_CLOCK"
python3 build.py > /dev/null
for nt in 1 2; do
  hipcc $F -mllvm -amdgpu-mfma-vgpr-form=1 -DGBNF_V_ARGS=0,14,3,$nt,0,0,0,1 -c variant_hx3.hip -o $OUT/h_$nt.o &
done
hipcc $F -c gbnf_api.hip -o $OUT/api.o
wait
objs=$(ls obj/*.o | grep -v "v_hx3_0_14_3_[12]_0_0_0_1.o" | grep -v "obj/gbnf_api.o")      # every other object of the shipped build
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libgbnf_hip_clock.so $OUT/api.o $OUT/h_1.o $OUT/h_2.o $objs -ldl
echo "built tools/libgbnf_hip_clock.so"
