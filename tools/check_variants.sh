#!/bin/bash
# Register / scratch use of every compiled split-kernel variant (reads the device code objects inside csrc/obj/*.o).
# A variant with scratch > 0 spills: its hx3_waves() estimate (gbnf_flow_kernel_hx3.hip.h) is too optimistic.
OBJ="$(dirname "$0")/../gradient-boosted-normalizing-flows_amd/csrc/obj"
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
bad=0
for f in "$OBJ"/v_hx3_*.o; do
  $LLVM/llvm-objcopy --dump-section .hip_fatbin=$T/fb.bin "$f" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --type=o --input=$T/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --unbundle --output=$T/dev.co 2>/dev/null || continue
  line=$($LLVM/llvm-readelf --notes $T/dev.co | awk '/\.private_segment_fixed_size:/{p=$2} /\.vgpr_count:/{v=$2} /\.max_flat_workgroup_size:/{w=$2} /\.sgpr_count:/{s=$2} END{printf "vgpr %d sgpr %d scratch %d threads %d", v, s, p, w}')
  echo "$(basename $f) $line"
  case "$line" in *"scratch 0 "*) ;; *) bad=$((bad+1));; esac
done
echo "variants with scratch: $bad"
rm -rf $T
