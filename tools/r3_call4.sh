#!/bin/bash
mkdir -p gpurun_out/r3c4
for m in "$@"; do
  GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_tail$m.so timeout 300 python tools/tail_repro.py --launches 200 > gpurun_out/r3c4/tail$m.txt 2>&1
  echo "== TAIL_MODE $m"; grep -v "amdgpu.ids" gpurun_out/r3c4/tail$m.txt | grep "wg_pairs=0\|TOTAL\|library"
done
