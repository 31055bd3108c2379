#!/bin/bash
mkdir -p gpurun_out/r3c5
timeout 200 ./tools/ubench/mfma_raw_latency > gpurun_out/r3c5/ubench_raw_latency.txt 2>&1; cat gpurun_out/r3c5/ubench_raw_latency.txt
for m in 7; do
  GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_tail$m.so timeout 300 python tools/tail_repro.py --launches 200 > gpurun_out/r3c5/tail$m.txt 2>&1
  echo "== TAIL_MODE $m"; grep -v "amdgpu.ids" gpurun_out/r3c5/tail$m.txt | grep "wg_pairs=0\|TOTAL\|library"
done
