#!/bin/bash
mkdir -p gpurun_out/r3c3
timeout 120 ./tools/ubench/mfma_srcc_war_dsta > gpurun_out/r3c3/ubench_srcc_war_dsta.txt 2>&1
cat gpurun_out/r3c3/ubench_srcc_war_dsta.txt
for m in 0 1 2 3; do
  GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_tail$m.so timeout 300 python tools/tail_repro.py --launches 400 > gpurun_out/r3c3/tail$m.txt 2>&1
  echo "== TAIL_MODE $m"; cat gpurun_out/r3c3/tail$m.txt | tail -15
done
