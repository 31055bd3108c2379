#!/bin/bash
D=gpurun_out/r4a; mkdir -p $D
python tools/ab_bench.py --names base,mixlo,runs,edge,dmaat2,all3,all4 --rounds 3 --check > $D/ab.txt 2>&1
tail -12 $D/ab.txt
for n in tl_base tl_all3; do
  GBNF_NO_WG_PAIRS=1 GBNF_FORCE_NT=2 GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_$n.so python tools/timeline.py > $D/${n}_8wave.txt 2>&1
  GBNF_FORCE_NT=2 GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_$n.so python tools/timeline.py > $D/${n}_4wave.txt 2>&1
  grep "stage durations" $D/${n}_8wave.txt $D/${n}_4wave.txt
done
