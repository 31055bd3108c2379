#!/bin/bash
D=gpurun_out/r4v; mkdir -p $D
python tools/bench_module_eval.py > $D/mod.json 2>$D/err.txt; cat $D/mod.json; tail -2 $D/err.txt
