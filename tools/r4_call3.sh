#!/bin/bash
D=gpurun_out/r4c; mkdir -p $D
python tools/ab_bench.py --names base,bias,barrier,dma,frag,act --rounds 2 --steps 1024 --extra "--math f16x3" > $D/ablate.txt 2>&1
tail -10 $D/ablate.txt
