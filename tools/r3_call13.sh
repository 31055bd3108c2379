#!/bin/bash
D=gpurun_out/r3c13; mkdir -p $D
for rep in 1 2; do
for lib in new oldwgrad; do
  if [ $lib = new ]; then unset GBNF_LIB_PATH; else export GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_oldwgrad.so; fi
  for N in 4096 65536; do python tools/bench_train.py --batch $N --cpu-steps 0 --steps 100 > $D/t_${lib}_$N.json 2>/dev/null; python -c "
import json; d=json.loads(open('$D/t_${lib}_$N.json').read().strip().splitlines()[-1]); print('$lib', $N, round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done
done; done
