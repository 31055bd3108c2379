#!/bin/bash
for m in 1 2 4 7; do
  export GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_bwd$m.so
  for N in 4096 65536; do python tools/bench_train.py --batch $N --cpu-steps 0 --steps 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate $m N $N', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done
done
