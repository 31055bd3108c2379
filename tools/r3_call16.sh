#!/bin/bash
D=gpurun_out/r3c16; mkdir -p $D
for T in "wg_pairs=-1" "wg_pairs=1"; do for N in 1024 4096 16384; do python tools/bench_train.py --batch $N --cpu-steps 0 --steps 100 --tune $T > $D/t.json 2>/dev/null; python -c "
import json; d=json.loads(open('$D/t.json').read().strip().splitlines()[-1]); print('$T', $N, round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done; done
