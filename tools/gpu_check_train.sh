#!/bin/bash
# GPU box: the training tests, then the training-step rate at N = 4096 / 65536 (what every kernel change of round 3 was checked with)
D=gpurun_out/check_train; mkdir -p $D
( time timeout 900 python -m pytest tests/test_hip_train.py -q -m gpu -x ) > $D/pytest_train.txt 2>&1
echo "pytest rc $?"; tail -12 $D/pytest_train.txt
for N in 4096 65536; do python tools/bench_train.py --batch $N --cpu-steps 0 --steps 100 > $D/t_$N.json 2>/dev/null; python -c "
import json; d=json.loads(open('$D/t_$N.json').read().strip().splitlines()[-1]); print($N, round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done
