#!/bin/bash
D=gpurun_out/r3c11; mkdir -p $D
( time timeout 900 python -m pytest tests/test_hip_train.py tests/test_hip_module.py -q -m gpu -x ) > $D/pytest_train.txt 2>&1
echo "pytest rc $?"; tail -8 $D/pytest_train.txt
for N in 4096 65536; do timeout 300 python tools/bench_train.py --batch $N --cpu-steps 0 > $D/train_n$N.json 2> $D/train_n$N.err; python -c "
import json; d=json.loads(open('$D/train_n$N.json').read().strip().splitlines()[-1]); print($N, round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$D/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --batch 65536 --cpu-steps 0 --steps 20 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; python tools/summarize_prof.py $D/prof 2>/dev/null | head -12 || find $D/prof -name "*stats*" | head
