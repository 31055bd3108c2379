#!/bin/bash
D=gpurun_out/r3c19; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --batch 65536 --cpu-steps 0 --steps 20 --no-torch-legs > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$D/prof -name "*kernel_stats.csv" | head -1); grep gbnf $f | cut -c1-150
