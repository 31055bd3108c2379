#!/bin/bash
D=gpurun_out/r3c15; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
for N in 4096 65536; do
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$D/prof$N -- python3 $GRAFT_REPO_ROOT/tools/bench_train.py --batch $N --cpu-steps 0 --steps 20 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/$D/prof$N -name "*kernel_stats.csv" | head -1); echo "== N=$N"; grep -i "gbnf" $f | cut -c1-110 | awk -F, '{print $1, $2, $4}'
done
