#!/bin/bash
D=gpurun_out/r4f; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$D/prof -o img --output-format csv -- python3 $R/tools/bench_image.py --batch 256 --steps 5 --warmup 2 --cpu-seconds 0 --no-graph > $R/$D/prof.log 2>&1
cd $R
f=$(find $D/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-90s calls=%5s avg_ns=%10.1f pct=%s" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
