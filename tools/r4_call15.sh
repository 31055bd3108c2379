#!/bin/bash
D=gpurun_out/r4t; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/$D/prof -o tr --output-format csv -- python3 $R/tools/bench_train.py --config hepmass_realnvp --batch 65536 --batch-stats --cpu-steps 0 --no-torch-legs --steps 20 > $R/$D/prof.log 2>&1
cd $R
f=$(find $D/prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:16]:
    print("%-100s calls=%5s avg_ns=%10.1f pct=%s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
