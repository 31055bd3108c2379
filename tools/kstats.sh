#!/bin/bash
# rocprofv3 kernel stats of one command, printed as a table: tools/kstats.sh <name> <python3 script args...>  (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; shift
O=gpurun_out/kstats_$name; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O -o s --output-format csv -- "$@" > $O/log.txt 2>&1
f=$(find $O -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print("%-64s calls %6s avg %9.1f us  %5.1f %%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
