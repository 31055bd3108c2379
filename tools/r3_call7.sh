#!/bin/bash
D=gpurun_out/r3c7; mkdir -p $D
( time timeout 1500 python -m pytest tests -q -m gpu ) > $D/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -8 $D/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$D/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --prewarm 0 --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > $GRAFT_REPO_ROOT/$D/trace_bench.json 2> $GRAFT_REPO_ROOT/$D/trace_bench.err
cd $GRAFT_REPO_ROOT
find $D/trace -name "*kernel_trace.csv" | head -2
python - <<'PY'
import csv,glob
fs=glob.glob('gpurun_out/r3c7/trace/**/*kernel_trace.csv', recursive=True)
rows=list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-60:]
t0=int(last[0]['Start_Timestamp'])
with open('gpurun_out/r3c7/timeline_tail.txt','w') as f:
    for r in last:
        f.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:10.1f} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} us  grid {r.get('Grid_Size','?'):>8} wg {r.get('Workgroup_Size','?'):>5}  {r['Kernel_Name'][:90]}\n")
print(open('gpurun_out/r3c7/timeline_tail.txt').read())
PY
