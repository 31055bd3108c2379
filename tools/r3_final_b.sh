#!/bin/bash
# Round 3 extras, part 1: the driver's invocation (line + profile), emulated per-rank lines, the module evaluate loop.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
T="timeout 420"
$T python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null; echo steps20 $?
$T python bench.py --steps 20 --warmup 5 --force-gather --components 1 --no-extra-legs --cpu-seconds 0 > $O/bench_emulated_c1_steps20.json 2>/dev/null; echo emu20 $?
$T python bench.py --force-gather --components 1 --no-extra-legs --cpu-seconds 0 > $O/bench_emulated_c1.json 2>/dev/null; echo emu $?
$T python tools/bench_module_eval.py > $O/module_eval.json 2>/dev/null; echo module $?
$T rocprofv3 --kernel-trace --stats -d $O/prof_steps20 -o s20 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs > $O/prof_steps20.log 2>&1; echo prof20 $?
$T rocprofv3 --kernel-trace --stats -d $O/prof_module -o mod --output-format csv -- python3 tools/bench_module_eval.py > $O/prof_module.log 2>&1; echo profmod $?
for d in prof_steps20 prof_module; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.kernel_stats.csv; done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
