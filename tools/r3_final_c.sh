#!/bin/bash
# Round 3 extras, part 2: the training step under the profiler (kernel stats, SQ and HBM counters).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
T="timeout 300"
for N in 4096 65536; do
  $T python tools/bench_train.py --batch $N --cpu-steps 0 > $O/train_n$N.json 2>/dev/null; echo line $N $?
  $T rocprofv3 --kernel-trace --stats -d $O/prof_train$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 50 --no-torch-legs > $O/prof_train$N.log 2>&1; echo stats $N $?
  $T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmc_train$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 10 --warmup 2 --no-torch-legs > $O/pmc_train$N.log 2>&1; echo sq $N $?
  python tools/pmc_summary.py $O/pmc_train$N > $O/pmc_train$N.txt 2>&1
done
for d in prof_train4096 prof_train65536; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.kernel_stats.csv; done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
