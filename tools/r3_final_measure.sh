#!/bin/bash
# Round 3: everything tools/final_measure.sh measures, plus the driver's own invocation under the profiler, the emulated
# 8-GPU per-rank load at that invocation, the module evaluate loop and the training step (kernel stats + PMC).  GPU box only.
bash tools/final_measure.sh > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --force-gather --components 1 --no-extra-legs --cpu-seconds 0 > $O/bench_emulated_c1_steps20.json 2>/dev/null
python bench.py --force-gather --components 1 --no-extra-legs --cpu-seconds 0 > $O/bench_emulated_c1.json 2>/dev/null
python tools/bench_module_eval.py > $O/module_eval.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/prof_steps20 -o s20 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs > $O/prof_steps20.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/prof_module -o mod --output-format csv -- python3 tools/bench_module_eval.py > $O/prof_module.log 2>&1
for N in 4096 65536; do
  rocprofv3 --kernel-trace --stats -d $O/prof_train$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 50 > $O/prof_train$N.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmc_train$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 20 > $O/pmc_train$N.log 2>&1
  python tools/pmc_summary.py $O/pmc_train$N > $O/pmc_train$N.txt 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $O/pmc_train_hbm$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 20 > $O/pmc_train_hbm$N.log 2>&1
  python tools/pmc_summary.py $O/pmc_train_hbm$N > $O/pmc_train_hbm$N.txt 2>&1
done
for d in prof_steps20 prof_module prof_train4096 prof_train65536; do f=$(find $O/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$d.kernel_stats.csv; done
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
ls $O | head -80
