#!/bin/bash
# HBM counters of the training step, FETCH_SIZE and WRITE_SIZE in separate passes (see tools/r3_image_pmc.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
T="timeout 200"
for N in 4096 65536; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    $T rocprofv3 --kernel-trace --pmc $CNT -d $O/pmc_train_${CNT}$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 10 --warmup 2 --no-torch-legs > $O/pmc_train_${CNT}$N.log 2>&1; echo train $N $CNT $?
    python tools/pmc_summary.py $O/pmc_train_${CNT}$N > $O/pmc_train_${CNT}$N.txt 2>&1
  done
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
