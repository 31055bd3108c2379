#!/bin/bash
# PMC passes of the image benchmark (BASELINE.json configs[3]) and the HBM counters of the training step.  FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md; in one pass rocprofv3 hung on this pool: two 300 s time-outs, round 3).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
T="timeout 200"
for B in 256 64; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    $T rocprofv3 --kernel-trace --pmc $CNT -d $O/pmc_image_${CNT}$B -o im --output-format csv -- python3 tools/bench_image.py --no-graph --batch $B --cpu-seconds 0 --steps 10 --warmup 2 > $O/pmc_image_${CNT}$B.log 2>&1; echo image $B $CNT $?
    python tools/pmc_summary.py $O/pmc_image_${CNT}$B > $O/pmc_image_${CNT}$B.txt 2>&1
  done
done
$T rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA -d $O/pmc_image_sq256 -o im --output-format csv -- python3 tools/bench_image.py --no-graph --batch 256 --cpu-seconds 0 --steps 10 --warmup 2 > $O/pmc_image_sq256.log 2>&1; echo image sq $?
python tools/pmc_summary.py $O/pmc_image_sq256 > $O/pmc_image_sq256.txt 2>&1
for N in 4096 65536; do
  for CNT in FETCH_SIZE WRITE_SIZE; do
    $T rocprofv3 --kernel-trace --pmc $CNT -d $O/pmc_train_${CNT}$N -o tr --output-format csv -- python3 tools/bench_train.py --batch $N --cpu-steps 0 --steps 10 --warmup 2 --no-torch-legs > $O/pmc_train_${CNT}$N.log 2>&1; echo train $N $CNT $?
    python tools/pmc_summary.py $O/pmc_train_${CNT}$N > $O/pmc_train_${CNT}$N.txt 2>&1
  done
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
