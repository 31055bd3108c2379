#!/bin/bash
# SQ counter passes of one command (GPU box): tools/pmc_sq.sh <name> <kernel substring> python3 script args...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
name=$1; kern=$2; shift; shift
O=gpurun_out/pmc_$name; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/sq1 -o q --output-format csv -- "$@" > $O/log1.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O/sq2 -o q --output-format csv -- "$@" > $O/log2.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC -d $O/sq3 -o q --output-format csv -- "$@" > $O/log3.txt 2>&1
for d in sq1 sq2 sq3; do python tools/pmc_summary.py $O/$d 2>&1 | grep -A12 "$kern" | head -14; done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
