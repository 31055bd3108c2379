set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O; : > $O/rc.txt
timeout 420 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc $? $O/bench_default.json" >> $O/rc.txt
timeout 420 python bench.py --config hepmass_realnvp --batch 65536 --no-extra-legs --cpu-seconds 0 > $O/bench_hepmass.json 2>/dev/null; echo "rc $? $O/bench_hepmass.json" >> $O/rc.txt
timeout 420 python bench.py --components 4 --no-extra-legs --cpu-seconds 0 > $O/bench_c4.json 2>/dev/null; echo "rc $? $O/bench_c4.json" >> $O/rc.txt
timeout 420 python bench.py --math bf16x6 --no-extra-legs --cpu-seconds 0 > $O/bench_bf16x6.json 2>/dev/null; echo "rc $? $O/bench_bf16x6.json" >> $O/rc.txt
timeout 420 python tools/bench_image.py --batch 256 --cpu-seconds 5 > $O/image_n256.json 2>/dev/null; echo "rc $? $O/image_n256.json" >> $O/rc.txt
timeout 420 python tools/bench_image.py --batch 64 --cpu-seconds 0 > $O/image_n64.json 2>/dev/null; echo "rc $? $O/image_n64.json" >> $O/rc.txt
timeout 420 python tools/bench_train.py --batch 4096 --cpu-steps 0 > $O/train_n4096.json 2>/dev/null; echo "rc $? $O/train_n4096.json" >> $O/rc.txt
timeout 420 python tools/bench_train.py --batch 65536 --cpu-steps 0 > $O/train_n65536.json 2>/dev/null; echo "rc $? $O/train_n65536.json" >> $O/rc.txt
timeout 420 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 640 --warmup 64 --prewarm 0.05 --no-extra-legs > $O/prof_stats.log 2>&1; echo "rc $? $O/prof_stats.log" >> $O/rc.txt
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs > $O/pmc_fetch.log 2>&1; echo "rc $? $O/pmc_fetch.log" >> $O/rc.txt
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs > $O/pmc_write.log 2>&1; echo "rc $? $O/pmc_write.log" >> $O/rc.txt
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/pmc_sq1 -o sq1 --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs > $O/pmc_sq1.log 2>&1; echo "rc $? $O/pmc_sq1.log" >> $O/rc.txt
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O/pmc_sq2 -o sq2 --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs > $O/pmc_sq2.log 2>&1; echo "rc $? $O/pmc_sq2.log" >> $O/rc.txt
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; done
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
ls -la $O
