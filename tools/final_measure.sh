# End-of-round measurement on ONE box (GPU side): bench lines of every configuration, rocprofv3 kernel stats of the headline
# command, and the separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ) of the headline AND of the other configurations, so that
# every line's roofline.traffic is a measured number.  Output: gpurun_out/final/ -> tools/collect_final_profiles.py r6
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O; : > $O/rc.txt
run() { name=$1; shift; timeout 420 "$@" > $O/$name.json 2> $O/$name.err; echo "rc $? $name" >> $O/rc.txt; }
run bench_default python bench.py
run bench_steps20 python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs
run bench_hepmass python bench.py --config hepmass_realnvp --batch 65536 --steps 256 --warmup 32 --no-extra-legs --cpu-seconds 0
run bench_c4 python bench.py --components 4 --steps 2048 --no-extra-legs --cpu-seconds 0
run bench_bf16x6 python bench.py --math bf16x6 --no-extra-legs --cpu-seconds 0
run bench_emul8_steps20 python bench.py --force-gather --components 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs
run bench_emul8_default python bench.py --force-gather --components 1 --cpu-seconds 0 --no-extra-legs
run bench_emul8_default_torch python bench.py --force-gather --components 1 --cpu-seconds 0 --no-extra-legs --pipeline torch
run image_n256 python tools/bench_image.py --batch 256 --cpu-seconds 5
run image_n64 python tools/bench_image.py --batch 64 --cpu-seconds 0
run image_1x28x28_n256 python tools/bench_image.py --batch 256 --input 1 28 28 --cpu-seconds 3
run image_1x28x20_n256 python tools/bench_image.py --batch 256 --input 1 28 20 --cpu-seconds 0
run image_h512_n256 python tools/bench_image.py --batch 256 --hidden 512 --cpu-seconds 0
run image_h384_n256 python tools/bench_image.py --batch 256 --hidden 384 --cpu-seconds 0
run image_depth0_n256 python tools/bench_image.py --batch 256 --depth 0 --cpu-seconds 0
run image_depth2_n256 python tools/bench_image.py --batch 256 --depth 2 --cpu-seconds 0
(python tools/bench_image_inverse.py; python tools/bench_image_inverse.py --batch 64; python tools/bench_image_inverse.py --input 1 28 28) > $O/image_inverse.txt 2>&1
run train_n4096 python tools/bench_train.py --batch 4096 --cpu-steps 0
run train_n65536 python tools/bench_train.py --batch 65536 --cpu-steps 2
run train_hepmass_bs_n65536 python tools/bench_train.py --config hepmass_realnvp --batch 65536 --batch-stats --cpu-steps 0 --no-torch-legs
run train_hepmass_n65536 python tools/bench_train.py --config hepmass_realnvp --batch 65536 --cpu-steps 0 --no-torch-legs
run train_glow_depth0_n65536 python tools/bench_train.py --config miniboone_glow_depth0 --batch 65536 --cpu-steps 0 --no-torch-legs
run train_glow_depth2_n65536 python tools/bench_train.py --config miniboone_glow_depth2 --batch 65536 --cpu-steps 0 --no-torch-legs
run train_hepmass_depth2_n65536 python tools/bench_train.py --config hepmass_realnvp_depth2 --batch 65536 --cpu-steps 0 --no-torch-legs
run train_hepmass_residual_n65536 python tools/bench_train.py --config hepmass_realnvp_residual --batch 65536 --cpu-steps 0 --no-torch-legs
run module_eval python tools/bench_module_eval.py
run boosted_step_n512 python tools/bench_boosted_step.py --batch 512 --steps 200
cp gpurun_out/bench_full.json $O/bench_full_record_last.json 2>/dev/null
(python tools/bench_latency.py; python tools/latency_ablate.py shipped:coop=0,shipped:coop=1,shipped:coop=2,shipped:coop=3,shipped --sizes 64,256,512,1024,2048,4096) > $O/latency.txt 2>&1
python tools/bench_coop_geometries.py 256 512 1024 > $O/coop_geometries.txt 2>&1
prof() { name=$1; shift; timeout 420 rocprofv3 --kernel-trace "$@" > $O/$name.log 2>&1; echo "rc $? $name" >> $O/rc.txt; }
HEAD="python3 bench.py --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs"
prof prof_stats --stats -d $O/prof_stats -o stats --output-format csv -- python3 bench.py --cpu-seconds 0 --steps 640 --warmup 64 --prewarm 0.05 --no-extra-legs
prof pmc_fetch --pmc FETCH_SIZE -d $O/pmc_fetch -o fetch --output-format csv -- $HEAD
prof pmc_write --pmc WRITE_SIZE -d $O/pmc_write -o write --output-format csv -- $HEAD
prof pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/pmc_sq1 -o sq1 --output-format csv -- $HEAD
prof pmc_sq2 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O/pmc_sq2 -o sq2 --output-format csv -- $HEAD
S20="python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs"
prof pmc_fetch_s20 --pmc FETCH_SIZE -d $O/pmc_fetch_s20 -o f --output-format csv -- $S20
prof pmc_write_s20 --pmc WRITE_SIZE -d $O/pmc_write_s20 -o w --output-format csv -- $S20
C4="python3 bench.py --components 4 --cpu-seconds 0 --steps 128 --warmup 32 --prewarm 0.01 --no-extra-legs"
prof pmc_fetch_c4 --pmc FETCH_SIZE -d $O/pmc_fetch_c4 -o f --output-format csv -- $C4
prof pmc_write_c4 --pmc WRITE_SIZE -d $O/pmc_write_c4 -o w --output-format csv -- $C4
HM="python3 bench.py --config hepmass_realnvp --batch 65536 --cpu-seconds 0 --steps 64 --warmup 32 --prewarm 0.0 --no-extra-legs"
prof pmc_fetch_hm --pmc FETCH_SIZE -d $O/pmc_fetch_hm -o f --output-format csv -- $HM
prof pmc_write_hm --pmc WRITE_SIZE -d $O/pmc_write_hm -o w --output-format csv -- $HM
IMG="python3 tools/bench_image.py --batch 256 --cpu-seconds 0 --steps 10 --warmup 2 --no-graph"
# (round 6: 120 steps, so that the one-time on-data checks of the repair kernel are < 5 % of the GPU time: VERDICT r5 weak 6)
prof stats_img --stats -d $O/stats_img -o s --output-format csv -- python3 tools/bench_image.py --batch 256 --cpu-seconds 0 --steps 120 --warmup 5 --no-graph
# the latency form: one log_prob call (flow + repair + recursion launches) at the reference's batch sizes, plain stream launches
prof stats_latency_n512 --stats -d $O/stats_latency_n512 -o s --output-format csv -- python3 tools/latency_one.py 512 -1 300 call
prof stats_latency_n1024 --stats -d $O/stats_latency_n1024 -o s --output-format csv -- python3 tools/latency_one.py 1024 -1 300 call
prof pmc_sq1_latency --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/pmc_sq1_latency -o q --output-format csv -- python3 tools/latency_one.py 512 -1 200
prof pmc_sq2_latency --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O/pmc_sq2_latency -o q --output-format csv -- python3 tools/latency_one.py 512 -1 200
prof pmc_fetch_latency --pmc FETCH_SIZE -d $O/pmc_fetch_latency -o q --output-format csv -- python3 tools/latency_one.py 512 -1 200
prof pmc_write_latency --pmc WRITE_SIZE -d $O/pmc_write_latency -o q --output-format csv -- python3 tools/latency_one.py 512 -1 200
prof pmc_fetch_img --pmc FETCH_SIZE -d $O/pmc_fetch_img -o f --output-format csv -- $IMG
prof pmc_write_img --pmc WRITE_SIZE -d $O/pmc_write_img -o w --output-format csv -- $IMG
prof pmc_sq_img --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $O/pmc_sq_img -o q --output-format csv -- $IMG
TR="python3 tools/bench_train.py --batch 65536 --cpu-steps 0 --no-torch-legs --steps 20 --warmup 3"
prof stats_train --stats -d $O/stats_train -o s --output-format csv -- $TR
prof pmc_fetch_train --pmc FETCH_SIZE -d $O/pmc_fetch_train -o f --output-format csv -- $TR
prof pmc_write_train --pmc WRITE_SIZE -d $O/pmc_write_train -o w --output-format csv -- $TR
TRB="python3 tools/bench_train.py --config hepmass_realnvp --batch 65536 --batch-stats --cpu-steps 0 --no-torch-legs --steps 20 --warmup 3"
prof stats_train_bs --stats -d $O/stats_train_bs -o s --output-format csv -- $TRB
for d in pmc_fetch pmc_write pmc_sq1 pmc_sq2 pmc_fetch_s20 pmc_write_s20 pmc_fetch_c4 pmc_write_c4 pmc_fetch_hm pmc_write_hm pmc_fetch_img pmc_write_img pmc_sq_img pmc_fetch_train pmc_write_train pmc_sq1_latency pmc_sq2_latency pmc_fetch_latency pmc_write_latency; do python tools/pmc_summary.py $O/$d > $O/$d.txt 2>&1; done
prof stats_hm --stats -d $O/stats_hm -o s --output-format csv -- $HM
prof stats_c4 --stats -d $O/stats_c4 -o s --output-format csv -- $C4
for d in prof_stats stats_img stats_train stats_train_bs stats_hm stats_c4 stats_latency_n512 stats_latency_n1024; do find $O/$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/$d.kernel_stats.csv; done
find $O -name "*.csv" -size +1M -delete
find $O -name "*.db" -delete
cat $O/rc.txt
