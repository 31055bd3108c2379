#!/bin/bash
# round 3, GPU call 1: hazard microbenchmark, the GPU suite + the new soak, baseline bench lines
mkdir -p gpurun_out/r3c1
cd $GRAFT_REPO_ROOT
timeout 300 ./tools/ubench/mfma_srcc_war > gpurun_out/r3c1/ubench_srcc_war.txt 2>&1
echo "ubench rc $?"
( time timeout 900 python -m pytest tests -x -q -m gpu --deselect tests/test_hip_soak.py ) > gpurun_out/r3c1/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r3c1/pytest_gpu.txt
( time timeout 1500 python -m pytest tests/test_hip_soak.py -q -m gpu ) > gpurun_out/r3c1/pytest_soak.txt 2>&1
echo "soak rc $?"; tail -15 gpurun_out/r3c1/pytest_soak.txt
timeout 600 python bench.py > gpurun_out/r3c1/bench_default.json 2> gpurun_out/r3c1/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3c1/bench_s20.json 2> gpurun_out/r3c1/bench_s20.err
timeout 300 python bench.py --steps 20 --warmup 5 --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > gpurun_out/r3c1/bench_s20_c1.json 2> gpurun_out/r3c1/bench_s20_c1.err
timeout 300 python bench.py --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > gpurun_out/r3c1/bench_c1.json 2> gpurun_out/r3c1/bench_c1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c1/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']/1e6,2),'M/s', d['ms_per_step'], d['dtype'], d['roofline']['launch_ms'], d.get('rccl'))
    except Exception as e: print(f,'ERR',e)
PY
