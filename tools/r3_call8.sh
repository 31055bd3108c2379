#!/bin/bash
D=gpurun_out/r3c8; mkdir -p $D
( time timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_module.py tests/test_sharded_gpu.py -q -m gpu -x ) > $D/pytest_parity.txt 2>&1
echo "pytest rc $?"; tail -12 $D/pytest_parity.txt
python tools/parity_report.py > $D/parity_report.txt 2>&1; grep -i "g17\|g15" $D/parity_report.txt | head -20
timeout 300 python bench.py --steps 20 --warmup 5 --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > $D/bench_s20_c1.json 2> $D/bench_s20_c1.err
for B in 4096 65536 1048576; do
  G=32; S=1024; if [ $B = 65536 ]; then S=256; fi; if [ $B = 1048576 ]; then G=4; S=32; fi
  timeout 600 python bench.py --batch $B --group $G --steps $S --warmup $G --prewarm 0.05 --cpu-seconds 0 --no-extra-legs > $D/sweep_n$B.json 2> $D/sweep_n$B.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c8/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        r=d['roofline']
        print(f, round(d['value']/1e6,2),'M/s', d['dtype'], 'S',d['config']['group'],'B',d['config']['global_batch'], 'launch_ms',round(r['launch_ms'],4),'frac',round(r['frac'],4),'exec',round(r['executed_frac'],3),'hbm_frac',round(r['hbm_frac'],5), d.get('rccl'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
