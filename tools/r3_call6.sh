#!/bin/bash
D=gpurun_out/r3c6; mkdir -p $D
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $D/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -12 $D/pytest_gpu.txt
timeout 600 python bench.py > $D/bench_default.json 2> $D/bench_default.err
timeout 300 python bench.py --steps 20 --warmup 5 > $D/bench_s20.json 2> $D/bench_s20.err
timeout 300 python bench.py --steps 20 --warmup 5 --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > $D/bench_s20_c1.json 2> $D/bench_s20_c1.err
timeout 300 python bench.py --force-gather --components 1 --cpu-seconds 0 --no-extra-legs > $D/bench_c1.json 2> $D/bench_c1.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3c6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us/step', d['dtype'], 'S',d['config']['group'], 'launch_ms',round(d['roofline']['launch_ms'],4), d.get('rccl'), d['timing'], d['numerics_guard'])
        if 'legs' in d: print('   legs', {k:(round(v.get('value',0)/1e6,2) if isinstance(v,dict) else v) for k,v in d['legs'].items()}, d['legs'].get('module_evaluate_loop'))
        if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('thread_probe_samples_per_s'), d.get('max_rel_err_vs_cpu'))
    except Exception as e: print(f,'ERR',e, open(f.replace('.json','.err')).read()[-600:])
PY
