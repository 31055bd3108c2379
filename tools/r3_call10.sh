#!/bin/bash
D=gpurun_out/r3c10; mkdir -p $D
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > $D/pytest_gpu.txt 2>&1
echo "pytest rc $?"; tail -15 $D/pytest_gpu.txt
timeout 600 python bench.py --cpu-seconds 0 --no-extra-legs > $D/bench_default.json 2> $D/bench_default.err; python -c "
import json; d=json.loads(open('$D/bench_default.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['roofline']['launch_ms'])"
python - <<'PY'
import torch, time, sys
sys.path.insert(0,'.')
from gbnf_amd import native, synth
dev=torch.device('cuda:0')
spec=synth.synth_boosted_specs('glow',1,43,215,5,seed=1)[0]
for math in ('f32','f16x3','bf16x6','default'):
    f=native.NativeFlow(spec,math=math)
    for n in (4096,65536):
        z=torch.randn(n,43,device=dev)
        for _ in range(5): f.inverse(z)
        torch.cuda.synchronize(); t0=time.perf_counter()
        for _ in range(50): f.inverse(z)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/50
        print(f"inverse {math:8s} N={n:6d}: {dt*1e6:8.1f} us  {n/dt/1e6:7.1f} M samples/s")
PY
