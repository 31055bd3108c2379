#!/bin/bash
# GPU box: parity + soak + training tests, then the same benchmarks on two builds of the library (A/B on ONE box; edit the list of libraries)
D=gpurun_out/r3c18; mkdir -p $D
( timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_soak.py tests/test_hip_train.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -3 $D/pytest.txt
for L in tools/ablate/libgbnf_hip_builtin_dma.so gradient-boosted-normalizing-flows_amd/libgbnf_hip.so; do
  for rep in 1 2; do
    GBNF_LIB_PATH=$PWD/$L python bench.py > $D/b.json 2>/dev/null
    python -c "
import json; d=json.loads(open('$D/b.json').read().strip().splitlines()[-1]); print('$L'[-24:], 'default', round(d['value']/1e6,2),'M/s frac',d['roofline']['frac'])"
  done
  GBNF_LIB_PATH=$PWD/$L python bench.py --steps 20 --warmup 5 > $D/b.json 2>/dev/null
  python -c "
import json; d=json.loads(open('$D/b.json').read().strip().splitlines()[-1]); print('$L'[-24:], 'steps20', round(d['value']/1e6,2),'M/s')"
  for N in 4096 65536; do GBNF_LIB_PATH=$PWD/$L python tools/bench_train.py --batch $N --cpu-steps 0 --steps 100 > $D/t.json 2>/dev/null; python -c "
import json; d=json.loads(open('$D/t.json').read().strip().splitlines()[-1]); print('$L'[-24:], 'train', $N, round(d['value']/1e6,2),'M/s', round(d['ms_per_step'],4),'ms fwd',round(d['forward_kernel_ms'],4),'bwd',round(d['backward_kernels_ms'],4))"; done
done
