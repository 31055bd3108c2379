#!/bin/bash
D=gpurun_out/r4q; mkdir -p $D
( timeout 600 python -m pytest tests/test_sharded_gpu.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -12 $D/pytest.txt
for P in library torch; do
  timeout 300 python bench.py --force-gather --components 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs --pipeline $P > $D/emul_c1_steps20_$P.json 2> $D/err_$P.txt; echo "rc $?"
  timeout 300 python bench.py --force-gather --components 1 --cpu-seconds 0 --no-extra-legs --pipeline $P > $D/emul_c1_default_$P.json 2>> $D/err_$P.txt; echo "rc $?"
done
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs > $D/one_gpu_steps20.json 2>/dev/null
python - <<'PY'
import json
for f in ("emul_c1_steps20_library", "emul_c1_steps20_torch", "emul_c1_default_library", "emul_c1_default_torch", "one_gpu_steps20"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4q/{f}.json") if l.startswith("{")][-1])
        print(f, round(d["value"] / 1e6, 1), "M/s", "ms/step", round(d["ms_per_step"], 5), "elapsed_med_us", round(1e3 * d["timing"]["elapsed_ms_median"], 1), d.get("rccl", {}).get("pipeline"), d.get("rccl", {}).get("graph_errors"))
    except Exception as e:
        print(f, "failed", e)
PY
tail -3 $D/err_library.txt
