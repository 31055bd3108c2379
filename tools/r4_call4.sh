#!/bin/bash
D=gpurun_out/r4d; mkdir -p $D
( timeout 1500 python -m pytest tests -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -5 $D/pytest.txt
python bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "bench rc $?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-config-legs > $D/bench_steps20.json 2> $D/bench_steps20.err; echo "bench20 rc $?"
python - <<'PY'
import json
for f in ("bench_default", "bench_steps20"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/r4d/{f}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no line", e); continue
    print(f, round(d["value"] / 1e6, 2), "M/s frac", round(d["roofline"]["frac"], 4), "exec", round(d["roofline"]["executed_frac"], 4))
    legs = d.get("legs", {})
    if "module_evaluate_loop" in legs: print("  module loop", legs["module_evaluate_loop"])
    for k, v in legs.get("configs", {}).items():
        print("  ", k, {kk: v.get(kk) for kk in ("value", "dtype", "error", "wall_s")}, (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
PY
