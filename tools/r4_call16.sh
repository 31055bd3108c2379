#!/bin/bash
D=gpurun_out/r4u; mkdir -p $D
python bench.py > $D/bench_default.json 2> $D/bench_default.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4u/bench_default.json") if l.startswith("{")][-1])
print(round(d["value"] / 1e6, 2), "M/s frac", round(d["roofline"]["frac"], 4), "exec", round(d["roofline"]["executed_frac"], 4))
legs = d.get("legs", {})
print("module loop", {k: legs["module_evaluate_loop"].get(k) for k in ("value", "ms_per_batch", "log_prob_one_call_value", "max_abs_diff_loop_vs_one_call", "max_rel_err_vs_pipeline", "error")})
for k, v in legs.get("configs", {}).items():
    print("  ", k, {kk: v.get(kk) for kk in ("value", "dtype", "error", "wall_s")}, (v.get("roofline") or {}).get("frac"), (v.get("cpu_baseline") or {}).get("value"))
PY
python tools/profile_module_host.py 2>/dev/null | tail -15
