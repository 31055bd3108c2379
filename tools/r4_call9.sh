#!/bin/bash
D=gpurun_out/r4n; mkdir -p $D
( timeout 900 python -m pytest tests/test_hip_image.py tests/test_hip_baseline_configs.py tests/test_abi.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -15 $D/pytest.txt
for b in 256 64; do
  python tools/bench_image.py --batch $b --steps 20 --warmup 3 --cpu-seconds 1 > $D/img_$b.json 2>$D/err_$b.txt
  GBNF_IMAGE_REPAIR=2 python tools/bench_image.py --batch $b --steps 20 --warmup 3 --cpu-seconds 1 > $D/img_norepair_$b.json 2>$D/err_nr_$b.txt
done
python - <<'PY'
import json
for b in (256, 64):
    for k in ("img", "img_norepair"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r4n/{k}_{b}.json") if l.startswith("{")][-1])
            print(k, b, round(d["value"]), "img/s  stream", round(d.get("stream_launches_value") or 0), "err", d.get("max_rel_err_vs_cpu"), "gpu_ms", round(d["roofline"]["gpu_ms_per_step"], 3))
        except Exception as e:
            print(k, b, "failed", e)
PY
