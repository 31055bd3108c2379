#!/bin/bash
D=gpurun_out/r4l; mkdir -p $D
( timeout 900 python -m pytest tests/test_hip_image.py tests/test_hip_baseline_configs.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -4 $D/pytest.txt
for b in 256 64; do
  python tools/bench_image.py --batch $b --steps 20 --warmup 3 --cpu-seconds 1 > $D/img_fused_$b.json 2>$D/err_f_$b.txt
done
python - <<'PY'
import json
for b in (256, 64):
    for k in ("fused",):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r4l/img_{k}_{b}.json") if l.startswith("{")][-1])
            print(k, b, round(d["value"]), "img/s  stream", round(d.get("stream_launches_value") or 0), "err", d.get("max_rel_err_vs_cpu"), "gpu_ms", round(d["roofline"]["gpu_ms_per_step"], 3))
        except Exception as e:
            print(k, b, "failed", e)
PY
cp tools/libgbnf_image_stamps16.so tools/libgbnf_image_stamps.so; python tools/image_stamps2.py 256 16 > $D/stamps16.txt 2>&1; cat $D/stamps16.txt | tail -8
cp tools/libgbnf_image_stamps8.so tools/libgbnf_image_stamps.so; python tools/image_stamps2.py 256 8 > $D/stamps8.txt 2>&1; cat $D/stamps8.txt | tail -8
