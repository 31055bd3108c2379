#!/bin/bash
D=gpurun_out/r4s; mkdir -p $D
( timeout 1200 python -m pytest tests/test_hip_train.py -q -m gpu -x ) > $D/pytest.txt 2>&1
echo "pytest rc $?"; tail -8 $D/pytest.txt
for N in 4096 16384 65536; do
  python tools/bench_train.py --config hepmass_realnvp --batch $N --batch-stats --cpu-steps 0 --no-torch-legs --steps 50 > $D/t_new_$N.json 2>$D/err.txt
  GBNF_TRAIN_PATH=old python tools/bench_train.py --config hepmass_realnvp --batch $N --batch-stats --cpu-steps 0 --no-torch-legs --steps 50 > $D/t_old_$N.json 2>>$D/err.txt
  python tools/bench_train.py --config hepmass_realnvp --batch $N --cpu-steps 0 --no-torch-legs --steps 50 > $D/t_eval_$N.json 2>>$D/err.txt
done
python - <<'PY'
import json
for N in (4096, 16384, 65536):
    for k in ("new", "old", "eval"):
        try:
            d = json.loads([l for l in open(f"gpurun_out/r4s/t_{k}_{N}.json") if l.startswith("{")][-1])
            print(k, N, round(d["value"] / 1e6, 2), "M/s", round(d["ms_per_step"], 4), "ms fwd", round(d["forward_kernel_ms"], 4), "bwd", round(d["backward_kernels_ms"], 4))
        except Exception as e:
            print(k, N, "failed", e)
PY
tail -3 $D/err.txt
