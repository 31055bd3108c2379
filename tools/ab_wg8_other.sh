cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for form in pairs wg8; do
    e=""; [ $form = wg8 ] && e="GBNF_NO_WG_PAIRS=1"
    a=$(env $e python bench.py --config hepmass_realnvp --steps 256 --warmup 32 --prewarm 0.02 --cpu-seconds 0 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    b=$(env $e python bench.py --force-gather --components 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    c=$(env $e python tools/bench_train.py --batch 65536 --steps 20 --warmup 5 --cpu-steps 0 --no-torch-legs 2>/dev/null | tail -1)
    python - "$form" "$a" "$b" "$c" <<'PY'
import json, sys
v = [json.loads(x) for x in sys.argv[2:]]
print(f"{sys.argv[1]:6s} hepmass {v[0]['value']/1e6:7.2f} M | rank-of-8 steps20 {v[1]['value']/1e6:7.2f} M ({v[1]['ms_per_step']*20*1e3:.1f} us/rep) | train n65536 {v[2]['value']/1e6:7.2f} M")
PY
  done
done
