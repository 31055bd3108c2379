#!/bin/bash
# shipped library: two 4-wave workgroups per CU (the policy since round 5) against one 8-wave workgroup per CU (GBNF_NO_WG_PAIRS=1), alternating
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for form in pairs wg8; do
    e=""; [ $form = wg8 ] && e="GBNF_NO_WG_PAIRS=1"
    a=$(env $e python bench.py --cpu-seconds 0 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    b=$(env $e python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    c=$(env $e python bench.py --components 4 --steps 2048 --cpu-seconds 0 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    python - "$form" "$a" "$b" "$c" <<'PY'
import json, sys
v = [json.loads(x) for x in sys.argv[2:]]
print(f"{sys.argv[1]:6s} default {v[0]['value']/1e6:7.2f} M ({v[0]['roofline']['launch_ms']:.4f} ms) | --steps 20 {v[1]['value']/1e6:7.2f} M ({v[1]['roofline']['launch_ms']:.4f} ms) | C=4 {v[2]['value']/1e6:7.2f} M")
PY
  done
done
