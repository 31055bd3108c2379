#!/bin/bash
# Upper bound of "one 8-wave workgroup shares one LDS weight copy" (VERDICT r5 item 4a): the headline launch as two 4-wave workgroups per CU
# (shipped) and as one 8-wave workgroup per CU (GBNF_NO_WG_PAIRS=1: half the L2 -> LDS DMA per CU), each on the shipped kernel and on builds
# with the stage barrier and / or the weight DMA removed (tools/build_ab.sh base= nobar=-DGBNF_ABLATE_BARRIER nodma=-DGBNF_ABLATE_DMA
# "nobar_nodma=-DGBNF_ABLATE_BARRIER -DGBNF_ABLATE_DMA"; ablated builds give wrong results: timing only).
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in base nobar nodma nobar_nodma; do
  for form in pairs wg8; do
    env="GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_$lib.so GBNF_NO_REPAIR=1"
    [ $form = wg8 ] && env="$env GBNF_NO_WG_PAIRS=1"
    line=$(env $env python bench.py --math f16x3 --cpu-seconds 0 --steps 1280 --warmup 64 --prewarm 0.1 --no-extra-legs --no-config-legs 2>/dev/null | tail -1)
    python - "$lib" "$form" "$line" <<'PY'
import json, sys
j = json.loads(sys.argv[3])
print(f"{sys.argv[1]:12s} {sys.argv[2]:6s} launch {j['roofline']['launch_ms']:.4f} ms  {j['value'] / 1e6:7.2f} M samples/s")
PY
  done
done
done
