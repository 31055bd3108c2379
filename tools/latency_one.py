#!/usr/bin/env python3
"""One policy, one batch size, plain stream launches of the bare flow launch (for rocprofv3 / PMC passes): python3 tools/latency_one.py <n> <coop form> [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gbnf_amd import native, synth
n, form = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
specs = synth.synth_boosted_specs("glow", 8, 43, 215, 5, seed=1)
mix = native.NativeMixture([native.NativeFlow(s, math="f16x3") for s in specs])
native.tuning_set("repair", 0); native.tuning_set("coop", form)
x = torch.from_numpy(synth.synth_batch(n, 43, seed=0)).to(dev)
ll = torch.empty((8, n), device=dev)
for _ in range(reps):
    mix.component_log_prob(x, out=ll)
torch.cuda.synchronize()
