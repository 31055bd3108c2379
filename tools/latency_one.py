#!/usr/bin/env python3
"""One policy, one batch size, plain stream launches (for rocprofv3 / PMC passes): python3 tools/latency_one.py <n> <coop form, -1 = the shipped policy> [launches] [call]
default: the bare flow launch (f16x3, no repair launch); `call`: whole model.log_prob calls (flow + repair + recursion launches, default math)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gbnf_amd import native, synth
n, form = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
dev = torch.device("cuda:0")
specs = synth.synth_boosted_specs("glow", 8, 43, 215, 5, seed=1)
call = len(sys.argv) > 4 and sys.argv[4] == "call"
mix = native.NativeMixture([native.NativeFlow(s, math="default" if call else "f16x3") for s in specs])
native.tuning_set("coop", form)
if not call:
    native.tuning_set("repair", 0)
x = torch.from_numpy(synth.synth_batch(n, 43, seed=0)).to(dev)
ll = torch.empty((8, n), device=dev)
G = torch.empty(n, device=dev)
rho = torch.from_numpy(np.maximum(1.0 / np.power(2.0, np.arange(8)), 0.05).astype(np.float32)).to(dev)
for _ in range(reps):
    if call:
        mix.log_prob(x, rho, ll_out=ll, out=G)
    else:
        mix.component_log_prob(x, out=ll)
torch.cuda.synchronize()
