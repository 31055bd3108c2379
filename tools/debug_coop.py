#!/usr/bin/env python3
"""Where do the cooperative kernel's 16- and 32-sample tile forms differ? (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
dev = torch.device("cuda:0")
kind, d, h, K, C = "glow", 43, 215, int(os.environ.get("K", "5")), 2
specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=11)
mix, flows = native.mixture_from_specs(specs, math="f16x3")
for n in (1, 16, 33):
    x = torch.from_numpy(synth.synth_batch(n, d, seed=12)).to(dev)
    out = {}
    for m in (1, 2, 0):
        native.tuning_set("coop", m)
        z, ldj, ll = flows[0].forward(x, want_ll=True)
        out[m] = (z.cpu().numpy(), ldj.cpu().numpy(), ll.cpu().numpy())
    for nm, k in (("z", 0), ("ldj", 1), ("ll", 2)):
        a, b, t = out[1][k], out[2][k], out[0][k]
        print(f"n={n} {nm}: max|nt1-nt2| = {np.abs(a-b).max():.3e}   max|nt1-thr| = {np.abs(a-t).max():.3e}", "cols differing:", np.unique(np.nonzero(a != b)[-1])[:20] if nm == "z" else "")
