#!/usr/bin/env python3
"""Randomised stress of the evaluation kernels (component log-density and the mixture) against the torch-CPU oracle,
weighted towards the wide-hidden variants (opt-in, GPU).  usage: python tools/stress_eval.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
dev = torch.device("cuda:0")
bad = 0
for k in range(cases):
    kind = "glow" if rng.randint(2) else "realnvp"
    d = int(rng.choice([2, 3, 6, 8, 13, 21, 43, 50, 63, 64]))
    h = int(rng.choice([30, 105, 215, 256, 257, 300, 315, 384, 385, 430, 500, 512]))
    K = int(rng.randint(1, 9))
    C = int(rng.randint(1, 5))
    n = int(rng.choice([1, 17, 33, 100, 333, 1000, 4096, 5000]))
    if kind == "glow":
        kw = dict(act=str(rng.choice(["tanh", "relu", "random"])), coupling=str(rng.choice(["affine", "additive"])),
                  permutation=str(rng.choice(["shuffle", "reverse"])))
    else:
        kw = dict(coupling_network=str(rng.choice(["tanh", "relu", "mixed", "random"])), batch_norm=bool(rng.randint(2)))
    tag = f"{kind} C={C} d={d} h={h} K={K} n={n} {kw}"
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=300 + k, **kw)
    try:
        mix = native.NativeMixture(native.flows_for_mixture(specs))
    except native.GbnfError as e:
        print("skip (unsupported):", tag, "|", str(e)[:90]); continue
    x = synth.synth_batch(n, d, seed=k)
    rho = oracle.rho_init(C)
    ll_ref, G_ref = oracle.mixture_log_prob(specs, rho, x)
    G, ll = mix.log_prob(torch.from_numpy(x).to(dev), torch.from_numpy(rho).to(dev))
    e1 = float(np.max(np.abs(ll.cpu().numpy() - ll_ref) / np.maximum(np.abs(ll_ref), 1.0)))
    e2 = float(np.max(np.abs(G.cpu().numpy() - G_ref) / np.maximum(np.abs(G_ref), 1.0)))
    ok = e1 < 1e-5 and e2 < 1e-5
    bad += 0 if ok else 1
    print("ok  " if ok else "FAIL", tag, f"| ll {e1:.1e} G {e2:.1e}")
print(f"{cases} cases, {bad} failures")
sys.exit(1 if bad else 0)
