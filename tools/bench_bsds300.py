#!/usr/bin/env python3
"""Density evaluation at the BSDS300 geometry (d = 63, the reference's h_size_factor 5: h = 315; C = 8, K = 5, N = 16384): Boosted-Glow and
Boosted-RealNVP mixtures, samples/s and the error against the oracle on 256 rows.  usage: python tools/bench_bsds300.py   (GPU)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
dev = torch.device("cuda:0")
for kind in ("glow", "realnvp"):
    specs = synth.synth_boosted_specs(kind, 8, 63, 315, 5, seed=5)
    mix, flows = native.mixture_from_specs(specs)
    rho = torch.from_numpy(oracle.rho_init(8)).to(dev)
    x = torch.from_numpy(synth.synth_batch(16384, 63, seed=1)).to(dev)
    for _ in range(5): mix.log_prob(x, rho)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): mix.log_prob(x, rho)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    ll_ref, G_ref = oracle.mixture_log_prob(specs, rho.cpu().numpy(), x[:256].cpu().numpy())
    G, ll = mix.log_prob(x[:256].contiguous(), rho)
    print(f"BSDS300-shaped {kind} C=8 d=63 h=315 K=5 N=16384: {16384 / dt / 1e6:.1f} M samples/s ({flows[0].info().math_mode}), G err {float(np.abs(G.cpu().numpy() - G_ref).max() / np.abs(G_ref).max()):.1e}")
