#!/usr/bin/env python3
"""Find models on which a math mode of the evaluation kernels misses the 1e-5 bar against the torch-f32 oracle (GPU, opt-in).
Un-normalised ReLU RealNVPs (no BatchNorm) at wide hidden layers are the known ill-conditioned family.

    python tools/find_offender.py [math=f16x3] [n_seeds=24]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle

math = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device("cuda:0")
rows = []
for d, h, K in ((21, 500, 8), (43, 500, 8), (21, 430, 8), (8, 500, 8)):
    for seed in range(n_seeds):
        for x_scale in (1.0, 2.0):
            spec = synth.synth_realnvp_spec(d, h, K, coupling_network="relu", batch_norm=False, flip_init=seed % 2, seed=7000 + seed)
            x = synth.synth_batch(512, d, seed=seed, scale=x_scale)
            ll_ref = oracle.component_log_prob(spec, x)
            z64, l64 = oracle.component_forward(spec, x, backend="numpy64")
            ll64 = (-0.5 * z64 ** 2 - 0.5 * np.log(2 * np.pi)).sum(1) + l64
            try:
                _, _, ll = native.NativeFlow(spec, math=math).forward(torch.from_numpy(x).to(dev), want_z=False, want_ldj=False, want_ll=True)
            except native.GbnfError as e:
                print("unsupported", d, h, K, str(e)[:80]); break
            ll = ll.cpu().numpy()
            den = np.maximum(np.abs(ll_ref), 1.0)
            e_gpu_ref = float(np.max(np.abs(ll - ll_ref) / den))
            e_gpu_64 = float(np.max(np.abs(ll - ll64) / np.maximum(np.abs(ll64), 1.0)))
            e_ref_64 = float(np.max(np.abs(ll_ref - ll64) / np.maximum(np.abs(ll64), 1.0)))
            rows.append((e_gpu_ref, d, h, K, seed, x_scale, e_gpu_64, e_ref_64, float(np.abs(ll_ref).max())))
rows.sort(reverse=True)
print(f"math={math}: worst cases (rel err vs torch-f32 oracle | d h K seed x_scale | kernel vs f64 | oracle vs f64 | max|ll|)")
for r in rows[:15]:
    print(f"  {r[0]:.2e} | d={r[1]} h={r[2]} K={r[3]} seed={r[4]} x_scale={r[5]} | {r[6]:.2e} | {r[7]:.2e} | {r[8]:.1f}")
print(f"{sum(1 for r in rows if r[0] > 1e-5)} of {len(rows)} beyond 1e-5")
