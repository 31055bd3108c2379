#!/usr/bin/env python3
"""Replay case(s) of tools/stress_train.py's stream and look at a deviation from the float64 oracle more closely: the same call on the
per-step kernels (another float32 implementation: a ReLU kink moves with the implementation, a kernel error does not), and the oracle's
ReLU-threshold bracket at a wider tolerance (pre-activations of 430-512 products carry ~1e-5 of float32 round-off).
usage: python tools/debug_train_case.py <cases> <seed> <k> [<k> ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
from test_hip_train import _dev_spec
import stress_train

cases, seed, want = int(sys.argv[1]), int(sys.argv[2]), set(int(v) for v in sys.argv[3:])
rng = np.random.RandomState(seed)
dev = torch.device("cuda:0")
for k in range(cases):
    kind, d, h, K, n, extra, spec = stress_train.gen_case(rng, k)
    try:                                   # (stress_train.py draws nothing more for a case it skips)
        native.NativeTrainer(_dev_spec(spec, dev))
    except native.GbnfError:
        continue
    gscale = np.float32(10.0 ** rng.uniform(-6, 4))
    x = synth.synth_batch(n, d, seed=k)
    g_z = rng.standard_normal(x.shape).astype(np.float32) * gscale
    g_l = rng.standard_normal(n).astype(np.float32) * gscale
    if k not in want:
        continue
    print(f"case {k}: {kind} d={d} h={h} K={K} n={n} {extra}")
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    xd = torch.from_numpy(x).to(dev)
    z, ldj, trace = tr.forward(xd, want_trace=True)
    gx, grads = tr.backward(xd, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True, trace=trace)
    gx0, grads0 = tr.backward(xd, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True)
    _, g64 = oracle.component_grads(spec, x, g_z, g_l)
    def worst(ga, gb):
        e = [(float(np.abs(a.cpu().numpy().reshape(b.shape).astype(np.float64) - (b.cpu().numpy().astype(np.float64) if torch.is_tensor(b) else b)).max()
                    / max(float(np.abs(b.cpu().numpy() if torch.is_tensor(b) else b).max()), 1e-3 * float(gscale))), i)
             for i, (a, b) in enumerate(zip(ga, gb)) if b is not None]
        return sorted(e, reverse=True)[:3]
    print("  chained vs oracle   ", [(f"{e:.1e}", i) for e, i in worst(grads, g64)])
    print("  per-step vs oracle  ", [(f"{e:.1e}", i) for e, i in worst(grads0, g64)])
    print("  chained vs per-step ", [(f"{e:.1e}", i) for e, i in worst(grads, grads0)])
    for tol in (5e-6, 2e-5, 1e-4):
        _, g_lo = oracle.component_grads(spec, x, g_z, g_l, relu_shift=-tol)
        _, g_hi = oracle.component_grads(spec, x, g_z, g_l, relu_shift=+tol)
        out = 0.0
        for i, (a, b) in enumerate(zip(grads, g64)):
            if b is None: continue
            a = a.cpu().numpy().reshape(b.shape).astype(np.float64)
            lo = np.minimum(np.minimum(g_lo[i], g_hi[i]), b); hi = np.maximum(np.maximum(g_lo[i], g_hi[i]), b)
            out = max(out, float(np.maximum(np.maximum(lo - a, a - hi), 0.0).max() / max(np.abs(b).max(), 1e-3 * float(gscale))))
        print(f"  outside the ReLU bracket at +-{tol:g}: {out:.1e}")
