#!/usr/bin/env python3
"""Replay case k of tools/stress_train.py (same random stream) with per-tensor errors, for several batch sizes (debug aid).
usage: python tools/debug_train_case.py k [n ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
from test_hip_train import _dev_spec
want = int(sys.argv[1]); ns = [int(v) for v in sys.argv[2:]]
rng = np.random.RandomState(2024)
for k in range(want + 1):
    kind = "glow" if rng.randint(3) else "realnvp"
    d = int(rng.choice([2, 3, 5, 6, 8, 13, 21, 33, 43, 50, 63, 64]))
    h = int(rng.choice([5, 16, 30, 33, 64, 105, 129, 215, 256, 257, 315, 430, 512]))
    K = int(rng.randint(1, 7))
    n = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 100, 257, 1000, 1537, 2049, 3000]))
    depth = int(rng.choice([0, 1, 1, 1, 2]))
    if kind == "glow":
        extra = dict(act=str(rng.choice(["tanh", "relu"])), coupling=str(rng.choice(["affine", "additive"])),
                     permutation=str(rng.choice(["shuffle", "reverse"])), depth=depth)
        spec = synth.synth_glow_spec(d, h, K, seed=5000 + k, **extra)
    else:
        extra = dict(coupling_network=str(rng.choice(["tanh", "relu", "mixed"])), batch_norm=bool(rng.randint(2)),
                     flip_init=int(rng.randint(2)), depth=depth)
        spec = synth.synth_realnvp_spec(d, h, K, seed=5000 + k, **extra)
    x = synth.synth_batch(n, d, seed=k)
    g_z = rng.standard_normal(x.shape).astype(np.float32)
    g_l = rng.standard_normal(n).astype(np.float32)
print(kind, d, h, K, n, extra)
dev = torch.device("cuda:0")
tr = native.NativeTrainer(_dev_spec(spec, dev))
for m in (ns or [n]):
    xs, gz, gl = x[:m], g_z[:m], g_l[:m]
    gx64, gr64 = oracle.component_grads(spec, xs, gz, gl)
    xd = torch.from_numpy(xs.copy()).to(dev)
    z, ldj, trace = tr.forward(xd, want_trace=True)
    gx, gr = tr.backward(xd, torch.from_numpy(gz.copy()).to(dev), torch.from_numpy(gl.copy()).to(dev), want_gx=True, trace=trace)
    errs = sorted(((float(np.abs(a.cpu().numpy().reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-3)), i, b.shape)
                   for i, (a, b) in enumerate(zip(gr, gr64)) if b is not None), reverse=True)[:4]
    print(f"n={m} NT={os.environ.get('GBNF_TRAIN_NT', 'auto')} chunk={os.environ.get('GBNF_WGRAD_CHUNK', 'auto')}: gx err "
          f"{np.abs(gx.cpu().numpy() - gx64).max() / np.abs(gx64).max():.1e}; worst grads {[(f'{e:.1e}', i, s) for e, i, s in errs]}")
