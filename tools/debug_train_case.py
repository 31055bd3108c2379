#!/usr/bin/env python3
"""Replay case k of `python tools/stress_train.py <cases> <seed>` (same random stream) with per-tensor errors, optionally
at other batch sizes (debug aid).   usage: python tools/debug_train_case.py seed k [n ...]   |   seed find kind d h K n"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
from test_hip_train import _dev_spec
from stress_train import gen_case

seed = int(sys.argv[1])
if sys.argv[2] == "find":              # ... seed find kind d h K n: the first case of the stream with that geometry
    target = (sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]))
    want, ns = 100000, []
else:
    target, want, ns = None, int(sys.argv[2]), [int(v) for v in sys.argv[3:]]
rng = np.random.RandomState(seed)
dev = torch.device("cuda:0")
for k in range(want + 1):
    kind, d, h, K, n, extra, spec = gen_case(rng, k)
    if target is not None and (kind, d, h, K, n) == target:
        want = k
    try:
        native.NativeTrainer(_dev_spec(spec, dev))
    except native.GbnfError:
        continue                       # (the stress tool draws g_z, g_l only for supported cases)
    x = synth.synth_batch(n, d, seed=k)
    gscale = np.float32(10.0 ** rng.uniform(-6, 4))
    g_z = rng.standard_normal(x.shape).astype(np.float32) * gscale
    g_l = rng.standard_normal(n).astype(np.float32) * gscale
    if k == want:
        break
print("case", k, kind, d, h, K, n, extra, "activations:", native.activation_pattern(spec))
tr = native.NativeTrainer(_dev_spec(spec, dev))
for m in (ns or [n]):
    xs, gz, gl = x[:m], g_z[:m], g_l[:m]
    gx64, gr64 = oracle.component_grads(spec, xs, gz, gl)
    z64, l64 = oracle.component_forward(spec, xs, backend="numpy64")
    xd = torch.from_numpy(xs.copy()).to(dev)
    z, ldj, trace = tr.forward(xd, want_trace=True)
    gx, gr = tr.backward(xd, torch.from_numpy(gz.copy()).to(dev), torch.from_numpy(gl.copy()).to(dev), want_gx=True, trace=trace)
    errs = sorted(((float(np.abs(a.cpu().numpy().reshape(b.shape) - b).max() / max(np.abs(b).max(), 1e-3)), i, b.shape)
                   for i, (a, b) in enumerate(zip(gr, gr64)) if b is not None), reverse=True)[:5]
    print(f"n={m} NT={os.environ.get('GBNF_TRAIN_NT', 'auto')}: z err {np.abs(z.cpu().numpy() - z64).max():.1e} ldj err "
          f"{np.abs(ldj.cpu().numpy() - l64).max():.1e} |ldj|max {np.abs(l64).max():.1e}; gx err {np.abs(gx.cpu().numpy() - gx64).max() / np.abs(gx64).max():.1e}; "
          f"finite {bool(torch.isfinite(gx).all())}; worst grads {[(f'{e:.1e}', i, s) for e, i, s in errs]}")
