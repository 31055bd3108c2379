"""GPU debugging aid: per-tensor gradient errors of the traced (register-chained) and untraced training paths against the
float64 oracle, for the golden cases named on the command line."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
torch.cuda.init()
from conftest import GoldenCase
from test_hip_train import _dev_spec
from gbnf_amd import native
from oracle import gbnf_oracle as oracle

dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    if name.startswith("g10_"):
        from conftest import load_grads_case
        import types
        cfg, spec, xx, nll, flat, g_x = load_grads_case(name)
        print(cfg)
        g = types.SimpleNamespace(x=xx)
    else:
        g = GoldenCase(name)
        spec = g.specs[0]
    rng = np.random.RandomState(7)
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    x = torch.from_numpy(g.x).to(dev)
    g_z = rng.standard_normal(g.x.shape).astype(np.float32)
    g_l = rng.standard_normal(g.x.shape[0]).astype(np.float32)
    gx64, grads64 = oracle.component_grads(spec, g.x, g_z, g_l)
    for traced in (False, True):
        trace = tr.forward(x, want_trace=True)[2] if traced else None
        gx, grads = tr.backward(x, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True, trace=trace)
        errs = []
        for a, b in zip(grads, grads64):
            if b is None: errs.append(None); continue
            a = a.cpu().numpy().reshape(b.shape)
            errs.append(float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-3)))
        print(name, "traced" if traced else "plain", "gx", float(np.abs(gx.cpu().numpy() - gx64).max() / np.abs(gx64).max()))
        print("   ", " ".join("-" if e is None else f"{e:.1e}" for e in errs))
        print("    shapes", [None if b is None else b.shape for b in grads64][:12])
        print("    gx err per feature", np.abs(gx.cpu().numpy() - gx64).max(axis=0) / np.abs(gx64).max())
