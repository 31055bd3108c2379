#!/usr/bin/env python3
"""Small batches (1, 17, 33 rows: the last workgroup of the backward sweep has spare waves) through the register-chained training
kernels of every net type (ResidualNets of one and two blocks, depth 0 / 1 / 2) and width (4 .. 32 hidden tiles), each case in its own
process (a GPU memory fault takes the process down): no fault, and the gradients next to the per-step kernels' on the same call.
usage: python tools/stress_train_small_batches.py      (GPU, opt-in; written for the fault of HISTORY round 5)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import numpy as np, torch
    from gbnf_amd import native, synth
    from test_hip_train import _dev_spec
    dev = torch.device("cuda:0")
    mode, d, h, K, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    kw = {"res1": dict(coupling_network="residual"), "res2": dict(coupling_network="residual", depth=2), "rnvp0": dict(depth=0), "rnvp1": dict(),
          "rnvp2": dict(depth=2), "relu2": dict(depth=2, coupling_network="relu")}[mode]
    spec = synth.synth_realnvp_spec(d, h, K, seed=7, **kw)
    x = torch.from_numpy(synth.synth_batch(n, d, seed=8)).to(dev)
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    z, ldj, trace = tr.forward(x, want_trace=True)
    g_z = torch.randn_like(x); g_l = torch.randn(n, device=dev)
    gx, grads = tr.backward(x, g_z, g_l, want_gx=True, trace=trace)
    gx0, grads0 = tr.backward(x, g_z, g_l, want_gx=True)            # per-step kernels
    torch.cuda.synchronize()
    err = max(float((a - b).abs().max() / max(float(b.abs().max()), 1e-3)) for a, b in zip(grads, grads0) if a is not None)
    print("ok", f"{err:.1e}")
    sys.exit(0)
bad = 0
for mode in ("res1", "res2", "rnvp0", "rnvp1", "rnvp2", "relu2"):
    for h in (64, 105, 215, 250, 300, 500):
        if mode == "res2" and h > 256: continue
        for K, n in ((2, 1), (3, 17), (2, 33)):
            r = subprocess.run([sys.executable, __file__, mode, "21", str(h), str(K), str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            out = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "no output"
            flag = r.returncode != 0 or not out.startswith("ok") or float(out.split()[1]) > 2e-1
            bad += flag
            if flag: print("BAD", mode, h, K, n, "rc", r.returncode, out, r.stderr.strip().splitlines()[-1][:100] if r.stderr.strip() else "")
print("spare-wave sweep:", bad, "bad")
