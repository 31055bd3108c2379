#!/usr/bin/env python3
"""The register-chained training kernels on ragged batches with POISONED uninitialised memory: before every step 1 GiB of allocator blocks
is filled with a pattern (quiet NaN / -1 / FLT_MAX as int32) and released, so that every `torch.empty` of the step -- the trace / operand
workspace first of all -- comes back poisoned.  A kernel that reads a row nobody wrote shows up as a non-finite gradient or a fault.
Each case in its own process.  usage: python tools/stress_train_poisoned_memory.py   (GPU, opt-in; 150 cases x 4 steps, ~3.5 min)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1:
    import numpy as np, torch
    from gbnf_amd import native, synth
    from test_hip_train import _dev_spec
    dev = torch.device("cuda:0")
    mode, h, K, n, pat = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    kw = {"res1": dict(coupling_network="residual"), "res2": dict(coupling_network="residual", depth=2), "relu2": dict(depth=2, coupling_network="relu"),
          "tanh1": dict(), "tanh0": dict(depth=0)}[mode]
    spec = synth.synth_realnvp_spec(21, h, K, seed=7, **kw)
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    x = torch.from_numpy(synth.synth_batch(n, 21, seed=8)).to(dev)
    for rep in range(4):
        blocks = [torch.empty(32 << 20, dtype=torch.int32, device=dev) for _ in range(8)]          # 1 GiB of allocator blocks
        for b in blocks:
            b.fill_({"nan": 0x7fc00000, "neg": -1, "big": 0x7f7fffff, "zero": 0}[pat])
        del blocks
        z, ldj, trace = tr.forward(x, want_trace=True)
        gx, grads = tr.backward(x, torch.randn_like(x), torch.randn(n, device=dev), want_gx=True, trace=trace)
        torch.cuda.synchronize()
        assert torch.isfinite(gx).all() and all(torch.isfinite(g).all() for g in grads if g is not None)
    print("ok")
    sys.exit(0)
bad = tot = 0
for mode in ("res1", "res2", "relu2", "tanh1", "tanh0"):
    for h in (250, 105):
        for K, n in ((2, 1), (3, 17), (2, 33), (2, 48), (3, 100)):
            for pat in ("nan", "neg", "big"):
                r = subprocess.run([sys.executable, __file__, mode, str(h), str(K), str(n), pat], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                tot += 1
                if r.returncode != 0 or "ok" not in r.stdout:
                    bad += 1
                    print("BAD", mode, h, K, n, pat, "rc", r.returncode, (r.stderr.strip().splitlines() or [""])[-1][:120])
print(f"poisoned-memory sweep: {bad} bad of {tot}")
