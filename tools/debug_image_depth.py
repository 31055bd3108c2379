#!/usr/bin/env python3
"""Which path do image coupling nets of depth 0 / 2 run on, and how far is it from the exact-f32 convolutions? (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
dev = torch.device("cuda:0")
if os.environ.get("NOCHECK"):
    native.tuning_set("check_every", -1)
cases = [((1, 28, 20), 256, 2, 1, {"depth": 2}), ((1, 28, 20), 256, 2, 1, {"depth": 0}), ((1, 28, 20), 256, 2, 1, {}), ((3, 32, 32), 256, 2, 2, {"depth": 2}),
         ((1, 28, 28), 32, 2, 2, {"depth": 0, "learn_top": False}), ((1, 28, 20), 64, 2, 1, {"depth": 2}), ((1, 28, 20), 256, 1, 1, {"depth": 2}), ((1, 28, 28), 256, 2, 1, {"depth": 2})]
for size, h, K, L, kw in cases:
    sp = synth.synth_image_glow_spec(size, h, K, L, seed=3, **kw)
    x, noise = synth.synth_image_batch(5, size, seed=4)
    flow = native.NativeImageFlow(sp)
    st = flow.numerics()
    z, ldj, ll = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    os.environ["GBNF_MATH"] = "f32"
    f32 = native.NativeImageFlow(sp)
    del os.environ["GBNF_MATH"]
    _, _, ll32 = f32.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    _, _, _, _, llo = oracle.image_component_forward(sp, x, noise)
    rel = lambda a, b: float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0)))
    torch.cuda.synchronize()
    d = np.abs(ll.cpu().numpy() - ll32.cpu().numpy())
    print("   per-image |ll - ll32|:", d)
    print(size, h, K, L, kw, "mode", native.MATH_NAME[int(st.math_mode)], "probe/worst", float(st.worst_rel_err), "checks", int(st.checks), "demoted", bool(st.demoted),
          "| ll vs f32 bit-equal", bool(torch.equal(ll, ll32)), "rel(ll, f32)", rel(ll.cpu().numpy(), ll32.cpu().numpy()), "rel(ll, oracle)", rel(ll.cpu().numpy(), llo),
          "counts", flow.repair_counts())
