#!/usr/bin/env python3
"""Per-call latency of the one-call evaluation forms at the reference's own batch sizes (density_experiment.py:80-81: 512 rows to
train on, 1024 to evaluate): model.log_prob(x) = one flow launch (+ its repair launch) + the recursion launch, MINIBOONE Boosted-Glow
C = 8.  Prints us per call (stream launches, and the same call replayed as a HIP graph) per batch size and launch policy.
usage: python tools/bench_latency.py [--policies]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth

dev = torch.device("cuda:0")
C, d, h, K = 8, 43, 215, 5
specs = synth.synth_boosted_specs("glow", C, d, h, K, seed=1)
flows = [native.NativeFlow(s) for s in specs]
mix = native.NativeMixture(flows)
rho = torch.from_numpy(np.maximum(1.0 / np.power(2.0, np.arange(C)), 0.05).astype(np.float32)).to(dev)
policies = [("default", -1, 0)]
if "--policies" in sys.argv:
    policies += [("4-wave workgroups", 1, 0), ("8-wave workgroups", 0, 0), ("4-wave, 16-sample waves", 1, 1), ("4-wave, 32-sample waves", 1, 2)]
for n in (64, 256, 512, 1024, 2048, 4096):
    x = torch.from_numpy(synth.synth_batch(n, d, seed=0)).to(dev)
    ll = torch.empty((C, n), device=dev); G = torch.empty(n, device=dev)
    row = [f"n = {n:5d}"]
    for name, pairs, nt in policies:
        native.tuning_set("wg_pairs", pairs); native.tuning_set("force_nt", nt)
        f = lambda: mix.log_prob(x, rho, ll_out=ll, out=G)
        for _ in range(20): f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): f()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 300 * 1e6
        # GPU time of the call: the same launches replayed as a graph, back to back
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                f()
        torch.cuda.synchronize()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(300): g.replay()
        torch.cuda.synchronize()
        ug = (time.perf_counter() - t0) / 300 * 1e6
        row.append(f"{name}: {us:6.1f} us stream / {ug:6.1f} us graph ({n / ug:5.1f} M samples/s)")
    print(" | ".join(row), flush=True)
native.tuning_set("wg_pairs", -1); native.tuning_set("force_nt", 0)
