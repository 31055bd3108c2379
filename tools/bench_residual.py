#!/usr/bin/env python3
"""Density evaluation of a Boosted-RealNVP whose coupling networks are one-block ResidualNets (`--coupling_network residual`,
models/layers.py:246-301): the exact-f32 kernel against the split kernels (round 3).  GPU box.

    python tools/bench_residual.py [--d 21 --hidden 105 --batch 65536]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=21)
    ap.add_argument("--hidden", type=int, default=105)
    ap.add_argument("--components", type=int, default=8)
    ap.add_argument("--K", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--blocks", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    specs = synth.synth_boosted_specs("realnvp", a.components, a.d, a.hidden, a.K, seed=1, coupling_network="residual", depth=a.blocks)
    x = torch.from_numpy(synth.synth_batch(a.batch, a.d, seed=0)).to(dev)
    rho = torch.from_numpy(oracle.rho_init(a.components)).to(dev)
    out = {"workload": f"Boosted-RealNVP, ResidualNet coupling ({a.blocks} block(s)), d={a.d} h={a.hidden} K={a.K} C={a.components} batch={a.batch}"}
    ref = None
    for math in ("f32", "f16x3", "bf16x6", "default"):
        flows = [native.NativeFlow(s, math=math) for s in specs]
        mix = native.NativeMixture(flows)
        for _ in range(3):
            G, ll = mix.log_prob(x, rho)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            G, ll = mix.log_prob(x, rho)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        g = G.cpu().numpy()
        if ref is None:
            ref = g
        out[math] = {"samples_per_s": a.batch / dt, "ms": dt * 1e3, "kernel": native.MATH_NAME[flows[0].info().math_mode],
                     "max_rel_err_vs_f32": float(np.max(np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
