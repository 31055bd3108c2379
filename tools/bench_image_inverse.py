#!/usr/bin/env python3
"""z -> x (sampling) throughput of one image Glow component: python tools/bench_image_inverse.py [--batch 256] [--input 3 32 32]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--input", type=int, nargs=3, default=[3, 32, 32])
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--K", type=int, default=8)
ap.add_argument("--L", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
size = tuple(a.input)
sp = synth.synth_image_glow_spec(size, a.hidden, a.K, a.L, seed=100)
fl = native.NativeImageFlow(sp)
z = 0.7 * torch.randn((a.batch,) + fl.z_shape, device=dev)
eps = [torch.randn((a.batch,) + tuple(s), device=dev) for s in fl.split_shapes()]
for _ in range(3):
    x = fl.inverse(z, eps, 0.9)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    x = fl.inverse(z, eps, 0.9)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
xn, nn_ = synth.synth_image_batch(a.batch, size, seed=0)
xd, nd = torch.from_numpy(xn).to(dev), torch.from_numpy(nn_).to(dev)
for _ in range(3):
    fl.forward(xd, nd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    fl.forward(xd, nd)
torch.cuda.synchronize()
df = (time.perf_counter() - t0) / a.steps
print(f"one component, {size}, batch {a.batch}: inverse {a.batch / dt:.0f} images/s ({dt * 1e3:.2f} ms), forward {a.batch / df:.0f} images/s ({df * 1e3:.2f} ms)")
