#!/usr/bin/env python3
"""GPU time of one flow launch (f16x3, C = 8, no repair launch) per kernel form for every geometry with a latency-form variant:
python tools/bench_coop_geometries.py [n ...]   (forms: 0 = throughput kernel, 1 / 2 / 3 = cooperative, -1 = the shipped policy)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
GEOMETRIES = [("glow", 43, 215, {}), ("realnvp", 21, 105, {}), ("glow", 43, 64, {}), ("glow", 21, 105, {}), ("glow", 63, 128, {}), ("glow", 63, 250, {}),
              ("glow", 63, 315, {}), ("glow", 43, 215, {"act": "relu"}), ("glow", 63, 250, {"act": "relu"}),
              ("realnvp", 21, 105, {"coupling_network": "relu"}), ("realnvp", 21, 105, {"coupling_network": "mixed"}),
              ("realnvp", 43, 215, {}), ("realnvp", 63, 250, {}), ("realnvp", 63, 315, {})]
dev = torch.device("cuda:0")
sizes = [int(v) for v in sys.argv[1:]] or [512, 1024]
native.tuning_set("repair", 0)
for kind, d, h, kw in GEOMETRIES:
    specs = synth.synth_boosted_specs(kind, 8, d, h, 5, seed=1, **kw)
    mix = native.NativeMixture([native.NativeFlow(s, math="f16x3") for s in specs])
    row = [f"{kind:8s} d={d:2d} h={h:3d} {str(kw):34s}"]
    for n in sizes:
        x = torch.from_numpy(synth.synth_batch(n, d, seed=0)).to(dev)
        ll = torch.empty((8, n), device=dev)
        cells = []
        for form in (0, 1, 2, 3, -1):
            native.tuning_set("coop", form)
            f = lambda: mix.component_log_prob(x, out=ll)
            try:
                f()
            except native.GbnfError:
                cells.append(" fail")
                continue
            for _ in range(5): f()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                f(); torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    for _ in range(10): f()
            torch.cuda.synchronize()
            for _ in range(3): g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(30): g.replay()
            torch.cuda.synchronize()
            cells.append(f"{(time.perf_counter() - t0) / 300 * 1e6:5.1f}")
        row.append(f"n={n}: thr {cells[0]} | f1 {cells[1]} | f2 {cells[2]} | f3 {cells[3]} | auto {cells[4]} us")
    native.tuning_set("coop", -1)
    print("  ".join(row), flush=True)
