#!/usr/bin/env python3
"""GPU time of ONE flow launch (f16x3, no repair launch, no recursion) of the MINIBOONE C = 8 model at small batch sizes, on every
library named (tools/build_ab.sh name=flags -> tools/ablate/libgbnf_hip_<name>.so; ablated builds give wrong results: timing only).
    python tools/latency_ablate.py base,dma,barrier [--sizes 64,1024,4096]
Each library runs in its own process (the library path is fixed at import)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("GBNF_LATENCY_CHILD"):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from gbnf_amd import native, synth
    dev = torch.device("cuda:0")
    C, d, h, K = 8, 43, 215, 5
    specs = synth.synth_boosted_specs("glow", C, d, h, K, seed=1)
    flows = [native.NativeFlow(s, math="f16x3") for s in specs]
    mix = native.NativeMixture(flows)
    native.tuning_set("repair", 0)
    for kv in os.environ.get("GBNF_LATENCY_TUNING", "").split(","):
        if "=" in kv:
            native.tuning_set(kv.split("=")[0], int(kv.split("=")[1]))
    out = []
    for n in [int(v) for v in os.environ["GBNF_LATENCY_SIZES"].split(",")]:
        x = torch.from_numpy(synth.synth_batch(n, d, seed=0)).to(dev)
        ll = torch.empty((C, n), device=dev)
        f = lambda: mix.component_log_prob(x, out=ll)
        for _ in range(10): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            f(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                for _ in range(10): f()
        torch.cuda.synchronize()
        for _ in range(5): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): g.replay()
        torch.cuda.synchronize()
        out.append(f"n={n}: {(time.perf_counter() - t0) / 500 * 1e6:6.1f} us")
    print(" | ".join(out), flush=True)
    sys.exit(0)
names = sys.argv[1].split(",")          # a name may carry tuning knobs: shipped:coop=0 | shipped:coop=1:coop_max_wgs=512
sizes = "64,1024,4096"
if "--sizes" in sys.argv: sizes = sys.argv[sys.argv.index("--sizes") + 1]
for full in names:
    name, knobs = full.split(":")[0], ",".join(full.split(":")[1:])
    lib = os.path.join(ROOT, "tools", "ablate", f"libgbnf_hip_{name}.so") if name != "shipped" else os.path.join(ROOT, "gradient-boosted-normalizing-flows_amd", "libgbnf_hip.so")
    env = dict(os.environ, GBNF_LIB_PATH=lib, GBNF_LATENCY_CHILD="1", GBNF_LATENCY_SIZES=sizes, GBNF_LATENCY_TUNING=knobs)
    r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    lines = [l for l in r.stdout.splitlines() if l.startswith("n=")]
    print(f"{full:28s} {lines[-1] if lines else 'FAILED ' + r.stderr[-300:]}", flush=True)
