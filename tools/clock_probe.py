#!/usr/bin/env python3
"""The clock the chip holds under the headline kernel (diagnostic, GPU box only).

    bash tools/build_clock.sh && GBNF_LIB_PATH=$PWD/tools/libgbnf_hip_clock.so python tools/clock_probe.py [--zeros] [--batches S]

The -DGBNF_CLOCK build stamps s_memtime (shader clock) and s_memrealtime (100 MHz) around every workgroup; their
quotient x 100 MHz is the clock that workgroup ran at (MI355X_MICROARCH.md, "DVFS give-back" item 6).  The probe
launches the headline workload (MINIBOONE Boosted-Glow C = 8, S batches of 4096 per launch, f16x3) back to back for
`--seconds`, then reads the stamps of the last launch: median / min / max over its workgroups, next to the launch
time by HIP events.  `--zeros` feeds all-zero inputs (the data-dependent part of the power draw: the guide's zero-fill
control)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gbnf_amd import native, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--components", type=int, default=8)
    args = ap.parse_args()
    os.environ["GBNF_NO_REPAIR"] = "1"
    C_, B, S = args.components, 4096, args.batches
    specs = synth.synth_boosted_specs("glow", C_, 43, 215, 5, seed=1)
    dev = torch.device("cuda:0")
    flows = [native.NativeFlow(s, math="f16x3") for s in specs]
    mix = native.NativeMixture(flows)
    xs = [torch.from_numpy(synth.synth_batch(B, 43, seed=k)).to(dev) for k in range(S)]
    if args.zeros:
        xs = [torch.zeros_like(x) for x in xs]
    table = torch.empty((C_, S * B), dtype=torch.float32, device=dev)
    launch = mix.prepared_group_log_prob(xs, table)
    n_wg_max = 1 << 16
    buf = torch.zeros(2 * n_wg_max, dtype=torch.int64, device=dev)
    L = native.lib()
    L.gbnf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    sp = native._stream_ptr()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        for _ in range(50):
            launch(sp)
        torch.cuda.synchronize()
        n += 50
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        launch(sp)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    t = buf.cpu().numpy().reshape(-1, 2)
    t = t[(t[:, 0] > 0) & (t[:, 1] > 0)]
    ghz = t[:, 0] / t[:, 1] * 0.1
    us = t[:, 1] / 100.0
    out = {"workload": f"miniboone_glow C={C_} batch={B} x {S} batches per launch, f16x3, {'zeros' if args.zeros else 'random'} inputs",
           "kernel": flows[0].info().kernel if hasattr(flows[0].info(), "kernel") else None,
           "warm_launches": n, "launch_ms": ms, "samples_per_s": B * S / (ms * 1e-3), "workgroups": int(len(t)),
           "clock_GHz_median": float(np.median(ghz)), "clock_GHz_min": float(ghz.min()), "clock_GHz_max": float(ghz.max()),
           "workgroup_us_median": float(np.median(us)), "workgroup_cycles_median": float(np.median(t[:, 0]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
