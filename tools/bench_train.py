#!/usr/bin/env python3
"""Training-step timing of ONE component on the HIP training path: forward + backward
(gbnf_trainer_forward / gbnf_trainer_backward) on synthetic data, next to (a) the same step in plain PyTorch on the GPU
(eager autograd over the oracle's op order) and (b) the torch-CPU oracle.  Prints one JSON line.

    python tools/bench_train.py [--config miniboone_glow|hepmass_realnvp] [--batch N] [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

CONFIGS = {"miniboone_glow": dict(kind="glow", d=43, h=215, K=5, kw={}),
           "hepmass_realnvp": dict(kind="realnvp", d=21, h=105, K=5, kw={}),
           # the other coupling networks the reference constructs (--coupling_network_depth 0 / 2, --coupling_network residual)
           "miniboone_glow_depth0": dict(kind="glow", d=43, h=215, K=5, kw={"depth": 0}),
           "miniboone_glow_depth2": dict(kind="glow", d=43, h=215, K=5, kw={"depth": 2}),
           "hepmass_realnvp_depth2": dict(kind="realnvp", d=21, h=105, K=5, kw={"depth": 2}),
           "hepmass_realnvp_residual": dict(kind="realnvp", d=21, h=105, K=5, kw={"coupling_network": "residual"}),
           "hepmass_realnvp_residual2": dict(kind="realnvp", d=21, h=105, K=5, kw={"coupling_network": "residual", "depth": 2})}


def torch_step(spec, x, dev, dtype=torch.float32):
    """One forward+backward of the same component in eager PyTorch (oracle op order) on `dev`."""
    from oracle import gbnf_oracle as oracle

    class Ops(oracle._TorchGradOps):
        def arr(self, a):
            if torch.is_tensor(a):
                return a
            key = id(a)
            if key not in self.leaves:
                self.leaves[key] = (a, torch.tensor(np.asarray(a), dtype=dtype, device=dev, requires_grad=True))
            return self.leaves[key][1]

        def idx(self, a):
            return torch.as_tensor(np.asarray(a), dtype=torch.long, device=dev)

        def zeros(self, n):
            return torch.zeros(n, dtype=dtype, device=dev)

    ops = Ops()

    def step():
        z, ld = x, ops.zeros(x.shape[0])
        for st in spec["steps"]:
            if spec["kind"] == "glow":
                z, ld = oracle.glow_step(ops, spec, st, z, ld)
            else:
                z, sld = oracle.realnvp_step(ops, spec, st, z)
                ld = ld + sld
        loss = torch.mean(-(torch.sum(-0.5 * np.log(2 * np.pi) - 0.5 * z * z, dim=1) + ld))
        for _, t in ops.leaves.values():
            t.grad = None
        loss.backward()
        return loss
    return step


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="miniboone_glow")
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--cpu-rows", type=int, default=4096, help="rows of the batch the CPU baseline leg runs on (bounded sample)")
    ap.add_argument("--tune", action="append", default=[], help="key=value for gbnf_tuning_set (A/B runs)")
    ap.add_argument("--graph", action="store_true", help="also time the step captured once in a HIP graph and replayed (torch.cuda.CUDAGraph)")
    ap.add_argument("--batch-stats", action="store_true", help="RealNVP: BatchNorm on batch statistics (the reference's train() mode, its default training configuration)")
    ap.add_argument("--no-torch-legs", action="store_true", help="skip the eager-PyTorch GPU leg (profiler runs: thousands of tiny dispatches)")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    from gbnf_amd import native, synth
    for kv in a.tune:
        native.tuning_set(kv.split("=")[0], int(kv.split("=")[1]))
    from test_hip_train import _dev_spec
    dev = torch.device("cuda:0")
    spec = synth.synth_boosted_specs(cfg["kind"], 1, cfg["d"], cfg["h"], cfg["K"], seed=1, **cfg["kw"])[0]
    dv = _dev_spec(spec, dev)
    if a.batch_stats:
        for st in dv["steps"]:
            if st.get("bn") is not None:
                st["bn"]["batch_mean"] = torch.zeros(cfg["d"], device=dev)
                st["bn"]["batch_var"] = torch.zeros(cfg["d"], device=dev)
    tr = native.NativeTrainer(dv)
    if a.batch_stats:
        tr.set_batch_stats(True)
    x = torch.from_numpy(synth.synth_batch(a.batch, cfg["d"], seed=0)).to(dev)
    n = a.batch

    def hip_step():
        z, ldj, trace = tr.forward(x, want_trace=True)
        g_z = z / n
        g_l = torch.full((n,), -1.0 / n, device=dev)
        return tr.backward(x, g_z, g_l, trace=trace)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    t_hip = timed(hip_step, a.steps, a.warmup)
    t_graph = None
    if a.graph:
        # the library only enqueues on the current stream (no host synchronisation, no allocation of its own after the first
        # call), so a whole training step -- pack, forward, loss gradient, backward, weight gradients -- captures into ONE graph
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                hip_step()
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out_static = hip_step()
        t_graph = timed(gr.replay, a.steps, a.warmup)
    # kernel-only: events around forward, backward
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    z, ldj, trace = tr.forward(x, want_trace=True)
    g_z = (z / n).contiguous(); g_l = torch.full((n,), -1.0 / n, device=dev)
    torch.cuda.synchronize()
    fws, bws = [], []
    for _ in range(a.steps):
        ev[0].record(); tr.forward(x, want_trace=True); ev[1].record(); tr.backward(x, g_z, g_l, trace=trace); ev[2].record()
        torch.cuda.synchronize()
        fws.append(ev[0].elapsed_time(ev[1])); bws.append(ev[1].elapsed_time(ev[2]))
    fw, bw = float(np.median(fws)), float(np.median(bws))          # (medians: one hiccup of the box must not move the kernel times)
    t_gpu_torch = None if a.no_torch_legs else timed(torch_step(spec, x, dev), a.steps, a.warmup)
    # CPU baseline: the same step (torch-CPU autograd over the oracle's op order) on a BOUNDED sample of the batch's rows;
    # threads: the fastest of a quick probe over {1, 4, 8, 16, 32, 64} on one step of that sample, as bench.py's headline leg
    n_cpu = min(n, a.cpu_rows)
    cpu = None
    host_cores = os.cpu_count() or 1
    cands = [t for t in (1, 4, 8, 16, 32, 64) if t <= host_cores] or [1]
    probe, best_t = {}, cands[0]
    if a.cpu_steps > 0:
        cpu_fn = torch_step(spec, x[:n_cpu].cpu(), torch.device("cpu"))
        cpu_fn()
        for t in cands:
            torch.set_num_threads(t)
            cpu_fn()
            t0 = time.perf_counter()
            cpu_fn()
            probe[t] = n_cpu / (time.perf_counter() - t0)
            if probe[t] >= probe[best_t]:
                best_t = t
        torch.set_num_threads(best_t)
        t0 = time.perf_counter()
        for _ in range(a.cpu_steps):
            cpu_fn()
        t_cpu = (time.perf_counter() - t0) / a.cpu_steps
        cpu = {"value": n_cpu / t_cpu, "unit": "samples/s", "ms_per_step": t_cpu * 1e3, "cores": torch.get_num_threads(),
               "host_cores": host_cores, "thread_probe_samples_per_s": {str(k): v for k, v in probe.items()},
               "kind": "port", "sample": f"{a.cpu_steps} step(s) over {n_cpu} of the {n} rows, forward + backward of one component, "
                                         f"torch-CPU autograd over the oracle's op order, {torch.get_num_threads()} threads (fastest of {cands})"}
    macs = sum((w.shape[0] * w.shape[1]) for st in spec["steps"] for net in ([st["net"]] if cfg["kind"] == "glow" else [st["t_net"], st["s_net"]]) for w, _ in net["layers"])
    flops = 2.0 * macs * n * 3        # forward + dgrad + wgrad
    kern_s = (fw + bw) * 1e-3
    print(json.dumps({
        "metric": f"training step (forward+backward) samples/sec, one component, {a.config}", "unit": "samples/s",
        "value": n / t_hip, "ms_per_step": t_hip * 1e3, "batch": n, "dtype": "f16x3", "data": "synthetic", "batch_stats": bool(a.batch_stats),
        "config": {"workload": f"{a.config}: one component, batch {n}: traced forward, loss gradient, backward (dgrad + wgrad), synthetic weights"},
        "hip_graph_replay": None if t_graph is None else {"value": n / t_graph, "ms_per_step": t_graph * 1e3},
        "forward_kernel_ms": fw, "backward_kernels_ms": bw,
        "algorithmic_tflops": flops / t_hip / 1e12, "algorithmic_tflops_kernels_only": flops / kern_s / 1e12,
        # forward + dgrad + wgrad contractions (3 x 2 x MACs per sample) over the kernels' own time (HIP events), against the
        # dense f16 pipe they run on (three f16 MFMAs per f32 product: executed = 3 x achieved)
        "roofline": {"kernel": "flow_kernel_hx3<TRAIN> + bwd_kernel_hx3 + wgrad_kernel", "bound": "mfma",
                     "achieved": flops / kern_s / 1e12, "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": flops / kern_s / 1e12 / F16_MFMA_PEAK_TFLOPS, "executed_frac": 3.0 * flops / kern_s / 1e12 / F16_MFMA_PEAK_TFLOPS,
                     "traffic": measured_traffic(a.config, n), "launch_ms": fw + bw, "flops_per_launch": flops},
        "torch_gpu_eager": None if t_gpu_torch is None else {"value": n / t_gpu_torch, "ms_per_step": t_gpu_torch * 1e3},
        "cpu_baseline": cpu,
        "speedup_vs_torch_gpu_eager": None if t_gpu_torch is None else t_gpu_torch / t_hip,
        "speedup_vs_cpu": None if cpu is None else (n / t_hip) / cpu["value"]}))


F16_MFMA_PEAK_TFLOPS = 2500.0      # /opt/skills/guides/MI355X_MICROARCH.md: dense bf16 / f16 matrix peak


def measured_traffic(config, n):
    """HBM-side bytes per training step (all kernels) from the PMC passes committed under profiles/ (2 x FETCH_SIZE + WRITE_SIZE,
    MI355X_MICROARCH.md) -- only for the exact workload they were taken on."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "train_traffic.json")))
    except (OSError, ValueError):
        return None
    for r in rec.get("workloads", []):
        if r.get("workload", {}).get("config") == config and r.get("workload", {}).get("batch") == n:
            return float(r["traffic_bytes_per_step"])
    return None


if __name__ == "__main__":
    main()
