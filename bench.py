#!/usr/bin/env python3
"""Headline benchmark: density-eval samples/sec of a Boosted-Glow mixture on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric): MINIBOONE-shaped d=43, Boosted-Glow C=8 (K=5 flow steps, coupling
net 21->215->215->44 tanh, shuffle permutation, affine coupling), batch 4096 synthetic N(0,1) rows
already resident in HBM, random-init synthetic weights.  One "step" = one pass of the hot path over
one batch: all C component flows + the mixture log-sum-exp  ->  G (N,).

N > 1: the C components are sharded over the ranks (contiguous blocks, C/N each), x is replicated,
one RCCL all-gather of float32[C/N, batch] per rank per step rebuilds (C, batch) before the
recursion; total work is fixed => "scaling": "strong".

Timing: 2000 x `--prewarm` (default 0.3, reported as "prewarm_s": about that many seconds on one GPU) untimed steps let
the clocks of a cold GPU settle, then W untimed warm-up steps, then exactly K timed steps between barriers + device synchronisations.

Prints ONE JSON line on rank 0 (see the driver contract), including
  "roofline":     dominant kernel (the fused flow kernel) against the dense f32-MFMA peak
  "cpu_baseline": the torch-CPU oracle ("port" of the reference path) timed on this box's cores.
"""
import argparse
import ctypes
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

F32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense f32 matrix peak
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: dense bf16/f16 matrix peak (no sparsity)
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E spec peak

CONFIGS = {
    # name: (kind, d, h, K, default C, default batch)
    "miniboone_glow": ("glow", 43, 215, 5, 8, 4096),
    "hepmass_realnvp": ("realnvp", 21, 105, 5, 8, 65536),
}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(specs, rho, x_np, budget_s):
    """Times the oracle (torch CPU, the reference's op order) on a bounded sample of the SAME workload.

    Thread count: the reference sets torch.set_num_threads(cores - 1) (density_experiment.py:280-287),
    which is pathological on a many-core host for these small GEMMs (256 threads: ~100 s per pass), so
    the baseline is given its best setting: a quick probe over {1, 4, 8, 16, 32, 64} picks the fastest,
    and `cores` reports the threads actually used.  Then whole passes (all components + recursion) are
    timed until ~budget_s seconds are spent (first pass untimed)."""
    import torch
    from oracle import gbnf_oracle as oracle
    host_cores = os.cpu_count() or 1
    cands = [t for t in (1, 4, 8, 16, 32, 64) if t <= host_cores] or [1]
    probe_x = x_np[: min(512, x_np.shape[0])]
    best_t, best_dt = cands[0], None
    one_thread = None
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            oracle.component_log_prob(specs[0], probe_x)             # warm
            t0 = time.perf_counter()
            oracle.component_log_prob(specs[0], probe_x)
            dt = time.perf_counter() - t0
            if t == 1:      # the reference scripts' own default is one worker: record that figure too (probe-sized sample)
                one_thread = probe_x.shape[0] / (dt * len(specs))
            if best_dt is None or dt < best_dt:
                best_t, best_dt = t, dt
            if dt > 2.0:
                break
        torch.set_num_threads(best_t)
        est = best_dt * len(specs) * x_np.shape[0] / probe_x.shape[0]
        n = x_np.shape[0]
        if est > budget_s / 2:                                        # keep the leg bounded: fewer rows
            n = max(256, int(x_np.shape[0] * (budget_s / 2) / est))
        xs = x_np[:n]
        ll, G = oracle.mixture_log_prob(specs, rho, xs)               # warm-up pass (untimed)
        t0 = time.perf_counter()
        passes = 0
        while True:
            ll, G = oracle.mixture_log_prob(specs, rho, xs)
            passes += 1
            el = time.perf_counter() - t0
            if el >= budget_s or passes >= 500:
                break
    return {
        "value": n * passes / el, "unit": "samples/s", "cores": best_t, "kind": "port",
        "host_cores": host_cores, "cpu_model": _cpu_model(), "one_thread_value": one_thread,
        "sample": f"{passes} pass(es) over {n} of the {x_np.shape[0]} rows, all {len(specs)} components + mixture "
                  f"recursion, torch-CPU oracle in the reference's op order, {best_t} threads "
                  f"(fastest of {cands}), {el:.1f} s",
    }, G, n


def measured_traffic(config, B, C, S, f16x3, world):
    """HBM-side bytes per launch of the dominant kernel from the PMC passes committed under profiles/ (FETCH_SIZE x 2
    + WRITE_SIZE, collected as MI355X_MICROARCH.md prescribes) -- only for the exact workload they were taken on."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "headline_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    w = rec.get("workload", {})
    same = (w.get("config") == config and w.get("batch") == B and w.get("components") == C and w.get("group") == S
            and w.get("math") == ("f16x3" if f16x3 else "f32") and w.get("n_gpus") == world)
    return float(rec["traffic_bytes_per_launch"]) if same else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--prewarm", type=float, default=0.3, help="untimed launches before the warm-up steps (clock ramp): 2000 steps per unit, i.e. about this many seconds on one GPU")
    ap.add_argument("--config", default="miniboone_glow", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--components", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline leg (0 = skip)")
    ap.add_argument("--math", default="default", choices=["default", "f32", "f16x3"],
                    help="matrix path: exact-f32 MFMA or split-f16 (3 f16 MFMAs per f32 product)")
    ap.add_argument("--force-gather", action="store_true",
                    help="testing aid: run the RCCL all-gather leg even with one rank")
    ap.add_argument("--group", type=int, default=0,
                    help="batches served per launch / per all-gather (0 = auto: 8..16, max 16)")
    args = ap.parse_args()

    # dmabuf IPC is the only form the host driver supports (without it RCCL fails with hipIpcGetMemHandle: invalid argument)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from gbnf_amd import native, sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank != 0:          # only rank 0 reports: nothing another rank (or a library it loads) writes may follow the JSON line
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    kind, d, h, K, C_def, B_def = CONFIGS[args.config]
    C = args.components or C_def
    B = args.batch or B_def

    # identical bytes on every rank: synthetic inputs and weights from fixed seeds
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=1)
    x_np = synth.synth_batch(B, d, seed=0)
    rho_np = np.maximum(1.0 / np.power(2.0, np.arange(C)), 0.05).astype(np.float32)   # "decreasing"
    parts = sharded.partition(C, world)
    c0, c1 = parts[rank]
    flows = [native.NativeFlow(specs[c], math=args.math) for c in range(c0, c1)]
    mix = native.NativeMixture(flows)
    info = flows[0].info()
    x = torch.from_numpy(x_np).to(dev)
    rho = torch.from_numpy(rho_np).to(dev)

    # ---- pipeline.  Consecutive steps (batches) are independent, so they are served in GROUPS of S: one flow
    # launch covers the S batches of a group (for a rank holding C/N components a single batch fills only 1/N
    # of its GPU), ONE RCCL all-gather rebuilds the (C, S*B) table of the group and ONE recursion launch
    # finishes it -- "bigger, fewer collectives".  The flow kernels run on one stream, gather + recursion on a
    # second one, double-buffered, so the exchange of group g overlaps the kernel of group g+1.
    gather = world > 1 or args.force_gather
    # auto: at least 8, and enough batches per launch for two co-resident workgroups per CU (>= 512 workgroups)
    S = args.group if args.group > 0 else max(8, (16 * world + C - 1) // C)
    S = max(1, min(S, 16, args.steps))
    xs_np = [x_np] + [synth.synth_batch(B, d, seed=100 + s) for s in range(1, S)]   # S distinct resident batches
    xs = [x] + [torch.from_numpy(a).to(dev) for a in xs_np[1:]]
    NBUF = 2
    main = torch.cuda.current_stream()
    post = torch.cuda.Stream() if gather else main
    mptr, pptr = ctypes.c_void_p(main.cuda_stream), ctypes.c_void_p(post.cuda_stream)
    flow_done = [torch.cuda.Event() for _ in range(NBUF)]
    post_done = [torch.cuda.Event() for _ in range(NBUF)]

    class Group:
        """Buffers + pre-bound launches for groups of `size` batches (double-buffered).  Every argument is bound
        once: per group the host does two ctypes calls (+ one RCCL call when sharded)."""

        def __init__(self, size):
            self.size = size
            self.local = [torch.empty((c1 - c0, size * B), dtype=torch.float32, device=dev) for _ in range(NBUF)]
            self.full = ([torch.empty((C, size * B), dtype=torch.float32, device=dev) for _ in range(NBUF)]
                         if gather else self.local)
            self.G = [torch.empty(size * B, dtype=torch.float32, device=dev) for _ in range(NBUF)]
            self.flow = [mix.prepared_group_log_prob(xs[:size], self.local[q]) for q in range(NBUF)]
            self.lse = [native.prepared_mixture_lse(self.full[q], rho, self.G[q]) for q in range(NBUF)]

    groups = {S: Group(S)}
    for t in {args.steps % S, args.warmup % S} - {0}:       # ragged last group of the timed / warm-up run
        groups[t] = Group(t)
    n_timed_groups = (args.steps + S - 1) // S
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(n_timed_groups)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(n_timed_groups)]
    if gather and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def run(n_steps, timed):
        """n_steps batches in ceil(n_steps / S) groups; every batch gets its flow pass, exchange and recursion."""
        gi = 0
        done = 0
        while done < n_steps:
            q = gi % NBUF
            grp = groups[min(S, n_steps - done)]
            if gather and gi >= NBUF:
                main.wait_event(post_done[q])          # group gi-2 has released this buffer pair
            if timed:
                ev0[gi].record(main)
            grp.flow[q](mptr)
            if timed:
                ev1[gi].record(main)
            if gather:
                flow_done[q].record(main)
                post.wait_event(flow_done[q])
                with torch.cuda.stream(post):
                    dist.all_gather_into_tensor(grp.full[q], grp.local[q])
                grp.lse[q](pptr)
                post_done[q].record(post)
            else:
                grp.lse[q](mptr)
            done += grp.size
            gi += 1

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock settle phase (untimed, on top of the W warm-up steps, reported as "prewarm_s"): a cold MI355X runs the first
    # few hundred steps ~10 % slower while its clocks ramp up; the timed region is meant to see the steady state
    # (a FIXED number of steps: every rank must issue the same collectives)
    run(int(args.prewarm * 2000) // S * S, False)
    run(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    full_groups = args.steps // S
    kern_ms = sum(ev0[i].elapsed_time(ev1[i]) for i in range(full_groups)) / max(1, full_groups)   # per GROUP launch
    G_gpu = groups[S].G[(full_groups - 1) % NBUF][:B].cpu().numpy()      # batch 0 of the last full group

    if rank == 0:
        value = B * args.steps / elapsed
        f16x3 = info.math_mode == native.MATH["f16x3"]
        flops_per_launch = 2.0 * info.macs_per_sample * (c1 - c0) * B * S      # one launch serves S batches
        achieved_tf = flops_per_launch / (kern_ms * 1e-3) / 1e12
        # what the matrix pipe executes: tile padding, and 3 f16 products per f32 product on the split path
        executed_tf = achieved_tf * info.padded_macs_per_sample / info.macs_per_sample * (3.0 if f16x3 else 1.0)
        peak = F16_MFMA_PEAK_TFLOPS if f16x3 else F32_MFMA_PEAK_TFLOPS
        alg_bytes = (4.0 * d + 4.0 * (c1 - c0)) * B * S      # read x once, write ll per component
        out = {
            "metric": "density-eval samples/sec, Boosted-Glow C=8 MINIBOONE d=43" if args.config == "miniboone_glow"
                      else f"density-eval samples/sec, {args.config}",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "prewarm_s": args.prewarm, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: d={d} h={h} K={K} C={C} batch={B}, x ~ N(0,1), synthetic weights",
                       "global_batch": B, "components": C,
                       "math": "f16x3: f32 operands split into two fp16 pieces, 3 f16 MFMAs per product, f32 accumulate"
                               if f16x3 else "f32: exact f32-input MFMA",
                       "parallelism": "single GPU, all components in one launch" if world == 1
                       else f"components sharded {C // world}/GPU + one RCCL all-gather of ll per group",
                       "group": f"{S} batches per flow launch / all-gather / recursion launch"},
            "roofline": {
                "kernel": "gbnf::flow_kernel_hx3" if f16x3 else "gbnf::flow_kernel",
                "bound": "mfma", "achieved": achieved_tf, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved_tf / peak, "traffic": measured_traffic(args.config, B, C, S, f16x3, world),
                "launch_ms": kern_ms, "flops_per_launch": flops_per_launch,
                "executed_mfma_tflops": executed_tf, "executed_frac": executed_tf / peak,
                "vs_f32_mfma_peak": achieved_tf / F32_MFMA_PEAK_TFLOPS,
                "hbm_algorithmic_GBs": alg_bytes / (kern_ms * 1e-3) / 1e9,
                "hbm_frac": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "achieved = ALGORITHMIC f32 flops / launch time; the split path runs 3x (+padding) that on "
                        "the f16 pipe. The binding limit is instruction issue: VALU (tanh, hi/mid split) and MFMA "
                        "serialise on a SIMD (tools/ubench), see DESIGN.md section 4",
            },
        }
        if args.cpu_seconds > 0 and world == 1:
            cb, G_cpu, n_cpu = cpu_baseline(specs, rho_np, x_np, args.cpu_seconds)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu"] = value / cb["value"]
            err = float(np.max(np.abs(G_gpu[:n_cpu].astype(np.float64) - G_cpu) / np.maximum(np.abs(G_cpu), 1.0)))
            out["max_rel_err_vs_cpu"] = err
        elif args.cpu_seconds > 0:
            out["cpu_baseline"] = None     # reported on the N=1 run only
        line = json.dumps(out)
    else:
        line = None

    # RCCL writes a version banner to (C-level, buffered) stdout: tear the process group down and flush that buffer first,
    # so that the JSON line is the LAST line on stdout
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)       # ... its banner sits in the C library's stdout buffer until then
    except OSError:
        pass
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
