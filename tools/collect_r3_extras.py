#!/usr/bin/env python3
"""gpurun_out/final/ (tools/r3_final_b.sh, tools/r3_final_c.sh) -> profiles/r3_final_*: the driver's own invocation under the profiler, the
emulated per-rank lines, the module evaluate loop, the training step (kernel stats, PMC).  Run after collect_final_profiles.py r3."""
import csv
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(REPO, "gpurun_out", "final")
P = os.path.join(REPO, "profiles")


def last_json(path):
    lines = [x for x in open(path).read().splitlines() if x.startswith("{")]
    return lines[-1], json.loads(lines[-1])


def stats_rows(name, only_gbnf=False, top=14):
    rows = list(csv.DictReader(open(os.path.join(F, name + ".kernel_stats.csv"))))
    out = []
    for r in rows:
        if only_gbnf and "gbnf" not in r["Name"] and len(out) >= top:
            continue
        out.append("%-96s calls=%6s avg_ns=%12.1f min_ns=%9s max_ns=%9s pct=%s"
                   % (r["Name"][:96], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
    return out


def main():
    # 1. the driver's invocation
    l, d = last_json(os.path.join(F, "bench_steps20.json"))
    open(os.path.join(P, "r3_final_bench_line_driver_invocation_steps20.json"), "w").write(l + "\n")
    _, dp = last_json(os.path.join(F, "prof_steps20.log"))
    out = ["# rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-extra-legs` -- the driver's own",
           "# invocation (MINIBOONE C=8, N=4096; the 20 timed steps are ONE launch of 20 batches, repeated %s times, value from the median)."
           % d.get("timing", {}).get("repetitions", "?"),
           "# bench.py's line in this profiled run: value %.1f M samples/s, roofline.launch_ms %.4f; unprofiled (profiles/r3_final_bench_line_driver_invocation_steps20.json):"
           % (dp["value"] / 1e6, dp["roofline"]["launch_ms"]),
           "# value %.1f M samples/s, launch_ms %.4f." % (d["value"] / 1e6, d["roofline"]["launch_ms"]),
           "# Name, Calls, AverageNs, MinNs, MaxNs, Percentage"]
    out += stats_rows("prof_steps20")
    open(os.path.join(P, "r3_final_driver_invocation_steps20.txt"), "w").write("\n".join(out) + "\n")
    # 2. emulated per-rank load
    for src, dst in (("bench_emulated_c1_steps20", "bench_line_emulated_8gpu_c1_steps20"), ("bench_emulated_c1", "bench_line_emulated_8gpu_c1")):
        l, dd = last_json(os.path.join(F, src + ".json"))
        open(os.path.join(P, f"r3_final_{dst}.json"), "w").write(l + "\n")
        print(src, "%.1f M samples/s" % (dd["value"] / 1e6), "emulated", dd.get("emulated"))
    # 3. module evaluate loop
    l, dm = last_json(os.path.join(F, "module_eval.json"))
    out = ["# rocprofv3 --kernel-trace --stats of `python3 tools/bench_module_eval.py` (density evaluation THROUGH THE MODULE, batch 1024, C = 8):",
           "# the reference's own loop (one model(x, components=c) call per component + the recursion in torch ops, density_experiment.py:561-573)",
           "# and the one-call form model.log_prob(x).  Unprofiled line: " + l[:400],
           "# Name, Calls, AverageNs, MinNs, MaxNs, Percentage"]
    out += stats_rows("prof_module", only_gbnf=True)
    open(os.path.join(P, "r3_final_module_evaluate_loop.txt"), "w").write("\n".join(out) + "\n")
    # 4. training step
    for n in (4096, 65536):
        l, dt = last_json(os.path.join(F, f"train_n{n}.json"))
        out = [f"# rocprofv3 --kernel-trace --stats of `python3 tools/bench_train.py --batch {n} --cpu-steps 0 --steps 50` (MINIBOONE Glow d=43 h=215 K=5, one component;",
               "# the register-chained training kernels of round 3).  Unprofiled line (profiles/r3_final_train_step_line_n%d.json): %.2f M samples/s, %.4f ms per step"
               % (n, dt["value"] / 1e6, dt["ms_per_step"]),
               "# (forward %.4f ms, backward %.4f ms of kernels).  The at::native / Cijk rows belong to the eager-PyTorch comparison leg of the same script."
               % (dt["forward_kernel_ms"], dt["backward_kernels_ms"]),
               "# Name, Calls, AverageNs, MinNs, MaxNs, Percentage"]
        out += [r for r in stats_rows(f"prof_train{n}") if "gbnf" in r]
        out.append("#")
        out.append("# PMC (separate passes, --steps 20), per dispatch:")
        for f in (f"pmc_train{n}.txt", f"pmc_train_FETCH_SIZE{n}.txt", f"pmc_train_WRITE_SIZE{n}.txt"):
            keep = False
            for line in open(os.path.join(F, f)).read().split("\n"):
                if "dispatches=" in line:
                    keep = "gbnf::" in line and ("bwd_kernel_hx3" in line or "flow_kernel_hx3" in line or "wgrad_kernel" in line)
                    if keep:
                        out.append("#  " + line[:110])
                elif keep and line.startswith("    "):
                    out.append("#      " + line.strip()[:100])
        out.append("# SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* quad-cycles (MI355X_MICROARCH.md); FETCH_SIZE / WRITE_SIZE in KB,")
        out.append("# HBM-side read traffic = 2 x FETCH_SIZE on gfx950.")
        open(os.path.join(P, f"r3_final_train_step_miniboone_n{n}.txt"), "w").write("\n".join(out) + "\n")
        print("train", n, "%.2f M samples/s" % (dt["value"] / 1e6))


if __name__ == "__main__":
    main()
