#!/usr/bin/env python3
"""gpurun_out/final/ (written by tools/final_measure.sh on the GPU box) -> the committed summaries under profiles/:
bench lines, the rocprofv3 kernel stats of the headline command, the PMC passes, headline_traffic.json.

    python tools/collect_final_profiles.py [round-tag, default r2]
"""
import csv
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F = os.path.join(REPO, "gpurun_out", "final")
P = os.path.join(REPO, "profiles")


def last_json(path):
    lines = [x for x in open(path).read().splitlines() if x.startswith("{")]
    return lines[-1], json.loads(lines[-1])


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
    rows = list(csv.DictReader(open(os.path.join(F, "kernel_stats.csv"))))
    _, prof = last_json(os.path.join(F, "prof_stats.log"))
    _, d = last_json(os.path.join(F, "bench_default.json"))
    group = d["config"]["group"]
    out = [
        "# rocprofv3 --kernel-trace --stats of `python3 bench.py --cpu-seconds 0 --steps 640 --warmup 64 --prewarm 0.05 --no-extra-legs` (MINIBOONE C=8, N=4096,",
        f"# {group} batches per launch, default math = f16x3 chosen by the probe), MI355X, the build shipped at the end of the round (tools/final_measure.sh).",
        "# bench.py's own HIP-event average in this SAME profiled run: launch_ms = %.4f (value %.1f M samples/s); unprofiled run of the same build on the same box:"
        % (prof["roofline"]["launch_ms"], prof["value"] / 1e6),
        "# launch_ms = %.4f (value %.1f M samples/s, profiles/%s_final_bench_line.json) -- the profiler costs a few per cent."
        % (d["roofline"]["launch_ms"], d["value"] / 1e6, tag),
        "# Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs",
    ]
    for r in rows:
        out.append("%-100s calls=%5s avg_ns=%12.1f min_ns=%9s max_ns=%9s pct=%s"
                   % (r["Name"], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
    out += [
        "# <0,14,3,2,0,0,0,4,1>: the f16x3 flow kernel (KIND glow, 14 hidden tiles, 3 output tiles, 32-sample waves, tanh, PREC f16x3, 4-wave workgroups in pairs per CU, depth 1);",
        "# <...,1,0,0,1,8,1>: the bf16x6 repair launch behind every f16x3 launch (returns at once: nothing was marked; min 3.9 us) and the bf16x6 side of the create-time probe;",
        "# <...,1,0,0,0,8,1>: the f16x3 side of the probe and the one-batch warm-up calls.",
        "#",
        f"# PMC, separate passes of the same command with --steps 128 (per dispatch of the {group}-batch flow kernel):",
    ]
    vals = {}
    for f in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        keep = False
        for l in open(os.path.join(F, f + ".txt")).read().split("\n"):
            if l.startswith("void gbnf::flow_kernel_hx3<0, 14, 3, 2"):
                keep = True
                continue
            if keep and l.startswith("    "):
                out.append("#   " + l.strip()[:100])
                k = l.split()
                vals[k[0]] = float(k[1])
            elif keep:
                break
    fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    traffic = fetch * 1024 * 2 + write * 1024
    out.append("# FETCH_SIZE / WRITE_SIZE are in KB; HBM-side traffic per launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE")
    out.append("#   = %.1f MB (x: every component = every XCD reads the batches of the group; packed weights 10.2 MB; the ll table) against %.1f MB algorithmic"
               % (traffic / 1e6, d["roofline"]["hbm_algorithmic_bytes_per_launch"] / 1e6))
    out.append("# VALU instructions per MFMA instruction: (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA = %.2f"
               % ((vals["SQ_INSTS_VALU"] - vals["SQ_INSTS_MFMA"]) / vals["SQ_INSTS_MFMA"]))
    name = f"{tag}_final_miniboone_c8_n4096_group{group}.txt"
    open(os.path.join(P, name), "w").write("\n".join(out) + "\n")
    t = json.load(open(os.path.join(P, "headline_traffic.json")))
    t["workload"]["group"] = group
    t["FETCH_SIZE_kb_per_launch"], t["WRITE_SIZE_kb_per_launch"], t["traffic_bytes_per_launch"] = fetch, write, traffic
    t["source"] = "profiles/" + name
    json.dump(t, open(os.path.join(P, "headline_traffic.json"), "w"), indent=1)
    lines = [("bench_default", "bench_line"), ("bench_steps20", "bench_line_driver_invocation_steps20"),
             ("bench_hepmass", "bench_line_hepmass_realnvp_n65536"), ("bench_c4", "bench_line_miniboone_c4"),
             ("bench_bf16x6", "bench_line_bf16x6"), ("image_n256", "image_cifar_c4_n256_line"), ("image_n64", "image_cifar_c4_n64_line"),
             ("train_n4096", "train_step_line_n4096"), ("train_n65536", "train_step_line_n65536")]
    for src, dst in lines:
        path = os.path.join(F, src + ".json")
        if not os.path.exists(path):
            continue
        l, dd = last_json(path)
        open(os.path.join(P, f"{tag}_final_{dst}.json"), "w").write(l + "\n")
        rf = dd.get("roofline", {})
        print(f"{src:16s} value {dd['value']:.4g}  launch_ms {rf.get('launch_ms')}  executed {rf.get('executed_frac')}  traffic {rf.get('traffic')}")


if __name__ == "__main__":
    main()
