#!/usr/bin/env python3
"""gpurun_out/final/ (written by tools/final_measure.sh on the GPU box) -> the committed summaries under profiles/:
bench lines of every configuration, rocprofv3 kernel stats, the PMC passes, and the traffic tables the bench lines read
(profiles/headline_traffic.json, image_traffic.json, train_traffic.json).

    python tools/collect_final_profiles.py [round-tag, default r6]
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


def pmc(name):
    """{kernel name: {"dispatches": n, counter: per-dispatch value}} of one pmc_summary.py output."""
    out, cur = {}, None
    path = os.path.join(F, name + ".txt")
    if not os.path.exists(path):
        return out
    for l in open(path).read().split("\n"):
        if l.startswith("TOTAL"):
            out["__total__"] = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in l.split("dispatches):", 1)[1].split()}
        elif l and not l.startswith(" "):
            nm, _, rest = l.rpartition("  dispatches=")
            cur = out.setdefault(nm, {"dispatches": int(rest)})
        elif l.startswith("    ") and cur is not None:
            k = l.split()
            cur[k[0]] = float(k[1])
    return out


def dominant(table, counter, prefix="void gbnf::flow_kernel_hx3<"):
    best = None
    for nm, d in table.items():
        if nm.startswith(prefix) and counter in d and (best is None or d[counter] > best[1][counter]):
            best = (nm, d)
    return best


def stats_rows(name, top=14):
    path = os.path.join(F, name + ".kernel_stats.csv")
    if not os.path.exists(path):
        return []
    rows = list(csv.DictReader(open(path)))
    return ["%-100s calls=%5s avg_ns=%12.1f min_ns=%9s max_ns=%9s pct=%s" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["MinNs"],
                                                                           r["MaxNs"], r["Percentage"]) for r in rows[:top]]


def traffic_entry(fetch_name, write_name, workload, how):
    f, w = dominant(pmc(fetch_name), "FETCH_SIZE"), dominant(pmc(write_name), "WRITE_SIZE")
    if not f or not w:
        return None
    fetch, write = f[1]["FETCH_SIZE"], w[1]["WRITE_SIZE"]
    return {"workload": workload, "kernel": f[0], "FETCH_SIZE_kb_per_launch": fetch, "WRITE_SIZE_kb_per_launch": write,
            "fetch_correction": 2.0, "traffic_bytes_per_launch": fetch * 1024 * 2 + write * 1024, "how": how}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
    _, d = last_json(os.path.join(F, "bench_default.json"))
    group = d["config"]["group"]
    # ---- headline: kernel stats + PMC
    out = [f"# rocprofv3 --kernel-trace --stats of `python3 bench.py --cpu-seconds 0 --steps 640 --warmup 64 --prewarm 0.05 --no-extra-legs` (MINIBOONE",
           f"# Boosted-Glow C = 8, N = 4096, {group} batches per launch, default math = f16x3 chosen by the probe), MI355X, the build shipped at the end of the round",
           "# (tools/final_measure.sh); unprofiled run of the same build on the same box: launch_ms = %.4f, %.1f M samples/s (profiles/%s_final_bench_line.json)."
           % (d["roofline"]["launch_ms"], d["value"] / 1e6, tag)]
    try:
        _, prof = last_json(os.path.join(F, "prof_stats.log"))
        out.append("# bench.py's own HIP-event average in this SAME profiled run: launch_ms = %.4f (%.1f M samples/s)" % (prof["roofline"]["launch_ms"], prof["value"] / 1e6))
    except Exception:
        pass
    out += stats_rows("prof_stats")
    out.append(f"# PMC, separate passes of the same command with --steps 128 (per dispatch of the {group}-batch flow kernel):")
    vals = {}
    for f in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2"):
        t = pmc(f)
        best = dominant(t, next((c for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVE_CYCLES") if any(c in v for v in t.values())), "SQ_WAVE_CYCLES"))
        if best:
            out.append(f"#  {best[0]}  dispatches={best[1]['dispatches']}")
            for k, v in best[1].items():
                if k != "dispatches":
                    out.append("#      %-32s %16.0f per dispatch" % (k, v))
                    vals[k] = v
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        traffic = vals["FETCH_SIZE"] * 2048 + vals["WRITE_SIZE"] * 1024
        out.append("# HBM-side traffic per launch = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE (KB) = %.1f MB against %.1f MB algorithmic"
                   % (traffic / 1e6, d["roofline"]["hbm_algorithmic_bytes_per_launch"] / 1e6))
    if "SQ_INSTS_VALU" in vals and "SQ_INSTS_MFMA" in vals:
        out.append("# vector instructions per MFMA: (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA = %.2f" % ((vals["SQ_INSTS_VALU"] - vals["SQ_INSTS_MFMA"]) / vals["SQ_INSTS_MFMA"]))
    name = f"{tag}_final_miniboone_c8_n4096_group{group}.txt"
    open(os.path.join(P, name), "w").write("\n".join(out) + "\n")
    # ---- traffic tables
    how = "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE (per-dispatch averages of the dominant flow kernel); FETCH_SIZE doubled as MI355X_MICROARCH.md 'HBM' prescribes for gfx950"
    head = traffic_entry("pmc_fetch", "pmc_write", {"config": "miniboone_glow", "batch": 4096, "components": 8, "group": group, "math": "f16x3", "n_gpus": 1},
                         how + "; `python3 bench.py --cpu-seconds 0 --steps 128 --no-extra-legs`")
    others = [traffic_entry("pmc_fetch_s20", "pmc_write_s20", {"config": "miniboone_glow", "batch": 4096, "components": 8, "group": 20, "math": "f16x3", "n_gpus": 1},
                            how + "; the driver's invocation `bench.py --gpus 1 --steps 20 --warmup 5` (one launch of 20 batches)"),
              traffic_entry("pmc_fetch_c4", "pmc_write_c4", {"config": "miniboone_glow", "batch": 4096, "components": 4, "group": 32, "math": "f16x3", "n_gpus": 1},
                            how + "; `bench.py --components 4` (BASELINE configs[1])"),
              traffic_entry("pmc_fetch_hm", "pmc_write_hm", {"config": "hepmass_realnvp", "batch": 65536, "components": 8, "group": 32, "math": "f16x3", "n_gpus": 1},
                            how + "; `bench.py --config hepmass_realnvp --batch 65536` (BASELINE configs[2])")]
    if head:
        head["source"] = "profiles/" + name
        head["taken"] = __import__("datetime").date.today().isoformat() + f" (round {tag}, tools/final_measure.sh)"
        head["other_group_sizes"] = [o for o in others if o]
        json.dump(head, open(os.path.join(P, "headline_traffic.json"), "w"), indent=1)
    # image: all gbnf:: kernels of the run / steps
    fi, wi = pmc("pmc_fetch_img").get("__total__"), pmc("pmc_write_img").get("__total__")
    if fi and wi:
        steps = 12.0
        rec = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE of tools/bench_image.py (--steps 10 --warmup 2 --no-graph), "
                      "summed over every gbnf:: kernel and divided by the 12 steps; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950",
               "workloads": [{"workload": {"batch": 256, "components": 4, "K": 8, "L": 2, "hidden": 256, "math": "default"},
                              "FETCH_SIZE_kb_per_step": fi["FETCH_SIZE"] / steps, "WRITE_SIZE_kb_per_step": wi["WRITE_SIZE"] / steps,
                              "traffic_bytes_per_step": (fi["FETCH_SIZE"] * 2048 + wi["WRITE_SIZE"] * 1024) / steps}]}
        json.dump(rec, open(os.path.join(P, "image_traffic.json"), "w"), indent=1)
        img = [f"# rocprofv3 --kernel-trace --stats of `python3 tools/bench_image.py --batch 256 --cpu-seconds 0 --steps 120 --warmup 5 --no-graph` (CIFAR-shaped Boosted-Glow,",
               "# C = 4 components on 4 HIP streams, K = 8, L = 2, h = 256), MI355X, end of the round; round 6: 120 + 5 steps, the one-time on-data checks of",
               "# img_repair_kernel are amortised as in a long run; then the PMC passes (--steps 10 --warmup 2):"]
        img += stats_rows("stats_img")
        img.append("# all gbnf:: kernels of the 12 steps: FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB => HBM-side traffic per step (2 x FETCH + WRITE) = %.1f MB = %.2f MB per image and component"
                   % (fi["FETCH_SIZE"], wi["WRITE_SIZE"], rec["workloads"][0]["traffic_bytes_per_step"] / 1e6, rec["workloads"][0]["traffic_bytes_per_step"] / 1e6 / 1024))
        for f in ("pmc_fetch_img", "pmc_write_img", "pmc_sq_img"):
            for nm, v in pmc(f).items():
                if nm.startswith("void gbnf::img_net_hx3") or nm.startswith("void gbnf::img_conv_kernel<1"):
                    img.append(f"#  {nm}  dispatches={v['dispatches']}")
                    img += ["#      %-32s %16.0f per dispatch" % (k, x) for k, x in v.items() if k != "dispatches"]
        open(os.path.join(P, f"{tag}_final_image_cifar_c4_n256.txt"), "w").write("\n".join(img) + "\n")
    ft, wt = pmc("pmc_fetch_train").get("__total__"), pmc("pmc_write_train").get("__total__")
    if ft and wt:
        steps = 23.0 + 20.0       # timed + warm-up steps of the step loop, plus the 20 event-bracketed repetitions of forward / backward
        rec = {"how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bench_train.py --batch 65536 --steps 20 --warmup 3, summed over every gbnf:: kernel "
                      "and divided by the 43 forward + backward sweeps the command runs; FETCH_SIZE doubled (gfx950)",
               "workloads": [{"workload": {"config": "miniboone_glow", "batch": 65536}, "FETCH_SIZE_kb_per_step": ft["FETCH_SIZE"] / steps,
                              "WRITE_SIZE_kb_per_step": wt["WRITE_SIZE"] / steps,
                              "traffic_bytes_per_step": (ft["FETCH_SIZE"] * 2048 + wt["WRITE_SIZE"] * 1024) / steps}]}
        json.dump(rec, open(os.path.join(P, "train_traffic.json"), "w"), indent=1)
    for nm, title in (("stats_hm", "configs[2] HEPMASS Boosted-RealNVP C = 8, N = 65536 (bench.py --config hepmass_realnvp --batch 65536 --steps 64)"),
                      ("stats_c4", "configs[1] MINIBOONE Boosted-Glow C = 4, N = 4096 (bench.py --components 4 --steps 128)"),
                      ("stats_train", "MINIBOONE Glow one component, N = 65536 (tools/bench_train.py --batch 65536)"),
                      ("stats_train_bs", "HEPMASS RealNVP one component, N = 65536, BatchNorm on batch statistics (tools/bench_train.py --config hepmass_realnvp --batch-stats)")):
        rows = stats_rows(nm, 12)
        if rows:
            open(os.path.join(P, f"{tag}_final_{nm}.txt"), "w").write(f"# rocprofv3 --kernel-trace --stats: {title}\n" + "\n".join(rows) + "\n")
    # ---- the latency form: kernel stats of whole log_prob calls + SQ / traffic counters of the bare launch at 512 rows
    lat = ["# rocprofv3 --kernel-trace --stats of `python3 tools/latency_one.py <n> -1 300 call`: 300 model.log_prob(x) calls (flow launch + its repair",
           "# launch + recursion launch, default math, the shipped launch policy) of the MINIBOONE Boosted-Glow C = 8 at the reference's batch sizes"]
    for n_ in (512, 1024):
        rows = stats_rows(f"stats_latency_n{n_}", 6)
        if rows:
            lat += [f"# n = {n_}:"] + rows
    for f in ("pmc_sq1_latency", "pmc_sq2_latency", "pmc_fetch_latency", "pmc_write_latency"):
        for nm, v in pmc(f).items():
            if nm.startswith("void gbnf::flow_kernel_coop"):
                lat.append(f"# {f}: {nm}  dispatches={v['dispatches']}")
                lat += ["#      %-32s %16.0f per dispatch" % (k, x) for k, x in v.items() if k != "dispatches"]
    if len(lat) > 2:
        txt = os.path.join(F, "latency.txt")
        if os.path.exists(txt):
            lat += ["# tools/bench_latency.py (one log_prob call, us) and tools/latency_ablate.py (bare flow launch by kernel form, us):"] + \
                   ["# " + l for l in open(txt).read().splitlines() if l.startswith(("n =", "shipped"))]
        open(os.path.join(P, f"{tag}_final_latency_form.txt"), "w").write("\n".join(lat) + "\n")
    cg = os.path.join(F, "coop_geometries.txt")
    if os.path.exists(cg):
        open(os.path.join(P, f"{tag}_final_coop_geometries.txt"), "w").write(
            "# tools/bench_coop_geometries.py: GPU time (us) of one flow launch per kernel form, geometry and batch size (see profiles/r6_coop_geometries.txt)\n" + open(cg).read())
    inv = os.path.join(F, "image_inverse.txt")
    if os.path.exists(inv):
        keep = [l for l in open(inv).read().splitlines() if l.startswith("one component")]
        open(os.path.join(P, f"{tag}_final_image_inverse.txt"), "w").write(
            "# tools/bench_image_inverse.py: z -> x (sampling) and x -> z of ONE image Glow component (K = 8, L = 2, h = 256), plain stream launches\n" + "\n".join(keep) + "\n")
    lines = [("bench_default", "bench_line"), ("bench_steps20", "bench_line_driver_invocation_steps20"),
             ("bench_hepmass", "bench_line_hepmass_realnvp_n65536"), ("bench_c4", "bench_line_miniboone_c4"), ("bench_bf16x6", "bench_line_bf16x6"),
             ("bench_emul8_steps20", "bench_line_emulated_8gpu_c1_steps20"), ("bench_emul8_default", "bench_line_emulated_8gpu_c1"),
             ("bench_emul8_default_torch", "bench_line_emulated_8gpu_c1_torch_pipeline"),
             ("image_n256", "image_cifar_c4_n256_line"), ("image_n64", "image_cifar_c4_n64_line"),
             ("image_1x28x28_n256", "image_1x28x28_c4_n256_line"), ("image_1x28x20_n256", "image_1x28x20_c4_n256_line"),
             ("image_h512_n256", "image_cifar_c4_h512_n256_line"), ("image_h384_n256", "image_cifar_c4_h384_n256_line"),
             ("image_depth0_n256", "image_cifar_c4_depth0_n256_line"), ("image_depth2_n256", "image_cifar_c4_depth2_n256_line"), ("train_n4096", "train_step_line_n4096"),
             ("train_n65536", "train_step_line_n65536"), ("train_hepmass_bs_n65536", "train_step_line_hepmass_batchstats_n65536"),
             ("train_hepmass_n65536", "train_step_line_hepmass_n65536"), ("module_eval", "module_evaluate_loop_line"),
             ("train_glow_depth0_n65536", "train_step_line_glow_depth0_n65536"), ("train_glow_depth2_n65536", "train_step_line_glow_depth2_n65536"),
             ("train_hepmass_depth2_n65536", "train_step_line_hepmass_depth2_n65536"),
             ("train_hepmass_residual_n65536", "train_step_line_hepmass_residual_n65536"),
             ("boosted_step_n512", "boosted_step_line_n512"), ("bench_full_record_last", "bench_full_record")]
    for src, dst in lines:
        path = os.path.join(F, src + ".json")
        if not os.path.exists(path):
            continue
        if src == "bench_full_record_last":       # the default run's FULL record (pretty-printed): every leg in full, notes, thread probes
            open(os.path.join(P, f"{tag}_final_{dst}.json"), "w").write(open(path).read())
            continue
        try:
            l, dd = last_json(path)
        except Exception as e:
            print(src, "no line:", e)
            continue
        open(os.path.join(P, f"{tag}_final_{dst}.json"), "w").write(l + "\n")
        rf = dd.get("roofline", {}) or {}
        print(f"{src:26s} value {dd.get('value', float('nan')):.4g}  launch_ms {rf.get('launch_ms')}  executed {rf.get('executed_frac')}  traffic {rf.get('traffic')}")


if __name__ == "__main__":
    main()
