#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/...) into the small summaries committed under profiles/.

    python tools/summarize_prof.py <round-tag> <stats_dir> [<pmc_dir> ...]
"""
import collections
import csv
import glob
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def main():
    tag, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    out = []
    ks = find(stats_dir, "_kernel_stats.csv")
    out.append(f"# rocprofv3 --kernel-trace --stats   ({stats_dir})")
    if ks:
        for r in csv.DictReader(open(ks)):
            out.append(f"{r['Name'][:110]:110s} calls={r['Calls']:>5s} avg_ns={float(r['AverageNs']):12.1f} "
                       f"min_ns={r['MinNs']:>9s} max_ns={r['MaxNs']:>9s} pct={r['Percentage']}")
    for d in pmc_dirs:
        cc = find(d, "_counter_collection.csv")
        if not cc:
            continue
        out.append("")
        out.append(f"# rocprofv3 --kernel-trace --pmc ...   ({d})   per-dispatch averages")
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(cc)):
            k = r["Kernel_Name"][:90]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"],
                       r["Accum_VGPR_Count"], r["SGPR_Count"])
        for k, v in agg.items():
            g = meta[k]
            out.append(f"{k}  grid={g[0]} wg={g[1]} lds={g[2]} vgpr={g[3]} agpr={g[4]} sgpr={g[5]}")
            for c, vals in sorted(v.items()):
                out.append(f"    {c:28s} {sum(vals) / len(vals):16.1f}   (n={len(vals)})")
    os.makedirs("profiles", exist_ok=True)
    path = os.path.join("profiles", tag + ".txt")
    open(path, "w").write("\n".join(out) + "\n")
    print("\n".join(out))
    print("->", path)


if __name__ == "__main__":
    main()
