#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counters per kernel name from the counter_collection CSVs under a directory."""
import csv, glob, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k].add(r["Dispatch_Id"])
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    n = len(cnt[k])
    print(f"{k}  dispatches={n}")
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    for c, v in sorted(d.items()):
        print(f"    {c:32s} {v / n:16.0f} per dispatch   {100 * v / wc / n * n:6.1f} % of wave cycles")
# totals over this library's kernels (all dispatches of the run): the HBM-side traffic of a whole benchmark step is
# (2 x FETCH_SIZE + WRITE_SIZE) KB summed over its kernels, divided by the number of steps the command ran
tot = collections.defaultdict(float)
for k, d in acc.items():
    if "gbnf::" in k:
        for c, v in d.items():
            tot[c] += v
if tot:
    print("TOTAL over gbnf:: kernels (all dispatches): " + "  ".join(f"{c}={v:.0f}" for c, v in sorted(tot.items())))
