#!/usr/bin/env python3
"""A/B timing of candidate builds of the headline kernel on ONE box (tools/build_ab.sh makes the libraries under tools/ablate/).

    python tools/ab_bench.py --names base,mixlo,all3 [--rounds 3] [--check]

Every library runs in its own process (the library path is fixed at import), round-robin over the names so that clock / thermal
drift hits all of them alike.  --check first runs the MINIBOONE-geometry parity tests on each library (results must stay correct:
these are candidate kernels, not ablations).  Prints min / median launch time and samples/s per build, relative to the first name."""
import argparse
import json
import os
import statistics
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--names", required=True)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--group", type=int, default=32)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--extra", default="", help="extra bench.py arguments")
    args = ap.parse_args()
    names = args.names.split(",")
    lib = lambda n: os.path.join(REPO, "tools", "ablate", f"libgbnf_hip_{n}.so")
    if args.check:
        for n in names:
            env = dict(os.environ, GBNF_LIB_PATH=lib(n))
            r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_hip_parity.py", "-k",
                                "g3 or g17 or full_size or inverse or repaired"], cwd=REPO, env=env, stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, text=True)
            print(f"[check] {n}: rc {r.returncode}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ''}", flush=True)
            if r.returncode != 0:
                print(r.stdout[-3000:], flush=True)
    res = {}
    for r in range(args.rounds):
        for n in names:
            env = dict(os.environ, GBNF_LIB_PATH=lib(n))
            out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--cpu-seconds", "0", "--steps", str(args.steps),
                                  "--warmup", "64", "--prewarm", "0.15", "--group", str(args.group), "--no-extra-legs"] + args.extra.split(),
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(n, "FAILED", out.stderr[-600:], flush=True)
                continue
            j = json.loads(line[-1])
            res.setdefault(n, []).append((j["roofline"]["launch_ms"], j["value"], j.get("max_rel_err_vs_cpu")))
    base = statistics.median(v[0] for v in res[names[0]])
    print(f"{'build':24s} {'launch ms min':>14s} {'median':>8s} {'vs ' + names[0]:>10s} {'M samples/s max':>16s} {'median':>8s}")
    for n in names:
        if n not in res:
            continue
        ms = [v[0] for v in res[n]]
        val = [v[1] / 1e6 for v in res[n]]
        print(f"{n:24s} {min(ms):14.4f} {statistics.median(ms):8.4f} {statistics.median(ms) / base:10.3f} {max(val):16.2f} {statistics.median(val):8.2f}")


if __name__ == "__main__":
    main()
