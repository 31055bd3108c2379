#!/usr/bin/env python3
"""Times the headline workload on every diagnostic library under tools/ablate/ (tools/build_ablate.sh): each one in its
own process (the library path is fixed at import), same GPU, back to back.  Outputs of ablated builds are wrong by
construction; read the launch time only.

    python tools/ablate_bench.py [--group 8] [--rounds 2]
"""
import argparse
import glob
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--group", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--dir", default=os.path.join(REPO, "tools", "ablate"))
    args = ap.parse_args()
    libs = sorted(glob.glob(os.path.join(args.dir, "libgbnf_hip_*.so")))
    res = {}
    for r in range(args.rounds):
        for lib in libs:
            name = os.path.basename(lib)[len("libgbnf_hip_"):-3]
            env = dict(os.environ, GBNF_LIB_PATH=lib)
            out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--cpu-seconds", "0", "--steps",
                                  str(args.steps), "--warmup", "40", "--prewarm", "0.1", "--group", str(args.group),
                                  "--math", "f16x3", "--no-extra-legs"],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(name, "FAILED", out.stderr[-400:], flush=True)
                continue
            j = json.loads(line[-1])
            res.setdefault(name, []).append((j["roofline"]["launch_ms"], j["value"]))
    base = min(v[0] for v in res.get("base", [(float("nan"), 0)]))
    print(f"{'build':44s} {'launch ms (min)':>16s} {'vs base':>8s} {'M samples/s (max)':>18s}")
    for name, v in sorted(res.items(), key=lambda kv: -min(x[0] for x in kv[1])):
        ms = min(x[0] for x in v)
        print(f"{name:44s} {ms:16.4f} {ms / base:8.3f} {max(x[1] for x in v) / 1e6:18.1f}")


if __name__ == "__main__":
    main()
