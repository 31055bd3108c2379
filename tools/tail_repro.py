#!/usr/bin/env python3
"""Launch-stability soak of ONE split-kernel geometry (glow HT=8 OT=4: the round-2 offender g5_glow_d63_h128) on the
library GBNF_LIB_PATH names (tools/build_tail_experiments.sh):  per (NT, workgroup form, N) the number of launches whose
log-likelihoods differ from launch 0 and the error of launch 0 against the exact-f32 kernel.

    GBNF_LIB_PATH=tools/ablate/libgbnf_hip_tail0.so python tools/tail_repro.py [--launches 400] [--steps 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=400)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--d", type=int, default=63)
    ap.add_argument("--h", type=int, default=128)
    args = ap.parse_args()
    import torch
    from gbnf_amd import native, synth
    dev = torch.device("cuda:0")
    spec = synth.synth_glow_spec(args.d, args.h, args.steps, seed=3)
    ref = native.NativeFlow(spec, math="f32")
    flow = native.NativeFlow(spec, math="f16x3")
    print("library:", native.LIB_PATH, " variant tiles:", flow.info().hidden_tiles, flow.info().out_tiles)
    total_bad = 0
    for n in (77, 1024, 4096):
        x = torch.from_numpy(synth.synth_batch(n, args.d, seed=11 + n)).to(dev)
        want = ref.forward(x, want_z=False, want_ldj=False, want_ll=True)[2].cpu().numpy()
        for nt in (1, 2):
            for pairs in (0, 1):
                native.tuning_set("force_nt", nt)
                native.tuning_set("wg_pairs", pairs)
                lls = torch.stack([flow.forward(x, want_z=False, want_ldj=False, want_ll=True)[2] for _ in range(args.launches)])
                got = lls.cpu().numpy()
                err = np.abs(got - want[None]) / np.maximum(np.abs(want[None]), 1.0)
                bad_launches = int((err.max(axis=1) > 1e-5).sum())
                differ = int((got != got[0:1]).any(axis=1).sum())
                total_bad += bad_launches
                print(f"  N={n:5d} NT={nt} wg_pairs={pairs}: {bad_launches:4d} of {args.launches} launches beyond 1e-5 "
                      f"(worst {err.max():.2e}), {differ} differ from launch 0", flush=True)
    print("TOTAL bad launches:", total_bad)


if __name__ == "__main__":
    main()
