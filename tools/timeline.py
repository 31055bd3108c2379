#!/usr/bin/env python3
"""Per-stage timeline of workgroup 0 of the f16x3 flow kernel (diagnostic, GPU box only).

    bash tools/build_timeline.sh && GBNF_NO_HX32=1 GBNF_FORCE_NT=2 GBNF_LIB_PATH=$PWD/tools/libgbnf_hip_timeline.so python tools/timeline.py
For every stage: when each of the 8 waves reached the stage-end wait (cycles since the workgroup's first stamp) and when
it left the barrier.  Waves w and w+4 share a SIMD (or whatever the dispatcher did: the pairing shows in the numbers)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gbnf_amd import native, synth


def main():
    os.environ["GBNF_NO_REPAIR"] = "1"
    C_, B, S = 8, 4096, 16
    specs = synth.synth_boosted_specs("glow", C_, 43, 215, 5, seed=1)
    dev = torch.device("cuda:0")
    flows = [native.NativeFlow(s, math="f16x3") for s in specs]
    mix = native.NativeMixture(flows)
    xs = [torch.from_numpy(synth.synth_batch(B, 43, seed=k)).to(dev) for k in range(S)]
    table = torch.empty((C_, S * B), dtype=torch.float32, device=dev)
    launch = mix.prepared_group_log_prob(xs, table)
    buf = torch.zeros(8 * 128 * 2, dtype=torch.int64, device=dev)
    L = native.lib()
    L.gbnf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(5):
        launch(native._stream_ptr())
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(8, 128, 2).astype(np.int64)
    n_st = int((t[0, :, 0] > 0).sum())
    t0 = t[:, :n_st, :].min()
    rel = t[:, :n_st, :] - t0
    print(f"stages recorded: {n_st}; total cycles of workgroup 0: {rel.max()}")
    print("stage | per wave: arrive(at wait) / leave(barrier)  [cycles since start]")
    for s in range(min(n_st, 40)):
        arr = rel[:, s, 0]
        lv = rel[:, s, 1]
        print(f"{s:3d} | arrive " + " ".join(f"{v:7d}" for v in arr) + f" | leave {lv.min():7d}..{lv.max():7d} | spread of arrivals {arr.max() - arr.min():5d}")
    # per stage: duration between consecutive barrier leaves
    lv = rel[:, :n_st, 1].max(axis=0)
    dur = np.diff(lv)
    print("stage durations (leave-to-leave), first 40:", " ".join(str(int(v)) for v in dur[:40]))
    arr_spread = (rel[:, :n_st, 0].max(axis=0) - rel[:, :n_st, 0].min(axis=0))
    print(f"mean stage duration {dur.mean():.0f}; mean arrival spread {arr_spread.mean():.0f}; mean (leave - last arrival) {(rel[:, :n_st, 1].max(axis=0) - rel[:, :n_st, 0].max(axis=0)).mean():.0f}")


if __name__ == "__main__":
    main()
