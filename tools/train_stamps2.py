#!/usr/bin/env python3
"""Where does a wave of the training kernels (traced forward, backward chain) spend its cycles?  (diagnostic, GPU box only)

    bash tools/build_train_stamps2.sh && GBNF_LIB_PATH=$PWD/tools/libgbnf_hip_tstamps.so python tools/train_stamps2.py [N]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
torch.cuda.init()

from gbnf_amd import native, synth
from test_hip_train import _dev_spec

FWD = ["0 step boundary (norm, split, saves)", "1 layer 0 (stages)", "2 -", "3 hidden passes (stages)", "4 drain stage",
       "5 coupling epilogue", "6 final z/ldj store", "7 stage-end wait + barrier"]
BWD = ["0 coupling backward (+ its loads)", "1 W3^T stages", "2 W2^T passes", "3 drain, g_in -> LDS", "4 norm backward, sums",
       "5 net start (g_o stores, h2 requests)", "6 -", "7 stage-end wait + barrier"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = 5
    dev = torch.device("cuda:0")
    spec = synth.synth_boosted_specs("glow", 1, 43, 215, K, seed=1)[0]
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    x = torch.from_numpy(synth.synth_batch(n, 43, seed=0)).to(dev)
    L = native.lib()
    L.gbnf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    nwaves = (n + 15) // 16 + 8
    buf = torch.zeros(nwaves * 8, dtype=torch.int64, device=dev)
    z, ldj, trace = tr.forward(x, want_trace=True)
    g_z = (z / n).contiguous(); g_l = torch.full((n,), -1.0 / n, device=dev)
    for which, names in (("forward", FWD), ("backward", BWD)):
        for rep in range(3):
            buf.zero_()
            torch.cuda.synchronize()
            L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
            if which == "forward":
                tr.forward(x, want_trace=True)
            else:
                L.gbnf_debug_set_stamp_buffer(C.c_void_p(0))
                z, ldj, trace = tr.forward(x, want_trace=True)
                torch.cuda.synchronize()
                L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
                tr.backward(x, g_z, g_l, trace=trace)
            torch.cuda.synchronize()
            L.gbnf_debug_set_stamp_buffer(C.c_void_p(0))
        st = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
        st = st[st.sum(axis=1) > 0]
        tot = st.sum(axis=1)
        print(f"{which}: N {n}; stamped waves/blocks {len(st)}; cycles: median {np.median(tot):.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
        med = np.median(st, axis=0)
        for k, name in enumerate(names):
            print(f"  {name:40s} {med[k]:10.0f}  {100 * med[k] / med.sum():5.1f} %   per step {med[k] / K:8.0f}")


if __name__ == "__main__":
    main()
