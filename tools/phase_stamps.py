#!/usr/bin/env python3
"""Where does a wave of the flow kernel spend its cycles?  (diagnostic, GPU box only)

    bash tools/build_stamps.sh && GBNF_LIB_PATH=$PWD/tools/libgbnf_hip_stamps.so python tools/phase_stamps.py [f32|f16x3]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gbnf_amd import native, synth

NAMES = {
    "f32": ["0 tables+in-prep", "1 layer-0 tile 0", "2 fused pass u=0 (+layer 0)", "3 fused passes u>=1",
            "4 last tile+out", "5 coupling epilogue", "6 final ll/z store", "7 -"],
    "f16x3": ["0 tables+in-prep+split", "1 layer 0 (stages)", "2 -", "3 hidden passes (stages)", "4 drain stage",
              "5 coupling epilogue", "6 final ll/z store", "7 stage-end wait + barrier"],
}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    C_, B = 8, 4096
    specs = synth.synth_boosted_specs("glow", C_, 43, 215, 5, seed=1)
    dev = torch.device("cuda:0")
    os.environ["GBNF_NO_REPAIR"] = "1"
    flows = [native.NativeFlow(s, math=mode) for s in specs]
    mix = native.NativeMixture(flows)
    x = torch.from_numpy(synth.synth_batch(B, 43, seed=0)).to(dev)
    nt = int(os.environ.get("GBNF_FORCE_NT", "1"))      # set GBNF_FORCE_NT=2 for the 32-sample waves of the grouped launches
    nwaves = C_ * ((B + 16 * nt - 1) // (16 * nt))
    buf = torch.zeros((nwaves + 64) * 2 * 8, dtype=torch.int64, device=dev)
    L = native.lib()
    L.gbnf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
    L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(5):
        mix.component_log_prob(x)
    torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(-1, 8)[:nwaves].astype(np.float64)
    tot = st.sum(axis=1)
    print(f"mode {mode}; waves {nwaves}; cycles per wave: median {np.median(tot):.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
    med = np.median(st, axis=0)
    for k, name in enumerate(NAMES[mode]):
        print(f"  {name:30s} {med[k]:10.0f} cycles  {100 * med[k] / med.sum():5.1f} %   per step {med[k] / 5:8.0f}")
    if mode == "f32":
        mf = 5 * 2004 * 32
        print(f"  ideal MFMA cycles per wave {mf}  -> {100 * mf / med.sum():.1f} % of stamped total")
    else:
        n_mfma = 5 * (14 * 6 + 14 * 7 * 6 + 7 * 3 * 6)
        print(f"  f16 MFMAs per wave {n_mfma} x 17.3 cycles = {n_mfma * 17.3:.0f}  -> {100 * n_mfma * 17.3 / med.sum():.1f} % of stamped total")


if __name__ == "__main__":
    main()
