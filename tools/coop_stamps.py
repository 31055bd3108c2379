#!/usr/bin/env python3
"""Where does a wave of the latency-form kernel (flow_kernel_coop) spend its cycles?  (diagnostic, GPU box only)
    bash tools/build_coop_stamps.sh && GBNF_LIB_PATH=$PWD/tools/ablate/libgbnf_hip_coop_stamps.so python tools/coop_stamps.py [n] [form 1|2|3]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gbnf_amd import native, synth
NAMES = ["0 prologue (tables, x tile, first fragments)", "1 net input + layer 0 + ACT stores", "2 barrier 1", "3 hidden layer", "4 output layer + partials",
         "5 barrier 2", "6 epilogue + barrier 3", "7 tail (ll / z stores)"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
C_, d = 8, 43
specs = synth.synth_boosted_specs("glow", C_, d, 215, 5, seed=1)
dev = torch.device("cuda:0")
flows = [native.NativeFlow(s, math="f16x3") for s in specs]
mix = native.NativeMixture(flows)
native.tuning_set("repair", 0); native.tuning_set("coop", mode)
x = torch.from_numpy(synth.synth_batch(n, d, seed=0)).to(dev)
rows = 16 if mode in (1, 4) else 32
W = 8 if mode >= 3 else 4
nwg = C_ * ((n + rows - 1) // rows)
buf = torch.zeros((nwg * W + 64) * 8, dtype=torch.int64, device=dev)
L = native.lib()
L.gbnf_debug_set_stamp_buffer.argtypes = [C.c_void_p]
L.gbnf_debug_set_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(5):
    mix.component_log_prob(x)
torch.cuda.synchronize()
st = buf.cpu().numpy().reshape(-1, 8)[: nwg * W].astype(np.float64).reshape(nwg, W, 8)
tot = st.sum(axis=2)
print(f"n = {n}, form {mode}: {rows}-sample tiles on {W} waves: {nwg} workgroups; shader cycles per wave: median {np.median(tot):.0f} min {tot.min():.0f} max {tot.max():.0f}")
for w in range(W):
    med = np.median(st[:, w, :], axis=0)
    print(f" wave {w}: " + " | ".join(f"{k}:{med[k]:6.0f}" for k in range(8)) + f" | total {med.sum():.0f}")
med = np.median(st.reshape(-1, 8), axis=0)
for k, name in enumerate(NAMES):
    print(f"  {name:48s} {med[k]:8.0f}  {100 * med[k] / med.sum():5.1f} %   per step {med[k] / 5:7.0f}")
