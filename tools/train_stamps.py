#!/usr/bin/env python3
"""Phase shares inside gbnf::train_kernel (forward mode), from a -DGBNF_TRAIN_STAMPS build (tools/build_train_stamps.sh).
Read the SHARES, not the absolute time (the stamps serialise the wave)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GBNF_LIB_PATH", os.path.join(ROOT, "tools", "libgbnf_train_stamps.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import native, synth
from test_hip_train import _dev_spec
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
spec = synth.synth_boosted_specs("glow", 1, 43, 215, 5, seed=1)[0]
tr = native.NativeTrainer(_dev_spec(spec, dev))
x = torch.from_numpy(synth.synth_batch(n, 43, seed=0)).to(dev)
buf = torch.zeros((n + 15) // 16 * 8 + 64, dtype=torch.int64, device=dev)
native.lib().gbnf_debug_set_train_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(3):
    tr.forward(x)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
a = a[a.sum(1) > 0]          # workgroups that ran (two tiles per workgroup halve their number)
names = ["setup (tables, x)", "norm + net input", "-", "coupling + barrier", "outputs", "dense: setup + first loads issued",
         "dense: MFMA stream", "dense: barrier wait"]
tot = a.sum(1).mean()
print(f"workgroups {a.shape[0]}, mean stamped time per workgroup (wave 0) {tot:.0f} shader cycles")
for k, nm in enumerate(names):
    print(f"  {nm:32s} {a[:, k].mean():10.0f} ticks  {100 * a[:, k].mean() / tot:5.1f} %")
