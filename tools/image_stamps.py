#!/usr/bin/env python3
"""Phase shares inside gbnf::img_mid_hx3_kernel<4> (level-1 maps) from a -DGBNF_IMG_STAMPS build (tools/build_image_stamps.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["GBNF_LIB_PATH"] = os.path.join(ROOT, "tools", "libgbnf_image_stamps.so")
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
sp = synth.synth_image_glow_spec((3, 32, 32), 256, 1, 1, seed=1)
flow = native.NativeImageFlow(sp)
x, noise = synth.synth_image_batch(n, seed=0)
x, noise = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
buf = torch.zeros(n * 4 * 4 * 6, dtype=torch.int64, device=dev)
native.lib().gbnf_debug_set_image_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(3):
    flow.forward(x, noise)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 6).astype(np.float64)
a = a[a.sum(1) > 0]
names = ["stage z1 (+ barrier)", "first 3x3 (f32 MFMA) + split -> LDS (+ barrier)", "1x1 (f16x3 MFMA stream)", "relu + split + stores"]
tot = a.sum(1).mean()
print(f"workgroups {a.shape[0]}, wave 0: {tot:.0f} shader cycles per workgroup")
for k, nm in enumerate(names):
    print(f"  {nm:50s} {a[:, k].mean():9.0f}  {100 * a[:, k].mean() / tot:5.1f} %")
