#!/usr/bin/env python3
"""Phase shares inside gbnf::img_net_hx3_kernel (the fused coupling net, round 4) from a -DGBNF_IMG_STAMPS=<W> build:
    tools/build_image_stamps.sh 16 && python tools/image_stamps2.py 256 16"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("GBNF_LIB_PATH", os.path.join(ROOT, "tools", "libgbnf_image_stamps.so"))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sp = synth.synth_image_glow_spec((3, 32, 32), 256, 1, 2 if W == 8 else 1, seed=1)
flow = native.NativeImageFlow(sp)
x, noise = synth.synth_image_batch(n, seed=0)
x, noise = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
wgs = n * (2 if W == 16 else 1)
buf = torch.zeros(wgs * 8 * 8, dtype=torch.int64, device=dev)
native.lib().gbnf_debug_set_image_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(3):
    flow.forward(x, noise)
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(wgs, 8, 8).astype(np.float64)
names = ["stage z1 (+ barrier)", "first 3x3 -> relu -> split -> LDS", "barrier waits", "1x1 MFMA stream", "1x1 relu + split + LDS stores",
         "last 3x3 MFMA stream", "reduction + coupling epilogue", "(diagnostic bucket 7: -DGBNF_IMG_STAMP_A = wait for the first 3x3's A fragments)"]
tot = a.sum(2).mean()
print(f"W = {W}, {wgs} workgroups x 8 waves: {tot:.0f} shader cycles per wave on average (max wave {a.sum(2).max():.0f})")
for k, nm in enumerate(names):
    print(f"  {nm:40s} {a[:, :, k].mean():9.0f}  {100 * a[:, :, k].mean() / tot:5.1f} %   (wave 0: {a[:, 0, k].mean():8.0f}, wave 7: {a[:, 7, k].mean():8.0f})")
