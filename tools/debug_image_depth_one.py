import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from gbnf_amd import native, synth
dev = torch.device("cuda:0")
size, h, K, L, depth = (1, 28, 20), 256, int(sys.argv[1]), 1, int(sys.argv[2])
sp = synth.synth_image_glow_spec(size, h, K, L, seed=3, depth=depth)
x, noise = synth.synth_image_batch(5, size, seed=4)
flow = native.NativeImageFlow(sp)
for _ in range(3):
    flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
torch.cuda.synchronize()
