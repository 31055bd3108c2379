import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_hip_train import _dev_spec
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
dev = torch.device("cuda:0")
for depth, coupling in ((0, "additive"), (0, "affine"), (1, "additive")):
    spec = synth.synth_glow_spec(6, 30, 1, depth=depth, coupling=coupling, seed=3)
    x = synth.synth_batch(16, 6, seed=1)
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    xd = torch.from_numpy(x).to(dev)
    rng = np.random.RandomState(0)
    g_z = rng.standard_normal(x.shape).astype(np.float32); g_l = np.zeros(16, np.float32)
    gx, grads = tr.backward(xd, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True)
    gx64, grads64 = oracle.component_grads(spec, x, g_z, g_l)
    print("depth", depth, coupling, "gx err", np.abs(gx.cpu().numpy() - gx64).max())
    for k, (a, b) in enumerate(zip(grads, grads64)):
        print("   ", k, b.shape, "err", np.abs(a.cpu().numpy().reshape(b.shape) - b).max(), "scale", np.abs(b).max())
