"""Debug aid: run the trainer's backward on one golden case and print gradient errors."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import GoldenCase
from test_hip_train import _dev_spec
from gbnf_amd import native
from oracle import gbnf_oracle as oracle

name = sys.argv[1]
g = GoldenCase(name)
spec = g.specs[0]
dev = torch.device("cuda:0")
tr = native.NativeTrainer(_dev_spec(spec, dev))
x = torch.from_numpy(g.x).to(dev)
z, ldj = tr.forward(x); torch.cuda.synchronize(); print("forward ok", flush=True)
rng = np.random.RandomState(7)
g_z = rng.standard_normal(g.x.shape).astype(np.float32); g_l = rng.standard_normal(g.x.shape[0]).astype(np.float32)
gx, grads = tr.backward(x, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True)
torch.cuda.synchronize(); print("backward ok", flush=True)
gx64, grads64 = oracle.component_grads(spec, g.x, g_z, g_l)
print("gx err", np.abs(gx.cpu().numpy() - gx64).max(), "scale", np.abs(gx64).max())
for k, (a, b) in enumerate(zip(grads, grads64)):
    if b is None: continue
    a = a.cpu().numpy().reshape(b.shape)
    print(k, b.shape, "err", np.abs(a - b).max(), "scale", np.abs(b).max())
