import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
torch.cuda.init()
from conftest import GoldenCase
from test_hip_train import _dev_spec
from gbnf_amd import native
from oracle import gbnf_oracle as oracle
np.set_printoptions(linewidth=250, precision=3, suppress=True)
dev = torch.device("cuda:0")
name = sys.argv[1]; si = int(sys.argv[2]); n = int(sys.argv[3])
g = GoldenCase(name)
full = g.specs[0]
st = full["steps"][si]
spec = dict(full, steps=[st])
xs = g.x[:n]
rng = np.random.RandomState(7)
tr = native.NativeTrainer(_dev_spec(spec, dev))
x = torch.from_numpy(xs).to(dev)
g_z = rng.standard_normal(xs.shape).astype(np.float32)
g_l = rng.standard_normal(xs.shape[0]).astype(np.float32)
gx64, grads64 = oracle.component_grads(spec, xs, g_z, g_l)
trace = tr.forward(x, want_trace=True)[2]
gx, grads = tr.backward(x, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True, trace=trace)
gx = gx.cpu().numpy().astype(np.float64)
d = spec["d"]; d1 = d // 2
X = xs.astype(np.float64)
lo, hi = X[:, :d1], X[:, d1:]
z1, z2 = (hi, lo) if st["flipped"] else (lo, hi)
nin = z1.shape[1]
def net_fb(net, gout_fn):
    (W1, b1), (W2, b2), (W3, b3) = [(np.asarray(w, np.float64), np.asarray(b, np.float64)) for w, b in net["layers"]]
    act = np.tanh if net["act"] == "tanh" else (lambda v: np.maximum(v, 0))
    a1 = act(z1 @ W1.T + b1); a2 = act(a1 @ W2.T + b2); out = a2 @ W3.T + b3
    return (W1, W2, W3, a1, a2, out)
T = net_fb(st["t_net"], None); S = net_fb(st["s_net"], None)
scale = S[5]
gz1, gz2 = g_z[:, :nin].astype(np.float64), g_z[:, nin:].astype(np.float64)
gsh = gz2; gsc = gz2 * z2 * np.exp(scale) + g_l[:, None]
def dact(net, a): return (1 - a * a) if net["act"] == "tanh" else (a > 0).astype(np.float64)
contrib = []
for nm, netd, (W1, W2, W3, a1, a2, out), go in (("t", st["t_net"], T, gsh), ("s", st["s_net"], S, gsc)):
    ga2 = (go @ W3) * dact(netd, a2)
    ga1 = (ga2 @ W2) * dact(netd, a1)
    h = W1.shape[0]
    for t in range((h + 15) // 16):
        contrib.append((nm, t, ga1[:, 16 * t:16 * t + 16] @ W1[16 * t:16 * t + 16, :]))
tot = gz1 + sum(c for _, _, c in contrib)
ref_in = gx64[:, d1:] if st["flipped"] else gx64[:, :d1]
dev_in = gx[:, d1:] if st["flipped"] else gx[:, :d1]
print("numpy vs oracle", np.abs(tot - ref_in).max())
diff = ref_in - dev_in      # what the device lacks
A = np.stack([c.reshape(-1) for _, _, c in contrib], axis=1)
coef, res, *_ = np.linalg.lstsq(A, diff.reshape(-1), rcond=None)
print("missing-contribution coefficients (net, tile):")
for (nm, t, _), c in zip(contrib, coef): print("  ", nm, t, round(float(c), 4))
print("residual", np.abs(A @ coef - diff.reshape(-1)).max(), "diff max", np.abs(diff).max())
print("diff sample0", diff[0]); print("diff sample1", diff[1])
