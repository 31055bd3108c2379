import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_hip_train import _dev_spec
from gbnf_amd import native, synth
dev = torch.device("cuda:0")
spec = synth.synth_glow_spec(6, 30, 1, depth=0, coupling="additive", seed=3)
x = synth.synth_batch(16, 6, seed=1)
tr = native.NativeTrainer(_dev_spec(spec, dev))
xd = torch.from_numpy(x).to(dev)
rng = np.random.RandomState(0)
g_z = rng.standard_normal(x.shape).astype(np.float32); g_l = np.zeros(16, np.float32)
gx, grads = tr.backward(xd, torch.from_numpy(g_z).to(dev), torch.from_numpy(g_l).to(dev), want_gx=True)
torch.cuda.synchronize()
ws = tr._ws.cpu().numpy()
np_ = 16; ip = 16; hp = 32; op = 16; nh = 1
X = ws[0:ip * np_].reshape(ip, np_); H0 = ws[ip * np_:(ip + hp) * np_].reshape(hp, np_)
D0 = ws[(ip + nh * hp) * np_:(ip + nh * hp + hp) * np_].reshape(hp, np_)
GO = ws[(ip + 2 * nh * hp) * np_:(ip + 2 * nh * hp + op) * np_].reshape(op, np_)
W1 = spec["steps"][0]["net"]["layers"][1][0]       # (3, 30)
exp = (W1.T @ GO[:3]) * (1 - H0[:30] ** 2)
print("D0 err", np.abs(D0[:30] - exp).max(), "scale", np.abs(exp).max())
print("D0 dev [0:4, 0:4]\n", D0[:4, :4], "\nexp\n", exp[:4, :4])
print("ratio", (D0[:6, :3] / exp[:6, :3]))
lin = W1.T @ GO[:3]
print("vs no-dact", np.abs(D0[:30] - lin).max())
dd = D0[:30] / lin[:30]
hh = 1 - H0[:32] ** 2
for r in (2, 3, 6, 7):
    best = np.argmin([np.abs(dd[r] - hh[q]).max() for q in range(32)])
    print("row", r, "dact matches H0 row", best, "err", np.abs(dd[r] - hh[best]).max())
