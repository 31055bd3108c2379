import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
dev = torch.device("cuda:0")
for (d, h, N) in [(21, 64, 32), (21, 96, 32), (21, 100, 32), (21, 112, 32), (43, 215, 32), (43, 215, 16)]:
    spec = synth.synth_glow_spec(d, h, 1, seed=3)
    x = synth.synth_batch(N, d, seed=1)
    try:
        f = native.NativeFlow(spec)
    except Exception as e:
        print(d, h, "create failed", e); continue
    z, ldj, ll = f.forward(torch.from_numpy(x).to(dev), want_ll=True)
    zr, lr = oracle.component_forward(spec, x)
    z = z.cpu().numpy(); ldj = ldj.cpu().numpy()
    err = np.abs(z - zr)
    print(f"d={d} h={h} N={N}: max|dz|={err.max():.3e} max|dldj|={np.abs(ldj-lr).max():.3e}  bad cols={np.where(err.max(axis=0)>1e-4)[0].tolist()} bad rows={np.where(err.max(axis=1)>1e-4)[0].tolist()[:8]}")
