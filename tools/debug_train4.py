import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_hip_train import _dev_spec
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle
dev = torch.device("cuda:0")
for (d, h, K, bn, flip) in ((6, 30, 1, False, 0), (6, 30, 1, False, 1), (21, 105, 1, False, 0), (21, 105, 2, True, 0)):
    spec = synth.synth_realnvp_spec(d, h, K, batch_norm=bn, flip_init=flip, seed=3)
    x = synth.synth_batch(16, d, seed=1)
    tr = native.NativeTrainer(_dev_spec(spec, dev))
    z, ldj = tr.forward(torch.from_numpy(x).to(dev))
    z64, l64 = oracle.component_forward(spec, x, backend="numpy64")
    print(d, h, K, bn, flip, "z err per feature", np.round(np.abs(z.cpu().numpy() - z64).max(0), 5), "ldj err", np.abs(ldj.cpu().numpy() - l64).max())
