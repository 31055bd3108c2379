#!/usr/bin/env python3
"""Host-side profile (cProfile) of the reference-style evaluate loop through the drop-in module: where do the ~150 us per
model(x, components=c) call go?  GPU box."""
import cProfile, pstats, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gbnf_amd import BoostedFlow, synth
from test_hip_train import _args

dev = torch.device("cuda:0")
d, h, K, C = 43, 215, 5, 8
m = BoostedFlow(_args("glow", d, h, K, C, dev)).to(dev)
for c, sp in enumerate(synth.synth_boosted_specs("glow", C, d, h, K, seed=1)):
    m.load_spec(c, sp)
m.component = C - 1
m.all_trained = True
m.eval()
x = torch.from_numpy(synth.synth_batch(1024, d, seed=0)).to(dev)


def loop(n):
    with torch.no_grad():
        for _ in range(n):
            for c in range(C):
                z, _, _, ldj, _ = m(x=x, components=c)


loop(20)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
loop(200)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
