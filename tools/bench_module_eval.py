#!/usr/bin/env python3
"""Host-inclusive time of density evaluation THROUGH THE MODULE (what density_experiment.evaluate costs per batch):
the reference's own loop (one model(x, components=c) call per component + the recursion in torch ops,
density_experiment.py:561-573) and the one-call form model.log_prob(x).

    python tools/bench_module_eval.py [--batch 1024] [--components 8]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gbnf_amd import BoostedFlow, synth
from test_hip_train import _args
from test_hip_module import _evaluate_like_reference


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--components", type=int, default=8)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    d, h, K, C = 43, 215, 5, a.components
    m = BoostedFlow(_args("glow", d, h, K, C, dev)).to(dev)
    for c, sp in enumerate(synth.synth_boosted_specs("glow", C, d, h, K, seed=1)):
        m.load_spec(c, sp)
    m.component = C - 1
    m.all_trained = True
    m.eval()
    x = torch.randn(a.batch, d, device=dev)

    def timed(fn):
        with torch.no_grad():
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                fn()
            torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.iters

    t_loop = timed(lambda: _evaluate_like_reference(m, x))
    t_one = timed(lambda: m.log_prob(x))
    print(json.dumps({"metric": "density evaluation through the module, MINIBOONE Glow", "batch": a.batch, "components": C,
                      "reference_loop_ms": t_loop * 1e3, "log_prob_ms": t_one * 1e3,
                      "reference_loop_samples_per_s": a.batch / t_loop, "log_prob_samples_per_s": a.batch / t_one}))


if __name__ == "__main__":
    main()
