#!/usr/bin/env python3
"""Host-inclusive time of density evaluation THROUGH THE MODULE (what density_experiment.evaluate costs per batch), taken apart:
the reference's own loop (one model(x, components=c) call per component + the base density and the recursion in torch ops,
density_experiment.py:561-573) over a cycle of DIFFERENT batch tensors, the same loop's torch ops alone (the module's outputs
precomputed), its module calls alone, and the one-call form model.log_prob(x).

    python tools/bench_module_eval.py [--batch 4096] [--components 8]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gbnf_amd import BoostedFlow, synth
from test_hip_train import _args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
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
    xs = [torch.from_numpy(synth.synth_batch(a.batch, d, seed=k)).to(dev) for k in range(4)]
    turn = [0]

    def recursion(outs):
        G_ll = None
        for c, (z_G, ldj_G) in enumerate(outs):
            ll = torch.sum(-0.5 * math.log(2 * math.pi) - 0.5 * z_G.pow(2), dim=-1) + ldj_G
            if c == 0:
                G_ll = ll
            else:
                rho_simplex = m.rho[0:(c + 1)] / torch.sum(m.rho[0:(c + 1)])
                last_ll = torch.log(1 - rho_simplex[c]) + G_ll
                next_ll = torch.log(rho_simplex[c]) + ll
                G_ll = torch.logsumexp(torch.cat([last_ll.view(-1, 1), next_ll.view(-1, 1)], dim=1), dim=1)
        return G_ll

    def calls():
        x = xs[turn[0] % len(xs)]
        turn[0] += 1
        outs = []
        for c in range(m.component + 1):
            z, _, _, ldj, _ = m(x=x, components=c)
            outs.append((z, ldj))
        return outs

    fixed = calls()

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.iters

    t_loop = timed(lambda: recursion(calls()))
    t_calls = timed(calls)
    t_ops = timed(lambda: recursion(fixed))
    t_one = timed(lambda: m.log_prob(xs[0]))
    m.SERVE_ALL_COMPONENTS = False
    t_loop_plain = timed(lambda: recursion(calls()))
    print(json.dumps({"metric": "density evaluation through the module, MINIBOONE Glow", "batch": a.batch, "components": C,
                      "reference_loop_ms": t_loop * 1e3, "module_calls_only_ms": t_calls * 1e3, "callers_torch_ops_only_ms": t_ops * 1e3,
                      "log_prob_ms": t_one * 1e3, "reference_loop_one_launch_per_call_ms": t_loop_plain * 1e3,
                      "reference_loop_samples_per_s": a.batch / t_loop, "log_prob_samples_per_s": a.batch / t_one}))


if __name__ == "__main__":
    main()
