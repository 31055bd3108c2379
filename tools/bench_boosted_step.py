#!/usr/bin/env python3
"""End-to-end boosted training step through the drop-in module, as density_experiment.train runs it
(density_experiment.py:366-374 + compute_kl_pq_loss :606-660): sample weights from the fixed mixture, resample,
forward of the component being trained, nll, backward, Adam -- wall time per step including every host-side cost,
next to the sum of the GPU kernel times and to the same step written with eager PyTorch modules on the same GPU.

    python tools/bench_boosted_step.py [--batch 512] [--steps 200] [--components 4]
"""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import BoostedFlow
from test_hip_train import _args


def lns(z):
    return torch.sum(-0.5 * math.log(2 * math.pi) - 0.5 * z.pow(2), dim=-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--components", type=int, default=4)
    ap.add_argument("--reference-style", action="store_true",
                    help="the reference's own statement sequence (one model(x, components=c) call per fixed component + the "
                         "recursion and the weights in torch ops, density_experiment.py:612-644) instead of boosting_weights()")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    d, h, K, C = 43, 215, 5, a.components
    m = BoostedFlow(_args("glow", d, h, K, C, dev)).to(dev)
    x = torch.randn(a.batch, d, device=dev)
    m.train()
    with torch.no_grad():
        for c in range(C):                       # ActNorm data-dependent init of every component
            m.component = c
            m(x=x, components="c")
    m.component = C - 1                          # train the last component against the C-1 fixed ones
    opt = torch.optim.Adam(m.flows[C - 1].parameters(), lr=1e-3)

    for name, p_ in m.named_parameters():                                   # init_boosted_lr, density_experiment.py:537-538
        p_.requires_grad = name.startswith(f"flows.{m.component}")
    lo, hi = torch.tensor([0.01], device=dev), torch.tensor([0.1], device=dev)

    def reference_weights():
        G_ll = None
        for c in range(m.component):
            z_G, _, _, ldj_G, _ = m(x=x, components=c)
            ll = lns(z_G) + ldj_G
            if c == 0:
                G_ll = ll
            else:
                rs = m.rho[0:(c + 1)] / torch.sum(m.rho[0:(c + 1)])
                G_ll = torch.logsumexp(torch.cat([(torch.log(1 - rs[c]) + G_ll).view(-1, 1),
                                                  (torch.log(rs[c]) + ll).view(-1, 1)], dim=1), dim=1)
        w = torch.softmax(-1.0 * G_ll, dim=0)
        if w.max() > 0.1:
            w = torch.max(torch.min(w, hi), lo)
        return w / torch.sum(w)

    def step():
        opt.zero_grad(set_to_none=False)
        if a.reference_style:
            w = reference_weights()
        else:
            with torch.no_grad():
                w, _ = m.boosting_weights(x)                                # softmax(-G) clamped, :624-640
        xr = x[torch.multinomial(w, x.size(0), replacement=True)]           # :642-644
        z, _, _, ldj, _ = m(x=xr, components="c")
        loss = torch.mean(-(lns(z) + ldj))
        loss.backward()
        opt.step()
        return loss

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.steps
    # GPU time of the same steps (events around the loop body exclude nothing but host gaps)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(a.steps):
        step()
    ev[1].record(); torch.cuda.synchronize()
    gpu = ev[0].elapsed_time(ev[1]) / a.steps * 1e-3
    prof_note = "GPU-side time between events around the loop (the stream is never empty if host < GPU)"
    # the library's own share of a step: its three calls in tight loops of their own (host launch cost + GPU time, synchronised at the end):
    # the sample weights from the fixed components (one mixture launch + the weights kernel), the recorded forward of the component in
    # training (live re-pack + TRAIN sweep) and its backward (gradient scaling, backward sweep, weight gradients)
    def loop_ms(fn, iters=200):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3
    tr = m.native_trainer(C - 1)
    g_z = torch.randn_like(x)
    g_l = torch.randn(a.batch, device=dev)
    with torch.no_grad():
        lib_w = loop_ms(lambda: m.boosting_weights(x))
        keep = {}

        def fwd():
            keep["t"] = tr.forward(x, want_trace=True)
        lib_f = loop_ms(fwd)
        lib_b = loop_ms(lambda: tr.backward(x, g_z, g_l, want_gx=False, trace=keep["t"][2]))
    print(json.dumps({"metric": "boosted training step (weights + resample + forward + backward + Adam), MINIBOONE Glow",
                      "value": a.batch / wall, "unit": "samples/s", "ms_per_step": wall * 1e3,
                      "batch": a.batch, "components": C, "reference_style": a.reference_style, "ms_per_step_wall": wall * 1e3, "ms_per_step_gpu_events": gpu * 1e3,
                      "library_ms_per_step": lib_w + lib_f + lib_b, "library_weights_ms": lib_w, "library_forward_ms": lib_f, "library_backward_ms": lib_b,
                      "samples_per_s": a.batch / wall, "data": "synthetic", "dtype": "f16x3",
                      "config": {"workload": f"one boosted training step of density_experiment.py:606-674 at the reference's batch {a.batch} "
                                             f"(--batch_size, density_experiment.py:80): MINIBOONE Glow d=43 h=215 K=5, component {C} of {C} in training"},
                      "note": prof_note + "; library_*: the library's three calls of a step in tight loops of their own (host launch cost + GPU time)"}))


if __name__ == "__main__":
    main()
