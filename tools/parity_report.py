#!/usr/bin/env python3
"""Per-fixture parity report of the HIP path in both math modes (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from conftest import GoldenCase, golden_names, rel_err
from gbnf_amd import native

dev = torch.device("cuda:0")
print(f"{'fixture':38s} {'mode':7s} {'ll rel':>9s} {'G rel':>9s} {'ldj abs':>9s} {'ldj/|ll|':>9s} {'z abs':>9s}")
for name in golden_names():
    g = GoldenCase(name)
    for mode in ("f32", "f16x3", "bf16x6", "default"):
        try:
            flows = native.flows_for_mixture(g.specs, math=mode)
        except native.GbnfError as e:
            print(f"{name:38s} {mode:7s} n/a ({str(e)[:60]})")
            continue
        mix = native.NativeMixture(flows)
        if g.base is not None:
            mix.set_base(*g.base)
        x = torch.from_numpy(g.x).to(dev)
        G, ll = mix.log_prob(x, torch.from_numpy(g.rho).to(dev), n_used=g.n_used)
        e_ldj = e_rel = e_z = 0.0
        for c in range(g.n_used):
            z, ldj, _ = flows[c].forward(x)
            d = np.abs(ldj.cpu().numpy() - g.ldj[c])
            e_ldj = max(e_ldj, d.max())
            e_rel = max(e_rel, (d / np.maximum(np.abs(g.ll[c]), 1.0)).max())
            if g.z(c) is not None:
                e_z = max(e_z, np.abs(z.cpu().numpy() - g.z(c)).max() / max(1.0, np.abs(g.z(c)).max()))
        print(f"{name:38s} {mode:7s} {rel_err(ll.cpu().numpy(), g.ll):9.2e} {rel_err(G.cpu().numpy(), g.G):9.2e} "
              f"{e_ldj:9.2e} {e_rel:9.2e} {e_z:9.2e}")
