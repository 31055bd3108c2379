#!/usr/bin/env python3
"""Replay case(s) of tools/stress_eval.py's stream and look at a deviation more closely (GPU, opt-in).

    python tools/debug_eval_case.py <cases> <seed> "<substring of the case's tag>" [...]

For a matching case: the forward error against the float64 oracle per math mode (and of the float32 oracle itself), the worst
row with |z|, ldj and the sizes of the nets' outputs there; the inverse of the LAST component (what stress_eval.py checks):
the kernel's x(z) against the float32 and the float64 oracle's inverse of the SAME z, per row.  A row whose float64 inverse
is as far from x as the kernel's is lost in z already (conditioning of the forward map in float32), not in the way back."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gbnf_amd import native, synth
from oracle import gbnf_oracle as oracle

cases, seed, wants = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
rng = np.random.RandomState(seed)
dev = torch.device("cuda:0")


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), 1.0)


for k in range(cases):
    kind = "glow" if rng.randint(2) else "realnvp"
    d = int(rng.choice([2, 3, 6, 8, 13, 21, 43, 50, 63, 64]))
    h = int(rng.choice([30, 105, 215, 256, 257, 300, 315, 384, 385, 430, 500, 512]))
    K = int(rng.randint(1, 9))
    C = int(rng.randint(1, 5))
    n = int(rng.choice([1, 17, 33, 100, 333, 1000, 4096, 5000]))
    depth = int(rng.choice([0, 1, 1, 1, 2]))
    if kind == "glow":
        kw = dict(depth=depth, act=str(rng.choice(["tanh", "relu", "random"])), coupling=str(rng.choice(["affine", "additive"])),
                  permutation=str(rng.choice(["shuffle", "reverse"])))
    else:
        kw = dict(depth=depth, coupling_network=str(rng.choice(["tanh", "relu", "mixed", "random", "residual"])), batch_norm=bool(rng.randint(2)))
    tag = f"{kind} C={C} d={d} h={h} K={K} n={n} {kw}"
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=300 + k, **kw)
    hit = any(w in tag for w in wants)
    try:                                   # (stress_eval.py draws the data scale only for a case it does not skip)
        mix, _ = native.mixture_from_specs(specs)
        del mix
    except native.GbnfError:
        continue
    scale = float(10.0 ** rng.uniform(-1, 0.5))
    if not hit:
        continue
    x = synth.synth_batch(n, d, seed=k, scale=scale)
    xd = torch.from_numpy(x).to(dev)
    print(f"=== case {k}: {tag}  (data scale {scale:.2f})")
    for c, sp in enumerate(specs):
        z64, l64 = oracle.component_forward(sp, x, backend="numpy64")
        ll64 = (-0.5 * z64 ** 2 - 0.5 * np.log(2 * np.pi)).sum(1) + l64
        z32, l32 = oracle.component_forward(sp, x)
        ll32 = oracle.component_log_prob(sp, x)
        line = f"  component {c}: f32 oracle vs f64: ll {rel(ll32, ll64).max():.1e}"
        for math in ("f32", "bf16x6", "f16x3", "default"):
            try:
                f = native.NativeFlow(sp, math=math)
            except native.GbnfError as e:
                line += f" | {math}: refused"
                continue
            z, ldj, ll = f.forward(xd, want_ll=True)
            e = rel(ll.cpu().numpy(), ll64)
            r = int(e.argmax())
            line += f" | {math} ({native.MATH_NAME[int(f.info().math_mode)]}): {e.max():.1e} at row {r}"
            f.close()
        print(line)
        r = int(rel(ll32, ll64).argmax())
        print(f"      worst f32-oracle row {r}: |z|max {np.abs(z64[r]).max():.3g} ldj {l64[r]:.6g} ll {ll64[r]:.6g}; |x|max {np.abs(x[r]).max():.3g}")
    if h <= 256:
        sp = specs[C - 1]
        f32 = native.NativeFlow(sp, math="f32")
        z, ldj, _ = f32.forward(xd)
        xr, ldj_inv = f32.inverse(z)
        zn = z.cpu().numpy()
        x_o32, _ = oracle.component_inverse(sp, zn, backend="torch")
        x_o64, _ = oracle.component_inverse(sp, zn.astype(np.float64), backend="numpy64")
        z64, l64 = oracle.component_forward(sp, x, backend="numpy64")
        sc = max(1.0, float(np.abs(x).max()))
        ek = np.abs(xr.cpu().numpy() - x).max(1) / sc
        e32 = np.abs(np.asarray(x_o32) - x).max(1) / sc
        e64 = np.abs(np.asarray(x_o64) - x).max(1) / sc
        ez = np.abs(zn - z64).max(1) / np.maximum(1.0, np.abs(z64).max(1))
        order = np.argsort(-ek)[:5]
        print(f"  inverse of component {C - 1} (exact-f32 kernel): worst rows (x error relative to max|x| = {sc:.3g})")
        for r in order:
            print(f"      row {r}: kernel {ek[r]:.2e} | f32 oracle of the same z {e32[r]:.2e} | f64 oracle of the same z {e64[r]:.2e} | "
                  f"that z vs the f64 forward {ez[r]:.1e}, |z|max {np.abs(zn[r]).max():.3g}, ldj {float(ldj[r]):.4g}")
        print(f"      rows with kernel error > 2e-4: {(ek > 2e-4).sum()} of {n}; of those the f64 inverse of the same z is also off (> 1e-4): "
              f"{((ek > 2e-4) & (e64 > 1e-4)).sum()}")
