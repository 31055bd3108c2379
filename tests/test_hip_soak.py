"""Repeated-launch soak of every compiled split-kernel (hx3) variant.

Round 2 met a fault that returned whole 16-sample tiles off by 5e-2 on a DIFFERENT set of waves per launch (an MFMA
register hazard the compiler's padding does not cover, csrc/gbnf_flow_kernel_hx3.hip.h `mfma_tail_guard`,
profiles/r3_mfma_srcc_war_ubench.txt).  A per-launch-random fault needs many launches to show, so for EVERY `hx3`
line of csrc/variants.list, both split precisions, 16- and 32-sample waves, the 8-wave and the paired 4-wave workgroup
forms, this runs GBNF_SOAK_LAUNCHES (default 200) launches at N = 77 and N = 4096 and requires

  * bit-identical outputs across all launches, and
  * <= 1e-5 relative against the exact-f32 kernel on the same inputs (the parity bar of BASELINE.json).

The same for the group launch (several batches per launch), for the z -> x direction of the exact-f32 kernel (its
epilogue carries a hand-padded hazard: DESIGN 4.6) and for the training kernels (branch drains: DESIGN 4.7).
`GBNF_SOAK_LAUNCHES=2000 pytest tests/test_hip_soak.py` is the 10^5-launch form.
"""
import os

import numpy as np
import pytest

from conftest import REPO, rel_err

pytestmark = pytest.mark.gpu

LAUNCHES = int(os.environ.get("GBNF_SOAK_LAUNCHES", "200"))


def hx3_variants():
    out = []
    with open(os.path.join(REPO, "gradient-boosted-normalizing-flows_amd", "csrc", "variants.list")) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if line.startswith("hx3"):
                v = [int(t) for t in line.split()[1:] if t.lstrip('-').isdigit()]       # (a trailing `eval`: no training sweeps, csrc/build.py)
                out.append(tuple(v[:5]) + ((v[5],) if len(v) > 5 else (1,)))
    return sorted(set(out))


def spec_for(kind, ht, ot, acta, actb, depth, seed=3):
    """A component whose exact geometry is the variant's: hidden width 16 HT - 1, d such that the coupled half needs OT tiles."""
    from gbnf_amd import synth
    h = 16 * ht - 1
    if kind == 0:
        d = 16 * ot - 1
        act = {(0, 0): "tanh", (1, 1): "relu", (3, 3): "random"}[(acta, actb)]
        return synth.synth_glow_spec(d, h, 2, depth=depth, act=act, seed=seed), d, acta == 3
    d = 21 if ot == 1 else 63
    net = {(0, 0): "tanh", (1, 1): "relu", (1, 0): "mixed", (3, 3): "random", (2, 2): "residual"}[(acta, actb)]
    if net == "residual":         # the variant's depth counts hidden -> hidden layers: two per ResidualNet block
        depth //= 2
    return synth.synth_realnvp_spec(d, h, 2, depth=depth, coupling_network=net, flip_init=1, seed=seed), d, acta == 3


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def policy():
    """Restores the library's launch policy after a test that forces it."""
    from gbnf_amd import native
    keys = ("force_nt", "wg_pairs", "repair", "check_every")
    saved = {k: native.tuning_get(k) for k in keys}
    yield native
    for k, v in saved.items():
        native.tuning_set(k, v)


def _soak_flow(flow, x, launches):
    """`launches` launches of flow.forward -> (ll of launch 0, ldj of launch 0); asserts bit-identical repeats."""
    import torch
    lls, ldjs = [], []
    for _ in range(launches):
        _, ldj, ll = flow.forward(x, want_z=False, want_ldj=True, want_ll=True)
        lls.append(ll)
        ldjs.append(ldj)
    L = torch.stack(lls)
    D = torch.stack(ldjs)
    same = bool((L == L[0]).all().item()) and bool((D == D[0]).all().item())
    if not same:
        bad = int(((L != L[0]).any(dim=1) | (D != D[0]).any(dim=1)).sum().item())
        worst = float((L - L[0]).abs().max().item())
        raise AssertionError(f"{bad} of {launches} launches differ from launch 0 (max |d ll| {worst:.3e})")
    return L[0], D[0]


@pytest.mark.parametrize("variant", hx3_variants(), ids=lambda v: "hx3_" + "_".join(str(a) for a in v))
def test_every_split_variant_is_launch_stable(variant, dev, policy):
    import torch
    from gbnf_amd import native, synth
    kind, ht, ot, acta, actb, depth = variant
    spec, d, per_step = spec_for(*variant)
    ref = native.NativeFlow(spec, math="f32", per_step_activation=per_step)
    xs = {n: torch.from_numpy(synth.synth_batch(n, d, seed=11 + n)).to(dev) for n in (77, 4096)}
    want = {}
    for n, x in xs.items():
        _, ldj, ll = ref.forward(x, want_z=False, want_ldj=True, want_ll=True)
        want[n] = (ll.cpu().numpy(), ldj.cpu().numpy())
    for math in ("f16x3", "bf16x6"):
        flow = native.NativeFlow(spec, math=math, per_step_activation=per_step)
        info = flow.info()
        assert (info.hidden_tiles, info.out_tiles) == (ht, ot), "the soak must run the variant it names"
        for nt in (1, 2):
            for pairs in (0, 1):
                native.tuning_set("force_nt", nt)
                native.tuning_set("wg_pairs", pairs)
                for n, x in xs.items():
                    ll, ldj = _soak_flow(flow, x, LAUNCHES)
                    e_ll, e_ldj = rel_err(ll.cpu().numpy(), want[n][0]), rel_err(ldj.cpu().numpy(), want[n][1])
                    assert e_ll < 1e-5 and e_ldj < 1e-5, (math, nt, pairs, n, e_ll, e_ldj)
        flow.close()
    ref.close()


@pytest.mark.parametrize("variant", [(0, 8, 4, 0, 0, 1), (0, 14, 3, 0, 0, 1), (1, 7, 1, 0, 0, 1)],
                         ids=lambda v: "hx3_" + "_".join(str(a) for a in v))
def test_group_launch_is_launch_stable(variant, dev, policy):
    """The grouped form (several batches x several components per launch: what bench.py and GroupPipeline run)."""
    import torch
    from gbnf_amd import native, synth
    kind, ht, ot, acta, actb, depth = variant
    specs = [spec_for(*variant, seed=3 + c)[0] for c in range(2)]
    d = specs[0]["d"]
    n = 1024
    xs = [torch.from_numpy(synth.synth_batch(n, d, seed=50 + b)).to(dev) for b in range(3)]
    ref_mix = native.NativeMixture([native.NativeFlow(s, math="f32") for s in specs])
    want = torch.cat([ref_mix.component_log_prob(x) for x in xs], dim=1).cpu().numpy()
    for math in ("f16x3", "bf16x6"):
        mix = native.NativeMixture([native.NativeFlow(s, math=math) for s in specs])
        for nt in (1, 2):
            for pairs in (0, 1):
                native.tuning_set("force_nt", nt)
                native.tuning_set("wg_pairs", pairs)
                tables = [torch.empty((2, 3 * n), device=dev) for _ in range(LAUNCHES)]
                for t in tables:
                    mix.prepared_group_log_prob(xs, t)(native._stream_ptr())
                T = torch.stack(tables)
                assert bool((T == T[0]).all().item()), (math, nt, pairs)
                assert rel_err(T[0].cpu().numpy(), want) < 1e-5, (math, nt, pairs)


@pytest.mark.parametrize("math", ["f32", "f16x3", "bf16x6"])
@pytest.mark.parametrize("kind,d,h", [("glow", 43, 215), ("glow", 63, 128), ("realnvp", 21, 105)])
def test_inverse_is_launch_stable(kind, d, h, math, dev, policy):
    """z -> x (exact-f32 kernel and, since round 3, the split kernels; 16- and 32-sample waves): bit-identical repeats, round trip."""
    import torch
    from gbnf_amd import native, synth
    spec = synth.synth_boosted_specs(kind, 1, d, h, 3, seed=9)[0]
    flow = native.NativeFlow(spec, math=math)
    for nt in (1, 2):
        native.tuning_set("force_nt", nt)
        for n in (77, 2048):
            x = torch.from_numpy(synth.synth_batch(n, d, seed=n)).to(dev)
            z, ldj, _ = flow.forward(x)
            outs = [flow.inverse(z) for _ in range(LAUNCHES)]
            X = torch.stack([o[0] for o in outs])
            D = torch.stack([o[1] for o in outs])
            assert bool((X == X[0]).all().item()) and bool((D == D[0]).all().item()), (nt, n)
            assert float((X[0] - x).abs().max().item()) < 2e-4 * max(1.0, float(x.abs().max().item()))
            assert rel_err((-D[0]).cpu().numpy(), ldj.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("kind,d,h", [("glow", 43, 215), ("realnvp", 21, 105)])
def test_training_kernels_are_launch_stable(kind, d, h, dev):
    """forward (z, ldj, trace) and backward g_x of the training kernels: bit-identical across launches (the parameter
    gradients are accumulated with float atomics and are compared with a tolerance instead)."""
    import torch
    from gbnf_amd import native, synth
    spec = synth.synth_boosted_specs(kind, 1, d, h, 3, seed=4)[0]
    from test_hip_train import _dev_spec as to_device_spec
    dspec = to_device_spec(spec, dev)
    tr = native.NativeTrainer(dspec)
    n_launch = max(20, LAUNCHES // 4)
    try:
        for n, force_nt in ((77, 0), (2048, 0), (77, 2), (2048, 2)):     # 2: the 32-sample-wave form of the traced forward (large batches)
            native.tuning_set("force_nt", force_nt)
            x = torch.from_numpy(synth.synth_batch(n, d, seed=n + 1)).to(dev)
            g_z = torch.from_numpy(synth.synth_batch(n, d, seed=n + 2)).to(dev)
            g_ldj = torch.ones(n, device=dev)
            z0 = ldj0 = gx0 = grads0 = None
            for it in range(n_launch):
                z, ldj, trace = tr.forward(x, want_trace=True)
                g_x, grads = tr.backward(x, g_z, g_ldj, want_gx=True, trace=trace)
                flat = torch.cat([g.reshape(-1) for g in grads if g is not None])
                if it == 0:
                    z0, ldj0, gx0, grads0 = z, ldj, g_x, flat
                else:
                    assert torch.equal(z, z0) and torch.equal(ldj, ldj0) and torch.equal(g_x, gx0), (n, it)
                    scale = float(grads0.abs().max().item())
                    assert float((flat - grads0).abs().max().item()) <= 2e-5 * scale, (n, it)
    finally:
        native.tuning_set("force_nt", 0)
