"""GPU: randomised shape sweep of the HIP path (both math modes) against the oracle -- exercises the packer's
padding / superset-variant selection (odd and even hidden tile counts, partial k-steps, every output-tile count,
ragged batches, d from 2 to 64) beyond the shapes the reference fixtures pin."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
LL_RTOL = 1e-5


def _cases():
    rng = np.random.RandomState(2024)
    cases = []
    for k in range(36):
        kind = "glow" if k % 3 else "realnvp"
        d = int(rng.choice([2, 3, 5, 6, 8, 13, 21, 32, 43, 50, 63, 64]))
        h = int(rng.choice([7, 16, 30, 33, 48, 64, 100, 105, 112, 129, 160, 200, 215, 240, 256]))
        K = int(rng.randint(1, 6))
        n = int(rng.choice([1, 15, 16, 17, 31, 33, 64, 100, 257, 1000]))
        act = str(rng.choice(["tanh", "relu"]))
        extra = {}
        if kind == "glow":
            extra = dict(coupling=str(rng.choice(["affine", "additive"])), permutation=str(rng.choice(["shuffle", "reverse"])))
        else:
            extra = dict(batch_norm=bool(rng.randint(2)), flip_init=int(rng.randint(2)))
            act = str(rng.choice(["tanh", "relu", "mixed"]))
        if k % 4 == 3:                   # every fourth case: coupling_network_depth 0 or 2
            extra["depth"] = int(rng.choice([0, 2]))
        cases.append((kind, d, h, K, n, act, extra, 500 + k))
    return cases


@pytest.mark.parametrize("kind,d,h,K,n,act,extra,seed", _cases())
def test_random_shape_against_oracle(kind, d, h, K, n, act, extra, seed):
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    if kind == "glow":
        spec = synth.synth_glow_spec(d, h, K, act=act, seed=seed, **extra)
    else:
        spec = synth.synth_realnvp_spec(d, h, K, coupling_network=act, seed=seed, **extra)
    x = synth.synth_batch(n, d, seed=seed + 1)
    zr, lr = oracle.component_forward(spec, x)
    llr = oracle.component_log_prob(spec, x)
    xd = torch.from_numpy(x).to(dev)
    for math in ("f32", "f16x3", "bf16x6"):
        z, ldj, ll = native.NativeFlow(spec, math=math).forward(xd, want_ll=True)
        assert np.isfinite(ll.cpu().numpy()).all()
        assert rel_err(ll.cpu().numpy(), llr) < LL_RTOL, (math, rel_err(ll.cpu().numpy(), llr))
        # ldj is a part of ll: judged on ll's scale
        assert float(np.max(np.abs(ldj.cpu().numpy() - lr) / np.maximum(np.abs(llr), 1.0))) < LL_RTOL, math
        np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=0, atol=2e-5 * max(1.0, np.abs(zr).max()), err_msg=math)


# 256 < h <= 512 beyond depth 1: the reference's CLI reaches these with --h_size_factor on MINIBOONE / BSDS300
# (utils/load_data.py:64-65) together with --coupling_network_depth 0 / 2 or --coupling_network residual
# (models/layers.py:208-301).  Exact-f32: the 32-tile variants (16-sample waves); depth 0 / 2 also on the split kernels.
WIDE_CASES = [
    ("glow", 43, 430, 0, "tanh", dict(coupling="affine", permutation="shuffle")),
    ("glow", 43, 430, 2, "tanh", dict(coupling="affine", permutation="shuffle")),
    ("glow", 63, 315, 2, "relu", dict(coupling="additive", permutation="reverse")),
    ("glow", 13, 300, 0, "random", dict(coupling="affine", permutation="shuffle")),
    ("realnvp", 43, 385, 2, "random", dict(batch_norm=False)),
    ("realnvp", 21, 400, 0, "mixed", dict(batch_norm=True)),
    ("realnvp", 43, 430, 1, "residual", dict(batch_norm=True)),
    ("realnvp", 8, 500, 2, "residual", dict(batch_norm=False)),
]


@pytest.mark.parametrize("kind,d,h,depth,act,extra", WIDE_CASES)
def test_wide_hidden_layers_beyond_depth_one(kind, d, h, depth, act, extra):
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    if kind == "glow":
        spec = synth.synth_glow_spec(d, h, 3, depth=depth, act=act, seed=900 + h, **extra)
    else:
        spec = synth.synth_realnvp_spec(d, h, 3, depth=depth, coupling_network=act, seed=900 + h, **extra)
    x = synth.synth_batch(100, d, seed=h)
    zr, lr = oracle.component_forward(spec, x)
    llr = oracle.component_log_prob(spec, x)
    xd = torch.from_numpy(x).to(dev)
    modes = ["f32"] if act == "residual" else ["f32", "f16x3", "bf16x6", "default"]
    for math in modes:
        z, ldj, ll = native.NativeFlow(spec, math=math).forward(xd, want_ll=True)
        assert rel_err(ll.cpu().numpy(), llr) < LL_RTOL, (math, rel_err(ll.cpu().numpy(), llr))
        assert float(np.max(np.abs(ldj.cpu().numpy() - lr) / np.maximum(np.abs(llr), 1.0))) < LL_RTOL, math
        np.testing.assert_allclose(z.cpu().numpy(), zr, rtol=0, atol=2e-5 * max(1.0, np.abs(zr).max()), err_msg=math)
    # and back: the z -> x direction runs on the exact-f32 variants
    flow = native.NativeFlow(spec, math="f32")
    z = flow.forward(xd)[0]
    xb, ldj_inv = flow.inverse(z)
    np.testing.assert_allclose(xb.cpu().numpy(), x, rtol=0, atol=1e-4 * max(1.0, np.abs(x).max()))
