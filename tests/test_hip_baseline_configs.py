"""GPU: the BASELINE.json configurations themselves, at their stated sizes, on the DEFAULT math mode.

configs[1]  MINIBOONE (d=43) Boosted-Glow C=4, batch 4096          -- whole batch against the torch-f32 oracle
configs[3]  CIFAR-10 3x32x32 multi-scale Boosted-Glow C=4, K=8, L=2, h=256 -- a small batch against the float64 oracle
            (0.6 GFLOP per image and component on the CPU), a larger one through the size-independent mixture property
configs[4]  the sharded C=8 pipeline: tests/test_sharded_gpu.py (RCCL, world size 1) -- the 8-GPU hardware run is the driver's

(configs[0] is the reference's CPU toy: fixture g1; configs[2] HEPMASS N=65536: tests/test_hip_parity.py.)
Tolerance: 1e-5 relative on log-likelihoods (north star).
"""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
LL_RTOL = 1e-5


def _lse64(ll, rho):
    w = rho.astype(np.float64) / rho.astype(np.float64).sum()
    a = ll.astype(np.float64) + np.log(w)[:, None]
    m = a.max(axis=0)
    return m + np.log(np.exp(a - m).sum(axis=0))


def test_miniboone_glow_c4_batch_4096_against_oracle():
    """configs[1]: C = 4 gives the block -> (component, tile) map another shape than C = 8 (2 XCDs per component)."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    specs = synth.synth_boosted_specs("glow", 4, 43, 215, 5, seed=1)
    xs = synth.synth_batch(4096, 43, seed=0)
    rho = oracle.rho_init(4)
    ll_ref, G_ref = oracle.mixture_log_prob(specs, rho, xs)
    mix, flows = native.mixture_from_specs(specs)                     # default math mode
    G, ll = mix.log_prob(torch.from_numpy(xs).to(dev), torch.from_numpy(rho).to(dev))
    assert rel_err(ll.cpu().numpy(), ll_ref) < LL_RTOL
    assert rel_err(G.cpu().numpy(), G_ref) < LL_RTOL
    # the group form bench.py times (16 batches in one launch): batch 0 agrees with the single launch (bit-identical
    # when both use the same samples-per-wave tile; here the single launch runs 16-sample tiles, the group 32-sample ones)
    table = torch.empty((4, 16 * 4096), dtype=torch.float32, device=dev)
    xb = [torch.from_numpy(xs).to(dev)] + [torch.from_numpy(synth.synth_batch(4096, 43, seed=100 + s)).to(dev) for s in range(1, 16)]
    mix.prepared_group_log_prob(xb, table)(native._stream_ptr())
    torch.cuda.synchronize()
    assert rel_err(table[:, :4096].cpu().numpy(), ll.cpu().numpy()) < 2e-6
    assert rel_err(table[:, :4096].cpu().numpy(), ll_ref) < LL_RTOL
    Gg = native.mixture_lse(table, torch.from_numpy(rho).to(dev)).cpu().numpy()
    assert rel_err(Gg, _lse64(table.cpu().numpy(), rho)) < 5e-6     # all 16 batches: G == LSE(ll + log w) in float64


def test_cifar_glow_c4_k8_l2_h256():
    """configs[3] at its full depth: K = 8 steps per level, L = 2 levels, h = 256, C = 4 components."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    C, K, L, h = 4, 8, 2, 256
    specs = [synth.synth_image_glow_spec((3, 32, 32), h=h, K=K, L=L, seed=40 + c) for c in range(C)]
    rho = oracle.rho_init(C)
    flows = [native.NativeImageFlow(sp) for sp in specs]
    # (1) a small batch, every component, against the float64 oracle
    n_small = 3
    x, noise = synth.synth_image_batch(n_small, seed=77)
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    lls = []
    for sp, flow in zip(specs, flows):
        z64, _, _, ld64, ll64 = oracle.image_component_forward(sp, x, noise, dtype=torch.float64)
        z, ldj, ll = flow.forward(xd, nd)
        assert rel_err(ll.cpu().numpy(), ll64) < LL_RTOL
        assert rel_err(ldj.cpu().numpy(), ld64) < LL_RTOL
        assert np.abs(z.cpu().numpy() - z64).max() <= 2e-4 * max(1.0, float(np.abs(z64).max()))
        lls.append(ll)
    G = native.mixture_lse(torch.stack(lls), torch.from_numpy(rho).to(dev)).cpu().numpy()
    assert rel_err(G, oracle.mixture_recursion(torch.stack(lls).cpu().numpy(), rho)) < 2e-6
    # (2) the benchmark batch (64 images): batch-size independence (rows 0..2 of the big batch == the small batch) and
    # the mixture property G == LSE(ll + log w)
    xb, nb = synth.synth_image_batch(64, seed=78)
    xb[:n_small], nb[:n_small] = x, noise
    xbd, nbd = torch.from_numpy(xb).to(dev), torch.from_numpy(nb).to(dev)
    big = torch.stack([flow.forward(xbd, nbd, want_z=False)[2] for flow in flows])
    assert torch.isfinite(big).all()
    assert rel_err(big[:, :n_small].cpu().numpy(), torch.stack(lls).cpu().numpy()) < 2e-6
    Gb = native.mixture_lse(big, torch.from_numpy(rho).to(dev)).cpu().numpy()
    assert rel_err(Gb, _lse64(big.cpu().numpy(), rho)) < 5e-6
