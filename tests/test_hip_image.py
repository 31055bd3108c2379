"""GPU: the image path (gbnf_image_flow_*, SURVEY.md section 8a a14 / BASELINE.json configs[3]) against the reference's own
image Glow outputs (fixtures g12_*) and the oracle.  Tolerance: 1e-5 relative on the log-likelihood (north star)."""
import numpy as np
import pytest

from conftest import IMAGE_CASES, load_image_case, rel_err

pytestmark = pytest.mark.gpu
LL_RTOL = 1e-5


@pytest.mark.parametrize("name", IMAGE_CASES)
def test_image_flow_matches_reference(name):
    import torch
    from gbnf_amd import native
    cfg, specs, x, noise, data = load_image_case(name)
    dev = torch.device("cuda:0")
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    lls = []
    for c, sp in enumerate(specs):
        flow = native.NativeImageFlow(sp)
        z, ldj, ll = flow.forward(xd, nd)
        assert rel_err(ldj.cpu().numpy(), data["ldj"][c]) < LL_RTOL
        assert rel_err(ll.cpu().numpy(), data["ll"][c]) < LL_RTOL
        zr = data["z"][c]
        assert np.abs(z.cpu().numpy() - zr).max() <= 2e-4 * max(1.0, float(np.abs(zr).max()))
        lls.append(ll)
    G = native.mixture_lse(torch.stack(lls), torch.from_numpy(data["rho"]).to(dev))
    assert rel_err(G.cpu().numpy(), data["G"]) < LL_RTOL


def test_image_flow_full_width_matches_oracle():
    """BASELINE.json configs[3] geometry: 3x32x32, L = 2, h = 256 (K reduced to 2 to keep the CPU oracle quick)."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    sp = synth.synth_image_glow_spec((3, 32, 32), h=256, K=2, L=2, seed=5)
    x, noise = synth.synth_image_batch(6, seed=9)
    z64, _, _, ld64, ll64 = oracle.image_component_forward(sp, x, noise, dtype=torch.float64)
    dev = torch.device("cuda:0")
    flow = native.NativeImageFlow(sp)
    z, ldj, ll = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    assert rel_err(ll.cpu().numpy(), ll64) < LL_RTOL
    assert rel_err(ldj.cpu().numpy(), ld64) < LL_RTOL
    assert np.abs(z.cpu().numpy() - z64).max() <= 2e-4 * max(1.0, float(np.abs(z64).max()))
    # no noise, odd batch, no z
    _, ldj1, ll1 = flow.forward(torch.from_numpy(x[:1]).to(dev), None, want_z=False)
    _, _, _, _, ll1_ref = oracle.image_component_forward(sp, x[:1], np.zeros_like(x[:1]), dtype=torch.float64)
    assert rel_err(ll1.cpu().numpy(), ll1_ref) < LL_RTOL
    mu, lv = flow.prior()
    assert mu.shape == (24,) and lv.shape == (24,)


def test_image_flow_argument_validation():
    import torch
    from gbnf_amd import native, synth
    sp = synth.synth_image_glow_spec((3, 32, 32), h=16, K=1, L=2, seed=1)
    flow = native.NativeImageFlow(sp)
    with pytest.raises(native.GbnfError):
        flow.forward(torch.zeros(2, 3, 32, 32))                    # CPU tensor: no CPU path
    with pytest.raises(native.GbnfError):
        flow.forward(torch.zeros(2, 3, 16, 16, device="cuda"))
    bad = synth.synth_image_glow_spec((3, 64, 64), h=16, K=1, L=1, seed=1)
    with pytest.raises(native.GbnfError):                          # 32-wide maps are not compiled
        native.NativeImageFlow(bad)
    assert native.lib().gbnf_image_flow_forward(None, None, None, 1, None, None, None, None, 0, None) == -1


def test_image_module_dropin_matches_reference():
    """BoostedFlow(args) with image input_size: model(x=x, components=c) and log_prob against the reference's outputs
    (fixture g12, noise injected) -- the drop-in module end to end."""
    import argparse
    import torch
    from gbnf_amd import BoostedFlow, image_glow
    cfg, specs, x, noise, data = load_image_case("g12_image_glow_invconv_affine")
    dev = torch.device("cuda:0")
    args = argparse.Namespace(
        num_flows=cfg["K"], z_size=3072, density_evaluation=True, device=dev, cuda=True, component_type="glow",
        num_components=cfg["C"], rho_init="decreasing", learn_top=cfg["learn_top"], y_classes=0, y_condition=False,
        sample_size=4, input_size=[3, 32, 32], h_size=cfg["h"], num_blocks=cfg["L"], actnorm_scale=1.0,
        flow_permutation=cfg["permutation"], flow_coupling=cfg["coupling"], LU_decomposed=False, num_dequant_blocks=0,
        coupling_network="tanh", coupling_network_depth=cfg["depth"], batch_norm=False)
    m = BoostedFlow(args)
    assert isinstance(m, image_glow.BoostedImageFlow)
    for c, sp in enumerate(specs):
        image_glow.load_image_spec(m.flows[c], sp)
    m.eval()
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    with torch.no_grad():
        ll = m.component_log_prob(xd, noise=nd)
        assert rel_err(ll.cpu().numpy().T, data["ll"]) < LL_RTOL
        assert rel_err(m.log_prob(xd, noise=nd).cpu().numpy(), data["G"]) < LL_RTOL
        # the same call captured once in a HIP graph and replayed: identical results; a changed parameter re-captures
        f = m.graphed_log_prob(xd.shape[0])
        G_eager = m.log_prob(xd, noise=nd)
        assert torch.equal(f(xd, nd), G_eager) and torch.equal(f(xd, nd), G_eager)
        next(m.flows[0].parameters()).add_(1e-3)                    # (in place: the version counter moves, the handle is rebuilt)
        G_new = m.log_prob(xd, noise=nd)
        assert not torch.equal(G_new, G_eager)
        assert torch.equal(f(xd, nd), G_new)
        z, z_mu, z_var, ldj, y = m(x=xd, components=1)              # fresh noise: shapes and the constant prior only
        assert y is None and z.shape == (cfg["N"], 24, 8, 8) and z_mu.shape == z.shape and z_var.shape == z.shape
        assert torch.isfinite(ldj).all()


def _random_image_cases():
    rng = np.random.RandomState(11)
    cases = []
    for k in range(10):
        cases.append(dict(h=int(rng.choice([8, 16, 24, 32, 48, 100])), K=int(rng.randint(1, 3)), L=int(rng.choice([1, 2])),
                          depth=int(rng.choice([0, 1, 1, 2])), coupling=str(rng.choice(["affine", "additive"])),
                          permutation=str(rng.choice(["invconv", "shuffle", "reverse"])), learn_top=bool(rng.randint(2)),
                          n=int(rng.choice([1, 3, 5])), seed=300 + k))
    return cases


@pytest.mark.parametrize("case", _random_image_cases(), ids=lambda c: f"h{c['h']}K{c['K']}L{c['L']}d{c['depth']}{c['coupling'][:3]}{c['permutation'][:3]}")
def test_image_random_geometry_against_oracle(case):
    """Hidden widths that are not multiples of 32 (padded split-f16 layout), depth 0 / 2 (exact-f32 kernels), one level
    (16x16 maps only), every permutation kind, both couplings -- against the float64 oracle."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    c = dict(case)
    n, seed = c.pop("n"), c.pop("seed")
    sp = synth.synth_image_glow_spec((3, 32, 32), seed=seed, **c)
    x, noise = synth.synth_image_batch(n, seed=seed + 1)
    z64, _, _, ld64, ll64 = oracle.image_component_forward(sp, x, noise, dtype=torch.float64)
    dev = torch.device("cuda:0")
    z, ldj, ll = native.NativeImageFlow(sp).forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    assert rel_err(ll.cpu().numpy(), ll64) < LL_RTOL
    assert rel_err(ldj.cpu().numpy(), ld64) < LL_RTOL
    assert np.abs(z.cpu().numpy() - z64).max() <= 2e-4 * max(1.0, float(np.abs(z64).max()))


# ---- the z -> x direction (gbnf_image_flow_inverse) -------------------------------------------
from conftest import IMAGE_DECODE_CASES, load_image_decode_case  # noqa: E402

X_ATOL = 5e-5        # x is a pixel intensity in [0, 1] (sigmoid of the logits): absolute tolerance


@pytest.mark.parametrize("name", IMAGE_DECODE_CASES)
def test_image_inverse_matches_reference_decode(name):
    """g16: the reference's Glow.decode(z, None, temperature) with Split2d's draws injected."""
    import torch
    from gbnf_amd import native
    cfg, spec, z, eps, x_ref = load_image_decode_case(name)
    dev = torch.device("cuda:0")
    flow = native.NativeImageFlow(spec)
    assert flow.split_shapes() == [tuple(e.shape[1:]) for e in eps]
    x = flow.inverse(torch.from_numpy(z).to(dev), [torch.from_numpy(e).to(dev) for e in eps], cfg["temperature"])
    torch.cuda.synchronize()
    assert np.abs(x.cpu().numpy() - x_ref).max() <= X_ATOL, float(np.abs(x.cpu().numpy() - x_ref).max())


def test_image_inverse_round_trip_at_config_size():
    """BASELINE.json configs[3] geometry (3x32x32, K = 8, L = 2, h = 256), 64 images: decode, then encode the decoded image
    (dequantisation noise chosen so that (255 x + noise) / 256 == x): z comes back.  Small batch also against the oracle."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    sp = synth.synth_image_glow_spec((3, 32, 32), h=256, K=8, L=2, seed=44)
    flow = native.NativeImageFlow(sp)
    rng = np.random.RandomState(3)
    n = 64
    z = (0.7 * rng.randn(n, *flow.z_shape)).astype(np.float32)
    eps = [rng.randn(n, *sh).astype(np.float32) for sh in flow.split_shapes()]
    zd, ed = torch.from_numpy(z).to(dev), [torch.from_numpy(e).to(dev) for e in eps]
    x = flow.inverse(zd, ed, 0.8)
    assert torch.isfinite(x).all()
    x_or = oracle.image_component_inverse(sp, z[:2], [e[:2] for e in eps], 0.8, dtype=torch.float64)
    assert np.abs(x[:2].cpu().numpy() - x_or).max() <= X_ATOL
    inside = (x > 0.02) & (x < 0.98)                 # pixels the logit of the forward direction can take back
    xs = torch.where(inside, x, torch.full_like(x, 0.5))
    z_back = flow.forward(xs, xs.clone())[0]
    z_ref = torch.from_numpy(oracle.image_component_forward(sp, xs[:2].cpu().numpy(), xs[:2].cpu().numpy(), dtype=torch.float64)[0])
    assert np.abs(z_back[:2].cpu().numpy() - z_ref.numpy()).max() <= 2e-4 * max(1.0, float(z_ref.abs().max()))
    ok = inside.flatten(1).all(dim=1)                 # images decoded entirely inside the range: z must come back
    if ok.any():
        err = (z_back[ok] - zd[ok]).abs().max().item()
        assert err <= 2e-3 * max(1.0, float(zd.abs().max())), err


def test_image_module_decode_and_sampling():
    """BoostedFlow(args) for images: model(z=z, components=c, reverse=True) == Glow.decode of the reference (g16), and
    z = None draws sample_size images from the top prior."""
    import argparse
    import torch
    from gbnf_amd import BoostedFlow, image_glow
    cfg, spec, z, eps, x_ref = load_image_decode_case("g16_image_decode_invconv_affine")
    dev = torch.device("cuda:0")
    args = argparse.Namespace(
        num_flows=cfg["K"], z_size=3072, density_evaluation=True, device=dev, cuda=True, component_type="glow",
        num_components=2, rho_init="decreasing", learn_top=True, y_classes=0, y_condition=False,
        sample_size=5, input_size=[3, 32, 32], h_size=cfg["h"], num_blocks=cfg["L"], actnorm_scale=1.0,
        flow_permutation=cfg["permutation"], flow_coupling=cfg["coupling"], LU_decomposed=False, num_dequant_blocks=0,
        coupling_network="tanh", coupling_network_depth=cfg["depth"], batch_norm=False)
    m = BoostedFlow(args)
    image_glow.load_image_spec(m.flows[1], spec)
    m.eval()
    with torch.no_grad():
        x = m.decode(torch.from_numpy(z).to(dev), None, cfg["temperature"], 1, eps=[torch.from_numpy(e).to(dev) for e in eps])
        assert np.abs(x.cpu().numpy() - x_ref).max() <= X_ATOL
        xs = m(z=None, temperature=0.7, components=1, reverse=True)
        assert xs.shape == (5, 3, 32, 32) and torch.isfinite(xs).all()


def _blow_up_hidden(sp, step=0, gain_logs=12.0):
    """Make the coupling net of one FlowStep leave the fp16 range: its first convolution's ActNorm2d scales every hidden
    channel by e^gain_logs (hidden activations of 1e5 .. 1e7 instead of O(1)); the last convolution's weights shrink by the
    same factor so that shift / scale -- and the exact result -- stay ordinary."""
    import copy
    sp = copy.deepcopy(sp)
    net = sp["levels"][0]["steps"][step]["convs"]
    net[0]["an_logs"] = net[0]["an_logs"] + np.float32(gain_logs)
    net[-1]["w"] = (net[-1]["w"] * np.float32(np.exp(-gain_logs))).astype(np.float32)
    return sp


def test_image_out_of_range_model_is_caught_by_the_probe():
    """a model whose hidden activations leave +-65504 must not reach the caller with clamped (wrong) values.
    When the MODEL is the cause the create-time probe meets it on its own images and the handle runs on the exact-f32 kernels."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    sp = _blow_up_hidden(synth.synth_image_glow_spec((3, 32, 32), h=64, K=2, L=2, seed=11))
    x, noise = synth.synth_image_batch(6, seed=3)
    _, _, _, ld32, ll32 = oracle.image_component_forward(sp, x, noise)           # the reference's arithmetic: torch-CPU float32
    assert np.isfinite(ll32).all()
    flow = native.NativeImageFlow(sp)
    st = flow.numerics()
    assert native.MATH_NAME[int(st.math_mode)] == "f32" and bool(st.demoted) and not (st.worst_rel_err <= st.tolerance)
    z, ldj, ll = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    assert rel_err(ll.cpu().numpy(), ll32) < LL_RTOL
    assert rel_err(ldj.cpu().numpy(), ld32) < LL_RTOL


def test_image_range_marks_and_repair_without_the_probe(monkeypatch):
    """The same model with the probe switched off (GBNF_IMAGE_NO_PROBE: the handle stays on split f16, as for a model that only
    the caller's DATA drives out of range).  EVERY image of the batch is out of range.  The very first call -- and every later
    one -- returns what the exact-f32 handle returns for every image: the repair launch behind the split-f16 pass walks each
    marked image through the exact-f32 sequence in the same call, without a capacity (no NaN where the
    reference is finite)."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    monkeypatch.setenv("GBNF_IMAGE_NO_PROBE", "1")
    native.tuning_set("check_every", -1)           # (range protocol alone: the on-data check has its own test)
    try:
        dev = torch.device("cuda:0")
        sp = _blow_up_hidden(synth.synth_image_glow_spec((3, 32, 32), h=64, K=2, L=2, seed=11))
        flow = native.NativeImageFlow(sp)
        assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3"
        x, noise = synth.synth_image_batch(11, seed=3)
        zo, _, _, ld32, ll32 = oracle.image_component_forward(sp, x, noise)
        native.saturation_count(reset=True)
        xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
        z, ldj, ll = flow.forward(xd, nd)                                  # FIRST call, 11 marked images (> the old capacity of 8)
        assert native.saturation_count(reset=True) > 0
        assert rel_err(ll.cpu().numpy(), ll32) < LL_RTOL and rel_err(ldj.cpu().numpy(), ld32) < LL_RTOL
        assert np.abs(z.cpu().numpy() - zo).max() <= 2e-5 * max(1.0, float(np.abs(zo).max()))
        rc = flow.repair_counts()
        assert int(flow.numerics().checks) == 1 and rc["marked_calls"] == 1 and rc["repaired_images"] == 11
        # ... exactly what the exact-f32 handle gives (the same code on the same image; log-det sums are atomic: not bit for bit)
        monkeypatch.setenv("GBNF_MATH", "f32")
        exact = native.NativeImageFlow(sp)
        monkeypatch.delenv("GBNF_MATH")
        z3, ldj3, ll3 = exact.forward(xd, nd)
        assert rel_err(ll.cpu().numpy(), ll3.cpu().numpy()) < 1e-6 and torch.equal(z, z3)
        # more images than repair workgroups (256): the workgroups loop
        xb, nb_ = synth.synth_image_batch(300, seed=4)
        z4, ldj4, ll4 = flow.forward(torch.from_numpy(xb).to(dev), torch.from_numpy(nb_).to(dev), want_z=False)
        _, _, ll5 = exact.forward(torch.from_numpy(xb).to(dev), torch.from_numpy(nb_).to(dev), want_z=False)
        assert torch.isfinite(ll4).all() and rel_err(ll4.cpu().numpy(), ll5.cpu().numpy()) < 1e-6
        assert flow.repair_counts()["repaired_images"] == 311
    finally:
        native.tuning_set("check_every", 256)


def test_image_graph_captured_before_any_mark_repairs_a_later_batch(monkeypatch):
    """a HIP graph of the call captured while the handle had never seen a mark must still repair an
    out-of-range batch it is replayed on later -- the repair launch is part of every call's sequence from the start, and the
    on-data check schedule is counted on the device."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    monkeypatch.setenv("GBNF_IMAGE_NO_PROBE", "1")
    dev = torch.device("cuda:0")
    sp = _blow_up_hidden(synth.synth_image_glow_spec((3, 32, 32), h=64, K=2, L=2, seed=11))
    flow = native.NativeImageFlow(sp)
    x, noise = synth.synth_image_batch(5, seed=3)
    _, _, _, ld32, ll32 = oracle.image_component_forward(sp, x, noise)
    xs, ns = torch.zeros(5, 3, 32, 32, device=dev), torch.zeros(5, 3, 32, 32, device=dev)
    xs.fill_(0.5); ns.fill_(0.5)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        flow.forward(xs, ns)                                            # warm-up (allocates the workspace) outside the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            z, ldj, ll = flow.forward(xs, ns)
    xs.copy_(torch.from_numpy(x)); ns.copy_(torch.from_numpy(noise))
    g.replay()
    torch.cuda.synchronize()
    assert rel_err(ll.cpu().numpy(), ll32) < LL_RTOL and rel_err(ldj.cpu().numpy(), ld32) < LL_RTOL


def test_image_on_data_check_demotes_and_repairs_the_failing_call(monkeypatch):
    """The periodic on-data precision check (first launch, every check_every-th): with the tolerance forced to zero the first
    call's check fails -> the repair launch of THAT call re-evaluates every image (results = the exact-f32 handle's), the pinned
    word is raised and the next call runs the exact-f32 kernels directly."""
    import torch
    from gbnf_amd import native, synth
    dev = torch.device("cuda:0")
    sp = synth.synth_image_glow_spec((3, 32, 32), h=64, K=2, L=2, seed=5)
    x, noise = synth.synth_image_batch(6, seed=9)
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    exact = native.NativeImageFlow(sp, math="f32")
    z3, ldj3, ll3 = exact.forward(xd, nd)
    flow = native.NativeImageFlow(sp)
    assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3"
    z0, ldj0, ll0 = flow.forward(xd, nd)                                  # ordinary tolerance: the check passes
    torch.cuda.synchronize()
    rc = flow.repair_counts()
    assert rc["data_checks"] == 2 and rc["failed_checks"] == 0 and rc["worst_check_rel_err"] < 2.5e-6 and rc["marked_calls"] == 0
    flow2 = native.NativeImageFlow(sp)
    native.tuning_set("check_tolerance_e9", 0)
    try:
        z, ldj, ll = flow2.forward(xd, nd)
        torch.cuda.synchronize()
    finally:
        native.tuning_set("check_tolerance_e9", 2500)
    rc = flow2.repair_counts()
    assert rc["failed_checks"] >= 1
    assert torch.equal(z, z3) and rel_err(ll.cpu().numpy(), ll3.cpu().numpy()) < 1e-6        # the failing call itself: all on exact f32
    st = flow2.numerics()
    assert bool(st.demoted) and native.MATH_NAME[int(st.math_mode)] == "f32"
    z2, ldj2, ll2 = flow2.forward(xd, nd)                                 # from now on: the exact-f32 kernels directly
    assert torch.equal(z2, z3)


def test_image_well_scaled_model_stays_on_split_f16_and_unmarked():
    import torch
    from gbnf_amd import native, synth
    dev = torch.device("cuda:0")
    sp = synth.synth_image_glow_spec((3, 32, 32), h=256, K=2, L=2, seed=5, trained_like=True)
    flow = native.NativeImageFlow(sp)
    st = flow.numerics()
    assert native.MATH_NAME[int(st.math_mode)] == "f16x3" and st.worst_rel_err <= st.tolerance
    native.saturation_count(reset=True)
    x, noise = synth.synth_image_batch(16, seed=9)
    flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    torch.cuda.synchronize()
    assert native.saturation_count(reset=True) == 0 and int(flow.numerics().checks) == 0
    rc = flow.repair_counts()                                   # nothing marked, nothing repaired; the first launch's on-data check passed
    assert rc["marked_calls"] == 0 and rc["repaired_images"] == 0 and rc["failed_checks"] == 0 and rc["data_checks"] == 2


@pytest.mark.parametrize("size,h,K,L,kw", [((1, 28, 28), 256, 3, 2, {}), ((1, 28, 20), 64, 2, 2, {}), ((1, 28, 20), 256, 2, 1, {"depth": 2}),
                                           ((3, 24, 16), 48, 2, 2, {"coupling": "additive", "permutation": "reverse"}),
                                           ((1, 28, 28), 32, 2, 2, {"depth": 0, "learn_top": False}), ((2, 8, 12), 16, 1, 1, {}),
                                           # three levels: the 4 x 4 (3 x 3) map of the last one in the corner of 8 x 8 storage
                                           ((3, 32, 32), 64, 2, 3, {}), ((1, 24, 24), 32, 1, 3, {"coupling": "additive"}), ((1, 16, 16), 32, 2, 2, {}),
                                           # hidden widths above 256 (the usual Glow width is 512): exact-f32 convolutions, the last 3 x 3's
                                           # 512-channel strip staged in two halves
                                           ((3, 32, 32), 512, 1, 2, {}), ((1, 28, 28), 384, 1, 2, {"coupling": "additive"}), ((3, 32, 32), 300, 1, 1, {"depth": 2}),
                                           # ... and since round 5 on the fused split-f16 kernel in two halves of the hidden channels (depth 1):
                                           # halves of 16, 12, 10 and 9.x tiles, both couplings, maps smaller than their storage
                                           ((3, 32, 32), 320, 2, 2, {}), ((1, 28, 20), 448, 1, 2, {}), ((3, 32, 32), 290, 1, 2, {"coupling": "additive"}),
                                           ((1, 28, 28), 512, 2, 2, {"permutation": "shuffle"}),
                                           # widths whose padding to 64 holds a whole empty 32-channel chunk of the contraction
                                           ((1, 32, 20), 257, 2, 1, {"coupling": "additive", "permutation": "reverse"}), ((3, 32, 32), 330, 1, 2, {}),
                                           ((1, 28, 28), 460, 1, 2, {}),
                                           # round 6: depth 0 / 2 on the fused kernel -- full maps (CIFAR-shaped), both couplings, padded hidden widths
                                           ((3, 32, 32), 256, 2, 2, {"depth": 0}), ((3, 32, 32), 256, 2, 2, {"depth": 2}),
                                           ((3, 32, 32), 100, 1, 2, {"depth": 2, "coupling": "additive"}), ((1, 28, 28), 200, 2, 2, {"depth": 0, "permutation": "shuffle"}),
                                           ((3, 32, 32), 64, 1, 3, {"depth": 2})])
@pytest.mark.parametrize("math", ["default", "f32"])
def test_image_inputs_smaller_than_the_storage_match_oracle(size, h, K, L, kw, math, monkeypatch):
    """The reference's other image loaders hand over 1 x 28 x 28 and 1 x 28 x 20 (utils/load_data.py:389-529).  Such a map lives in
    the corner of the 16- / 8-wide storage the kernels are built for; everything outside must behave as the map's zero padding:
    the fused split-f16 coupling-net kernel (default) and the exact-f32 convolutions against the float32 oracle."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    if math == "f32":
        monkeypatch.setenv("GBNF_MATH", "f32")
    dev = torch.device("cuda:0")
    sp = synth.synth_image_glow_spec(size, h, K, L, seed=3, **kw)
    x, noise = synth.synth_image_batch(5, size, seed=4)
    flow = native.NativeImageFlow(sp)
    if math == "default" and kw.get("depth", 1) == 1:      # every depth-1 width up to 512 runs on split f16 (round 5: above 256 in two halves)
        assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3", flow.numerics().worst_rel_err
    z, ldj, ll = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    zo, _, _, ldo, llo = oracle.image_component_forward(sp, x, noise)
    assert tuple(z.shape) == zo.shape
    assert rel_err(ll.cpu().numpy(), llo) < LL_RTOL and rel_err(ldj.cpu().numpy(), ldo) < LL_RTOL
    assert np.abs(z.cpu().numpy() - zo).max() <= 2e-5 * max(1.0, float(np.abs(zo).max()))
    # a second call on the same workspace (stale data outside the map would show up here) is bit-identical
    z2, ldj2, ll2 = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
    assert torch.equal(ll, ll2) or rel_err(ll2.cpu().numpy(), llo) < LL_RTOL
    if math == "default" and kw.get("depth", 1) != 1 and h <= 256:
        # round 6: coupling nets of depth 0 / 2 run on the fused split-f16 kernel too (hidden widths to 256) -- seen as
        # results that differ from the exact-f32 convolutions' in the last bits while both meet the oracle
        torch.cuda.synchronize()
        rc = flow.repair_counts()          # nothing was marked, no on-data check failed: what came back IS the split-f16 pass's result
        assert rc["failed_checks"] == 0 and rc["repaired_images"] == 0 and rc["data_checks"] >= 1, rc
        assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3" and not flow.numerics().demoted
        if size == (3, 32, 32):            # (small maps often agree with the exact-f32 convolutions to the last bit: no evidence either way)
            monkeypatch.setenv("GBNF_MATH", "f32")
            flow32 = native.NativeImageFlow(sp)
            monkeypatch.delenv("GBNF_MATH")
            _, _, ll32 = flow32.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
            assert rel_err(ll32.cpu().numpy(), llo) < LL_RTOL
            assert not torch.equal(ll32, ll), "depth 0 / 2 did not leave the exact-f32 convolutions"


def test_image_module_dropin_on_28x28_matches_reference():
    """BoostedFlow(args) with input_size [1, 28, 28] against the reference's own outputs (fixture g19)."""
    import argparse
    import torch
    from gbnf_amd import BoostedFlow, image_glow
    cfg, specs, x, noise, data = load_image_case("g19_image_glow_1x28x28")
    dev = torch.device("cuda:0")
    args = argparse.Namespace(
        num_flows=cfg["K"], z_size=784, density_evaluation=True, device=dev, cuda=True, component_type="glow",
        num_components=cfg["C"], rho_init="decreasing", learn_top=cfg["learn_top"], y_classes=0, y_condition=False,
        sample_size=4, input_size=[1, 28, 28], h_size=cfg["h"], num_blocks=cfg["L"], actnorm_scale=1.0,
        flow_permutation=cfg["permutation"], flow_coupling=cfg["coupling"], LU_decomposed=False, num_dequant_blocks=0,
        coupling_network="tanh", coupling_network_depth=cfg["depth"], batch_norm=False)
    m = BoostedFlow(args)
    assert isinstance(m, image_glow.BoostedImageFlow)
    for c, sp in enumerate(specs):
        image_glow.load_image_spec(m.flows[c], sp)
    m.eval()
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    with torch.no_grad():
        ll = m.component_log_prob(xd, noise=nd)
        assert rel_err(ll.cpu().numpy().T, data["ll"]) < LL_RTOL
        assert rel_err(m.log_prob(xd, noise=nd).cpu().numpy(), data["G"]) < LL_RTOL
        z, z_mu, z_var, ldj, y = m(x=xd, components=1)
        assert z.shape == (cfg["N"], 8, 7, 7) and z_mu.shape == z.shape and torch.isfinite(ldj).all()


@pytest.mark.parametrize("size,h,K,L,kw", [((1, 28, 28), 64, 2, 2, {}), ((1, 28, 20), 32, 2, 2, {"coupling": "additive", "permutation": "shuffle"}),
                                           ((1, 28, 28), 32, 2, 1, {}), ((3, 24, 16), 32, 1, 2, {"depth": 2}),
                                           ((3, 32, 32), 32, 2, 3, {}), ((1, 24, 24), 32, 1, 3, {}), ((3, 32, 32), 512, 1, 2, {}),
                                           # round 6: depth 0 / 2 on the fused kernel, the z -> x direction
                                           ((3, 32, 32), 256, 2, 2, {"depth": 0}), ((3, 32, 32), 128, 2, 2, {"depth": 2}), ((1, 28, 28), 64, 2, 2, {"depth": 0})])
def test_image_inverse_on_inputs_smaller_than_the_storage(size, h, K, L, kw):
    """z -> x for the padded maps against the float64 oracle's Glow.decode (Split2d draws injected)."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    dev = torch.device("cuda:0")
    sp = synth.synth_image_glow_spec(size, h, K, L, seed=8, **kw)
    flow = native.NativeImageFlow(sp)
    rng = np.random.RandomState(9)
    n = 3
    z = (0.7 * rng.standard_normal((n,) + flow.z_shape)).astype(np.float32)
    shapes = oracle.image_split_shapes(sp, size)
    assert [tuple(s) for s in shapes] == [tuple(s) for s in flow.split_shapes()]
    eps = [rng.standard_normal((n,) + tuple(sh)).astype(np.float32) for sh in shapes]
    x = flow.inverse(torch.from_numpy(z).to(dev), [torch.from_numpy(e).to(dev) for e in eps], 0.9)
    x_or = oracle.image_component_inverse(sp, z, eps, 0.9, dtype=torch.float64)
    assert tuple(x.shape) == x_or.shape
    assert np.abs(x.cpu().numpy() - x_or).max() <= 2e-5


def test_image_range_marks_and_repair_on_a_map_smaller_than_its_storage(monkeypatch):
    """The marks -> repair launch protocol with x, noise and z of the map's own size (1 x 28 x 28 in 16 x 16 storage): the first
    call is repaired to the float32 oracle."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    monkeypatch.setenv("GBNF_IMAGE_NO_PROBE", "1")
    dev = torch.device("cuda:0")
    size = (1, 28, 28)
    sp = _blow_up_hidden(synth.synth_image_glow_spec(size, h=64, K=2, L=2, seed=11))
    flow = native.NativeImageFlow(sp)
    assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3"
    x, noise = synth.synth_image_batch(6, size, seed=3)
    zo, _, _, ld32, ll32 = oracle.image_component_forward(sp, x, noise)
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    for _ in range(2):
        z, ldj, ll = flow.forward(xd, nd)
        assert tuple(z.shape) == zo.shape
        assert rel_err(ll.cpu().numpy(), ll32) < LL_RTOL and rel_err(ldj.cpu().numpy(), ld32) < LL_RTOL
        assert np.abs(z.cpu().numpy() - zo).max() <= 2e-5 * max(1.0, float(np.abs(zo).max()))


from conftest import IMAGE_ACTNORM_INIT_CASES, load_image_actnorm_init_case  # noqa: E402


@pytest.mark.parametrize("name", IMAGE_ACTNORM_INIT_CASES)
def test_image_actnorm_data_init_matches_reference(name):
    """g20: BoostedImageFlow.initialize_actnorms on a freshly constructed model (the reference's own parameter initialisation,
    loaded through load_state_dict) against what the reference's first training-mode forward left in every ActNorm2d -- the
    FlowSteps' own and the ones inside the coupling nets -- and the initialised model's output against the reference's output of
    that call.  The statistics come from gbnf_image_flow_actnorm_stats (exact-f32 kernels), layer by layer."""
    import torch
    cfg, m, x, noise, after, data = load_image_actnorm_init_case(name, device="cuda:0")
    dev = torch.device("cuda:0")
    xd, nd = torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev)
    glow = m.flows[0]
    assert not any(a.inited for a in glow._actnorms())
    with pytest.raises(ValueError):
        m.eval(); m.component_log_prob(xd, noise=nd)            # un-initialised: loud, like the reference (models/layers.py:473-475)
    m.initialize_actnorms(xd, noise=nd)
    acts = glow._actnorms()
    assert all(a.inited for a in acts) and len(acts) == len(after)
    for a, (rb, rl) in zip(acts, after):
        assert np.abs(a.bias.detach().cpu().numpy().reshape(-1) - rb).max() <= 1e-5 * max(1.0, float(np.abs(rb).max()))
        assert np.abs(a.logs.detach().cpu().numpy().reshape(-1) - rl).max() <= 1e-5 * max(1.0, float(np.abs(rl).max()))
    m.eval()
    zz, ldj, ll = m.native_flow(0).forward(xd, nd)
    assert rel_err(ldj.cpu().numpy(), data["ldj"]) < LL_RTOL
    assert np.abs(zz.cpu().numpy() - data["z"]).max() <= 2e-5 * max(1.0, float(np.abs(data["z"]).max()))
    # a second call leaves an initialised model alone
    before = [a.logs.detach().clone() for a in acts]
    m.initialize_actnorms(xd * 0.5, noise=nd)
    assert all(torch.equal(b, a.logs.detach()) for b, a in zip(before, acts))


@pytest.mark.parametrize("size,L", [((3, 32, 32), 2), ((1, 28, 28), 2), ((3, 32, 32), 3)])
def test_image_inverse_on_split_f16_repairs_in_the_same_call(size, L, monkeypatch):
    """z -> x runs the fused split-f16 coupling-net kernel like the forward.  An image whose hidden activation leaves the fp16
    range is re-evaluated on the exact-f32 sequence by the repair launch of the same call: the first call already matches the
    oracle (round 4 returned NaN once)."""
    import torch
    from gbnf_amd import native, synth
    from oracle import gbnf_oracle as oracle
    monkeypatch.setenv("GBNF_IMAGE_NO_PROBE", "1")
    dev = torch.device("cuda:0")
    sp = _blow_up_hidden(synth.synth_image_glow_spec(size, h=64, K=2, L=L, seed=11))
    flow = native.NativeImageFlow(sp)
    assert native.MATH_NAME[int(flow.numerics().math_mode)] == "f16x3"
    rng = np.random.RandomState(5)
    n = 4
    z = (0.7 * rng.standard_normal((n,) + flow.z_shape)).astype(np.float32)
    eps = [rng.standard_normal((n,) + tuple(sh)).astype(np.float32) for sh in flow.split_shapes()]
    zd, ed = torch.from_numpy(z).to(dev), [torch.from_numpy(e).to(dev) for e in eps]
    x_or = oracle.image_component_inverse(sp, z, eps, 0.9, dtype=torch.float64)
    for _ in range(2):
        x1 = flow.inverse(zd, ed, 0.9)
        assert np.abs(x1.cpu().numpy() - x_or).max() <= 2e-5
    assert int(flow.numerics().checks) == 2 and flow.repair_counts()["repaired_images"] == 2 * n
    if size == (3, 32, 32) and L == 2:
        # a well-scaled model stays on the fast kernels and agrees with the exact-f32 handle
        sp2 = synth.synth_image_glow_spec(size, h=64, K=2, L=2, seed=12)
        fast = native.NativeImageFlow(sp2)
        exact = native.NativeImageFlow(sp2, math="f32")
        xa, xb = fast.inverse(zd, ed, 0.9), exact.inverse(zd, ed, 0.9)
        assert int(fast.numerics().checks) == 0 and float((xa - xb).abs().max()) <= 2e-5
