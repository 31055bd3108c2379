"""CPU: host-side mirror of the reference's BoostedFlow -- names, rho, component selection, side-car,
spec export, and the "no fallback" contract.  No compute happens here (no GPU in this container)."""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from gbnf_amd import native, spec as gspec, synth
from gbnf_amd.boosted_flow import BoostedFlow


def make_args(kind="glow", d=7, h=12, K=3, C=2, depth=1, coupling_network="tanh", coupling="affine",
              permutation="shuffle", batch_norm=True, rho_init="decreasing"):
    return argparse.Namespace(
        num_flows=K, z_size=d, density_evaluation=True, device=torch.device("cpu"), cuda=False,
        component_type=kind, num_components=C, rho_init=rho_init, learn_top=False, y_classes=0,
        y_condition=False, sample_size=4, input_size=[d], h_size=h, num_blocks=1, actnorm_scale=1.0,
        flow_permutation=permutation, flow_coupling=coupling, LU_decomposed=False,
        num_dequant_blocks=0, coupling_network=coupling_network, coupling_network_depth=depth,
        batch_norm=batch_norm)


LAYOUT = json.load(open(os.path.join(GOLDEN_DIR, "state_dict_layout.json")))


@pytest.mark.parametrize("case,kw", [
    ("glow", dict(kind="glow")),
    ("realnvp", dict(kind="realnvp")),
    ("realnvp_mixed", dict(kind="realnvp", coupling_network="mixed")),
    ("realnvp_residual", dict(kind="realnvp", coupling_network="residual", depth=2)),
    ("glow_depth2_additive", dict(kind="glow", depth=2, coupling="additive", permutation="reverse")),
])
def test_state_dict_layout_matches_reference(case, kw):
    """Same keys, same shapes, same parameter order as the reference (fixture made from the reference)."""
    m = BoostedFlow(make_args(**kw))
    mine = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert mine == LAYOUT[case]
    assert [n for n, _ in m.named_parameters()] == LAYOUT[case + "::named_parameters"]
    # optimization/optimizers.py:29-35 and density_experiment.py:538-539 parse "flows.<c>." prefixes
    for n, _ in m.named_parameters():
        assert n.startswith("flows.") and n.split(".")[1].isdigit()


@pytest.mark.parametrize("name,kind,d,h,K", [("g13_glow_random_d43_h64", "glow", 43, 64, 6),
                                              ("g13_realnvp_random_d21_h32", "realnvp", 21, 32, 5)])
def test_random_coupling_network_draws_like_the_reference(name, kind, d, h, K):
    """`--coupling_network random`: the constructor draws TanhNet / ReLUNet from numpy's global RNG in the reference's
    order (per step for Glow, models/glow.py:295-296; per net for RealNVP, models/realnvp.py:59-60), so the same seed
    builds the same architecture -- the fixture records what the reference drew with seed 13."""
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    cfg = json.loads(bytes(g["config"]).decode())
    np.random.seed(13)
    m = BoostedFlow(make_args(kind=kind, d=d, h=h, K=K, C=2, coupling_network="random"))
    got = [gspec.activation_pattern_of_component(m.flows[c]) for c in range(2)]
    want = [tuple(tuple(st) if kind == "realnvp" else st[0] for st in comp) for comp in cfg["activations"]]
    assert got == want
    assert any(len(set(p)) > 1 for p in got)                  # a mixed component exists
    assert m._per_step_activation()                            # -> the per-step-activation kernel variants
    specs = [gspec.unflatten_spec(g, prefix=f"c{c}.") for c in range(2)]
    assert [native.activation_pattern(sp) for sp in specs] == got
    assert native.needs_per_step_activation(specs)
    if kind == "glow":        # its first component happened to draw tanh six times: uniform on its own
        assert not native.needs_per_step_activation(specs[:1] * 2)


def test_side_car_restores_randomly_drawn_activations():
    """A `--coupling_network random` model draws its activations at construction: the side-car records them and a freshly
    built model (other draws) takes them over together with the state_dict."""
    np.random.seed(5)
    a = BoostedFlow(make_args(kind="glow", d=9, h=10, K=6, C=2, coupling_network="random"))
    np.random.seed(6)
    b = BoostedFlow(make_args(kind="glow", d=9, h=10, K=6, C=2, coupling_network="random"))
    pa = [gspec.activation_pattern_of_component(f) for f in a.flows]
    assert pa != [gspec.activation_pattern_of_component(f) for f in b.flows]
    b.load_state_dict(a.state_dict())
    b.load_permutation_state(a.permutation_state())
    assert [gspec.activation_pattern_of_component(f) for f in b.flows] == pa
    np.random.seed(7)
    r = BoostedFlow(make_args(kind="realnvp", d=9, h=10, K=5, C=2, coupling_network="random"))
    np.random.seed(8)
    r2 = BoostedFlow(make_args(kind="realnvp", d=9, h=10, K=5, C=2, coupling_network="random"))
    r2.load_permutation_state(r.permutation_state())
    assert [gspec.activation_pattern_of_component(f) for f in r2.flows] == [gspec.activation_pattern_of_component(f) for f in r.flows]


def test_rho_init_and_increment():
    m = BoostedFlow(make_args(C=8))
    np.testing.assert_array_equal(m.rho.numpy(), np.array([1, .5, .25, .125, .0625, .05, .05, .05], np.float32))
    u = BoostedFlow(make_args(C=4, rho_init="uniform"))
    np.testing.assert_allclose(u.rho.numpy(), 0.25)
    assert m.component == 0 and not m.all_trained
    for expect in range(1, 8):
        m.increment_component()
        assert m.component == expect and not m.all_trained
    m.increment_component()          # wraps: models/boosted_flow.py:52-59
    assert m.component == 0 and m.all_trained


def test_sample_component_semantics():
    torch.manual_seed(0)
    m = BoostedFlow(make_args(C=4))
    m.component = 2
    assert m._sample_component("c") == 2
    draws = [m._sample_component("1:c-1") for _ in range(200)]
    assert set(draws) <= {0, 1} and len(set(draws)) == 2
    draws = [m._sample_component("1:c") for _ in range(300)]
    assert set(draws) == {0, 1, 2}
    m.all_trained = True
    draws = [m._sample_component("-c") for _ in range(300)]
    assert 2 not in draws and set(draws) == {0, 1, 3}
    assert set(m._sample_component("1:c") for _ in range(400)) == {0, 1, 2, 3}
    with pytest.raises(ValueError):
        m._sample_component("bogus")


def test_no_cpu_fallback_and_guards():
    m = BoostedFlow(make_args())
    m.eval()
    x = torch.zeros(4, 7)
    with pytest.raises(ValueError):            # ActNorm not initialised: same error as the reference
        m(x=x.cuda() if torch.cuda.is_available() else _FakeCuda(x), components=0)
    for f in m.flows:
        f.set_actnorm_init()
    with pytest.raises(native.GbnfError):      # CPU tensor: the product has no CPU path
        m(x=x, components=0)
    with pytest.raises(native.GbnfError):
        m.log_prob(x)
    with pytest.raises(native.GbnfError):      # the inverse direction has no CPU path either
        m(z=x, components=0, reverse=True)
    with pytest.raises(NotImplementedError):
        BoostedFlow(make_args(coupling_network="residual"))
    with pytest.raises(NotImplementedError):
        BoostedFlow(make_args(permutation="invconv"))


class _FakeCuda(torch.Tensor):
    """A CPU tensor that claims to be on the device, to reach the ActNorm guard without a GPU."""
    @staticmethod
    def __new__(cls, t):
        return torch.Tensor._make_subclass(cls, t)

    @property
    def is_cuda(self):
        return True


@pytest.mark.parametrize("kind,kw", [("glow", {}), ("glow", dict(coupling="additive", permutation="reverse")),
                                      ("realnvp", {}), ("realnvp", dict(coupling_network="mixed"))])
def test_load_spec_and_export_roundtrip(kind, kw):
    d, h, K, C = 9, 20, 3, 3
    m = BoostedFlow(make_args(kind=kind, d=d, h=h, K=K, C=C, **kw))
    skw = {}
    if kind == "glow":
        skw = dict(coupling=kw.get("coupling", "affine"), permutation=kw.get("permutation", "shuffle"))
    else:
        skw = dict(coupling_network=kw.get("coupling_network", "tanh"))
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=4, **skw)
    for c in range(C):
        m.load_spec(c, specs[c])
        back = gspec.spec_from_component(m.flows[c])
        fa, fb = gspec.flatten_spec(specs[c]), gspec.flatten_spec(back)
        assert fa.keys() == fb.keys()
        for k in fa:
            assert np.array_equal(fa[k], fb[k]), k
    # npz round trip of the spec container
    again = gspec.unflatten_spec(gspec.flatten_spec(specs[0], prefix="c0."), prefix="c0.")
    fa, fb = gspec.flatten_spec(specs[0]), gspec.flatten_spec(again)
    assert all(np.array_equal(fa[k], fb[k]) for k in fa)


def test_permutation_sidecar_roundtrip():
    torch.manual_seed(1)
    a = BoostedFlow(make_args(C=2, K=3))
    for f in a.flows:
        f.set_actnorm_init()
    a.component, a.all_trained = 1, True
    side = a.permutation_state()
    torch.manual_seed(2)
    b = BoostedFlow(make_args(C=2, K=3))
    b.load_state_dict(a.state_dict())
    # state_dict alone does NOT carry the permutations ...
    assert any(not torch.equal(la.permutation.indices, lb.permutation.indices)
               for fa, fb in zip(a.flows, b.flows) for la, lb in zip(fa.flow.layers, fb.flow.layers))
    b.load_permutation_state(side)
    for fa, fb in zip(a.flows, b.flows):
        for la, lb in zip(fa.flow.layers, fb.flow.layers):
            assert torch.equal(la.permutation.indices, lb.permutation.indices)           # bit-exact indices
            inv = lb.permutation.indices_inverse
            assert torch.equal(inv[lb.permutation.indices], torch.arange(lb.permutation.num_dim))
            assert lb.actnorm.inited
    assert b.component == 1 and b.all_trained
    with pytest.raises(ValueError):
        b.flows[0].flow.layers[0].permutation.set_indices([0, 0, 1, 2, 3, 4, 5])


def test_default_permutation_is_reversed_arange():
    m = BoostedFlow(make_args(permutation="reverse", d=5))
    for f in m.flows:
        for layer in f.flow.layers:
            assert layer.permutation.indices.tolist() == [4, 3, 2, 1, 0]   # models/layers.py:636


# ------------------------------------------------------------------ checkpoint files
def _ckpt_fixture():
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g11_reference_checkpoint.npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    return cfg, data, os.path.join(GOLDEN_DIR, "g11_reference_checkpoint.pt")


def test_loads_a_checkpoint_written_by_the_reference():
    """g11: a file written by the reference's utils.utilities.save.  Parameters, component and all_trained come back;
    the file has no permutation indices (S5) -> a warning, unless the caller supplies the side-car."""
    import warnings
    from gbnf_amd import checkpoint
    cfg, data, path = _ckpt_fixture()
    args = make_args(kind="glow", d=cfg["d"], h=cfg["h"], K=cfg["K"], C=cfg["C"])
    m = BoostedFlow(args)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ckpt = checkpoint.load(m, opt, path, args)
    assert any("permutation" in str(x.message) for x in w)
    assert m.component == cfg["component"] and m.all_trained == cfg["all_trained"]
    for k, v in ckpt["model"].items():
        assert torch.equal(m.state_dict()[k], v), k
    # with the indices exported from the live reference model: no warning, indices installed
    side = {"component": cfg["component"], "all_trained": cfg["all_trained"],
            "indices": {f"{c}.{k}": torch.from_numpy(data["indices"][c, k]) for c in range(cfg["C"]) for k in range(cfg["K"])},
            "actnorm_inited": {f"{c}.{k}": True for c in range(cfg["C"]) for k in range(cfg["K"])}}
    m2 = BoostedFlow(args)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        checkpoint.load(m2, None, path, args, side_car=side)
    assert not any("permutation" in str(x.message) for x in w)
    for c in range(cfg["C"]):
        for k in range(cfg["K"]):
            assert m2.flows[c].flow.layers[k].permutation.indices.tolist() == data["indices"][c, k].tolist()


def test_checkpoint_round_trip_keeps_permutations(tmp_path):
    from gbnf_amd import checkpoint
    args = make_args(kind="glow", d=7, h=12, K=3, C=2)
    m = BoostedFlow(args)
    for f in m.flows:
        f.set_actnorm_init()
    m.component, m.all_trained = 1, True
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.StepLR(opt, 3)
    path = str(tmp_path / "ck.pt")
    checkpoint.save(m, opt, path, scheduler=sched)
    raw = torch.load(path)
    assert {"model", "optimizer", "scheduler", "all_trained", "component"} <= set(raw)      # the reference's keys
    m2 = BoostedFlow(args)
    checkpoint.load(m2, torch.optim.Adam(m2.parameters(), lr=1e-3), path, args)
    assert m2.component == 1 and m2.all_trained is True
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    for f1, f2 in zip(m.flows, m2.flows):
        for l1, l2 in zip(f1.flow.layers, f2.flow.layers):
            assert torch.equal(l1.permutation.indices, l2.permutation.indices) and bool(l2.actnorm.inited)
    # the reference's "initialise from arguments" branch (utils/utilities.py:53-64)
    args.boosted, args.loaded_init_component, args.loaded_all_trained, args.loaded_num_components = True, 0, False, None
    checkpoint.load(m2, None, path, args, init_with_args=True)
    assert m2.component == 0 and m2.all_trained is False


# ------------------------------------------------------------------ image components (BASELINE.json configs[3])
def _image_args(permutation="invconv", LU=False, learn_top=True, coupling="affine", depth=1, h=8, K=2, L=2, C=2,
                device=torch.device("cpu")):
    a = make_args(kind="glow", d=3 * 32 * 32, h=h, K=K, C=C, depth=depth, coupling=coupling, permutation=permutation)
    a.input_size = [3, 32, 32]; a.num_blocks = L; a.learn_top = learn_top; a.LU_decomposed = LU; a.device = device
    return a


@pytest.mark.parametrize("case,kw", [("image_invconv", dict(permutation="invconv", LU=False, learn_top=True)),
                                     ("image_lu", dict(permutation="invconv", LU=True, learn_top=True)),
                                     ("image_shuffle_additive", dict(permutation="shuffle", learn_top=False,
                                                                     coupling="additive", depth=2))])
def test_image_state_dict_layout_matches_reference(case, kw):
    """BoostedFlow(args) with a 3-d input_size builds the image mirror with the reference's parameter names/shapes."""
    from gbnf_amd import image_glow
    m = BoostedFlow(_image_args(**kw))
    assert isinstance(m, image_glow.BoostedImageFlow)
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == LAYOUT[case]
    assert [n for n, _ in m.named_parameters()] == LAYOUT[case + "::named_parameters"]


def test_image_spec_export_round_trips_synthetic_parameters():
    """Install a synthetic image spec into the mirror module by module and export it again."""
    from gbnf_amd import image_glow, synth
    sp = synth.synth_image_glow_spec((3, 32, 32), h=8, K=2, L=2, seed=3)
    m = BoostedFlow(_image_args(h=8, K=2, L=2, C=1))
    g = m.flows[0]
    with pytest.raises(ValueError):                 # ActNorm not initialised
        image_glow.image_spec_from_glow_module(g)
    image_glow.load_image_spec(g, sp)
    out = image_glow.image_spec_from_glow_module(g)
    for lv_a, lv_b in zip(sp["levels"], out["levels"]):
        for a, b in zip(lv_a["steps"], lv_b["steps"]):
            np.testing.assert_allclose(a["perm_w"], b["perm_w"], rtol=0, atol=1e-7)
            np.testing.assert_array_equal(a["an_logs"], b["an_logs"])
            for ca, cb in zip(a["convs"], b["convs"]):
                np.testing.assert_array_equal(ca["w"], cb["w"])
                assert (ca["logs"] is None) == (cb["logs"] is None)
        assert (lv_a["split"] is None) == (lv_b["split"] is None)
    np.testing.assert_array_equal(sp["learn_top"]["b"], out["learn_top"]["b"])
    with pytest.raises(native.GbnfError):           # CPU tensor: no CPU path
        m.eval()
        m(x=torch.zeros(2, 3, 32, 32), components=0)
