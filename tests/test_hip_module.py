"""GPU: the drop-in host module against the reference's golden vectors, driven exactly the way
density_experiment.evaluate drives the reference (density_experiment.py:561-573)."""
import argparse
import math

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
LL_RTOL = 1e-5


def _args(cfg, dev):
    import torch
    skw = cfg.get("synth_kw", {})
    kind = cfg["kind"]
    return argparse.Namespace(
        num_flows=cfg["K"], z_size=cfg["d"], density_evaluation=True, device=dev, cuda=True,
        component_type=kind, num_components=cfg["C"], rho_init="decreasing", learn_top=False, y_classes=0,
        y_condition=False, sample_size=4, input_size=[cfg["d"]], h_size=cfg["h"], num_blocks=1,
        actnorm_scale=1.0, flow_permutation=skw.get("permutation", "shuffle"),
        flow_coupling=skw.get("coupling", "affine"), LU_decomposed=False, num_dequant_blocks=0,
        coupling_network="random" if "activations" in cfg else skw.get("act", skw.get("coupling_network", "tanh")),
        coupling_network_depth=skw.get("depth", 1), batch_norm=skw.get("batch_norm", True))


def _model_from_case(g, dev):
    import torch
    from gbnf_amd import BoostedFlow
    if "activations" in g.cfg:     # `--coupling_network random`: the constructor draws from numpy's global RNG like the
        np.random.seed(13)         # reference does (tests/golden/make_golden.py native_random_case seeds it with 13)
    m = BoostedFlow(_args(g.cfg, dev))
    for c, spec in enumerate(g.specs):
        m.load_spec(c, spec)
    with torch.no_grad():
        m.rho.copy_(torch.from_numpy(g.rho).to(dev))
    m.eval()
    return m


def _evaluate_like_reference(model, x):
    """The caller's loop, verbatim in structure (density_experiment.py:561-573)."""
    import torch
    G_ll = None
    for c in range(model.component + 1):
        z_G, _, _, ldj_G, _ = model(x=x, components=c)
        ll = torch.sum(-0.5 * math.log(2 * math.pi) - 0.5 * z_G.pow(2), dim=-1) + ldj_G
        if c == 0:
            G_ll = ll
        else:
            rho_simplex = model.rho[0:(c + 1)] / torch.sum(model.rho[0:(c + 1)])
            last_ll = torch.log(1 - rho_simplex[c]) + G_ll
            next_ll = torch.log(rho_simplex[c]) + ll
            G_ll = torch.logsumexp(torch.cat([last_ll.view(-1, 1), next_ll.view(-1, 1)], dim=1), dim=1)
    return G_ll


@pytest.mark.parametrize("name", ["g2_glow_native_d43_h32_c3", "g4_realnvp_d21_h105_c8",
                                  "g5_glow_d43_h64_c2_additive", "g4_realnvp_d21_h105_c2_mixed",
                                  "g13_glow_random_d43_h64", "g13_realnvp_random_d21_h32",
                                  "g14_realnvp_residual_d21_h64_c2", "g14_realnvp_residual2_d8_h40_c2"])
def test_module_dropin_matches_reference(name, golden_case):
    import torch
    dev = torch.device("cuda:0")
    g = golden_case(name)
    m = _model_from_case(g, dev)
    x = torch.from_numpy(g.x).to(dev)
    m.component = g.n_used - 1
    # (1) the reference's own calling convention and 5-tuple
    z, z_mu, z_var, ldj, y_logits = m(x=x, components=0)
    assert y_logits is None and z_mu.shape == z.shape and z_var.shape == z.shape
    assert float(z_mu.abs().max()) == 0.0 and float(z_var.abs().max()) == 0.0
    assert rel_err(ldj.cpu().numpy(), g.ldj[0]) < LL_RTOL
    np.testing.assert_allclose(z.cpu().numpy(), g.z(0), rtol=0, atol=2e-5 * max(1.0, np.abs(g.z(0)).max()))
    # (2) evaluate()'s loop, unchanged, on top of the module
    G = _evaluate_like_reference(m, x)
    assert rel_err(G.cpu().numpy(), g.G) < LL_RTOL
    # (3) the convenience API
    assert rel_err(m.log_prob(x).cpu().numpy(), g.G) < LL_RTOL
    assert rel_err(m.component_log_prob(x).cpu().numpy().T, g.ll) < LL_RTOL
    zc, ldc = m.component_forward(x, g.n_used - 1)
    assert rel_err(ldc.cpu().numpy(), g.ldj[g.n_used - 1]) < LL_RTOL
    # string component selectors draw a component and return its transform
    torch.manual_seed(0)
    zs, _, _, ls, _ = m(x=x, components="1:c")
    assert any(rel_err(ls.cpu().numpy(), g.ldj[c]) < LL_RTOL for c in range(g.n_used))


def test_handles_follow_parameter_updates(golden_case):
    """Packed device copies are refreshed when parameters / permutations / rho change."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g5_glow_d43_h64_c2_reverse_relu")
    m = _model_from_case(g, dev)
    m.component = 1
    x = torch.from_numpy(g.x).to(dev)
    base = m.log_prob(x).clone()
    assert rel_err(base.cpu().numpy(), g.G) < LL_RTOL
    with torch.no_grad():
        m.flows[0].flow.layers[0].actnorm.bias.add_(0.25)
    moved = m.log_prob(x)
    assert float((moved - base).abs().max()) > 1e-3
    with torch.no_grad():
        m.flows[0].flow.layers[0].actnorm.bias.sub_(0.25)
    assert rel_err(m.log_prob(x).cpu().numpy(), g.G) < LL_RTOL
    # state_dict + side-car round trip into a freshly constructed model
    from gbnf_amd import BoostedFlow
    m2 = BoostedFlow(_args(g.cfg, dev))
    m2.load_state_dict(m.state_dict())
    m2.load_permutation_state(m.permutation_state())
    m2.eval()
    assert torch.equal(m2.log_prob(x), m.log_prob(x))


def test_training_mode_with_grad_is_recorded(golden_case):
    """Train mode with gradients enabled goes through the training path (autograd-recorded, live parameters); the
    no-grad loops of the reference keep using the packed evaluation kernels.  Same numbers either way."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g6_glow_d43_h64_n77")
    m = _model_from_case(g, dev)
    x = torch.from_numpy(g.x).to(dev)
    m.train()
    z, _, _, ldj, _ = m(x=x, components=0)
    assert z.requires_grad and ldj.requires_grad
    assert rel_err(ldj.detach().cpu().numpy(), g.ldj[0]) < LL_RTOL
    with torch.no_grad():
        z2, _, _, ldj2, _ = m(x=x, components=0)
    assert not ldj2.requires_grad
    assert rel_err(ldj2.cpu().numpy(), g.ldj[0]) < LL_RTOL


def test_sharded_single_rank_equals_mixture(golden_case):
    """world_size 1: the sharded path (gather degenerate) must equal the single-launch mixture."""
    import torch
    from gbnf_amd import native, sharded
    dev = torch.device("cuda:0")
    g = golden_case("g3_glow_d43_h215_c8")
    flows = [native.NativeFlow(s) for s in g.specs]
    mix = native.NativeMixture(flows)
    x = torch.from_numpy(g.x).to(dev)
    rho = torch.from_numpy(g.rho).to(dev)
    sm = sharded.ShardedMixture(8, lambda xx: mix.component_log_prob(xx), native.mixture_lse)
    G, ll = sm.log_prob(x, rho)
    G2, ll2 = mix.log_prob(x, rho)
    assert torch.equal(G, G2) and torch.equal(ll, ll2)
    outs = sm.log_prob_pipelined([x, x], rho)
    assert all(torch.equal(o, G2) for o in outs)
    assert rel_err(G.cpu().numpy(), g.G) < LL_RTOL


def test_actnorm_data_dependent_init_matches_reference():
    """G7: un-initialised model, train mode, first batch under no_grad -- as
    density_experiment.py:346-356 drives the reference -- must reproduce the reference's ActNorm parameters."""
    import torch
    from conftest import load_actnorm_init_case
    from gbnf_amd import BoostedFlow, native
    dev = torch.device("cuda:0")
    cfg, data, specs, x = load_actnorm_init_case()
    m = BoostedFlow(_args(cfg, dev))
    for c, spec in enumerate(specs):
        m.load_spec(c, spec)
        for layer in m.flows[c].flow.layers:
            layer.actnorm.inited = False
            with torch.no_grad():
                layer.actnorm.bias.zero_()
                layer.actnorm.logs.zero_()
    xd = torch.from_numpy(x).to(dev)
    m.eval()
    with pytest.raises(ValueError):                 # eval mode + un-initialised: the reference raises too
        m(x=xd, components=0)
    m.train()
    with torch.no_grad():
        for c in range(cfg["C"]):
            m(x=xd, components=c)
    for c in range(cfg["C"]):
        for k, layer in enumerate(m.flows[c].flow.layers):
            assert layer.actnorm.inited
            np.testing.assert_allclose(layer.actnorm.bias.detach().cpu().numpy().reshape(-1), data["an_bias"][c, k],
                                       rtol=0, atol=5e-6)
            np.testing.assert_allclose(layer.actnorm.logs.detach().cpu().numpy().reshape(-1), data["an_logs"][c, k],
                                       rtol=0, atol=5e-6)
    m.eval()
    m.component = cfg["C"] - 1
    assert rel_err(m.log_prob(xd).cpu().numpy(), data["G"]) < LL_RTOL
    # the statistics kernel on its own, incl. a single row and many blocks
    big = torch.randn(70000, 21, device=dev) * 3 + 1
    b, l = native.actnorm_init(big, 1.0)
    np.testing.assert_allclose(b.cpu().numpy(), -big.mean(0).cpu().numpy(), rtol=0, atol=2e-5)
    ref_logs = torch.log(1.0 / (torch.sqrt(((big - big.mean(0)) ** 2).mean(0)) + 1e-6)).cpu().numpy()
    np.testing.assert_allclose(l.cpu().numpy(), ref_logs, rtol=0, atol=2e-5)
    b2, l2 = native.actnorm_init(big, 1.0)
    assert torch.equal(b, b2) and torch.equal(l, l2)      # fixed-order reduction: bit-reproducible


def test_update_rho_runs_on_the_device_path(golden_case):
    """update_rho (models/boosted_flow.py:141-207) driven like the reference: rho of the current component moves
    by SGD on mean(g_nll - G_nll); rho_iters == 0 is a no-op."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g6_glow_d43_h64_c3_rho")
    m = _model_from_case(g, dev)
    m.component = 1
    m.args.rho_iters, m.args.rho_lr = 0, 0.1
    before = m.rho.clone()
    loader = [(torch.from_numpy(g.x), None)]
    m.update_rho(loader)
    assert torch.equal(m.rho, before)
    m.args.rho_iters = 12
    m.update_rho(loader)
    assert not torch.equal(m.rho, before) and 0.01 <= float(m.rho[1]) <= 100.0
    assert torch.equal(m.rho[0], before[0]) and torch.equal(m.rho[2], before[2])
    new_ll, fixed_ll, full_ll = m._rho_gradients(torch.from_numpy(g.x).to(dev))
    assert rel_err(new_ll.cpu().numpy(), g.ll[1]) < LL_RTOL          # g^c of the reference's recursion
    assert rel_err(fixed_ll.cpu().numpy(), g.ll[0]) < LL_RTOL         # G^(c-1) with one fixed component = ll_0


def test_update_rho_second_boosting_pass_component_zero(golden_case):
    """component == 0 with all_trained (the second pass over the components): the reference's recursion leaves
    new_ll = fixed_ll = zeros (models/boosted_flow.py:120-122), so the gradient is 0 and rho[0] does not move."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g6_glow_d43_h64_c3_rho")
    m = _model_from_case(g, dev)
    m.component, m.all_trained = 0, True
    m.args.rho_iters, m.args.rho_lr = 12, 0.1
    before = m.rho.clone()
    new_ll, fixed_ll, full_ll = m._rho_gradients(torch.from_numpy(g.x).to(dev))
    assert float(new_ll.abs().max()) == 0.0 and float(fixed_ll.abs().max()) == 0.0
    assert rel_err(full_ll.cpu().numpy(), g.ll[0]) < LL_RTOL
    m.update_rho([(torch.from_numpy(g.x), None)])
    assert torch.equal(m.rho, before)


def test_rho_gradient_helpers_of_the_reference_class(golden_case):
    """models/boosted_flow.py:98-118 (`_rho_gradient_g`, `_rho_gradient_G`: part of the class's surface, their callers are commented
    out upstream): the new component's log-density, and that of ONE fixed component drawn from rho."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g6_glow_d43_h64_c3_rho")
    m = _model_from_case(g, dev)
    x = torch.from_numpy(g.x).to(dev)
    m.component = 2
    g_ll = m._rho_gradient_g(x)
    assert not g_ll.requires_grad and rel_err(g_ll.cpu().numpy(), g.ll[2]) < LL_RTOL
    torch.manual_seed(5)
    seen = set()
    for _ in range(12):           # "1:c-1": a fixed component j < c, drawn by torch.multinomial over rho[0:c]
        G_ll = m._rho_gradient_G(x).cpu().numpy()
        j = int(np.argmin([rel_err(G_ll, g.ll[k]) for k in range(3)]))
        assert j < 2 and rel_err(G_ll, g.ll[j]) < LL_RTOL
        seen.add(j)
    assert seen == {0, 1}
    m.all_trained = True          # "-c": any component but the current one
    m.component = 0
    for _ in range(6):
        G_ll = m._rho_gradient_G(x).cpu().numpy()
        assert min(rel_err(G_ll, g.ll[k]) for k in (1, 2)) < LL_RTOL


def test_boosting_weights_match_reference(golden_case):
    """G8: sample weights for the next component; kernel vs the reference's own statements, and the
    module method on top of the fixed components' mixture density."""
    import os
    import torch
    from conftest import GOLDEN_DIR
    from gbnf_amd import native
    dev = torch.device("cuda:0")
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g8_boosting_weights.npz")))
    for key in ("flat", "peaked", "tiny", "beta"):
        w = native.boosting_weights(torch.from_numpy(data[key + ".G"]).to(dev), float(data[key + ".beta"]))
        np.testing.assert_allclose(w.cpu().numpy(), data[key + ".w"], rtol=2e-5, atol=0)
        w2 = native.boosting_weights(torch.from_numpy(data[key + ".G"]).to(dev), float(data[key + ".beta"]))
        assert torch.equal(w, w2)
    g = golden_case("g3_glow_d43_h215_c8")
    m = _model_from_case(g, dev)
    x = torch.from_numpy(g.x).to(dev)
    with pytest.raises(ValueError):
        m.boosting_weights(x)                     # component 0 trains without weights
    m.component = 3
    w, G = m.boosting_weights(x)
    from oracle import gbnf_oracle as oracle
    ll_ref, G_ref = oracle.mixture_log_prob(g.specs, g.rho, g.x, n_used=3)
    assert rel_err(G.cpu().numpy(), G_ref) < LL_RTOL
    np.testing.assert_allclose(w.cpu().numpy(), oracle.boosting_weights(G_ref), rtol=2e-4, atol=1e-9)
    idx = torch.multinomial(w, x.shape[0], replacement=True)           # the caller's resampling step works on it
    assert idx.shape == (x.shape[0],) and abs(float(w.sum()) - 1.0) < 1e-5


def test_decode_inverts_encode(golden_case):
    """model(z=..., reverse=True) undoes model(x=...) for a fixed component; z=None samples sample_size rows."""
    import torch
    dev = torch.device("cuda:0")
    for name in ("g2_glow_native_d43_h32_c3", "g4_realnvp_d21_h105_c8"):
        g = golden_case(name)
        m = _model_from_case(g, dev)
        x = torch.from_numpy(g.x).to(dev)
        for c in (0, g.cfg["C"] - 1):
            z = m(x=x, components=c)[0]
            xr = m(z=z, temperature=1.0, components=c, reverse=True)
            assert (xr - x).abs().max().item() <= 2e-5 * max(1.0, float(x.abs().max()))
        m.component = g.cfg["C"] - 1
        s = m(z=None, temperature=0.7, components="1:c", reverse=True)
        assert s.shape == (4, g.cfg["d"]) and torch.isfinite(s).all()


def test_reference_checkpoint_evaluates_like_the_reference():
    """g11: load the reference-written checkpoint + the exported indices, evaluate on the device, compare with the outputs
    the reference produced from the very model it saved."""
    import json, os
    import torch
    from conftest import GOLDEN_DIR
    from gbnf_amd import BoostedFlow, checkpoint, synth
    dev = torch.device("cuda:0")
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g11_reference_checkpoint.npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    args = _args(dict(cfg, synth_kw={}), dev)
    m = BoostedFlow(args)
    side = {"component": cfg["component"], "all_trained": cfg["all_trained"],
            "indices": {f"{c}.{k}": torch.from_numpy(data["indices"][c, k]) for c in range(cfg["C"]) for k in range(cfg["K"])},
            "actnorm_inited": {f"{c}.{k}": True for c in range(cfg["C"]) for k in range(cfg["K"])}}
    checkpoint.load(m, None, os.path.join(GOLDEN_DIR, "g11_reference_checkpoint.pt"), args, side_car=side)
    m.eval()
    x = torch.from_numpy(synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"])).to(dev)
    assert rel_err(m.component_log_prob(x).cpu().numpy().T, data["ll"]) < LL_RTOL
    assert rel_err(m.log_prob(x).cpu().numpy(), data["G"]) < LL_RTOL


def test_evaluate_loop_is_served_from_one_launch(golden_case):
    """in eval() mode the first ``model(x=x, components=c)`` of a batch launches every component in use
    (gbnf_mixture_component_forward) and the loop's following calls are answered from that table; the table is dropped when x
    is written in place, when another tensor comes in, and when a parameter of the asked component changes."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g2_glow_native_d43_h32_c3")
    m = _model_from_case(g, dev)
    m.component = g.n_used - 1
    m.eval()
    x = torch.from_numpy(g.x).to(dev)
    m.SERVE_ALL_COMPONENTS = "any"                          # the liberal form (what `with m.serving_loop():` sets); the default: next test
    G = _evaluate_like_reference(m, x)                      # grad mode ON, as in the reference's evaluate()
    assert rel_err(G.cpu().numpy(), g.G) < LL_RTOL
    tab = m.__dict__["_component_table"]
    assert tab.z.shape == (g.n_used, x.shape[0], x.shape[1])
    z1, _, _, l1, _ = m(x=x, components=1)
    assert m.__dict__["_component_table"] is tab            # served: no new table
    assert m(x=x, components=1) is m(x=x, components=1)     # ... by the early look-up of forward(): the batch's ready tuple
    xv = x.view_as(x)                                       # another tensor object on the same storage and version: the full key
    z1v, _, _, l1v, _ = m(x=xv, components=1)
    assert m.__dict__["_component_table"] is tab and torch.equal(z1v, z1) and torch.equal(l1v, l1)
    assert not z1.requires_grad and not l1.requires_grad
    # the table holds what the per-component launches return, bit for bit
    m.SERVE_ALL_COMPONENTS = False
    z1p, _, _, l1p, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = "any"
    assert torch.equal(z1, z1p) and torch.equal(l1, l1p)
    # (1) x written in place: same address, new version -> a new table with the new values
    x.add_(0.125)
    z1b, _, _, l1b, _ = m(x=x, components=1)
    assert m.__dict__["_component_table"] is not tab
    m.SERVE_ALL_COMPONENTS = False
    z1q, _, _, l1q, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = "any"
    assert torch.equal(z1b, z1q) and torch.equal(l1b, l1q) and not torch.equal(l1b, l1)
    # (2) an optimiser step on component 1 between two calls of the same batch
    tab2 = m.__dict__["_component_table"]
    opt = torch.optim.SGD(m.flows[1].parameters(), lr=0.5)
    for p_ in m.flows[1].parameters():
        p_.grad = torch.full_like(p_, 1e-2)
    opt.step()
    z1c, _, _, l1c, _ = m(x=x, components=1)
    assert m.__dict__["_component_table"] is not tab2
    assert not torch.equal(l1c, l1b)
    z0c, _, _, l0c, _ = m(x=x, components=0)                # component 0 did not change: served from the rebuilt table
    m.SERVE_ALL_COMPONENTS = False
    _, _, _, l0p, _ = m(x=x, components=0)
    _, _, _, l1r, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = "any"
    assert torch.equal(l0c, l0p) and torch.equal(l1c, l1r)
    # (2b) a write through .data moves neither the version counter nor the address (PyTorch semantics): invalidate_packed() is the
    #      documented way to make such a write visible to the evaluation handles
    w = next(iter(m.flows[1].parameters()))
    w.data.mul_(1.5)
    _, _, _, l1stale, _ = m(x=x, components=1)
    assert torch.equal(l1stale, l1c)                        # (unseen, as documented)
    m.invalidate_packed()
    _, _, _, l1new, _ = m(x=x, components=1)
    assert not torch.equal(l1new, l1c)
    w.data.div_(1.5)
    m.invalidate_packed()
    # (2c) round 5: the key keeps the permutations' index TENSORS instead of walking `perm.indices` per call.  A re-assigned
    #      permutation (set_indices: a new tensor object, the serial moves) and an in-place edit of the index tensor (its version
    #      moves) must both re-pack component 1 and be served with the new values; component 0 stays untouched
    _, _, _, l1a, _ = m(x=x, components=1)
    perm = m.flows[1].flow.layers[0].permutation
    old_idx = perm.indices.clone()
    new_idx = torch.roll(old_idx, 1)
    perm.set_indices(new_idx.tolist())
    _, _, _, l1b2, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = False
    _, _, _, l1b2p, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = "any"
    assert torch.equal(l1b2, l1b2p) and not torch.equal(l1b2, l1a)
    with torch.no_grad():
        perm.indices.copy_(old_idx.to(perm.indices.device))          # in place: same tensor object, new version
        perm._rebuild_inverse()
    _, _, _, l1back, _ = m(x=x, components=1)
    assert torch.equal(l1back, l1a)
    # (3) train() mode is never served (the call may be recorded by autograd)
    m.train()
    m.drop_component_table()
    m(x=x, components=1)
    assert "_component_table" not in m.__dict__


def test_the_default_cache_serves_one_loop_and_nothing_else(golden_case):
    """VERDICT r5 item 9 / ADVICE r4: by default the table of a batch lives for ONE pass of the evaluate loop -- started by the call
    for component 0, every entry handed out once in the loop's order, dropped with the last one.  A write that bypasses the version
    counter between two loops (x.data.copy_) is therefore seen; repeated and out-of-order calls take the plain path; the views a
    caller gets are never handed out again."""
    import torch
    dev = torch.device("cuda:0")
    g = golden_case("g2_glow_native_d43_h32_c3")
    m = _model_from_case(g, dev)
    m.component = g.n_used - 1
    m.eval()
    assert m.SERVE_ALL_COMPONENTS is True
    x = torch.from_numpy(g.x).to(dev)
    # one loop: one table, gone when the last component has been served
    z0, _, _, l0, _ = m(x=x, components=0)
    tab = m.__dict__["_component_table"]
    assert tab.next_c == 1 and tab.z.shape[0] == g.n_used
    outs = [(z0, l0)]
    for c in range(1, g.n_used):
        zc, _, _, lc, _ = m(x=x, components=c)
        assert zc.data_ptr() == tab.z[c].data_ptr()         # served from the table (a view of it)
        outs.append((zc, lc))
    assert "_component_table" not in m.__dict__
    G = _evaluate_like_reference(m, x)
    assert rel_err(G.cpu().numpy(), g.G) < LL_RTOL and "_component_table" not in m.__dict__
    # the same loop again after a write that moves NO version counter: fresh values
    x.data.copy_(x.data + 0.25)
    l_new = [m(x=x, components=c)[3] for c in range(g.n_used)]
    m.SERVE_ALL_COMPONENTS = False
    l_plain = [m(x=x, components=c)[3] for c in range(g.n_used)]
    m.SERVE_ALL_COMPONENTS = True
    for c in range(g.n_used):
        assert torch.equal(l_new[c], l_plain[c]) and not torch.equal(l_new[c], outs[c][1])
    # `x.data = other`: the same tensor object and version on other storage, in the MIDDLE of a loop -- the early look-up compares the address
    m(x=x, components=0)
    other = (x.data * 0.5).contiguous()
    keep = x.data
    x.data = other
    _, _, _, l1_other, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = False
    _, _, _, l1_ref, _ = m(x=x, components=1)
    m.SERVE_ALL_COMPONENTS = True
    assert torch.equal(l1_other, l1_ref)
    x.data = keep
    # a repeated / out-of-order component is not served from a table (and ends the loop's table)
    m(x=x, components=0)
    assert "_component_table" in m.__dict__
    a = m(x=x, components=2)
    assert "_component_table" not in m.__dict__
    b = m(x=x, components=2)
    assert a[0].data_ptr() != b[0].data_ptr() and torch.equal(a[0], b[0])
    # a served view is the caller's own: editing it changes nothing anybody else will ever get
    z0, _, _, _, _ = m(x=x, components=0)
    z1, _, _, _, _ = m(x=x, components=1)
    z1_before = z1.clone()
    z1.zero_()
    z1_again = m(x=x, components=1)[0]                      # (a repeated call: plain path)
    assert torch.equal(z1_again, z1_before)
    # the scoped liberal form
    with m.serving_loop():
        m(x=x, components=1)
        p = m(x=x, components=1)
        assert m(x=x, components=1) is p and "_component_table" in m.__dict__
    assert "_component_table" not in m.__dict__ and m.SERVE_ALL_COMPONENTS is True


def test_check_numerics_is_the_strict_form(golden_case):
    """BoostedFlow.check_numerics(): raises once any launch since the last check met an operand beyond the fp16 range -- saying how many
    waves of evaluation launches (repaired in the same call: the result is right) and of training launches (saturated) -- and resets."""
    import torch
    from gbnf_amd import BoostedFlow, native
    dev = torch.device("cuda:0")
    g = golden_case("g6_glow_d43_h64_c3_rho")
    m = _model_from_case(g, dev)
    x = torch.from_numpy(g.x).to(dev)
    native.saturation_count(reset=True)
    G = m.log_prob(x)
    BoostedFlow.check_numerics()                        # z-scored data on a sane model: silent
    xbig = x.clone()
    xbig[5] = 2.0e5
    G2 = m.log_prob(xbig)
    assert bool(torch.isfinite(G2[torch.arange(len(x), device=dev) != 5]).all())
    assert rel_err(G2[:5].cpu().numpy(), G[:5].cpu().numpy()) == 0.0          # the other rows are untouched
    with pytest.raises(FloatingPointError, match="0 of training launches"):
        BoostedFlow.check_numerics()
    BoostedFlow.check_numerics()                        # the check reset the counter


def test_the_module_methods_no_other_test_calls(golden_case):
    """`base_dist` (the reference class's property: models/generative_flow.py:38-41), `numerics_status`, `verify_numerics`,
    `native_flow_exact`: each once, on a reference fixture."""
    import torch
    from gbnf_amd import native
    dev = torch.device("cuda:0")
    g = golden_case("g4_realnvp_d21_h105_c8")
    m = _model_from_case(g, dev).to(dev)
    m.component = g.n_used - 1
    x = torch.from_numpy(g.x).to(dev)
    bd = m.base_dist                                  # Normal(base_dist_mean, base_dist_var): registered buffers, as upstream
    assert isinstance(bd, torch.distributions.Normal) and tuple(bd.loc.shape) == (m.z_size,) and bd.loc.is_cuda
    assert float(bd.scale.min()) == 3.0 and "base_dist_mean" in dict(m.named_buffers())
    G = m.log_prob(x)
    assert rel_err(G.cpu().numpy(), g.G) < LL_RTOL
    st = m.numerics_status()
    assert set(st) == {"math_mode", "demoted", "checks", "worst_rel_err", "tolerance"} and st["math_mode"] in ("f16x3", "bf16x6", "f32")
    assert st["demoted"] is False and st["worst_rel_err"] <= st["tolerance"]
    worst = m.verify_numerics(x)                      # f16x3 against bf16x6 on the caller's rows: a sane model stays where it is
    assert 0.0 <= worst < 2.5e-6 and not m.__dict__.get("_math_override")
    exact = m.native_flow_exact(0)
    assert exact.info().math_mode == native.MATH["f32"] and m.native_flow_exact(0) is exact      # cached per parameter state
    z, ldj, ll = exact.forward(x, want_ll=True)
    assert rel_err(ll.cpu().numpy(), g.ll[0]) < LL_RTOL
