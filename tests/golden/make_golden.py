#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running THE REFERENCE itself.

Run in the build container only (needs /root/reference, torch CPU):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference's ``BoostedFlow`` (models/boosted_flow.py), builds models
from a hand-made ``args`` Namespace (the fields its constructors read, SURVEY.md
section 5), installs parameters, runs ``model(x=x, components=c)`` exactly as
``density_experiment.evaluate`` does (density_experiment.py:561-573) and stores
inputs + outputs as small ``.npz`` files.  Only data is committed -- no reference
source or bytecode.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4); these files are the parity pin for ``oracle/`` and, through
it and directly, for the HIP path.

Two kinds of case:
  * "synth" -- parameters come from this repo's portable generator
    (``gbnf_amd.synth``) and are loaded INTO the reference modules; the fixture
    stores only the generator arguments + x + the reference's outputs.
  * "native" -- the reference initialises itself (its own nn.Linear init, ActNorm
    data-dependent init on a batch in train mode, then a small perturbation); the
    fixture stores the exported parameters too (exercises the module->spec export).
"""
import argparse
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gbnf_amd import spec as gspec  # noqa: E402
from gbnf_amd import synth  # noqa: E402
from models.boosted_flow import BoostedFlow as RefBoostedFlow  # noqa: E402  (the reference)
from utils.distributions import log_normal_standard  # noqa: E402  (the reference)


def ref_args(kind, d, h, K, C, depth=1, coupling_network="tanh", coupling="affine",
             permutation="shuffle", batch_norm=True, rho_init="decreasing"):
    return argparse.Namespace(
        num_flows=K, z_size=d, density_evaluation=True, device=torch.device("cpu"), cuda=False,
        component_type=kind, num_components=C, rho_init=rho_init, learn_top=False, y_classes=0,
        y_condition=False, sample_size=4, input_size=[d], h_size=h, num_blocks=1, actnorm_scale=1.0,
        flow_permutation=permutation, flow_coupling=coupling, LU_decomposed=False,
        num_dequant_blocks=0, coupling_network=coupling_network, coupling_network_depth=depth,
        batch_norm=batch_norm)


def _load_net(ref_net, net):
    linears = gspec.linears_of(ref_net)
    assert len(linears) == len(net["layers"])
    for m, (w, b) in zip(linears, net["layers"]):
        assert tuple(m.weight.shape) == w.shape, (m.weight.shape, w.shape)
        m.weight.data.copy_(torch.from_numpy(w))
        m.bias.data.copy_(torch.from_numpy(b))


def install_spec(ref_component, spec):
    """Load a flow spec's numbers into a reference Glow / RealNVPFlow module."""
    if spec["kind"] == "glow":
        for layer, st in zip(ref_component.flow.layers, spec["steps"]):
            layer.actnorm.bias.data.copy_(torch.from_numpy(st["an_bias"]).view(1, -1))
            layer.actnorm.logs.data.copy_(torch.from_numpy(st["an_logs"]).view(1, -1))
            layer.actnorm.inited = True
            perm_mod = layer.shuffle if hasattr(layer, "shuffle") else layer.reverse
            perm_mod.indices = torch.from_numpy(st["perm"]).long()
            for i in range(perm_mod.num_dim):
                perm_mod.indices_inverse[perm_mod.indices[i]] = i
            _load_net(layer.block, st["net"])
    else:
        for k, (mods, st) in enumerate(zip(ref_component.flow_param, spec["steps"])):
            assert (((k + ref_component.flip_init) % 2) > 0) == st["flipped"]
            _load_net(mods[0], st["t_net"])
            _load_net(mods[1], st["s_net"])
            if st["bn"] is None:
                assert mods[2] is None
            else:
                bn = mods[2]
                bn.log_gamma.data.copy_(torch.from_numpy(st["bn"]["log_gamma"]))
                bn.beta.data.copy_(torch.from_numpy(st["bn"]["beta"]))
                bn.running_mean.copy_(torch.from_numpy(st["bn"]["running_mean"]))
                bn.running_var.copy_(torch.from_numpy(st["bn"]["running_var"]))


@torch.no_grad()
def run_reference(model, x, n_used, base="standard", per_step=False):
    """density_experiment.evaluate's exact loop (density_experiment.py:561-573);
    base="toy" uses model.base_dist as toy_experiment.py:413-429 does."""
    model.eval()
    xt = torch.from_numpy(x)
    zs, ldjs, lls = [], [], []
    G = None
    for c in range(n_used):
        z, _, _, ldj, _ = model(x=xt.clone(), components=c)
        if base == "toy":
            ll = model.base_dist.log_prob(z).sum(1) + ldj
        else:
            ll = log_normal_standard(z, reduce=True, dim=-1, device=torch.device("cpu")) + ldj
        if c == 0:
            G = ll
        else:
            rho_simplex = model.rho[0:(c + 1)] / torch.sum(model.rho[0:(c + 1)])
            last_ll = torch.log(1 - rho_simplex[c]) + G
            next_ll = torch.log(rho_simplex[c]) + ll
            G = torch.logsumexp(torch.cat([last_ll.view(-1, 1), next_ll.view(-1, 1)], dim=1), dim=1)
        zs.append(z.numpy().copy())
        ldjs.append(ldj.numpy().copy())
        lls.append(ll.numpy().copy())
    return np.stack(zs), np.stack(ldjs), np.stack(lls), G.numpy().copy()


@torch.no_grad()
def glow_step_trace(component, x):
    """Per-FlowStep (z, logdet) of one reference Glow component (kernel bring-up aid)."""
    z = torch.from_numpy(x).clone()
    ld = torch.zeros(z.shape[0])
    zs, lds = [], []
    for layer in component.flow.layers:
        z, ld = layer(z, ld, reverse=False)
        zs.append(z.numpy().copy())
        lds.append(ld.numpy().copy())
    return np.stack(zs), np.stack(lds)


def synth_case(name, kind, d, h, K, C, N, x_seed=0, w_seed=1, x_scale=1.0, n_used=None,
               rho_init="decreasing", rho_override=None, **kw):
    ref_kw = {}
    synth_kw = {}
    depth = kw.get("depth", 1)
    if kind == "glow":
        synth_kw = dict(depth=depth, act=kw.get("act", "tanh"), coupling=kw.get("coupling", "affine"),
                        permutation=kw.get("permutation", "shuffle"), gain=kw.get("gain", 1.0))
        ref_kw = dict(depth=depth, coupling_network=synth_kw["act"], coupling=synth_kw["coupling"],
                      permutation=synth_kw["permutation"])
    else:
        synth_kw = dict(depth=depth, coupling_network=kw.get("coupling_network", "tanh"),
                        batch_norm=kw.get("batch_norm", True), gain=kw.get("gain", 1.0))
        ref_kw = dict(depth=depth, coupling_network=synth_kw["coupling_network"],
                      batch_norm=synth_kw["batch_norm"])
    torch.manual_seed(1234)
    model = RefBoostedFlow(ref_args(kind, d, h, K, C, rho_init=rho_init, **ref_kw))
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=w_seed, **synth_kw)
    for c in range(C):
        install_spec(model.flows[c], specs[c])
        # round-trip: exporting the reference module must give back the same numbers
        back = gspec.spec_from_component(model.flows[c])
        fa, fb = gspec.flatten_spec(specs[c]), gspec.flatten_spec(back)
        assert fa.keys() == fb.keys()
        for key in fa:
            assert np.array_equal(fa[key], fb[key]), key
    if rho_override is not None:
        model.rho.copy_(torch.tensor(rho_override, dtype=torch.float32))
    x = synth.synth_batch(N, d, seed=x_seed, scale=x_scale)
    n_used = C if n_used is None else n_used
    z, ldj, ll, G = run_reference(model, x, n_used)
    cfg = dict(case="synth", kind=kind, d=d, h=h, K=K, C=C, N=N, x_seed=x_seed, w_seed=w_seed,
               x_scale=x_scale, n_used=n_used, synth_kw=synth_kw)
    out = dict(config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
               rho=model.rho.numpy().copy(), ldj=ldj, ll=ll, G=G)
    # z for every component is bulky at full width: keep all of it for small cases only
    if z.nbytes <= 400_000:
        out["z"] = z
    else:
        out["z_c0"] = z[0]
    if kind == "glow" and h <= 64:
        zs, lds = glow_step_trace(model.flows[0], x)
        out["trace_z_c0"] = zs
        out["trace_ld_c0"] = lds
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: ll[0,:3]={ll[0, :3]}  G[:3]={G[:3]}  finite={np.isfinite(G).all()}")


def specs_case(name, kind, entries, N, x_seed, x_scale=1.0, x_dist="normal", x_clip=None, **ref_kw):
    """Components given one by one as calls of this repo's generator (``entries`` = [(function name, kwargs)]): the
    fixture stores the calls, x's seed and the reference's outputs.  Used for the stress offender found on the GPU
    (tools/find_offender.py): an un-normalised ReLU RealNVP (no BatchNorm), K = 8, h = 500 -- ill-conditioned in f32."""
    specs = [getattr(synth, fn)(**kw) for fn, kw in entries]
    d, K, h = specs[0]["d"], len(specs[0]["steps"]), int(np.asarray(specs[0]["steps"][0]["t_net" if kind == "realnvp" else "net"]["layers"][0][0]).shape[0])
    torch.manual_seed(1234)
    model = RefBoostedFlow(ref_args(kind, d, h, K, len(specs), **ref_kw))
    for c, sp in enumerate(specs):
        install_spec(model.flows[c], sp)
    x = synth.synth_batch(N, d, seed=x_seed, scale=x_scale, dist=x_dist, clip=x_clip)
    z, ldj, ll, G = run_reference(model, x, len(specs))
    cfg = dict(case="synth_specs", kind=kind, d=d, h=h, K=K, C=len(specs), N=N, x_seed=x_seed, x_scale=x_scale,
               x_dist=x_dist, x_clip=x_clip, specs=[dict(fn=fn, kwargs=kw) for fn, kw in entries])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                        rho=model.rho.numpy().copy(), ldj=ldj, ll=ll, G=G, z=z)
    print(f"{name}: ll[:, :3]={ll[:, :3]}  G[:3]={G[:3]}  finite={np.isfinite(G).all()}  max|ll|={np.abs(ll).max():.1f}")


def stress_cases():
    common = dict(d=8, h=500, K=8, coupling_network="relu", batch_norm=False)
    specs_case("g15_stress_realnvp_relu_nobn_d8_h500", "realnvp",
               [("synth_realnvp_spec", dict(common, flip_init=0, seed=7004)),
                ("synth_realnvp_spec", dict(common, flip_init=1, seed=7005))],     # tools/find_offender.py: d=8 h=500 K=8 seed=5 x_scale=2
               N=512, x_seed=5, x_scale=2.0, coupling_network="relu", batch_norm=False)


def native_glow_case(name, d=43, h=32, K=5, C=3, N=256):
    """G2: the reference initialises itself (ActNorm data init + perturbed Linear weights)."""
    torch.manual_seed(7)
    np.random.seed(7)
    model = RefBoostedFlow(ref_args("glow", d, h, K, C))
    x = synth.synth_batch(N, d, seed=11)
    model.train()
    with torch.no_grad():
        for c in range(C):   # ActNorm data-dependent init, density_experiment.py:346-356
            model(x=torch.from_numpy(x).clone(), components=c)
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    model.eval()
    specs = [gspec.spec_from_component(model.flows[c]) for c in range(C)]
    z, ldj, ll, G = run_reference(model, x, C)
    zs, lds = glow_step_trace(model.flows[0], x)
    cfg = dict(case="native", kind="glow", d=d, h=h, K=K, C=C, N=N, x_seed=11)
    out = dict(config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
               rho=model.rho.numpy().copy(), x=x, z=z, ldj=ldj, ll=ll, G=G,
               trace_z_c0=zs, trace_ld_c0=lds)
    for c in range(C):
        out.update(gspec.flatten_spec(specs[c], prefix=f"c{c}."))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: ll[0,:3]={ll[0, :3]}  G[:3]={G[:3]}")


def native_random_case(name, kind, d, h, K, C, N=96):
    """G13: `--coupling_network random` -- the reference draws TanhNet / ReLUNet per step (Glow, glow.py:295-296) or per
    net (RealNVP, realnvp.py:59-60) from numpy's global RNG; it initialises itself, the parameters are perturbed."""
    torch.manual_seed(13)
    np.random.seed(13)
    model = RefBoostedFlow(ref_args(kind, d, h, K, C, coupling_network="random"))
    x = synth.synth_batch(N, d, seed=13)
    model.train()
    with torch.no_grad():
        if kind == "glow":
            for c in range(C):   # ActNorm data-dependent init, density_experiment.py:346-356
                model(x=torch.from_numpy(x).clone(), components=c)
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
        for b_name, b in model.named_buffers():
            if b_name.endswith("running_var"):
                b.mul_(1.0 + 0.3 * torch.rand_like(b))
            elif b_name.endswith("running_mean"):
                b.add_(0.1 * torch.randn_like(b))
    model.eval()
    specs = [gspec.spec_from_component(model.flows[c]) for c in range(C)]
    acts = [[(st["net"]["act"],) if kind == "glow" else (st["t_net"]["act"], st["s_net"]["act"]) for st in sp["steps"]]
            for sp in specs]
    assert any(len({a for st in comp for a in st}) == 2 for comp in acts), "the draw gave no mixed component: change the seed"
    z, ldj, ll, G = run_reference(model, x, C)
    cfg = dict(case="native", kind=kind, d=d, h=h, K=K, C=C, N=N, x_seed=13, activations=acts)
    out = dict(config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
               rho=model.rho.numpy().copy(), x=x, z=z, ldj=ldj, ll=ll, G=G)
    for c in range(C):
        out.update(gspec.flatten_spec(specs[c], prefix=f"c{c}."))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: activations {acts}  ll[0,:3]={ll[0, :3]}  G[:3]={G[:3]}")


def toy_case(name, N=64):
    """G1: 8-Gaussians-shaped toy config: d=2 RealNVP C=2 K=1 h=64, uniform rho,
    base density = model.base_dist = Normal(base_dist_mean, 3.0) (toy_experiment.py:413-429)."""
    torch.manual_seed(3)
    d, h, K, C = 2, 64, 1, 2
    model = RefBoostedFlow(ref_args("realnvp", d, h, K, C, batch_norm=False, rho_init="uniform"))
    specs = synth.synth_boosted_specs("realnvp", C, d, h, K, seed=5, batch_norm=False)
    for c in range(C):
        install_spec(model.flows[c], specs[c])
    x = synth.synth_batch(N, d, seed=2, scale=2.0)
    z, ldj, ll, G = run_reference(model, x, C, base="toy")
    cfg = dict(case="toy", kind="realnvp", d=d, h=h, K=K, C=C, N=N, x_seed=2, w_seed=5, x_scale=2.0,
               n_used=C, synth_kw=dict(batch_norm=False))
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"),
        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), rho=model.rho.numpy().copy(),
        base_mean=model.base_dist_mean.numpy().copy(), base_std=model.base_dist_var.numpy().copy(),
        z=z, ldj=ldj, ll=ll, G=G)
    print(f"{name}: ll[0,:3]={ll[0, :3]}  G[:3]={G[:3]}")


def state_dict_layout_case():
    """Names + shapes of the reference's state_dict for both component kinds (the host mirror must
    expose exactly these so reference checkpoints load and optimizers.py:29-35's name parsing works)."""
    out = {}
    for kind, kw in (("glow", {}), ("realnvp", {}), ("realnvp_mixed", dict(coupling_network="mixed")),
                     ("realnvp_residual", dict(coupling_network="residual", depth=2)),
                     ("glow_depth2_additive", dict(depth=2, coupling="additive", permutation="reverse"))):
        base = kind.split("_")[0]
        m = RefBoostedFlow(ref_args(base, 7, 12, 3, 2, **kw))
        out[kind] = {k: list(v.shape) for k, v in m.state_dict().items()}
        out[kind + "::named_parameters"] = [n for n, _ in m.named_parameters()]
    for kind, kw in (("image_invconv", dict(permutation="invconv", LU=False, learn_top=True)),
                     ("image_lu", dict(permutation="invconv", LU=True, learn_top=True)),
                     ("image_shuffle_additive", dict(permutation="shuffle", LU=False, learn_top=False, coupling="additive",
                                                     depth=2))):
        a = ref_args("glow", 3 * 32 * 32, 8, 2, 2, depth=kw.get("depth", 1), coupling=kw.get("coupling", "affine"),
                     permutation=kw["permutation"])
        a.input_size = [3, 32, 32]; a.num_blocks = 2; a.learn_top = kw["learn_top"]; a.LU_decomposed = kw["LU"]
        m = RefBoostedFlow(a)
        out[kind] = {k: list(v.shape) for k, v in m.state_dict().items()}
        out[kind + "::named_parameters"] = [n for n, _ in m.named_parameters()]
    with open(os.path.join(HERE, "state_dict_layout.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("state_dict_layout.json:", {k: len(v) for k, v in out.items()})


def actnorm_init_case(name, d=43, h=64, K=4, C=2, N=300):
    """G7: ActNorm data-dependent initialisation by the reference itself: synthetic Linear weights and
    permutations installed, ActNorm left un-initialised, one train-mode forward per component under no_grad
    (density_experiment.py:346-356).  Stores the resulting bias/logs of every layer + the eval outputs."""
    torch.manual_seed(5)
    model = RefBoostedFlow(ref_args("glow", d, h, K, C))
    specs = synth.synth_boosted_specs("glow", C, d, h, K, seed=21)
    for c in range(C):
        install_spec(model.flows[c], specs[c])
        for layer in model.flows[c].flow.layers:
            layer.actnorm.inited = False
            layer.actnorm.bias.data.zero_()
            layer.actnorm.logs.data.zero_()
    x = synth.synth_batch(N, d, seed=13, scale=1.7) + 0.4
    model.train()
    with torch.no_grad():
        for c in range(C):
            model(x=torch.from_numpy(x).clone(), components=c)
    bias = np.stack([np.stack([l.actnorm.bias.detach().numpy().reshape(-1) for l in f.flow.layers]) for f in model.flows])
    logs = np.stack([np.stack([l.actnorm.logs.detach().numpy().reshape(-1) for l in f.flow.layers]) for f in model.flows])
    assert all(l.actnorm.inited for f in model.flows for l in f.flow.layers)
    z, ldj, ll, G = run_reference(model, x, C)
    cfg = dict(case="actnorm_init", kind="glow", d=d, h=h, K=K, C=C, N=N, w_seed=21, x_seed=13, x_scale=1.7, x_shift=0.4,
               synth_kw=dict())
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                        rho=model.rho.numpy().copy(), an_bias=bias, an_logs=logs, ldj=ldj, ll=ll, G=G)
    print(f"{name}: logs[0,0,:3]={logs[0, 0, :3]} ll[0,:3]={ll[0, :3]}")


def boosting_weights_case(name):
    """G8: the sample weights of compute_kl_pq_loss (density_experiment.py:624-640), produced with the reference's
    own ``utils.utilities.softmax`` and the very statements of that function, for several G shapes: a peaked one
    (max weight > 0.1 -> clamp branch), a flat one (no clamp), and beta != 1."""
    from utils.utilities import softmax as ref_softmax
    rng = np.random.RandomState(3)
    out = {}
    cases = {"flat": (-60 + 0.5 * rng.standard_normal(4096)).astype(np.float32),
             "peaked": (-60 + 6.0 * rng.standard_normal(512)).astype(np.float32),
             "tiny": np.array([-3.0, -2.5], dtype=np.float32),
             "beta": (-40 + 2.0 * rng.standard_normal(1000)).astype(np.float32)}
    for key, G_ll in cases.items():
        beta = 0.5 if key == "beta" else 1.0
        G_nll = -1.0 * torch.from_numpy(G_ll)
        weights = ref_softmax(G_nll)
        weights = torch.pow(weights, beta)
        if weights.max() > 0.1:
            weights = torch.max(torch.min(weights, torch.tensor([0.1])), torch.tensor([0.01]))
        if weights.sum() != 1.0:
            weights = weights / torch.sum(weights)
        out[key + ".G"] = G_ll
        out[key + ".w"] = weights.numpy().copy()
        out[key + ".beta"] = np.float32(beta)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: float(v.max()) for k, v in out.items() if k.endswith(".w")})


def decode_case(name, d=43, h=64, K=5, C=2, N=96):
    """G9: the z -> x direction, by the reference itself.  On tabular data only ``Glow.decode`` with ADDITIVE coupling
    runs in the reference: the affine branch sums its log-scale over image dims (models/glow.py:357 -> IndexError on
    2-D input), ``RealNVP.inverse`` conditions on the wrong half (models/transformations.py:581-599: it is not the
    inverse of ``forward``), and ``BoostedFlow.decode`` passes a misspelt keyword (models/boosted_flow.py:216).  So the
    fixture calls ``model.flows[c].decode(z, None, None)`` (models/glow.py:112-123) on additive components; the other
    kinds are pinned by the round trip inverse(forward(x)) == x in the tests."""
    specs = synth.synth_boosted_specs("glow", C, d, h, K, seed=31, coupling="additive")
    model = RefBoostedFlow(ref_args("glow", d, h, K, C, coupling="additive")).eval()
    for c in range(C):
        install_spec(model.flows[c], specs[c])
    z = synth.synth_batch(N, d, seed=17, scale=1.3)
    xs = []
    with torch.no_grad():
        for c in range(C):
            xs.append(model.flows[c].decode(torch.from_numpy(z).clone(), None, None).numpy().copy())
    cfg = dict(case="decode", kind="glow", d=d, h=h, K=K, C=C, N=N, w_seed=31, z_seed=17, z_scale=1.3,
               synth_kw=dict(coupling="additive"))
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8), x=np.stack(xs))
    print(f"{name}: x[0,0,:3]={xs[0][0, :3]}")


def _ref_params_in_flat_order(component, kind):
    """Reference parameters of one component in the order of the library's flat gradient buffer (include/gbnf.h)."""
    out = []
    if kind == "glow":
        for layer in component.flow.layers:
            out += [layer.actnorm.bias, layer.actnorm.logs]
            for m in layer.block.network:
                if isinstance(m, torch.nn.Linear):
                    out += [m.weight, m.bias]
    else:
        for mods in component.flow_param:
            bn = mods[2] if len(mods) > 2 else None
            out += [None, None] if bn is None else [bn.log_gamma, bn.beta]
            for net in (mods[0], mods[1]):
                for m in gspec.linears_of(net):
                    out += [m.weight, m.bias]
    return out


def grads_train_bn_case(name, d=21, h=32, K=4, N=80):
    """G10b: RealNVP in TRAIN mode (the reference's default batch_norm=True while training): BatchNorm normalises with the
    batch statistics, the log-det uses them, nll.backward() differentiates through them, and the running statistics are
    updated with momentum 0.9 (models/layers.py:338-358).  Stores nll, the gradients and the updated running statistics."""
    specs = synth.synth_boosted_specs("realnvp", 1, d, h, K, seed=43)
    model = RefBoostedFlow(ref_args("realnvp", d, h, K, 1))
    install_spec(model.flows[0], specs[0])
    model.train()
    x = torch.from_numpy(synth.synth_batch(N, d, seed=27, scale=1.2)).requires_grad_(True)
    z, _, _, ldj, _ = model(x=x, components=0)
    nll = torch.mean(-1.0 * (log_normal_standard(z, reduce=True, dim=-1) + ldj))
    nll.backward()
    flat = []
    for p in _ref_params_in_flat_order(model.flows[0], "realnvp"):
        flat.append(np.zeros(d, dtype=np.float32) if p is None else p.grad.detach().numpy().reshape(-1).astype(np.float32))
    bns = [mods[2] for mods in model.flows[0].flow_param if len(mods) > 2 and mods[2] is not None]
    cfg = dict(case="grads_train_bn", kind="realnvp", d=d, h=h, K=K, C=1, N=N, w_seed=43, x_seed=27, x_scale=1.2, synth_kw={})
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                        nll=np.float32(nll.item()), grads=np.concatenate(flat), g_x=x.grad.numpy().copy(),
                        z=z.detach().numpy().copy(), ldj=ldj.detach().numpy().copy(),
                        running_mean=np.stack([b.running_mean.numpy() for b in bns]),
                        running_var=np.stack([b.running_var.numpy() for b in bns]))
    print(f"{name}: nll={nll.item():.5f} |grads|={np.abs(np.concatenate(flat)).max():.4f}")


def grads_case(name, kind, d, h, K, N, **synth_kw):
    """G10: the training step's gradients by the reference itself: nll = mean(-(log_normal_standard(z) + ldj)) of ONE
    component (density_experiment.py:655-659, the non-boosted / first-component branch of compute_kl_pq_loss), then
    nll.backward() (density_experiment.py:366-368).  The model is in eval() mode so that RealNVP's BatchNorm uses its
    running statistics (the form the build differentiates); ActNorm is initialised.  Stores nll and every parameter
    gradient, concatenated in the order of the library's flat gradient buffer, plus d nll / d x."""
    C = 1
    specs = synth.synth_boosted_specs(kind, C, d, h, K, seed=41, **synth_kw)
    ref_kw = dict(depth=synth_kw.get("depth", 1), coupling=synth_kw.get("coupling", "affine"),
                  batch_norm=synth_kw.get("batch_norm", True),
                  coupling_network=synth_kw.get("act", synth_kw.get("coupling_network", "tanh")))
    model = RefBoostedFlow(ref_args(kind, d, h, K, C, **ref_kw)).eval()
    install_spec(model.flows[0], specs[0])
    x = torch.from_numpy(synth.synth_batch(N, d, seed=23, scale=1.2)).requires_grad_(True)
    z, _, _, ldj, _ = model(x=x, components=0)
    g_nll = -1.0 * (log_normal_standard(z, reduce=True, dim=-1) + ldj)
    nll = torch.mean(g_nll)
    nll.backward()
    flat = []
    for p in _ref_params_in_flat_order(model.flows[0], kind):
        if p is None:
            flat.append(np.zeros(d, dtype=np.float32))
        else:
            flat.append(p.grad.detach().numpy().reshape(-1).astype(np.float32))
    cfg = dict(case="grads", kind=kind, d=d, h=h, K=K, C=C, N=N, w_seed=41, x_seed=23, x_scale=1.2, synth_kw=synth_kw)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        config=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8),
                        nll=np.float32(nll.item()), grads=np.concatenate(flat), g_x=x.grad.numpy().copy())
    print(f"{name}: nll={nll.item():.5f} |grads|={np.abs(np.concatenate(flat)).max():.4f} n={sum(f.size for f in flat)}")


def checkpoint_case(name, d=6, h=8, K=2, C=2, N=16):
    """G11: a checkpoint FILE written by the reference's own utils.utilities.save (utils/utilities.py:78-93) from a
    reference model + Adam optimiser, together with what the file does not hold (the permutation indices of the live
    model) and the model's outputs on a batch.  The .pt file is data: tensors and python scalars."""
    from utils.utilities import save as ref_save
    torch.manual_seed(3)
    model = RefBoostedFlow(ref_args("glow", d, h, K, C))
    specs = synth.synth_boosted_specs("glow", C, d, h, K, seed=51)
    for c in range(C):
        install_spec(model.flows[c], specs[c])
    model.component = 1
    model.all_trained = False
    model.eval()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ref_save(model, opt, os.path.join(HERE, name + ".pt"))
    x = synth.synth_batch(N, d, seed=29)
    z, ldj, ll, G = run_reference(model, x, C)
    np.savez_compressed(os.path.join(HERE, name + ".npz"),
                        config=np.frombuffer(json.dumps(dict(case="checkpoint", kind="glow", d=d, h=h, K=K, C=C, N=N, x_seed=29,
                                                             component=1, all_trained=False)).encode(), dtype=np.uint8),
                        indices=np.stack([np.stack([l.shuffle.indices.numpy() for l in f.flow.layers]) for f in model.flows]),
                        ll=ll, G=G)
    print(f"{name}: ll[0,:3]={ll[0, :3]} file={os.path.getsize(os.path.join(HERE, name + '.pt'))} bytes")


def _install_conv(ref_conv, c):
    """Conv2d (+ActNorm2d) / Conv2dZeros of the reference <- conv dict of the image spec."""
    ref_conv.conv.weight.data.copy_(torch.from_numpy(c["w"]))
    if c["b"] is not None:
        ref_conv.conv.bias.data.copy_(torch.from_numpy(c["b"]))
    if c["an_bias"] is not None:
        ref_conv.actnorm.bias.data.copy_(torch.from_numpy(c["an_bias"]).view(1, -1, 1, 1))
        ref_conv.actnorm.logs.data.copy_(torch.from_numpy(c["an_logs"]).view(1, -1, 1, 1))
        ref_conv.actnorm.inited = True
    if c["logs"] is not None:
        ref_conv.logs.data.copy_(torch.from_numpy(c["logs"]).view(-1, 1, 1))


def install_image_spec(ref_glow, spec, keep_invconv=False):
    from models.glow import FlowStep
    from models.layers import Split2d
    layers = list(ref_glow.flow.layers)
    steps = [st for lvl in spec["levels"] for st in lvl["steps"]]
    splits = [lvl["split"] for lvl in spec["levels"] if lvl["split"] is not None]
    si = pi = 0
    for layer in layers:
        if isinstance(layer, FlowStep):
            st = steps[si]; si += 1
            layer.actnorm.bias.data.copy_(torch.from_numpy(st["an_bias"]).view(1, -1, 1, 1))
            layer.actnorm.logs.data.copy_(torch.from_numpy(st["an_logs"]).view(1, -1, 1, 1))
            layer.actnorm.inited = True
            if hasattr(layer, "invconv"):
                if keep_invconv:     # LU parameterisation: keep the reference's own factors, export the composed matrix
                    w, _ = layer.invconv.get_weight(torch.zeros(1, layer.invconv.w_shape[0], 1, 1), False)
                    st["perm_w"] = w.detach().view(*layer.invconv.w_shape).numpy().copy()
                else:
                    layer.invconv.weight.data.copy_(torch.from_numpy(st["perm_w"]))
            else:
                pm = layer.shuffle if hasattr(layer, "shuffle") else layer.reverse
                pm.indices = torch.from_numpy(st["perm"]).long()
                for i in range(pm.num_dim):
                    pm.indices_inverse[pm.indices[i]] = i
            convs = [m for m in layer.block.network if not isinstance(m, torch.nn.ReLU)]
            assert len(convs) == len(st["convs"])
            for m, c in zip(convs, st["convs"]):
                _install_conv(m, c)
        elif isinstance(layer, Split2d):
            _install_conv(layer.conv, splits[pi]); pi += 1
    if spec["learn_top"] is not None:
        _install_conv(ref_glow.learn_top_fn, spec["learn_top"])


def image_case(name, h=32, K=2, L=2, C=2, N=4, depth=1, coupling="affine", permutation="invconv", learn_top=True,
               LU=False, trained_like=False, input_size=(3, 32, 32)):
    """G12: the image path (BASELINE.json configs[3] at toy size) by the reference itself:
    BoostedFlow with input_size (3,32,32) -> Glow.encode (dequantise with the fixture's noise injected through
    Tensor.uniform_, to_logits, squeeze / FlowStep / Split2d levels, learned top prior), then
    ll_c = log_normal_diag(z, z_mu, z_var) + logdet (image_experiment.py:227) and the boosted recursion over c."""
    from utils.distributions import log_normal_diag
    input_size = tuple(input_size)      # g19 (round 4): the reference's 1 x 28 x 28 and 1 x 28 x 20 loaders (utils/load_data.py:389-529)
    a = ref_args("glow", int(np.prod(input_size)), h, K, C, depth=depth, coupling=coupling, permutation=permutation)
    a.input_size = list(input_size); a.num_blocks = L; a.learn_top = learn_top; a.LU_decomposed = LU
    torch.manual_seed(7)
    model = RefBoostedFlow(a).eval()
    specs = [synth.synth_image_glow_spec(input_size, h, K, L, depth=depth, coupling=coupling, permutation=permutation,
                                         learn_top=learn_top, seed=61 + c, trained_like=trained_like) for c in range(C)]
    for c in range(C):
        install_image_spec(model.flows[c], specs[c], keep_invconv=LU)
    x, noise = synth.synth_image_batch(N, input_size, seed=31)
    orig = torch.Tensor.uniform_
    noise_t = torch.from_numpy(noise)

    def injected(self, a=0.0, b=1.0):
        self.copy_(noise_t)
        return self
    lls, zs, ldjs = [], [], []
    G = None
    try:
        torch.Tensor.uniform_ = injected
        with torch.no_grad():
            for c in range(C):
                z, mu, var, ldj, _ = model(x=torch.from_numpy(x).clone(), components=c)
                ll = log_normal_diag(z, mu, var, dim=[1, 2, 3]) + ldj
                zs.append(z.numpy().copy()); ldjs.append(ldj.numpy().copy()); lls.append(ll.numpy().copy())
                if c == 0:
                    G = ll
                else:
                    r = model.rho[0:(c + 1)] / torch.sum(model.rho[0:(c + 1)])
                    G = torch.logsumexp(torch.stack([torch.log(1 - r[c]) + G, torch.log(r[c]) + ll], dim=1), dim=1)
    finally:
        torch.Tensor.uniform_ = orig
    out = dict(config=np.frombuffer(json.dumps(dict(case="image", h=h, K=K, L=L, C=C, N=N, depth=depth, coupling=coupling,
                                                    permutation=permutation, learn_top=learn_top, LU=LU, w_seed=61,
                                                    x_seed=31, trained_like=trained_like, input_size=list(input_size))).encode(),
                                         dtype=np.uint8),
               rho=model.rho.numpy().copy(), z=np.stack(zs), ldj=np.stack(ldjs), ll=np.stack(lls), G=G.numpy().copy())
    if LU:   # the composed invconv matrices of the reference's own LU factors (not reproducible from the generator)
        for c in range(C):
            k = 0
            for lvl in specs[c]["levels"]:
                for st in lvl["steps"]:
                    out[f"c{c}.perm_w.{k}"] = st["perm_w"]; k += 1
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: ll[:, :2]={np.stack(lls)[:, :2]} bpd={-np.stack(lls).mean() / (np.log(2) * np.prod(input_size)):.4f}")


def image_actnorm_init_case(name, h=32, K=2, L=2, N=6, input_size=(3, 32, 32)):
    """G20 (round 4): the data-dependent ActNorm2d initialisation of an image Glow by the reference itself -- a freshly
    constructed BoostedFlow (the reference's own parameter initialisation), ONE forward in training mode on a batch
    (models/layers.py:473-486 runs inside every un-initialised ActNorm2d), the dequantisation noise injected.  Saved: every
    tensor of component 0's state_dict BEFORE the call, x, noise, and bias / logs of every ActNorm2d AFTER it, in module order."""
    input_size = tuple(input_size)
    a = ref_args("glow", int(np.prod(input_size)), h, K, 1, permutation="invconv")
    a.input_size = list(input_size); a.num_blocks = L; a.learn_top = True; a.LU_decomposed = False
    torch.manual_seed(11)
    model = RefBoostedFlow(a)
    model.train()
    before = {k: v.detach().clone().numpy() for k, v in model.flows[0].state_dict().items()}
    x, noise = synth.synth_image_batch(N, input_size, seed=37)
    orig = torch.Tensor.uniform_
    noise_t = torch.from_numpy(noise)

    def injected(self, a=0.0, b=1.0):
        self.copy_(noise_t)
        return self
    try:
        torch.Tensor.uniform_ = injected
        with torch.no_grad():
            z, mu, var, ldj, _ = model(x=torch.from_numpy(x).clone(), components=0)
    finally:
        torch.Tensor.uniform_ = orig
    from models.layers import ActNorm2d as RefActNorm2d
    acts = [m for m in model.flows[0].modules() if isinstance(m, RefActNorm2d)]
    assert all(m.inited for m in acts)
    out = dict(config=np.frombuffer(json.dumps(dict(case="image_actnorm_init", h=h, K=K, L=L, N=N, input_size=list(input_size),
                                                    x_seed=37, n_actnorm=len(acts))).encode(), dtype=np.uint8),
               ldj=ldj.numpy().copy(), z=z.numpy().copy())
    for k, v in before.items():
        out["before." + k] = v
    for i, m in enumerate(acts):
        out[f"after.bias.{i}"] = m.bias.detach().numpy().reshape(-1).copy()
        out[f"after.logs.{i}"] = m.logs.detach().numpy().reshape(-1).copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: {len(acts)} ActNorm2d layers, logs[0][:3]={out['after.logs.0'][:3]}, ldj[:2]={ldj.numpy()[:2]}")


def image_decode_case(name, h=32, K=2, L=2, N=3, depth=1, coupling="affine", permutation="invconv", LU=False,
                      temperature=0.7):
    """G16: the image z -> x direction by the reference itself: Glow.decode(z, None, temperature) (models/glow.py:112-123;
    FlowNet.decode :254-260, FlowStep.decode :344-366, Split2d reverse models/layers.py:695-699, to_logits reverse
    models/glow.py:151-158).  The draws of Split2d's torch.normal are injected (mean + std * eps with the fixture's eps),
    so the result is a function of (z, eps, temperature)."""
    input_size = (3, 32, 32)
    a = ref_args("glow", 3 * 32 * 32, h, K, 1, depth=depth, coupling=coupling, permutation=permutation)
    a.input_size = list(input_size); a.num_blocks = L; a.learn_top = True; a.LU_decomposed = LU
    torch.manual_seed(11)
    model = RefBoostedFlow(a).eval()
    spec = synth.synth_image_glow_spec(input_size, h, K, L, depth=depth, coupling=coupling, permutation=permutation,
                                       learn_top=True, seed=71)
    install_image_spec(model.flows[0], spec, keep_invconv=LU)
    rng = np.random.RandomState(17)
    shapes = []                 # (C/2, H, W) of the half each Split2d drops, first level first
    zc, zh, zw = input_size
    for l in range(L):
        zc, zh, zw = zc * 4, zh // 2, zw // 2
        if l < L - 1:
            shapes.append((zc // 2, zh, zw))
            zc //= 2
    z = (0.8 * rng.randn(N, zc, zh, zw)).astype(np.float32)
    eps = [rng.randn(N, *sh).astype(np.float32) for sh in shapes]
    queue = [torch.from_numpy(e) for e in eps]          # FlowNet.decode meets the deepest Split2d first
    orig = torch.normal

    def injected(mean, std, *args, **kw):
        e = queue.pop()
        assert tuple(e.shape) == tuple(mean.shape)
        return mean + std * e
    try:
        torch.normal = injected
        with torch.no_grad():
            x = model.flows[0].decode(torch.from_numpy(z).clone(), None, temperature)
    finally:
        torch.normal = orig
    assert not queue
    out = dict(config=np.frombuffer(json.dumps(dict(case="image_decode", h=h, K=K, L=L, N=N, depth=depth, coupling=coupling,
                                                    permutation=permutation, LU=LU, w_seed=71,
                                                    temperature=temperature)).encode(), dtype=np.uint8),
               z=z, x=x.numpy().copy())
    for l, e in enumerate(eps):
        out[f"eps.{l}"] = e
    if LU:
        k = 0
        for lvl in spec["levels"]:
            for st in lvl["steps"]:
                out[f"c0.perm_w.{k}"] = st["perm_w"]; k += 1
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(f"{name}: x[0,0,0,:4]={x.numpy()[0, 0, 0, :4]} range [{float(x.min()):.3f}, {float(x.max()):.3f}]")


def image_decode_cases():
    image_decode_case("g16_image_decode_invconv_affine")
    image_decode_case("g16_image_decode_shuffle_additive", coupling="additive", permutation="shuffle", depth=2, temperature=1.0)
    image_decode_case("g16_image_decode_lu", LU=True, K=1, h=16, temperature=0.5)


def trained_like_cases():
    """magnitudes of TRAINED flows (last-layer rows of 1e-3 .. 1e-4, ActNorm log-scales of +-3, BatchNorm
    variances over four decades) on heavy-tailed inputs (Student t, 3 d.o.f., |x| <= 50): the reference's own outputs."""
    t = dict(x_dist="student_t3", x_clip=50.0)
    specs_case("g17_trained_like_glow_d43_h215_c4", "glow",
               [("synth_glow_spec", dict(d=43, h=215, K=5, seed=9100 + c, trained_like=True)) for c in range(4)],
               N=256, x_seed=31, **t)
    specs_case("g17_trained_like_glow_relu_d21_h105_c2", "glow",
               [("synth_glow_spec", dict(d=21, h=105, K=5, act="relu", seed=9200 + c, trained_like=True)) for c in range(2)],
               N=192, x_seed=32, coupling_network="relu", **t)
    specs_case("g17_trained_like_realnvp_d21_h105_c4", "realnvp",
               [("synth_realnvp_spec", dict(d=21, h=105, K=5, flip_init=c, seed=9300 + c, trained_like=True)) for c in range(4)],
               N=256, x_seed=33, **t)
    specs_case("g17_heavy_tails_glow_d43_h215_c2", "glow",        # init-like weights, only the inputs are harsh
               [("synth_glow_spec", dict(d=43, h=215, K=5, seed=9400 + c)) for c in range(2)], N=256, x_seed=34, **t)


def main():
    torch.set_num_threads(4)
    if "--trained-only" in sys.argv:
        trained_like_cases()
        return
    if "--image-decode-only" in sys.argv:
        image_decode_cases()
        return
    if "--stress-only" in sys.argv:
        stress_cases()
        return
    if "--image-trained-only" in sys.argv:
        # g18 (round 4): trained-like magnitudes on the image path -- ActNorm2d logs +-3, coupling-net ActNorm2d logs +-1.5,
        # Conv2dZeros logs +-0.5 (gains exp(3 logs) up to e^1.5) -- at the full hidden width of BASELINE.json configs[3]
        image_case("g18_image_glow_trained_like_h256", h=256, K=2, L=2, C=2, N=4, trained_like=True)
        return
    if "--image-actnorm-init-only" in sys.argv:
        image_actnorm_init_case("g20_image_actnorm_init_3x32x32")
        image_actnorm_init_case("g20_image_actnorm_init_1x28x28", h=16, K=2, L=2, N=5, input_size=(1, 28, 28))
        return
    if "--image-small-only" in sys.argv:
        # g19 (round 4): the reference's other image shapes -- 1 x 28 x 28 (MNIST, Omniglot, Caltech) and 1 x 28 x 20 (Frey faces)
        image_case("g19_image_glow_1x28x28", h=64, K=2, L=2, C=2, N=4, input_size=(1, 28, 28))
        image_case("g19_image_glow_1x28x20_additive", h=32, K=3, L=2, C=2, N=4, input_size=(1, 28, 20), coupling="additive",
                   permutation="shuffle", learn_top=False)
        image_case("g19_image_glow_1x28x28_one_level_h256", h=256, K=2, L=1, C=2, N=3, input_size=(1, 28, 28))
        return
    if "--image-only" in sys.argv:
        image_case("g12_image_glow_invconv_affine")
        image_case("g12_image_glow_shuffle_additive", coupling="additive", permutation="shuffle", learn_top=False, depth=2)
        image_case("g12_image_glow_lu", LU=True, K=1, h=16)
        return
    if "--checkpoint-only" in sys.argv:
        checkpoint_case("g11_reference_checkpoint")
        return
    if "--train-bn-only" in sys.argv:
        grads_train_bn_case("g10_realnvp_grads_train_bn_d21_h32")
        return
    if "--grads-only" in sys.argv:
        grads_case("g10_glow_grads_d43_h64", "glow", 43, 64, 3, 96)
        grads_case("g10_glow_grads_additive_relu_d8", "glow", 8, 40, 3, 50, coupling="additive", act="relu")
        grads_case("g10_realnvp_grads_d21_h32", "realnvp", 21, 32, 4, 80)
        grads_case("g10_realnvp_residual_grads_d21_h32", "realnvp", 21, 32, 3, 80, coupling_network="residual")
        return
    if "--residual-only" in sys.argv:
        synth_case("g14_realnvp_residual_d21_h64_c2", "realnvp", 21, 64, 4, 2, 128, coupling_network="residual")
        synth_case("g14_realnvp_residual2_d8_h40_c2", "realnvp", 8, 40, 3, 2, 96, coupling_network="residual", depth=2,
                   batch_norm=False)
        return
    if "--random-only" in sys.argv:
        native_random_case("g13_glow_random_d43_h64", "glow", 43, 64, 6, 2)
        native_random_case("g13_realnvp_random_d21_h32", "realnvp", 21, 32, 5, 2)
        return
    if "--decode-only" in sys.argv:
        decode_case("g9_glow_additive_decode")
        return
    if "--boosting-only" in sys.argv:
        boosting_weights_case("g8_boosting_weights")
        return
    if "--actnorm-only" in sys.argv:
        actnorm_init_case("g7_glow_actnorm_data_init")
        return
    if "--layout-only" in sys.argv:
        state_dict_layout_case()
        return
    state_dict_layout_case()
    actnorm_init_case("g7_glow_actnorm_data_init")
    boosting_weights_case("g8_boosting_weights")
    decode_case("g9_glow_additive_decode")
    checkpoint_case("g11_reference_checkpoint")
    image_case("g12_image_glow_invconv_affine")
    image_case("g12_image_glow_shuffle_additive", coupling="additive", permutation="shuffle", learn_top=False, depth=2)
    image_case("g12_image_glow_lu", LU=True, K=1, h=16)
    image_decode_cases()
    grads_case("g10_glow_grads_d43_h64", "glow", 43, 64, 3, 96)
    grads_case("g10_glow_grads_additive_relu_d8", "glow", 8, 40, 3, 50, coupling="additive", act="relu")
    grads_case("g10_realnvp_grads_d21_h32", "realnvp", 21, 32, 4, 80)
    grads_case("g10_realnvp_residual_grads_d21_h32", "realnvp", 21, 32, 3, 80, coupling_network="residual")
    grads_train_bn_case("g10_realnvp_grads_train_bn_d21_h32")
    toy_case("g1_toy_realnvp_c2")
    native_glow_case("g2_glow_native_d43_h32_c3")
    native_random_case("g13_glow_random_d43_h64", "glow", 43, 64, 6, 2)
    native_random_case("g13_realnvp_random_d21_h32", "realnvp", 21, 32, 5, 2)
    # G3: MINIBOONE full width (BASELINE.json metric config), synthetic weights
    synth_case("g3_glow_d43_h215_c8", "glow", 43, 215, 5, 8, 256)
    # G4: HEPMASS RealNVP, flip_init 0..7, BN with non-trivial running stats
    synth_case("g4_realnvp_d21_h105_c8", "realnvp", 21, 105, 5, 8, 256)
    synth_case("g4_realnvp_d21_h105_c2_mixed", "realnvp", 21, 105, 5, 2, 128, coupling_network="mixed")
    # G14: ResidualNet coupling networks (RealNVP only: models/realnvp.py:57, models/layers.py:246-301)
    synth_case("g14_realnvp_residual_d21_h64_c2", "realnvp", 21, 64, 4, 2, 128, coupling_network="residual")
    synth_case("g14_realnvp_residual2_d8_h40_c2", "realnvp", 8, 40, 3, 2, 96, coupling_network="residual", depth=2,
               batch_norm=False)
    synth_case("g4_realnvp_d21_h105_c2_relu_nobn", "realnvp", 21, 105, 3, 2, 128,
               coupling_network="relu", batch_norm=False)
    # G5: variants
    synth_case("g5_glow_d43_h64_c2_additive", "glow", 43, 64, 5, 2, 128, coupling="additive")
    synth_case("g5_glow_d43_h64_c2_reverse_relu", "glow", 43, 64, 5, 2, 128, permutation="reverse", act="relu")
    synth_case("g5_glow_d43_h64_c2_depth2", "glow", 43, 64, 3, 2, 128, depth=2)
    synth_case("g5_glow_d43_h64_c2_depth0", "glow", 43, 64, 3, 2, 128, depth=0)
    synth_case("g5_glow_d6_h30_c2", "glow", 6, 30, 5, 2, 128)        # POWER
    synth_case("g5_glow_d8_h40_c2", "glow", 8, 40, 5, 2, 128)        # GAS
    synth_case("g5_glow_d21_h105_c2", "glow", 21, 105, 5, 2, 128)    # HEPMASS with Glow
    synth_case("g5_glow_d63_h128_c2", "glow", 63, 128, 3, 2, 128)    # BSDS300 width class
    synth_case("g5_realnvp_d43_h215_c2", "realnvp", 43, 215, 5, 2, 128)
    synth_case("g5_realnvp_d6_h30_c3", "realnvp", 6, 30, 5, 3, 128)
    # G6: edge cases
    synth_case("g6_glow_d43_h64_n1", "glow", 43, 64, 5, 2, 1)
    synth_case("g6_glow_d43_h64_n77", "glow", 43, 64, 5, 2, 77)      # not a tile multiple
    synth_case("g6_glow_d43_h64_bigx", "glow", 43, 64, 5, 2, 96, x_scale=6.0, gain=3.0)  # saturating tanh/sigmoid
    synth_case("g6_glow_d43_h64_c4_used2", "glow", 43, 64, 5, 4, 64, n_used=2)           # loaded < C
    synth_case("g6_realnvp_d21_h64_n33", "realnvp", 21, 64, 5, 3, 33)
    synth_case("g6_glow_d43_h64_c3_rho", "glow", 43, 64, 5, 3, 64, rho_override=[0.7, 3.0, 0.01])
    stress_cases()
    trained_like_cases()


if __name__ == "__main__":
    main()
