import glob
import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN_DIR = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_names():
    """Forward/mixture fixtures (g1..g6).  g7 (ActNorm data-dependent init), g8 (boosting weights) and g9 (decode)
    have their own tests."""
    names = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in names if not n.startswith(("g7_", "g8_", "g9_", "g10_", "g11_", "g12_", "g16_", "g18_", "g19_", "g20_"))]


IMAGE_CASES = ("g12_image_glow_invconv_affine", "g12_image_glow_shuffle_additive", "g12_image_glow_lu",
               "g18_image_glow_trained_like_h256", "g19_image_glow_1x28x28", "g19_image_glow_1x28x20_additive",
               "g19_image_glow_1x28x28_one_level_h256")


def load_image_case(name):
    """g12: (cfg, specs, x, noise, data) -- the reference's image Glow outputs."""
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    size = tuple(cfg.get("input_size", (3, 32, 32)))
    specs = [synth.synth_image_glow_spec(size, cfg["h"], cfg["K"], cfg["L"], depth=cfg["depth"], coupling=cfg["coupling"],
                                         permutation=cfg["permutation"], learn_top=cfg["learn_top"], seed=cfg["w_seed"] + c,
                                         trained_like=cfg.get("trained_like", False))
             for c in range(cfg["C"])]
    if cfg["LU"]:
        for c, sp in enumerate(specs):
            k = 0
            for lvl in sp["levels"]:
                for st in lvl["steps"]:
                    st["perm_w"] = data[f"c{c}.perm_w.{k}"]; k += 1
    x, noise = synth.synth_image_batch(cfg["N"], size, seed=cfg["x_seed"])
    return cfg, specs, x, noise, data


IMAGE_ACTNORM_INIT_CASES = ("g20_image_actnorm_init_3x32x32", "g20_image_actnorm_init_1x28x28")


def load_image_actnorm_init_case(name, device="cpu"):
    """g20: (cfg, mirror module with the reference's freshly constructed parameters loaded and every ActNorm2d un-initialised,
    x, noise, [(bias, logs) of every ActNorm2d after the reference's first training-mode forward, module order])."""
    import argparse
    import torch
    from gbnf_amd import BoostedFlow
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    size = list(cfg["input_size"])
    args = argparse.Namespace(
        num_flows=cfg["K"], z_size=int(np.prod(size)), density_evaluation=True, device=torch.device(device), cuda=device != "cpu",
        component_type="glow", num_components=1, rho_init="decreasing", learn_top=True, y_classes=0, y_condition=False,
        sample_size=4, input_size=size, h_size=cfg["h"], num_blocks=cfg["L"], actnorm_scale=1.0, flow_permutation="invconv",
        flow_coupling="affine", LU_decomposed=False, num_dequant_blocks=0, coupling_network="tanh", coupling_network_depth=1,
        batch_norm=False)
    m = BoostedFlow(args)
    sd = {k[len("before."):]: torch.from_numpy(v) for k, v in data.items() if k.startswith("before.")}
    m.flows[0].load_state_dict(sd)
    for a in m.flows[0]._actnorms():
        a.inited = False
    if device != "cpu":
        m = m.to(device)
    x, noise = synth_image_batch_of(cfg)
    after = [(data[f"after.bias.{i}"], data[f"after.logs.{i}"]) for i in range(cfg["n_actnorm"])]
    return cfg, m, x, noise, after, data


def synth_image_batch_of(cfg):
    from gbnf_amd import synth
    return synth.synth_image_batch(cfg["N"], tuple(cfg["input_size"]), seed=cfg["x_seed"])


IMAGE_DECODE_CASES = ("g16_image_decode_invconv_affine", "g16_image_decode_shuffle_additive", "g16_image_decode_lu")


def load_image_decode_case(name):
    """g16: (cfg, spec, z, eps list, x of the reference's Glow.decode)."""
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    spec = synth.synth_image_glow_spec((3, 32, 32), cfg["h"], cfg["K"], cfg["L"], depth=cfg["depth"], coupling=cfg["coupling"],
                                       permutation=cfg["permutation"], learn_top=True, seed=cfg["w_seed"])
    if cfg["LU"]:
        k = 0
        for lvl in spec["levels"]:
            for st in lvl["steps"]:
                st["perm_w"] = data[f"c0.perm_w.{k}"]; k += 1
    eps = [data[f"eps.{l}"] for l in range(cfg["L"] - 1)]
    return cfg, spec, data["z"], eps, data["x"]


GRADS_CASES = ("g10_glow_grads_d43_h64", "g10_glow_grads_additive_relu_d8", "g10_realnvp_grads_d21_h32",
               "g10_realnvp_residual_grads_d21_h32")


def load_train_bn_case():
    """g10 (train-mode BatchNorm): (cfg, spec, x, data)."""
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g10_realnvp_grads_train_bn_d21_h32.npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    spec = synth.synth_boosted_specs("realnvp", 1, cfg["d"], cfg["h"], cfg["K"], seed=cfg["w_seed"])[0]
    x = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"], scale=cfg["x_scale"])
    return cfg, spec, x, data


def load_grads_case(name):
    """g10: (cfg, spec, x, nll, flat parameter gradients, g_x) -- the reference's own nll.backward()."""
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    spec = synth.synth_boosted_specs(cfg["kind"], 1, cfg["d"], cfg["h"], cfg["K"], seed=cfg["w_seed"], **cfg["synth_kw"])[0]
    x = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"], scale=cfg["x_scale"])
    return cfg, spec, x, float(data["nll"]), data["grads"], data["g_x"]


def load_decode_case():
    """g9: (specs, z, x_ref (C,N,d)) -- the reference's own Glow.decode on additive tabular components."""
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g9_glow_additive_decode.npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    specs = synth.synth_boosted_specs("glow", cfg["C"], cfg["d"], cfg["h"], cfg["K"], seed=cfg["w_seed"], **cfg["synth_kw"])
    z = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["z_seed"], scale=cfg["z_scale"])
    return specs, z, data["x"]


def load_actnorm_init_case():
    from gbnf_amd import synth
    data = dict(np.load(os.path.join(GOLDEN_DIR, "g7_glow_actnorm_data_init.npz")))
    cfg = json.loads(bytes(data["config"]).decode())
    specs = synth.synth_boosted_specs("glow", cfg["C"], cfg["d"], cfg["h"], cfg["K"], seed=cfg["w_seed"])
    x = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"], scale=cfg["x_scale"]) + np.float32(cfg["x_shift"])
    return cfg, data, specs, x.astype(np.float32)


class GoldenCase:
    """A fixture written by tests/golden/make_golden.py (outputs of the reference itself)."""

    def __init__(self, name):
        from gbnf_amd import spec as gspec
        from gbnf_amd import synth
        self.name = name
        self.data = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
        self.cfg = json.loads(bytes(self.data["config"]).decode())
        cfg = self.cfg
        self.rho = self.data["rho"]
        self.n_used = cfg.get("n_used", cfg["C"])
        if cfg["case"] == "native":
            self.x = self.data["x"]
            self.specs = [gspec.unflatten_spec(self.data, prefix=f"c{c}.") for c in range(cfg["C"])]
        elif cfg["case"] == "synth_specs":      # components given as generator calls (the stress offender g15)
            self.x = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"], scale=cfg.get("x_scale", 1.0),
                                       dist=cfg.get("x_dist", "normal"), clip=cfg.get("x_clip"))
            self.specs = [getattr(synth, e["fn"])(**e["kwargs"]) for e in cfg["specs"]]
            cfg.setdefault("synth_kw", {})
        else:
            self.x = synth.synth_batch(cfg["N"], cfg["d"], seed=cfg["x_seed"], scale=cfg.get("x_scale", 1.0))
            self.specs = synth.synth_boosted_specs(cfg["kind"], cfg["C"], cfg["d"], cfg["h"], cfg["K"],
                                                   seed=cfg["w_seed"], **cfg["synth_kw"])
        self.base = None
        if cfg["case"] == "toy":
            self.base = (self.data["base_mean"], self.data["base_std"])
        self.ll = self.data["ll"]
        self.ldj = self.data["ldj"]
        self.G = self.data["G"]

    def z(self, c):
        if "z" in self.data:
            return self.data["z"][c]
        return self.data["z_c0"] if c == 0 else None


@pytest.fixture(scope="session")
def golden_case():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = GoldenCase(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max |a-b| / max(|b|, 1) -- the 1e-5 relative log-likelihood bar of BASELINE.json."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) if a.size else 0.0
