"""The "flow spec": plain-data description of one boosted component.

It is the neutral format between (a) live ``nn.Module`` components (this
package's host mirror or the reference's own modules -- both expose the same
attribute names), (b) the C-ABI descriptors of ``include/gbnf.h`` and (c) the
``.npz`` golden fixtures.

  spec         = {"kind": "glow" | "realnvp", "d": int,
                  "coupling": "affine" | "additive"   (glow only),
                  "steps": [step, ...]}
  glow step    = {"an_bias": f32 (d,), "an_logs": f32 (d,), "perm": i64 (d,), "net": net}
  realnvp step = {"flipped": bool,
                  "bn": None | {"log_gamma","beta","running_mean","running_var": f32 (d,), "eps": float},
                  "t_net": net, "s_net": net}
  net          = {"act": "tanh" | "relu", "layers": [(W f32 (out,in), b f32 (out,)), ...]}

Reference anchors: FlowStep (models/glow.py:264-342), RealNVPFlow ctor
(models/realnvp.py:34-78), TanhNet/ReLUNet (models/layers.py:208-243),
BatchNorm (models/layers.py:320-358), PermuteNd (models/layers.py:633-651).
"""
from __future__ import annotations

import json

import numpy as np


def _np32(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def linears_of(net_module):
    """The Linear modules of a coupling network in evaluation order: TanhNet / ReLUNet keep them in ``network``
    (models/layers.py:208-243), a ResidualNet in ``initial_layer``, ``blocks[b].linear_layers[0..1]``, ``final_layer``
    (models/layers.py:246-301)."""
    if hasattr(net_module, "initial_layer"):
        return ([net_module.initial_layer] + [lin for blk in net_module.blocks for lin in blk.linear_layers]
                + [net_module.final_layer])
    return [m for m in net_module.network if type(m).__name__ == "Linear"]


def _net_from_module(seq_owner):
    """TanhNet / ReLUNet / ResidualNet -> net dict.  ``seq_owner.network`` is an nn.Sequential of
    Linear / activation modules (models/layers.py:208-243)."""
    if hasattr(seq_owner, "initial_layer"):
        return {"act": "residual", "layers": [(_np32(m.weight), _np32(m.bias)) for m in linears_of(seq_owner)]}
    layers = []
    act = None
    for m in seq_owner.network:
        cls = type(m).__name__
        if cls == "Linear":
            layers.append((_np32(m.weight), _np32(m.bias)))
        elif cls == "Tanh":
            act = "tanh" if act in (None, "tanh") else _mixed_act_error()
        elif cls == "ReLU":
            act = "relu" if act in (None, "relu") else _mixed_act_error()
        else:
            raise NotImplementedError(f"unsupported coupling-network module {cls}")
    return {"act": act or "tanh", "layers": layers}


def _mixed_act_error():
    raise NotImplementedError("coupling network mixes activations inside one net")


def spec_from_glow_module(glow, coupling=None, upto=None):
    """Export a tabular Glow component (this package's or the reference's module).

    Permutation indices are read from the LIVE module because the reference keeps
    them as plain attributes outside ``state_dict`` (models/layers.py:633-651).
    """
    steps = []
    d = None
    for k, layer in enumerate(glow.flow.layers):
        if upto is not None and k >= upto:
            break
        an = layer.actnorm
        if not bool(an.inited):
            raise ValueError("ActNorm not initialised (models/layers.py:473-475 raises in eval mode too)")
        d = int(an.bias.numel())
        if hasattr(layer, "shuffle"):
            perm = layer.shuffle.indices
        elif hasattr(layer, "reverse"):
            perm = layer.reverse.indices
        else:
            raise NotImplementedError("invconv permutation is image-only in the reference (models/layers.py:750)")
        steps.append({
            "an_bias": _np32(an.bias).reshape(-1),
            "an_logs": _np32(an.logs).reshape(-1),
            "perm": np.asarray(perm.detach().cpu().numpy() if hasattr(perm, "detach") else perm, dtype=np.int64),
            "net": _net_from_module(layer.block),
        })
        coupling = coupling or layer.flow_coupling
    return {"kind": "glow", "d": d, "coupling": coupling, "steps": steps}


def spec_from_realnvp_module(flow):
    """Export a RealNVPFlow component: flow_param[k] = [t_net, s_net, bn|None] and
    flipped = (k + flip_init) % 2 (models/realnvp.py:38, 115-119)."""
    steps = []
    for k, mods in enumerate(flow.flow_param):
        t_net, s_net, bn = mods[0], mods[1], (mods[2] if len(mods) > 2 else None)
        bn_spec = None
        if bn is not None:
            bn_spec = {
                "log_gamma": _np32(bn.log_gamma), "beta": _np32(bn.beta),
                "running_mean": _np32(bn.running_mean), "running_var": _np32(bn.running_var),
                "eps": float(bn.eps),
            }
        steps.append({
            "flipped": bool(((k + int(flow.flip_init)) % 2) > 0),
            "bn": bn_spec,
            "t_net": _net_from_module(t_net),
            "s_net": _net_from_module(s_net),
        })
    return {"kind": "realnvp", "d": int(flow.z_size), "steps": steps}


def spec_from_component(module):
    if hasattr(module, "flow_param"):
        return spec_from_realnvp_module(module)
    if hasattr(module, "flow"):
        return spec_from_glow_module(module)
    raise TypeError(f"not a boosted-flow component: {type(module).__name__}")


def activation_pattern_of_component(module):
    """The activations of a component's coupling nets, step by step, read from the module STRUCTURE only (no parameter
    is touched: usable before ActNorm's data-dependent initialisation).  Same value as
    ``native.activation_pattern(spec_from_component(module))``."""
    def act_of(seq_owner):
        if hasattr(seq_owner, "initial_layer"):
            return "residual"
        acts = {type(m).__name__.lower() for m in seq_owner.network if type(m).__name__ in ("Tanh", "ReLU")}
        if len(acts) > 1:
            _mixed_act_error()
        return acts.pop() if acts else "tanh"
    if hasattr(module, "flow_param"):
        return tuple((act_of(mods[0]), act_of(mods[1])) for mods in module.flow_param)
    if hasattr(module, "flow"):
        return tuple(act_of(layer.block) for layer in module.flow.layers)
    raise TypeError(f"not a boosted-flow component: {type(module).__name__}")


# ------------------------------------------------------------------ live (device) view for the training path
def _dev_net(seq_owner):
    if hasattr(seq_owner, "initial_layer"):
        return {"act": "residual", "layers": [(m.weight, m.bias) for m in linears_of(seq_owner)]}
    layers, act = [], None
    for m in seq_owner.network:
        cls = type(m).__name__
        if cls == "Linear":
            layers.append((m.weight, m.bias))
        elif cls in ("Tanh", "ReLU"):
            a = cls.lower()
            act = a if act in (None, a) else _mixed_act_error()
        else:
            raise NotImplementedError(f"unsupported coupling-network module {cls}")
    return {"act": act or "tanh", "layers": layers}


def device_spec_from_component(module):
    """Like ``spec_from_component`` but the float arrays are the module's own parameter / buffer TENSORS (no copy):
    the input of ``native.NativeTrainer``, which binds their device addresses."""
    if hasattr(module, "flow_param"):
        steps = []
        for k, mods in enumerate(module.flow_param):
            bn = mods[2] if len(mods) > 2 else None
            steps.append({
                "flipped": bool(((k + int(module.flip_init)) % 2) > 0),
                "bn": None if bn is None else {"log_gamma": bn.log_gamma, "beta": bn.beta, "running_mean": bn.running_mean,
                                               "running_var": bn.running_var, "eps": float(bn.eps),
                                               "batch_mean": getattr(bn, "batch_mean", None),
                                               "batch_var": getattr(bn, "batch_var", None)},
                "t_net": _dev_net(mods[0]), "s_net": _dev_net(mods[1])})
        return {"kind": "realnvp", "d": int(module.z_size), "steps": steps}
    steps, coupling, d = [], None, None
    for layer in module.flow.layers:
        an = layer.actnorm
        if not bool(an.inited):
            raise ValueError("ActNorm not initialised (models/layers.py:473-475 raises in eval mode too)")
        d = int(an.bias.numel())
        perm = layer.shuffle.indices if hasattr(layer, "shuffle") else layer.reverse.indices
        steps.append({"an_bias": an.bias, "an_logs": an.logs,
                      "perm": np.asarray(perm.detach().cpu().numpy() if hasattr(perm, "detach") else perm, dtype=np.int64),
                      "net": _dev_net(layer.block)})
        coupling = coupling or layer.flow_coupling
    return {"kind": "glow", "d": d, "coupling": coupling, "steps": steps}


# ------------------------------------------------------------------ (de)serialisation
def flatten_spec(spec, prefix=""):
    """spec -> {name: ndarray} (+ a json header) for ``np.savez``."""
    out = {}
    header = {"kind": spec["kind"], "d": spec["d"], "coupling": spec.get("coupling"), "steps": []}
    for k, st in enumerate(spec["steps"]):
        p = f"{prefix}s{k}."
        h = {}
        if spec["kind"] == "glow":
            out[p + "an_bias"] = st["an_bias"]
            out[p + "an_logs"] = st["an_logs"]
            out[p + "perm"] = st["perm"]
            nets = {"net": st["net"]}
        else:
            h["flipped"] = bool(st["flipped"])
            h["bn"] = st["bn"] is not None
            if st["bn"] is not None:
                h["eps"] = st["bn"]["eps"]
                for key in ("log_gamma", "beta", "running_mean", "running_var"):
                    out[p + "bn." + key] = st["bn"][key]
            nets = {"t_net": st["t_net"], "s_net": st["s_net"]}
        h["nets"] = {}
        for name, net in nets.items():
            h["nets"][name] = {"act": net["act"], "n": len(net["layers"])}
            for i, (w, b) in enumerate(net["layers"]):
                out[f"{p}{name}.{i}.w"] = w
                out[f"{p}{name}.{i}.b"] = b
        header["steps"].append(h)
    out[prefix + "header"] = np.frombuffer(json.dumps(header).encode(), dtype=np.uint8)
    return out


def unflatten_spec(arrs, prefix=""):
    header = json.loads(bytes(arrs[prefix + "header"]).decode())
    spec = {"kind": header["kind"], "d": header["d"], "steps": []}
    if header["kind"] == "glow":
        spec["coupling"] = header["coupling"]
    for k, h in enumerate(header["steps"]):
        p = f"{prefix}s{k}."
        st = {}
        for name, nh in h["nets"].items():
            st[name] = {"act": nh["act"],
                        "layers": [(np.asarray(arrs[f"{p}{name}.{i}.w"], np.float32),
                                    np.asarray(arrs[f"{p}{name}.{i}.b"], np.float32)) for i in range(nh["n"])]}
        if header["kind"] == "glow":
            st["an_bias"] = np.asarray(arrs[p + "an_bias"], np.float32)
            st["an_logs"] = np.asarray(arrs[p + "an_logs"], np.float32)
            st["perm"] = np.asarray(arrs[p + "perm"], np.int64)
        else:
            st["flipped"] = h["flipped"]
            st["bn"] = None
            if h["bn"]:
                st["bn"] = {key: np.asarray(arrs[p + "bn." + key], np.float32)
                            for key in ("log_gamma", "beta", "running_mean", "running_var")}
                st["bn"]["eps"] = h["eps"]
        spec["steps"].append(st)
    return spec


def spec_shape_key(spec):
    """Architecture signature shared by all components of one BoostedFlow."""
    st = spec["steps"][0]
    net = st["net"] if spec["kind"] == "glow" else st["t_net"]
    h = net["layers"][0][0].shape[0]
    depth = len(net["layers"]) - 2
    return (spec["kind"], spec["d"], h, len(spec["steps"]), depth, spec.get("coupling"))
