"""Portable synthetic parameters for boosted-flow components.

There are no datasets or checkpoints on the build / GPU boxes, so benchmarks,
tests and the golden-fixture generator all draw component parameters from this
one deterministic generator (legacy ``numpy.random.RandomState`` streams, which
are stable across numpy versions).  The output is the plain "flow spec" data
model described in ``spec.py``; the fixture generator loads the very same
numbers into the reference modules with ``load_state_dict``.

Parameter scales follow ``nn.Linear``'s default init U(-1/sqrt(in), 1/sqrt(in))
(the reference never overrides it for TanhNet/ReLUNet, models/layers.py:208-243)
times ``gain`` so shift/scale are non-trivial, and ActNorm / BatchNorm
statistics are perturbed away from identity.
"""
from __future__ import annotations

import numpy as np


def _linear(rng, out_f, in_f, gain):
    bound = gain / np.sqrt(in_f)
    w = rng.uniform(-bound, bound, size=(out_f, in_f)).astype(np.float32)
    b = rng.uniform(-bound, bound, size=(out_f,)).astype(np.float32)
    return w, b


def _net(rng, in_f, out_f, h, depth, act, gain):
    layers = [_linear(rng, h, in_f, gain)]
    for _ in range(depth):
        layers.append(_linear(rng, h, h, gain))
    layers.append(_linear(rng, out_f, h, gain))
    return {"act": act, "layers": layers}


def _res_net(rng, in_f, out_f, h, blocks, gain):
    """ResidualNet (models/layers.py:276-301): initial, `blocks` x (Linear, Linear), final."""
    layers = [_linear(rng, h, in_f, gain)]
    for _ in range(blocks):
        layers.append(_linear(rng, h, h, gain))
        layers.append(_linear(rng, h, h, 0.3 * gain))
    layers.append(_linear(rng, out_f, h, gain))
    return {"act": "residual", "layers": layers}


def _trained_like_net(rng, net):
    """Magnitudes a TRAINED flow ends up with rather than nn.Linear's init: the coupling net's last layer shrinks towards
    the identity transform -- every output row gets its own factor 10^-3 .. 10^-4 (weights and bias): fp16-subnormal
    territory for the `mid` piece of a split operand."""
    w, b = net["layers"][-1]
    f = np.power(10.0, rng.uniform(-4.0, -3.0, size=(w.shape[0], 1))).astype(np.float32)
    net["layers"][-1] = ((w * f).astype(np.float32), (b * f[:, 0]).astype(np.float32))
    return net


def synth_glow_spec(d, h, K, depth=1, act="tanh", coupling="affine", permutation="shuffle",
                    seed=0, gain=1.0, trained_like=False):
    """One tabular Glow component (models/glow.py FlowStep x K) with synthetic parameters.
    ``act``: "tanh" | "relu" | "random" (each step draws one of the two, models/glow.py:295-296)."""
    rng = np.random.RandomState(seed)
    act_rng = np.random.RandomState(seed + 7919)
    d1 = d // 2
    d2 = d - d1
    steps = []
    for _ in range(K):
        perm = np.arange(d - 1, -1, -1, dtype=np.int64)
        if permutation == "shuffle":
            perm = perm[rng.permutation(d)]
        out_f = d2 * 2 if coupling == "affine" else d2
        steps.append({
            "an_bias": (0.1 * rng.standard_normal(d)).astype(np.float32),
            "an_logs": (0.1 * rng.standard_normal(d)).astype(np.float32),
            "perm": perm.astype(np.int64),
            "net": _net(rng, d1, out_f, h, depth, ["tanh", "relu"][act_rng.randint(2)] if act == "random" else act, gain),
        })
        if trained_like:       # ActNorm after data-dependent init on un-normalised data + training: log-scales ~ N(0,1) out to +-3, O(1) shifts
            steps[-1]["an_bias"] = rng.standard_normal(d).astype(np.float32)
            steps[-1]["an_logs"] = np.clip(rng.standard_normal(d), -3.0, 3.0).astype(np.float32)
            _trained_like_net(rng, steps[-1]["net"])
    return {"kind": "glow", "d": int(d), "coupling": coupling, "steps": steps}


def synth_realnvp_spec(d, h, K, depth=1, coupling_network="tanh", batch_norm=True, flip_init=0,
                       seed=0, gain=1.0, trained_like=False):
    """One RealNVPFlow component (models/realnvp.py:34-78) with synthetic parameters.

    ``coupling_network``: "tanh" | "relu" | "mixed" (t_net ReLU, s_net Tanh, realnvp.py:47-51) | "random" (every net
    of every step draws one of the two, realnvp.py:59-60) | "residual" (ResidualNet of ``depth`` blocks, realnvp.py:57).
    BatchNorm is present on every step but the last when ``batch_norm`` (realnvp.py:71-74).
    """
    rng = np.random.RandomState(seed)
    act_rng = np.random.RandomState(seed + 7919)
    steps = []
    for k in range(K):
        flipped = ((k + flip_init) % 2) > 0
        if flipped:
            out_f, in_f = d // 2, d - d // 2
        else:
            in_f, out_f = d // 2, d - d // 2
        if coupling_network == "mixed":
            t_act, s_act = "relu", "tanh"
        elif coupling_network == "random":
            t_act, s_act = (["tanh", "relu"][act_rng.randint(2)] for _ in range(2))
        else:
            t_act = s_act = coupling_network
        if coupling_network == "residual":       # depth = number of residual blocks (realnvp.py:62-65)
            t_net = _res_net(rng, in_f, out_f, h, depth, gain)
            s_net = _res_net(rng, in_f, out_f, h, depth, 0.5 * gain)
        else:
            t_net = _net(rng, in_f, out_f, h, depth, t_act, gain)
            # keep log-scales tame: the s-net's last layer is scaled down a little
            s_net = _net(rng, in_f, out_f, h, depth, s_act, gain)
        bn = None
        if batch_norm and k < K - 1:
            bn = {
                "log_gamma": (0.1 * rng.standard_normal(d)).astype(np.float32),
                "beta": (0.1 * rng.standard_normal(d)).astype(np.float32),
                "running_mean": (0.1 * rng.standard_normal(d)).astype(np.float32),
                "running_var": rng.uniform(0.5, 1.5, size=d).astype(np.float32),
                "eps": 1e-5,
            }
        if trained_like:
            _trained_like_net(rng, t_net)
            _trained_like_net(rng, s_net)
            if bn is not None:     # running statistics of un-normalised activations: variances over four decades
                bn["log_gamma"] = rng.uniform(-1.0, 1.0, size=d).astype(np.float32)
                bn["running_mean"] = rng.standard_normal(d).astype(np.float32)
                bn["running_var"] = np.power(10.0, rng.uniform(-2.0, 2.0, size=d)).astype(np.float32)
        steps.append({"flipped": bool(flipped), "bn": bn, "t_net": t_net, "s_net": s_net})
    return {"kind": "realnvp", "d": int(d), "steps": steps}


def synth_boosted_specs(kind, C, d, h, K, seed=1, **kw):
    """C components; component c uses stream ``seed*1000 + c`` (and flip_init=c for RealNVP,
    as models/boosted_flow.py:46 does)."""
    specs = []
    for c in range(C):
        if kind == "glow":
            specs.append(synth_glow_spec(d, h, K, seed=seed * 1000 + c, **kw))
        elif kind == "realnvp":
            specs.append(synth_realnvp_spec(d, h, K, flip_init=c, seed=seed * 1000 + c, **kw))
        else:
            raise ValueError(kind)
    return specs


def synth_batch(N, d, seed=0, scale=1.0, dist="normal", clip=None):
    """z-scored-like N(0,1) inputs (the loaders z-score the data: utils/miniboone.py:57-67).  ``dist="student_t3"``:
    heavy-tailed rows (Student t, 3 degrees of freedom: what z-scored real data with outliers looks like), clipped to
    +-``clip``."""
    rng = np.random.RandomState(seed)
    if dist == "normal":
        x = rng.standard_normal((N, d))
    elif dist == "student_t3":
        x = rng.standard_t(3.0, size=(N, d))
    else:
        raise ValueError(dist)
    x = scale * x
    if clip is not None:
        x = np.clip(x, -clip, clip)
    return x.astype(np.float32)


# ------------------------------------------------------------------ image Glow (multi-scale, conv coupling nets)
def _conv(rng, out_ch, in_ch, k, std, bias=False, actnorm=False, zeros_logs=False, logs_spread=None, an_spread=None):
    """One Conv2d / Conv2dZeros of models/layers.py:577-630 as plain data:
    w (out,in,k,k); b (out,) | None; an_bias/an_logs (out,) | None (the ActNorm2d behind a Conv2d);
    logs (out,) | None (Conv2dZeros' output scale exp(3*logs))."""
    c = {"w": (std * rng.standard_normal((out_ch, in_ch, k, k))).astype(np.float32), "b": None,
         "an_bias": None, "an_logs": None, "logs": None}
    if bias:
        c["b"] = (0.05 * rng.standard_normal(out_ch)).astype(np.float32)
    if actnorm:
        c["an_bias"] = (0.1 * rng.standard_normal(out_ch)).astype(np.float32)
        c["an_logs"] = (0.1 * rng.standard_normal(out_ch)).astype(np.float32)
        if an_spread is not None:      # trained-like: per-channel scales over orders of magnitude
            c["an_logs"] = rng.uniform(-an_spread, an_spread, out_ch).astype(np.float32)
    if zeros_logs:
        c["logs"] = (0.05 * rng.standard_normal(out_ch)).astype(np.float32)
        if logs_spread is not None:    # Conv2dZeros: output scale exp(3 logs), models/layers.py:608-630
            c["logs"] = rng.uniform(-logs_spread, logs_spread, out_ch).astype(np.float32)
    return c


def synth_image_glow_spec(input_size=(3, 32, 32), h=32, K=2, L=2, depth=1, coupling="affine", permutation="invconv",
                          learn_top=True, seed=0, gain=1.0, trained_like=False):
    """One image Glow component (models/glow.py:192-233 FlowNet image branch: L x [squeeze, K FlowSteps, Split2d])
    with synthetic parameters.  ``permutation``: "invconv" (a random well-conditioned C x C matrix: what
    InvertibleConv1x1.get_weight returns for either parameterisation), "shuffle" or "reverse"."""
    rng = np.random.RandomState(seed)
    C, H, W = input_size
    levels = []
    # trained_like (fixture g18): magnitudes a trained model shows -- the coupling nets' ActNorm2d logs over +-3 (per-channel
    # scales 0.05 .. 20 on the hidden activations: what the fp16 range of the split kernels has to carry), the steps' ActNorm2d
    # logs over +-1 (+-3 there compounds over the steps to |z| ~ 1e3 and log-likelihoods of -1e9 .. -1e11: no model trains to
    # that), Conv2dZeros logs over +-0.5 (output gains exp(3 logs) up to e^1.5) with small weights
    an_step = 1.0 if trained_like else None
    an_net = 3.0 if trained_like else None
    zl = 0.5 if trained_like else None
    for lvl in range(L):
        C, H, W = C * 4, H // 2, W // 2
        steps = []
        for _ in range(K):
            st = {"an_bias": (0.1 * rng.standard_normal(C)).astype(np.float32),
                  "an_logs": (0.1 * rng.standard_normal(C)).astype(np.float32), "perm_w": None, "perm": None}
            if trained_like:
                st["an_logs"] = rng.uniform(-an_step, an_step, C).astype(np.float32)
            if permutation == "invconv":
                q, _ = np.linalg.qr(rng.standard_normal((C, C)))
                st["perm_w"] = (q * np.exp(0.1 * rng.standard_normal(C))[None, :]).astype(np.float32)
            else:
                perm = np.arange(C - 1, -1, -1, dtype=np.int64)
                st["perm"] = perm[rng.permutation(C)] if permutation == "shuffle" else perm
            cin, cout = C // 2, C - C // 2
            out_ch = 2 * cout if coupling == "affine" else cout
            convs = [_conv(rng, h, cin, 3, gain * 0.6 / np.sqrt(9 * cin), actnorm=True, an_spread=an_net)]
            for _ in range(depth):
                convs.append(_conv(rng, h, h, 1, gain * 1.0 / np.sqrt(h), actnorm=True, an_spread=an_net))
            convs.append(_conv(rng, out_ch, h, 3, gain * (0.05 if trained_like else 0.5) / np.sqrt(9 * h), bias=True, zeros_logs=True,
                               logs_spread=zl))
            st["convs"] = convs
            steps.append(st)
        split = None
        if lvl < L - 1:
            split = _conv(rng, C, C // 2, 3, gain * 0.3 / np.sqrt(9 * (C // 2)), bias=True, zeros_logs=True)
            C = C // 2
        levels.append({"steps": steps, "split": split})
    top = None
    if learn_top:
        top = _conv(rng, 2 * C, 2 * C, 3, 0.05, bias=True, zeros_logs=True)
    return {"kind": "glow_image", "input_size": [int(v) for v in input_size], "hidden": int(h), "coupling": coupling,
            "bounds": 0.9, "levels": levels, "learn_top": top}


def synth_image_batch(N, input_size=(3, 32, 32), seed=0):
    """Images in [0, 1] quantised to 256 levels (what the CIFAR loader hands over) + the dequantisation noise."""
    rng = np.random.RandomState(seed)
    x = rng.randint(0, 256, size=(N,) + tuple(input_size)).astype(np.float32) / 255.0
    noise = rng.uniform(0.0, 1.0, size=x.shape).astype(np.float32)
    return x, noise
