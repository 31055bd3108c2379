"""Host mirror of the reference's IMAGE Boosted-Glow (BASELINE.json configs[3]) on the density-evaluation path.

``BoostedFlow(args)`` returns a ``BoostedImageFlow`` when ``args.input_size`` has more than one dimension.  The module
tree reproduces the reference's parameter names and shapes (tests/golden/state_dict_layout.json: ``image_*``), so a
reference checkpoint loads with ``load_state_dict``; every call that computes goes to ``libgbnf_hip.so``
(``native.NativeImageFlow``) -- there is no eager / CPU path.

Reference anchors: Glow (models/glow.py:12-110), FlowNet image branch (:192-252), FlowStep (:264-342), ActNorm2d /
Conv2d / Conv2dZeros / Permute2d / Split2d / SqueezeLayer / InvertibleConv1x1 (models/layers.py:548-796), BoostedFlow
(models/boosted_flow.py), the image likelihood ll = log_normal_diag(z, z_mu, z_var) + logdet (image_experiment.py:227).
Not on the path: y_condition, learned dequantisation flows, training.  An un-initialised model (fresh from the constructor)
is initialised from data with ``BoostedImageFlow.initialize_actnorms(x)`` -- what the reference's first training-mode forward
does (models/layers.py:473-486) -- or by loading a trained checkpoint / ``set_actnorm_init``.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import native


def _no_eager(*a, **k):
    raise RuntimeError("image layers are evaluated by the HIP kernels, not module by module")


class ActNorm2d(nn.Module):
    def __init__(self, num_features, scale=1.0):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.logs = nn.Parameter(torch.zeros(1, num_features, 1, 1))
        self.num_features, self.scale, self.inited, self.image_input = num_features, scale, False, True


class Conv2d(nn.Module):
    """models/layers.py:577-606: conv without bias + ActNorm2d."""

    def __init__(self, in_dim, out_dim, kernel_size=(3, 3), weight_std=0.05):
        super().__init__()
        pad = ((kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2)
        self.conv = nn.Conv2d(in_dim, out_dim, kernel_size, 1, pad, bias=False)
        self.conv.weight.data.normal_(mean=0.0, std=weight_std)
        self.actnorm = ActNorm2d(out_dim)
        self.do_actnorm = True
    forward = _no_eager


class Conv2dZeros(nn.Module):
    """models/layers.py:609-630."""

    def __init__(self, in_dim, out_dim, logscale_factor=3):
        super().__init__()
        self.conv = nn.Conv2d(in_dim, out_dim, (3, 3), 1, (1, 1))
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()
        self.logscale_factor = logscale_factor
        self.logs = nn.Parameter(torch.zeros(out_dim, 1, 1))
    forward = _no_eager


class ConvNet(nn.Module):
    """models/layers.py:304-317."""

    def __init__(self, in_dim, out_dim, hidden_dim, num_layers=1):
        super().__init__()
        layers = [Conv2d(in_dim, hidden_dim)]
        for _ in range(num_layers):
            layers += [nn.ReLU(), Conv2d(hidden_dim, hidden_dim, kernel_size=(1, 1))]
        layers += [nn.ReLU(), Conv2dZeros(hidden_dim, out_dim)]
        self.network = nn.Sequential(*layers)
    forward = _no_eager


class Permute2d(nn.Module):
    """models/layers.py:633-680: indices are plain attributes (not in state_dict)."""

    def __init__(self, num_dim, shuffle):
        super().__init__()
        self.num_dim = num_dim
        self.indices = torch.arange(num_dim - 1, -1, -1, dtype=torch.long)
        if shuffle:
            self.indices = self.indices[torch.randperm(num_dim)]

    def __setattr__(self, name, value):
        if name == "indices":        # every (re)assignment gets a new serial number: the handle keys use it
            object.__setattr__(self, "indices_serial", getattr(self, "indices_serial", 0) + 1)
        super().__setattr__(name, value)

    def set_indices(self, indices):
        idx = torch.as_tensor(np.asarray(indices), dtype=torch.long).clone()
        if idx.shape != (self.num_dim,) or sorted(idx.tolist()) != list(range(self.num_dim)):
            raise ValueError("indices must be a permutation of range(num_dim)")
        self.indices = idx


class InvertibleConv1x1(nn.Module):
    """models/layers.py:722-749: either the plain (C,C) weight or its LU factors (p, sign_s buffers; lower, log_s, upper)."""

    def __init__(self, num_dim, LU_decomposed):
        super().__init__()
        w_init = torch.linalg.qr(torch.randn(num_dim, num_dim))[0]
        self.LU_decomposed = LU_decomposed
        self.w_shape = [num_dim, num_dim]
        if not LU_decomposed:
            self.weight = nn.Parameter(w_init)
        else:
            p, lower, upper = torch.linalg.lu(w_init)
            s = torch.diag(upper)
            self.register_buffer("p", p)
            self.register_buffer("sign_s", torch.sign(s))
            self.lower = nn.Parameter(lower)
            self.log_s = nn.Parameter(torch.log(torch.abs(s)))
            self.upper = nn.Parameter(torch.triu(upper, 1))

    def composed_weight(self):
        """What get_weight returns in the forward direction (models/layers.py:751-776), as float64 numpy."""
        if not self.LU_decomposed:
            return self.weight.detach().double().cpu().numpy()
        n = self.w_shape[0]
        l_mask = torch.tril(torch.ones(n, n, dtype=torch.float64), -1)
        lower = self.lower.detach().double().cpu() * l_mask + torch.eye(n, dtype=torch.float64)
        u = self.upper.detach().double().cpu() * l_mask.t() + torch.diag(
            self.sign_s.double().cpu() * torch.exp(self.log_s.detach().double().cpu()))
        return (self.p.double().cpu() @ (lower @ u)).numpy()


class FlowStep(nn.Module):
    def __init__(self, in_dim, hidden_dim, actnorm_scale, flow_permutation, flow_coupling, LU_decomposed, args):
        super().__init__()
        self.image_input, self.flow_coupling = True, flow_coupling
        self.actnorm = ActNorm2d(in_dim, actnorm_scale)
        if flow_permutation == "invconv":
            self.invconv = InvertibleConv1x1(in_dim, LU_decomposed)
        elif flow_permutation == "shuffle":
            self.shuffle = Permute2d(in_dim, shuffle=True)
        else:
            self.reverse = Permute2d(in_dim, shuffle=False)
        cin, cout = in_dim // 2, in_dim - in_dim // 2
        if flow_coupling not in ("additive", "affine"):
            raise ValueError(f"flow_coupling={flow_coupling!r}")
        self.block = ConvNet(cin, cout if flow_coupling == "additive" else 2 * cout, hidden_dim,
                             args.coupling_network_depth)


class SqueezeLayer(nn.Module):
    def __init__(self, factor):
        super().__init__()
        self.factor = factor


class Split2d(nn.Module):
    def __init__(self, in_dim):
        super().__init__()
        self.conv = Conv2dZeros(in_dim // 2, in_dim)


class FlowNet(nn.Module):
    """models/glow.py:192-233, image branch."""

    def __init__(self, args):
        super().__init__()
        self.image_input, self.K, self.L = True, args.num_flows, args.num_blocks
        self.layers = nn.ModuleList()
        self.output_shapes = []
        C, H, W = args.input_size
        for i in range(self.L):
            C, H, W = C * 4, H // 2, W // 2
            self.layers.append(SqueezeLayer(2))
            self.output_shapes.append([-1, C, H, W])
            for _ in range(self.K):
                self.layers.append(FlowStep(C, args.h_size, args.actnorm_scale, args.flow_permutation, args.flow_coupling,
                                            args.LU_decomposed, args))
            self.output_shapes.append([-1, C, H, W])
            if i < self.L - 1:
                self.layers.append(Split2d(C))
                self.output_shapes.append([-1, C // 2, H, W])
                C = C // 2


class ImageGlow(nn.Module):
    """models/glow.py:12-58 for image input."""

    def __init__(self, args):
        super().__init__()
        if getattr(args, "y_condition", False):
            raise NotImplementedError("y_condition is not on the supported path")
        if getattr(args, "num_dequant_blocks", 0) > 0:
            raise NotImplementedError("learned dequantisation flows are not on the supported path")
        self.learn_top, self.y_condition, self.y_classes = bool(args.learn_top), False, args.y_classes
        self.sample_size, self.image_input = args.sample_size, True
        self.input_size = [int(v) for v in args.input_size]
        self.hidden = int(args.h_size)
        self.flow = FlowNet(args)
        Cz, Hz, Wz = self.flow.output_shapes[-1][1:]
        if self.learn_top:
            self.learn_top_fn = Conv2dZeros(Cz * 2, Cz * 2)
        self.register_buffer("prior_h", torch.zeros([1, Cz * 2, Hz, Wz]))
        self.register_buffer("bounds", torch.tensor([0.9], dtype=torch.float32))
        self.dequant_flows = None

    def _actnorms(self):
        return [m for m in self.modules() if isinstance(m, ActNorm2d)]

    def set_actnorm_init(self):
        """models/glow.py:181-187."""
        for m in self._actnorms():
            m.inited = True


def _np(t):
    return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _conv_spec(m):
    zeros = isinstance(m, Conv2dZeros)
    return {"w": _np(m.conv.weight), "b": _np(m.conv.bias) if m.conv.bias is not None else None,
            "an_bias": None if zeros else _np(m.actnorm.bias).reshape(-1),
            "an_logs": None if zeros else _np(m.actnorm.logs).reshape(-1),
            "logs": _np(m.logs).reshape(-1) if zeros else None}


def image_spec_from_glow_module(glow):
    """ImageGlow (this mirror; the reference's module has the same attributes) -> image flow spec (synth.py)."""
    if not all(bool(m.inited) for m in glow._actnorms()):
        raise ValueError("ActNorm not initialised (models/layers.py:473-475 raises in eval mode too)")
    levels, steps = [], []
    coupling = None
    for layer in glow.flow.layers:
        if isinstance(layer, FlowStep):
            st = {"an_bias": _np(layer.actnorm.bias).reshape(-1), "an_logs": _np(layer.actnorm.logs).reshape(-1),
                  "perm_w": None, "perm": None}
            if hasattr(layer, "invconv"):
                st["perm_w"] = layer.invconv.composed_weight().astype(np.float32)
            else:
                pm = layer.shuffle if hasattr(layer, "shuffle") else layer.reverse
                st["perm"] = np.asarray(pm.indices.cpu().numpy(), dtype=np.int64)
            st["convs"] = [_conv_spec(m) for m in layer.block.network if not isinstance(m, nn.ReLU)]
            coupling = layer.flow_coupling
            steps.append(st)
        elif isinstance(layer, Split2d):
            levels.append({"steps": steps, "split": _conv_spec(layer.conv)})
            steps = []
    levels.append({"steps": steps, "split": None})
    return {"kind": "glow_image", "input_size": list(glow.input_size), "hidden": glow.hidden, "coupling": coupling,
            "bounds": float(glow.bounds.item()), "levels": levels,
            "learn_top": _conv_spec(glow.learn_top_fn) if glow.learn_top else None}


def _put(dst, src):
    with torch.no_grad():
        dst.copy_(torch.as_tensor(np.asarray(src), dtype=dst.dtype).reshape(dst.shape).to(dst.device))


def _load_conv(m, c):
    _put(m.conv.weight, c["w"])
    if c["b"] is not None:
        _put(m.conv.bias, c["b"])
    if c["an_bias"] is not None:
        _put(m.actnorm.bias, c["an_bias"])
        _put(m.actnorm.logs, c["an_logs"])
        m.actnorm.inited = True
    if c["logs"] is not None:
        _put(m.logs, c["logs"])


def load_image_spec(glow, spec):
    """Install an image flow spec's numbers into an ImageGlow (plain-weight invconv or Permute2d components)."""
    steps = [st for lv in spec["levels"] for st in lv["steps"]]
    splits = [lv["split"] for lv in spec["levels"] if lv["split"] is not None]
    si = pi = 0
    for layer in glow.flow.layers:
        if isinstance(layer, FlowStep):
            st = steps[si]
            si += 1
            _put(layer.actnorm.bias, st["an_bias"])
            _put(layer.actnorm.logs, st["an_logs"])
            layer.actnorm.inited = True
            if hasattr(layer, "invconv"):
                if layer.invconv.LU_decomposed:
                    raise NotImplementedError("load_image_spec installs a composed matrix: build the module with LU_decomposed=False")
                _put(layer.invconv.weight, st["perm_w"])
            else:
                (layer.shuffle if hasattr(layer, "shuffle") else layer.reverse).set_indices(st["perm"])
            convs = [m for m in layer.block.network if not isinstance(m, nn.ReLU)]
            if len(convs) != len(st["convs"]):
                raise ValueError("coupling network depth mismatch")
            for m, c in zip(convs, st["convs"]):
                _load_conv(m, c)
        elif isinstance(layer, Split2d):
            _load_conv(layer.conv, splits[pi])
            pi += 1
    if spec["learn_top"] is not None:
        _load_conv(glow.learn_top_fn, spec["learn_top"])


class BoostedImageFlow(nn.Module):
    """models/boosted_flow.py:BoostedFlow for image components, density-evaluation path."""

    def __init__(self, args):
        super().__init__()
        if args.component_type != "glow":
            raise NotImplementedError("image components are Glow (models/boosted_flow.py:44-50 builds nothing else for images)")
        self.args = args
        self.num_flows, self.z_size = args.num_flows, args.z_size
        self.density_evaluation = args.density_evaluation
        self.all_trained, self.component_type = False, args.component_type
        self.num_components, self.component = args.num_components, 0
        self.register_buffer("base_dist_mean", torch.randn(self.z_size).normal_(0, 0.1))
        self.register_buffer("base_dist_var", 3.0 * torch.ones(self.z_size))
        if args.rho_init == "decreasing":
            rho = torch.clamp(1.0 / torch.pow(2.0, torch.arange(self.num_components * 1.0)), min=0.05)
        else:
            rho = torch.full((self.num_components,), 1.0 / self.num_components)
        self.register_buffer("rho", rho.float())
        self.flows = nn.ModuleList([ImageGlow(args) for _ in range(self.num_components)])
        self._handles = {}
        dev = getattr(args, "device", None)
        if dev is not None:
            self.to(dev)

    # ---- reference API
    def increment_component(self):
        if self.component == self.num_components - 1:
            self.component, self.all_trained = 0, True
        else:
            self.component = min(self.component + 1, self.num_components - 1)

    def _sample_component(self, sampling_components):
        """models/boosted_flow.py:61-96."""
        if sampling_components == "c":
            return min(self.component, self.num_components - 1)
        if sampling_components in ("1:c", "1:c-1"):
            n = self.component if sampling_components == "1:c-1" else (
                self.num_components if self.all_trained else self.component + 1)
            n = min(max(n, 1), self.num_components)
            simplex = self.rho[0:n] / torch.sum(self.rho[0:n])
            return int(torch.multinomial(simplex, 1, replacement=True).item())
        if sampling_components == "-c":
            simplex = self.rho.clone().detach()
            simplex[self.component] = 0.0
            return int(torch.multinomial(simplex / simplex.sum(), 1, replacement=True).item())
        raise ValueError("z_k can only be sampled from ['c', '1:c-1', '1:c', '-c']")

    def _component_tensors(self, c):
        """(parameters, buffers, Permute2d modules, ActNorm modules) of component c, collected once: walking the module
        tree per call costs about as much as a small launch.  `_apply` (.to / .cuda / .float) drops the cache."""
        cache = self.__dict__.setdefault("_tensor_cache", {})
        if c not in cache:
            flow = self.flows[c]
            cache[c] = (list(flow.parameters()), list(flow.buffers()),
                        [m for m in flow.modules() if isinstance(m, Permute2d)], list(flow._actnorms()))
        return cache[c]

    def _apply(self, fn, *a, **k):
        self.__dict__.pop("_tensor_cache", None)
        return super()._apply(fn, *a, **k)

    def native_flow(self, c):
        flow = self.flows[c]
        params, buffers, perms, actnorms = self._component_tensors(c)
        key = tuple(int(t._version) for t in params) + tuple(int(t._version) for t in buffers) + \
            tuple(t.data_ptr() for t in params) + \
            tuple((m.indices_serial, int(m.indices._version)) for m in perms) + \
            tuple(bool(m.inited) for m in actnorms)
        cached = self._handles.get(c)
        if cached is None or cached[0] != key:
            self._handles[c] = (key, native.NativeImageFlow(image_spec_from_glow_module(flow)))
        return self._handles[c][1]

    def _check(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise native.GbnfError("x must live on the MI355X (cuda) device: this module has no CPU path")
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("the image path is density evaluation only: call .eval() or use torch.no_grad()")

    def component_forward(self, x, c, noise=None, want_z=True):
        """x (N,C,H,W) in [0,1] -> z, ldj, ll of component c.  ``noise``: the U(0,1) dequantisation noise
        (models/glow.py:135); drawn on the device when None."""
        self._check(x)
        x = x.contiguous().float()
        if noise is None:
            noise = torch.rand_like(x)
        with torch.cuda.device(x.device):
            return self.native_flow(int(c)).forward(x, noise.contiguous().float(), want_z=want_z)

    def _prior_on(self, handle, device):
        prior = handle.__dict__.get("_prior_dev")            # (mean, logvar) of the top prior on the device, per handle
        if prior is None or prior[0].device != device:
            mu, lv = handle.prior()
            prior = (torch.from_numpy(mu).to(device).view(1, -1, 1, 1), torch.from_numpy(lv).to(device).view(1, -1, 1, 1))
            handle.__dict__["_prior_dev"] = prior
        return prior

    def decode(self, z, y_onehot=None, temperature=1.0, components="c", eps=None):
        """models/boosted_flow.py:208-218 -> Glow.decode (models/glow.py:112-123) of one component (the reference's own
        BoostedFlow.decode dies on a misspelt keyword; the component-level decode is the parity target,
        fixtures g16_*).  z None: ``sample_size`` draws from the top prior, z = Normal(z_mu, exp(z_var) * temperature) as the
        reference writes it.  ``eps``: the standard-normal draws of the Split2d levels (native.NativeImageFlow.inverse)."""
        c = self._sample_component(components) if isinstance(components, str) else int(components)
        handle = self.native_flow(c)
        temperature = 1.0 if temperature is None else float(temperature)
        if z is None:
            dev = self.rho.device
            mu, lv = self._prior_on(handle, dev)
            n = int(self.flows[c].sample_size)
            z = mu + torch.exp(lv) * temperature * torch.randn((n,) + tuple(handle.z_shape), device=dev)
        if not z.is_cuda:
            raise native.GbnfError("z must live on the MI355X (cuda) device: this module has no CPU path")
        with torch.cuda.device(z.device):
            return handle.inverse(z.contiguous().float(), eps, temperature)

    def forward(self, x=None, y_onehot=None, z=None, temperature=None, components=None, reverse=False):
        if reverse:
            return self.decode(z, y_onehot, temperature, components)
        c = self._sample_component(components) if isinstance(components, str) else int(components)
        zz, ldj, _ = self.component_forward(x, c)
        handle = self.native_flow(c)
        prior = self._prior_on(handle, x.device)
        shape = (x.shape[0],) + tuple(zz.shape[1:])
        return zz, prior[0].expand(shape), prior[1].expand(shape), ldj, None      # broadcast views: same values, no copies

    @torch.no_grad()
    def initialize_actnorms(self, x, components=None, noise=None):
        """The data-dependent initialisation of every ActNorm2d of a component (the step's own and the ones behind the coupling
        nets' Conv2d layers) from the batch ``x``: what the reference's FIRST forward in training mode does, layer by layer
        (models/layers.py:473-486: bias = -mean, logs = log(scale / (sqrt(mean((x + bias)^2)) + 1e-6)) over (N, H, W), each layer
        seeing the outputs of the layers initialised before it).  The statistics come from the library
        (gbnf_image_flow_actnorm_stats, exact-f32 kernels); layers already initialised are left alone.  ``components``: an index,
        a list, or None = all.  ``noise``: the dequantisation noise (None: drawn once, as the reference's forward would)."""
        self._check(x)
        x = x.contiguous().float()
        noise = torch.rand_like(x) if noise is None else noise.contiguous().float()
        comps = range(self.num_components) if components is None else ([int(components)] if isinstance(components, int) else list(components))
        for c in comps:
            glow = self.flows[c]
            acts = glow._actnorms()
            for idx, m in enumerate(acts):
                if m.inited:
                    continue
                # every layer not yet initialised is the identity (bias = logs = 0): mark all as usable for packing, pack on exact f32
                flags = [a.inited for a in acts]
                for a in acts:
                    a.inited = True
                try:
                    with torch.cuda.device(x.device):
                        handle = native.NativeImageFlow(image_spec_from_glow_module(glow), math="f32")
                        mean, var = handle.actnorm_stats(x, noise, idx)
                finally:
                    for a, fl in zip(acts, flags):
                        a.inited = fl
                if mean.numel() != m.num_features:
                    raise native.GbnfError(f"ActNorm2d {idx} of component {c} has {m.num_features} channels, the library counted {mean.numel()}")
                m.bias.data.copy_((-mean).view_as(m.bias))
                m.logs.data.copy_(torch.log(float(m.scale) / (torch.sqrt(var) + 1e-6)).view_as(m.logs))
                m.inited = True
        self._handles.clear()

    def component_log_prob(self, x, n_used=None, noise=None):
        """(N, C_used): ll_c = log_normal_diag(z, z_mu, z_var) + logdet (image_experiment.py:227); the SAME noise for every
        component unless the caller passes one per call."""
        n_used = self.num_components if n_used is None else int(n_used)
        if noise is None:
            noise = torch.rand_like(x.float())
        if n_used == 1:
            return self.component_forward(x, 0, noise, want_z=False)[2].unsqueeze(1)
        # the components are independent until the recursion and a component is a chain of ~50 latency-bound launches:
        # one HIP stream per component lets the chains overlap (measured: +26 % at batch 64, +13 % at 256)
        self._check(x)
        x = x.contiguous().float()
        noise = noise.contiguous().float()
        streams = self.__dict__.setdefault("_streams", {})
        cur = torch.cuda.current_stream(x.device)
        lls = []
        for c in range(n_used):
            st = streams.get((x.device, c))
            if st is None:
                st = streams[(x.device, c)] = torch.cuda.Stream(x.device)
            st.wait_stream(cur)
            with torch.cuda.device(x.device), torch.cuda.stream(st):
                ll = self.native_flow(c).forward(x, noise, want_z=False)[2]
            ll.record_stream(cur)
            lls.append(ll)
        for c in range(n_used):
            cur.wait_stream(streams[(x.device, c)])
        return torch.stack(lls, dim=1)

    def graphed_log_prob(self, batch, n_used=None):
        """-> ``f(x, noise) -> G (N,)``: ``log_prob`` for batches of ``batch`` images captured ONCE in a HIP graph and
        replayed (static input / output buffers, the components' streams forked and joined inside the graph).  A component is a
        chain of ~50 short launches, so an evaluation call is host-bound as plain stream launches: the replay is 12 % faster at
        batch 256 (tools/bench_image.py, 62 -> 69.5 k images/s).  The graph holds the packed handles of the parameters it was
        captured with: ``f`` re-captures by itself when a parameter, permutation or ActNorm flag has changed since.  The
        returned tensor is the graph's output buffer -- overwritten by the next call of ``f``.  (Numerics protocol, include/gbnf.h
        gbnf_image_flow_numerics: the range marks, the repair launch and the periodic on-data check are part of every call's
        launch sequence and are counted on the device, so a replay repairs an out-of-range image exactly like a stream call; when
        an on-data check has demoted a handle, ``f`` re-captures so that the graph runs the exact-f32 kernels directly instead of
        the split-f16 pass + full repair.)"""
        n_used = self.num_components if n_used is None else int(n_used)
        dev = self.rho.device
        state = {}

        def handles():
            return [self.native_flow(c) for c in range(n_used)]

        def capture():
            c_, h_, w_ = self.flows[0].input_size
            xs = torch.zeros((batch, c_, h_, w_), dtype=torch.float32, device=dev)
            ns = torch.zeros_like(xs)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():
                self.log_prob(xs, n_used, ns)                    # warm-up: stream creation, first-use attributes
            torch.cuda.current_stream(dev).wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                out = self.log_prob(xs, n_used, ns)
            # the handles the graph's kernels read are HELD here: a re-pack cannot free them (and hand their address, hence their
            # id(), to a later handle) while this graph may still be replayed
            state.update(handles=handles(), graph=g, x=xs, noise=ns, out=out, demoted=[bool(h.numerics().demoted) for h in handles()])

        def f(x, noise):
            if tuple(x.shape) != (batch,) + tuple(self.flows[0].input_size) or tuple(noise.shape) != tuple(x.shape):
                raise native.GbnfError(f"graphed_log_prob was captured for batches of shape {(batch,) + tuple(self.flows[0].input_size)}: "
                                       f"got x {tuple(x.shape)}, noise {tuple(noise.shape)}")
            held = state.get("handles")
            now = handles()
            if (held is None or len(held) != len(now) or any(a is not b for a, b in zip(held, now))
                    or state["demoted"] != [bool(h.numerics().demoted) for h in now]):
                capture()
            state["x"].copy_(x)
            state["noise"].copy_(noise)
            state["graph"].replay()
            return state["out"]

        return f

    def log_prob(self, x, n_used=None, noise=None):
        """(N,): log mixture density with the reference's recursion over rho (density_experiment.py:561-573)."""
        ll = self.component_log_prob(x, n_used, noise)
        with torch.cuda.device(x.device):
            return native.mixture_lse(ll.t().contiguous(), self.rho)
