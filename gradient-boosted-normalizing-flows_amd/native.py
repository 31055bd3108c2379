"""ctypes binding of ``libgbnf_hip.so`` (C ABI: ``include/gbnf.h``).

There is NO fallback: if the shared library is missing, or a call fails, this
module raises.  Device buffers are PyTorch-ROCm tensors; only their
``data_ptr()`` and the current HIP stream cross the boundary.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GBNF_LIB_PATH: diagnostic builds only (e.g. the -DGBNF_STAMPS library of tools/build_stamps.sh)
LIB_PATH = os.environ.get("GBNF_LIB_PATH") or os.path.join(_HERE, "libgbnf_hip.so")

KIND = {"glow": 0, "realnvp": 1}
ACT = {"tanh": 0, "relu": 1, "residual": 2}     # GBNF_ACT_TANH / _RELU / _RESIDUAL_RELU
COUPLING = {"affine": 0, "additive": 1}
MATH = {"default": -1, "f32": 0, "f16x3": 1, "bf16x6": 2}
MATH_NAME = {v: k for k, v in MATH.items()}

# every symbol include/gbnf.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = (
    "gbnf_version", "gbnf_last_error", "gbnf_saturation_count", "gbnf_training_saturation_count",
    "gbnf_flow_create", "gbnf_flow_create_mode", "gbnf_flow_create_ex", "gbnf_flow_destroy", "gbnf_flow_info", "gbnf_flow_forward",
    "gbnf_flow_inverse",
    "gbnf_mixture_create", "gbnf_mixture_destroy", "gbnf_mixture_set_base",
    "gbnf_mixture_component_log_prob", "gbnf_mixture_component_log_prob_strided",
    "gbnf_mixture_component_log_prob_multi", "gbnf_mixture_component_forward", "gbnf_mixture_lse",
    "gbnf_mixture_log_prob",
    "gbnf_actnorm_init", "gbnf_boosting_weights",
    "gbnf_flow_validate", "gbnf_trainer_create", "gbnf_trainer_destroy", "gbnf_trainer_forward",
    "gbnf_trainer_grad_floats", "gbnf_trainer_workspace_bytes", "gbnf_trainer_backward", "gbnf_trainer_trace_floats",
    "gbnf_trainer_bind_batch_stats", "gbnf_trainer_set_batch_stats",
    "gbnf_image_flow_create", "gbnf_image_flow_destroy", "gbnf_image_flow_info", "gbnf_image_flow_workspace_bytes",
    "gbnf_image_flow_forward", "gbnf_image_flow_prior", "gbnf_image_flow_eps_floats", "gbnf_image_flow_inverse",
    "gbnf_image_flow_numerics", "gbnf_image_flow_repair_counts", "gbnf_image_flow_create_mode", "gbnf_image_flow_actnorm_stats",
    "gbnf_comm_unique_id", "gbnf_comm_create", "gbnf_comm_destroy", "gbnf_comm_info", "gbnf_mixture_group_log_prob",
    "gbnf_group_graph_create", "gbnf_group_graph_launch", "gbnf_group_graph_destroy",
    "gbnf_flow_numerics", "gbnf_mixture_numerics", "gbnf_tuning_set", "gbnf_tuning_get",
)


class GbnfError(RuntimeError):
    pass


class _Linear(C.Structure):
    _fields_ = [("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("out_features", C.c_int32), ("in_features", C.c_int32)]


class _Net(C.Structure):
    _fields_ = [("activation", C.c_int32), ("n_layers", C.c_int32), ("layers", C.POINTER(_Linear))]


class _GlowStep(C.Structure):
    _fields_ = [("actnorm_bias", C.POINTER(C.c_float)), ("actnorm_logs", C.POINTER(C.c_float)),
                ("perm_indices", C.POINTER(C.c_int64)), ("block", _Net)]


class _RealNVPStep(C.Structure):
    _fields_ = [("flipped", C.c_int32), ("has_batch_norm", C.c_int32),
                ("bn_log_gamma", C.POINTER(C.c_float)), ("bn_beta", C.POINTER(C.c_float)),
                ("bn_running_mean", C.POINTER(C.c_float)), ("bn_running_var", C.POINTER(C.c_float)),
                ("bn_eps", C.c_float), ("t_net", _Net), ("s_net", _Net)]


class _FlowDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("d", C.c_int32), ("n_steps", C.c_int32), ("coupling", C.c_int32),
                ("glow_steps", C.POINTER(_GlowStep)), ("realnvp_steps", C.POINTER(_RealNVPStep))]


class _Conv(C.Structure):
    _fields_ = [("weight", C.POINTER(C.c_float)), ("bias", C.POINTER(C.c_float)),
                ("actnorm_bias", C.POINTER(C.c_float)), ("actnorm_logs", C.POINTER(C.c_float)),
                ("logs", C.POINTER(C.c_float)),
                ("out_channels", C.c_int32), ("in_channels", C.c_int32), ("kernel_size", C.c_int32)]


class _ImageStep(C.Structure):
    _fields_ = [("actnorm_bias", C.POINTER(C.c_float)), ("actnorm_logs", C.POINTER(C.c_float)),
                ("perm_weight", C.POINTER(C.c_float)), ("perm_indices", C.POINTER(C.c_int64)),
                ("n_convs", C.c_int32), ("convs", C.POINTER(_Conv))]


class _ImageLevel(C.Structure):
    _fields_ = [("n_steps", C.c_int32), ("steps", C.POINTER(_ImageStep)), ("split_prior", C.POINTER(_Conv))]


class _ImageFlowDesc(C.Structure):
    _fields_ = [("channels", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("n_levels", C.c_int32),
                ("coupling", C.c_int32), ("hidden", C.c_int32), ("bounds", C.c_float),
                ("levels", C.POINTER(_ImageLevel)), ("learn_top", C.POINTER(_Conv))]


class KernelInfo(C.Structure):
    _fields_ = [("hidden_tiles", C.c_int32), ("out_tiles", C.c_int32), ("samples_per_wave", C.c_int32),
                ("n_steps", C.c_int32), ("macs_per_sample", C.c_double),
                ("padded_macs_per_sample", C.c_double), ("packed_bytes", C.c_int64),
                ("math_mode", C.c_int32), ("probe_rel_err", C.c_float)]


class NumericsStatus(C.Structure):
    """gbnf_numerics_status: the library's own re-check of a DEFAULT handle's f16x3 choice on the caller's data."""
    _fields_ = [("math_mode", C.c_int32), ("demoted", C.c_int32), ("checks", C.c_int64),
                ("worst_rel_err", C.c_float), ("tolerance", C.c_float)]


_lib = None


def tuning_set(key, value):
    """Launch-policy knob of the library (include/gbnf.h, gbnf_tuning_set): tests, soak runs, tuning."""
    _check(lib().gbnf_tuning_set(key.encode(), int(value)))


def tuning_get(key):
    v = C.c_int32()
    _check(lib().gbnf_tuning_get(key.encode(), C.byref(v)))
    return v.value


def lib():
    """Load the library (once).  Raises if it has not been built -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GbnfError(
            f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
            "(or gradient-boosted-normalizing-flows_amd/csrc/build.py). There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    L.gbnf_version.restype = C.c_int
    L.gbnf_last_error.restype = C.c_char_p
    L.gbnf_flow_create.argtypes = [C.POINTER(_FlowDesc), C.POINTER(vp)]
    L.gbnf_saturation_count.argtypes = [C.POINTER(C.c_int64), i32]
    L.gbnf_training_saturation_count.argtypes = [C.POINTER(C.c_int64), i32]
    L.gbnf_flow_create_mode.argtypes = [C.POINTER(_FlowDesc), i32, C.POINTER(vp)]
    L.gbnf_flow_create_ex.argtypes = [C.POINTER(_FlowDesc), i32, i32, C.POINTER(vp)]
    L.gbnf_flow_destroy.argtypes = [vp]
    L.gbnf_flow_info.argtypes = [vp, C.POINTER(KernelInfo)]
    L.gbnf_flow_forward.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.gbnf_flow_inverse.argtypes = [vp, vp, i64, vp, vp, vp]
    L.gbnf_mixture_create.argtypes = [C.POINTER(vp), i32, C.POINTER(vp)]
    L.gbnf_mixture_destroy.argtypes = [vp]
    L.gbnf_mixture_set_base.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gbnf_mixture_component_log_prob.argtypes = [vp, vp, i64, i32, i32, vp, vp]
    L.gbnf_mixture_component_log_prob_strided.argtypes = [vp, vp, i64, i32, i32, vp, i64, vp]
    L.gbnf_mixture_component_log_prob_multi.argtypes = [vp, C.POINTER(vp), i32, i64, i32, i32, vp, i64, vp]
    L.gbnf_mixture_component_forward.argtypes = [vp, vp, i64, i32, i32, vp, vp, vp, vp]
    L.gbnf_mixture_lse.argtypes = [vp, i64, vp, i32, i64, vp, vp]
    L.gbnf_mixture_log_prob.argtypes = [vp, vp, i64, i32, vp, vp, vp, vp]
    L.gbnf_actnorm_init.argtypes = [vp, i64, i32, C.c_float, vp, vp, vp]
    L.gbnf_boosting_weights.argtypes = [vp, i64, C.c_float, vp, vp]
    L.gbnf_flow_validate.argtypes = [C.POINTER(_FlowDesc)]
    L.gbnf_flow_numerics.argtypes = [vp, C.POINTER(NumericsStatus)]
    L.gbnf_mixture_numerics.argtypes = [vp, C.POINTER(NumericsStatus)]
    L.gbnf_tuning_set.argtypes = [C.c_char_p, i32]
    L.gbnf_tuning_get.argtypes = [C.c_char_p, C.POINTER(i32)]
    L.gbnf_trainer_create.argtypes = [C.POINTER(_FlowDesc), C.POINTER(vp)]
    L.gbnf_trainer_destroy.argtypes = [vp]
    L.gbnf_trainer_forward.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.gbnf_trainer_trace_floats.argtypes = [vp, i64, C.POINTER(i64)]
    L.gbnf_trainer_bind_batch_stats.argtypes = [vp, i32, vp, vp]
    L.gbnf_trainer_set_batch_stats.argtypes = [vp, i32]
    L.gbnf_trainer_grad_floats.argtypes = [vp, C.POINTER(i64)]
    L.gbnf_trainer_workspace_bytes.argtypes = [vp, i64, C.POINTER(i64)]
    L.gbnf_trainer_backward.argtypes = [vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, vp]
    L.gbnf_image_flow_create.argtypes = [C.POINTER(_ImageFlowDesc), C.POINTER(vp)]
    L.gbnf_image_flow_create_mode.argtypes = [C.POINTER(_ImageFlowDesc), i32, C.POINTER(vp)]
    L.gbnf_image_flow_actnorm_stats.argtypes = [vp, vp, vp, i64, i32, vp, vp, C.POINTER(i32), vp, i64, vp]
    L.gbnf_image_flow_destroy.argtypes = [vp]
    L.gbnf_image_flow_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(C.c_double)]
    L.gbnf_image_flow_workspace_bytes.argtypes = [vp, i64, C.POINTER(i64)]
    L.gbnf_image_flow_forward.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, i64, vp]
    L.gbnf_image_flow_prior.argtypes = [vp, C.POINTER(C.c_float)]
    L.gbnf_image_flow_numerics.argtypes = [vp, C.POINTER(NumericsStatus)]
    L.gbnf_image_flow_repair_counts.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(C.c_float)]
    L.gbnf_comm_unique_id.argtypes = [C.POINTER(C.c_uint8)]
    L.gbnf_comm_create.argtypes = [C.POINTER(C.c_uint8), i32, i32, C.POINTER(vp)]
    L.gbnf_comm_destroy.argtypes = [vp]
    L.gbnf_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.gbnf_mixture_group_log_prob.argtypes = [vp, vp, C.POINTER(vp), i32, i64, i32, vp, vp, vp, vp, vp]
    L.gbnf_group_graph_create.argtypes = [vp, vp, C.POINTER(vp), i32, i64, i32, vp, vp, vp, vp, C.POINTER(vp)]
    L.gbnf_group_graph_launch.argtypes = [vp, vp]
    L.gbnf_group_graph_destroy.argtypes = [vp]
    L.gbnf_image_flow_eps_floats.argtypes = [vp, C.POINTER(i64)]
    L.gbnf_image_flow_inverse.argtypes = [vp, vp, vp, C.c_float, i64, vp, vp, i64, vp]
    for name in ABI_SYMBOLS:
        if name not in ("gbnf_version", "gbnf_last_error"):
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise GbnfError(f"libgbnf_hip error {rc}: {lib().gbnf_last_error().decode(errors='replace')}")


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class _Keep:
    """Keeps the numpy arrays / ctypes arrays referenced by a descriptor alive."""

    def __init__(self):
        self.refs = []

    def f32(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        self.refs.append(a)
        return _fptr(a)

    def i64(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.int64))
        self.refs.append(a)
        return a.ctypes.data_as(C.POINTER(C.c_int64))

    def net(self, net):
        arr = (_Linear * len(net["layers"]))()
        for k, (w, b) in enumerate(net["layers"]):
            w = np.asarray(w)
            arr[k] = _Linear(self.f32(w), self.f32(b), int(w.shape[0]), int(w.shape[1]))
        self.refs.append(arr)
        return _Net(ACT[net["act"]], len(net["layers"]), arr)


def flow_desc_from_spec(spec):
    """flow spec (see spec.py) -> (ctypes gbnf_flow_desc, keep-alive object)."""
    keep = _Keep()
    K = len(spec["steps"])
    desc = _FlowDesc()
    desc.kind = KIND[spec["kind"]]
    desc.d = int(spec["d"])
    desc.n_steps = K
    desc.coupling = COUPLING[spec.get("coupling") or "affine"]
    if spec["kind"] == "glow":
        steps = (_GlowStep * K)()
        for k, st in enumerate(spec["steps"]):
            steps[k] = _GlowStep(keep.f32(st["an_bias"]), keep.f32(st["an_logs"]), keep.i64(st["perm"]),
                                 keep.net(st["net"]))
        desc.glow_steps = steps
    else:
        steps = (_RealNVPStep * K)()
        for k, st in enumerate(spec["steps"]):
            s = _RealNVPStep()
            s.flipped = int(bool(st["flipped"]))
            bn = st["bn"]
            s.has_batch_norm = int(bn is not None)
            if bn is not None:
                s.bn_log_gamma = keep.f32(bn["log_gamma"])
                s.bn_beta = keep.f32(bn["beta"])
                s.bn_running_mean = keep.f32(bn["running_mean"])
                s.bn_running_var = keep.f32(bn["running_var"])
                s.bn_eps = float(bn["eps"])
            s.t_net = keep.net(st["t_net"])
            s.s_net = keep.net(st["s_net"])
            steps[k] = s
        desc.realnvp_steps = steps
    keep.refs.append(steps)
    return desc, keep


def _require_device_f32(t, name):
    import torch
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GbnfError(f"{name} must be a tensor on the MI355X (cuda) device; there is no CPU path")
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise GbnfError(f"{name} must be contiguous float32")
    return t


def _stream_ptr():
    """The current HIP stream of the current device as a void*.  torch.cuda.current_stream() builds a Stream object through
    three Python layers (15 us per call, a fifth of a module call's host time); the raw-handle accessor is one C call."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeFlow:
    """One packed component on the device (gbnf_flow)."""

    def __init__(self, spec, math="default", per_step_activation=False):
        """``per_step_activation``: pack for the kernel variants that read the activation per step although this
        component uses one activation throughout (GBNF_CREATE_PER_STEP_ACTIVATION): lets it share a NativeMixture with
        components that do not (`--coupling_network random`); see ``flows_for_mixture``."""
        desc, keep = flow_desc_from_spec(spec)
        h = C.c_void_p()
        _check(lib().gbnf_flow_create_ex(C.byref(desc), MATH[math], 1 if per_step_activation else 0, C.byref(h)))
        del keep
        self.handle = h
        self.d = int(spec["d"])

    def info(self):
        ki = KernelInfo()
        _check(lib().gbnf_flow_info(self.handle, C.byref(ki)))
        return ki

    def numerics(self):
        """gbnf_flow_numerics: (math mode of the next launch, demoted?, completed checks, worst relative error); no sync."""
        st = NumericsStatus()
        _check(lib().gbnf_flow_numerics(self.handle, C.byref(st)))
        return st

    def forward(self, x, want_z=True, want_ldj=True, want_ll=False):
        """x (n,d) cuda f32 -> (z|None, ldj|None, ll|None); enqueued on the current stream."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 2 or x.shape[1] != self.d:
            raise GbnfError(f"x must be (n,{self.d}), got {tuple(x.shape)}")
        n = x.shape[0]
        z = torch.empty_like(x) if want_z else None
        ldj = torch.empty(n, dtype=torch.float32, device=x.device) if want_ldj else None
        ll = torch.empty(n, dtype=torch.float32, device=x.device) if want_ll else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)
        _check(lib().gbnf_flow_forward(self.handle, ptr(x), n, ptr(z), ptr(ldj), ptr(ll), _stream_ptr()))
        return z, ldj, ll

    def inverse(self, z, want_ldj=True):
        """z (n,d) -> (x (n,d), log|det dx/dz| (n,) | None): the flow's steps last to first (gbnf_flow_inverse), in the handle's math
        mode (the same kernels under a runtime flag)."""
        import torch
        _require_device_f32(z, "z")
        if z.dim() != 2 or z.shape[1] != self.d:
            raise GbnfError(f"z must be (n,{self.d}), got {tuple(z.shape)}")
        n = z.shape[0]
        x = torch.empty_like(z)
        ldj = torch.empty(n, dtype=torch.float32, device=z.device) if want_ldj else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)
        _check(lib().gbnf_flow_inverse(self.handle, ptr(z), n, ptr(x), ptr(ldj), _stream_ptr()))
        return x, ldj

    def close(self):
        if getattr(self, "handle", None):
            lib().gbnf_flow_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def image_flow_desc_from_spec(spec):
    """image flow spec (synth.synth_image_glow_spec / spec.image_spec_from_glow_module) -> (ctypes gbnf_image_flow_desc, keep-alive object)."""
    keep = _Keep()

    def conv(c):
        w = np.asarray(c["w"])
        cc = _Conv()
        cc.weight = keep.f32(w)
        for field, key in (("bias", "b"), ("actnorm_bias", "an_bias"), ("actnorm_logs", "an_logs"), ("logs", "logs")):
            if c[key] is not None:
                setattr(cc, field, keep.f32(np.asarray(c[key]).reshape(-1)))
        cc.out_channels, cc.in_channels, cc.kernel_size = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        return cc

    desc = _ImageFlowDesc()
    desc.channels, desc.height, desc.width = (int(v) for v in spec["input_size"])
    desc.n_levels = len(spec["levels"])
    desc.coupling = COUPLING[spec.get("coupling") or "affine"]
    desc.hidden = int(spec["hidden"])
    desc.bounds = float(spec.get("bounds", 0.9))
    levels = (_ImageLevel * desc.n_levels)()
    for l, lv in enumerate(spec["levels"]):
        steps = (_ImageStep * len(lv["steps"]))()
        for k, st in enumerate(lv["steps"]):
            s_ = _ImageStep()
            s_.actnorm_bias = keep.f32(np.asarray(st["an_bias"]).reshape(-1))
            s_.actnorm_logs = keep.f32(np.asarray(st["an_logs"]).reshape(-1))
            if st.get("perm_w") is not None:
                s_.perm_weight = keep.f32(st["perm_w"])
            else:
                s_.perm_indices = keep.i64(st["perm"])
            arr = (_Conv * len(st["convs"]))(*[conv(c) for c in st["convs"]])
            keep.refs.append(arr)
            s_.n_convs, s_.convs = len(st["convs"]), arr
            steps[k] = s_
        keep.refs.append(steps)
        levels[l].n_steps, levels[l].steps = len(lv["steps"]), steps
        if lv["split"] is not None:
            sp = conv(lv["split"])
            keep.refs.append(sp)
            levels[l].split_prior = C.pointer(sp)
    desc.levels = levels
    if spec.get("learn_top") is not None:
        top = conv(spec["learn_top"])
        keep.refs.append(top)
        desc.learn_top = C.pointer(top)
    return desc, keep


class NativeImageFlow:
    """One packed image Glow component (gbnf_image_flow).  ``spec``: the image flow spec of ``synth.synth_image_glow_spec``
    / ``spec.image_spec_from_glow_module`` (numpy arrays)."""

    def __init__(self, spec, math="default"):
        desc, keep = image_flow_desc_from_spec(spec)
        h = C.c_void_p()
        if math not in ("default", "f32", "f16x3"):
            raise GbnfError(f"image components run in math mode default | f32 | f16x3, not {math!r}")
        _check(lib().gbnf_image_flow_create_mode(C.byref(desc), MATH[math], C.byref(h)))
        del keep
        self.handle = h
        self.input_size = tuple(int(v) for v in spec["input_size"])
        self.n_levels = len(spec["levels"])
        zc, zh, zw, macs = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
        _check(lib().gbnf_image_flow_info(h, C.byref(zc), C.byref(zh), C.byref(zw), C.byref(macs)))
        self.z_shape = (zc.value, zh.value, zw.value)
        self.macs_per_image = macs.value
        self._ws = None

    def numerics(self):
        """gbnf_image_flow_numerics: math mode the handle runs in, the create-time probe's verdict (worst_rel_err, demoted),
        calls that repaired an out-of-range image so far (checks); no synchronisation."""
        st = NumericsStatus()
        _check(lib().gbnf_image_flow_numerics(self.handle, C.byref(st)))
        return st

    def repair_counts(self):
        """gbnf_image_flow_repair_counts: what the same-call protocol behind the split-f16 pass has done so far (pinned
        words, no synchronisation): calls that marked an image, images re-evaluated on exact f32, on-data checks completed /
        failed, the worst relative difference a check has seen."""
        a, b, c, d, w = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64(), C.c_float()
        _check(lib().gbnf_image_flow_repair_counts(self.handle, C.byref(a), C.byref(b), C.byref(c), C.byref(d), C.byref(w)))
        return {"marked_calls": a.value, "repaired_images": b.value, "data_checks": c.value, "failed_checks": d.value,
                "worst_check_rel_err": w.value}

    def actnorm_stats(self, x, noise, index):
        """gbnf_image_flow_actnorm_stats: (mean (C,), var (C,)) device tensors of the tensor that reaches ActNorm2d number
        ``index`` (module order) on the batch -- the two numbers the reference's data-dependent initialisation needs
        (models/layers.py:473-486)."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 4 or tuple(x.shape[1:]) != self.input_size:
            raise GbnfError(f"x must be (n,{self.input_size}), got {tuple(x.shape)}")
        if noise is not None:
            _require_device_f32(noise, "noise")
        n = x.shape[0]
        nb = C.c_int64()
        _check(lib().gbnf_image_flow_workspace_bytes(self.handle, n, C.byref(nb)))
        if self._ws is None or self._ws.numel() * 4 < nb.value or self._ws.device != x.device:
            self._ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=x.device)
        mean = torch.empty(512, dtype=torch.float32, device=x.device)
        var = torch.empty(512, dtype=torch.float32, device=x.device)
        ch = C.c_int32()
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
        _check(lib().gbnf_image_flow_actnorm_stats(self.handle, ptr(x), ptr(noise), n, int(index), ptr(mean), ptr(var), C.byref(ch),
                                                   ptr(self._ws), self._ws.numel() * 4, _stream_ptr()))
        return mean[: ch.value], var[: ch.value]

    def prior(self):
        """(mean (Cz,), log-variance (Cz,)) of the top prior, numpy."""
        buf = (C.c_float * (2 * self.z_shape[0]))()
        _check(lib().gbnf_image_flow_prior(self.handle, buf))
        a = np.frombuffer(buf, dtype=np.float32).copy()
        return a[: self.z_shape[0]], a[self.z_shape[0]:]

    def forward(self, x, noise=None, want_z=True):
        """x (n,C,H,W) in [0,1] (+ dequantisation noise) -> (z | None, ldj (n,), ll (n,)); current stream."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 4 or tuple(x.shape[1:]) != self.input_size:
            raise GbnfError(f"x must be (n,{self.input_size}), got {tuple(x.shape)}")
        if noise is not None:
            _require_device_f32(noise, "noise")
            if noise.shape != x.shape:
                raise GbnfError("noise must have the shape of x")
        n = x.shape[0]
        z = torch.empty((n,) + self.z_shape, dtype=torch.float32, device=x.device) if want_z else None
        ldj = torch.empty(n, dtype=torch.float32, device=x.device)
        ll = torch.empty(n, dtype=torch.float32, device=x.device)
        if n:
            nb = C.c_int64()
            _check(lib().gbnf_image_flow_workspace_bytes(self.handle, n, C.byref(nb)))
            if self._ws is None or self._ws.numel() * 4 < nb.value or self._ws.device != x.device:
                self._ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=x.device)
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            _check(lib().gbnf_image_flow_forward(self.handle, ptr(x), ptr(noise), n, ptr(z), ptr(ldj), ptr(ll),
                                                 ptr(self._ws), self._ws.numel() * 4, _stream_ptr()))
        return z, ldj, ll

    def split_shapes(self):
        """(C_l/2, H_l, W_l) of the half each Split2d level drops (and re-draws on the way back), first level first."""
        c, h, w = self.input_size
        out = []
        for l in range(self.n_levels):
            c, h, w = c * 4, h // 2, w // 2
            if l < self.n_levels - 1:
                out.append((c // 2, h, w))
                c //= 2
        return out

    def inverse(self, z, eps=None, temperature=1.0):
        """z (n,Cz,Hz,Wz) -> x (n,C,H,W): the reference's Glow.decode(z, None, temperature) (gbnf_image_flow_inverse).
        ``eps``: one standard-normal tensor (n, C_l/2, H_l, W_l) per Split2d level, first level first (``split_shapes``);
        None draws them from torch's generator on z's device."""
        import torch
        _require_device_f32(z, "z")
        if z.dim() != 4 or tuple(z.shape[1:]) != self.z_shape:
            raise GbnfError(f"z must be (n,{self.z_shape}), got {tuple(z.shape)}")
        n = z.shape[0]
        shapes = self.split_shapes()
        if eps is None:
            eps = [torch.randn((n,) + sh, dtype=torch.float32, device=z.device) for sh in shapes]
        if len(eps) != len(shapes):
            raise GbnfError(f"eps needs {len(shapes)} tensors, got {len(eps)}")
        for e, sh in zip(eps, shapes):
            _require_device_f32(e, "eps")
            if tuple(e.shape) != (n,) + sh:
                raise GbnfError(f"eps tensor must be {(n,) + sh}, got {tuple(e.shape)}")
        per = C.c_int64()
        _check(lib().gbnf_image_flow_eps_floats(self.handle, C.byref(per)))
        flat = torch.cat([e.reshape(-1) for e in eps]) if eps else None
        if flat is not None and flat.numel() != per.value * n:
            raise GbnfError("eps size disagrees with gbnf_image_flow_eps_floats")
        x = torch.empty((n,) + self.input_size, dtype=torch.float32, device=z.device)
        if n:
            nb = C.c_int64()
            _check(lib().gbnf_image_flow_workspace_bytes(self.handle, n, C.byref(nb)))
            if self._ws is None or self._ws.numel() * 4 < nb.value or self._ws.device != z.device:
                self._ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=z.device)
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            _check(lib().gbnf_image_flow_inverse(self.handle, ptr(z.contiguous()), ptr(flat), float(temperature), n, ptr(x),
                                                 ptr(self._ws), self._ws.numel() * 4, _stream_ptr()))
        return x

    def close(self):
        if getattr(self, "handle", None):
            lib().gbnf_image_flow_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeTrainer:
    """Training path of one component (gbnf_trainer): forward on the LIVE device parameters and the backward pass.

    ``dev_spec`` has the shape of a flow spec (spec.py) but every float array is a contiguous float32 CUDA tensor --
    the caller's parameter / buffer storage itself (``perm`` stays a host int64 array).  Nothing is copied: the
    library keeps the addresses, this object keeps the tensors alive."""

    def __init__(self, dev_spec):
        import torch
        self._tensors = []        # keep-alive; also the identity check for re-creation
        self.params = []          # tensors in the order of the flat gradient buffer (None = reserved, unused region)
        self._sizes = []
        keep = []
        K = len(dev_spec["steps"])
        self.d = d = int(dev_spec["d"])
        fp = C.POINTER(C.c_float)

        def dptr(t, numel):
            _require_device_f32(t, "parameter")
            if t.numel() != numel:
                raise GbnfError(f"parameter has {t.numel()} elements, expected {numel}")
            self._tensors.append(t)
            return C.cast(C.c_void_p(t.data_ptr()), fp)

        def region(t, numel):
            self.params.append(t)
            self._sizes.append(numel)

        def net(n):
            arr = (_Linear * len(n["layers"]))()
            for k, (w, b) in enumerate(n["layers"]):
                out_f, in_f = int(w.shape[0]), int(w.shape[1])
                arr[k] = _Linear(dptr(w, out_f * in_f), dptr(b, out_f), out_f, in_f)
                region(w, out_f * in_f)
                region(b, out_f)
            keep.append(arr)
            return _Net(ACT[n["act"]], len(n["layers"]), arr)

        desc = _FlowDesc()
        desc.kind = KIND[dev_spec["kind"]]
        desc.d = d
        desc.n_steps = K
        desc.coupling = COUPLING[dev_spec.get("coupling") or "affine"]
        hk = _Keep()
        if dev_spec["kind"] == "glow":
            steps = (_GlowStep * K)()
            for k, st in enumerate(dev_spec["steps"]):
                region(st["an_bias"], d)
                region(st["an_logs"], d)
                steps[k] = _GlowStep(dptr(st["an_bias"], d), dptr(st["an_logs"], d), hk.i64(st["perm"]), net(st["net"]))
            desc.glow_steps = steps
        else:
            steps = (_RealNVPStep * K)()
            for k, st in enumerate(dev_spec["steps"]):
                s = _RealNVPStep()
                s.flipped = int(bool(st["flipped"]))
                bn = st["bn"]
                s.has_batch_norm = int(bn is not None)
                if bn is not None:
                    s.bn_log_gamma = dptr(bn["log_gamma"], d)
                    s.bn_beta = dptr(bn["beta"], d)
                    s.bn_running_mean = dptr(bn["running_mean"], d)
                    s.bn_running_var = dptr(bn["running_var"], d)
                    s.bn_eps = float(bn["eps"])
                    region(bn["log_gamma"], d)
                    region(bn["beta"], d)
                else:
                    region(None, d)
                    region(None, d)
                s.t_net = net(st["t_net"])
                s.s_net = net(st["s_net"])
                steps[k] = s
            desc.realnvp_steps = steps
        h = C.c_void_p()
        _check(lib().gbnf_trainer_create(C.byref(desc), C.byref(h)))
        self.handle = h
        # BatchNorm on batch statistics (the reference's train() mode): bind the buffers that receive them
        self.has_batch_stats = False
        if dev_spec["kind"] == "realnvp":
            bound = 0
            for k, st in enumerate(dev_spec["steps"]):
                bn = st["bn"]
                if bn is not None and bn.get("batch_mean") is not None:
                    for t in (bn["batch_mean"], bn["batch_var"]):
                        _require_device_f32(t, "batch statistics buffer")
                        self._tensors.append(t)
                    _check(lib().gbnf_trainer_bind_batch_stats(h, k, C.c_void_p(bn["batch_mean"].data_ptr()),
                                                               C.c_void_p(bn["batch_var"].data_ptr())))
                    bound += 1
            n_bn = sum(1 for st in dev_spec["steps"] if st["bn"] is not None)
            self.has_batch_stats = bound == n_bn and n_bn > 0
        self.batch_stats = False
        nf = C.c_int64()
        _check(lib().gbnf_trainer_grad_floats(h, C.byref(nf)))
        self.grad_floats = int(nf.value)
        if self.grad_floats != sum(self._sizes):
            raise GbnfError("gradient-buffer layout mismatch between the library and the binding")
        self._ws = None

    def key(self):
        return tuple(t.data_ptr() for t in self._tensors)

    def set_batch_stats(self, on):
        """RealNVP BatchNorm on batch statistics (True = the reference's train() mode) or running statistics (False)."""
        on = bool(on)
        if on != self.batch_stats:
            _check(lib().gbnf_trainer_set_batch_stats(self.handle, int(on)))
            self.batch_stats = on

    def forward(self, x, want_trace=False):
        """-> (z, ldj) or (z, ldj, trace): ``trace`` holds every step's normalised state; passing it to ``backward``
        saves that call the forward sweep (valid while the parameters are unchanged)."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 2 or x.shape[1] != self.d:
            raise GbnfError(f"x must be (n,{self.d}), got {tuple(x.shape)}")
        n = x.shape[0]
        z = torch.empty_like(x)
        ldj = torch.empty(n, dtype=torch.float32, device=x.device)
        trace = None
        if want_trace:
            nf = C.c_int64()
            _check(lib().gbnf_trainer_trace_floats(self.handle, n, C.byref(nf)))
            trace = torch.empty(nf.value, dtype=torch.float32, device=x.device)
        if n:
            _check(lib().gbnf_trainer_forward(self.handle, C.c_void_p(x.data_ptr()), n, C.c_void_p(z.data_ptr()),
                                              C.c_void_p(ldj.data_ptr()),
                                              C.c_void_p(trace.data_ptr() if trace is not None else 0), _stream_ptr()))
        return (z, ldj, trace) if want_trace else (z, ldj)

    def backward(self, x, g_z=None, g_ldj=None, want_gx=False, trace=None):
        """-> (g_x | None, [gradient per entry of ``self.params`` (views of one flat buffer; None for reserved regions)])."""
        import torch
        _require_device_f32(x, "x")
        n = x.shape[0]
        for t, name in ((g_z, "g_z"), (g_ldj, "g_ldj")):
            if t is not None:
                _require_device_f32(t, name)
        flat = torch.zeros(self.grad_floats, dtype=torch.float32, device=x.device)
        g_x = torch.empty_like(x) if want_gx else None
        if n:
            nb = C.c_int64()
            _check(lib().gbnf_trainer_workspace_bytes(self.handle, n, C.byref(nb)))
            if self._ws is None or self._ws.numel() * 4 < nb.value or self._ws.device != x.device:
                self._ws = torch.empty((nb.value + 3) // 4, dtype=torch.float32, device=x.device)
            ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
            _check(lib().gbnf_trainer_backward(self.handle, ptr(x), n, ptr(trace), ptr(g_z), ptr(g_ldj), ptr(g_x), ptr(flat),
                                               ptr(self._ws), self._ws.numel() * 4, _stream_ptr()))
        grads, off = [], 0
        for t, size in zip(self.params, self._sizes):
            grads.append(None if t is None else flat[off:off + size].view(t.shape))
            off += size
        return g_x, grads

    def close(self):
        if getattr(self, "handle", None):
            lib().gbnf_trainer_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def saturation_count(reset=False):
    """How many waves stored a split-f16 operand beyond +-65504 (it saturates there) since the last reset: 0 for
    z-scored data on a trained flow.  Synchronises with the device (gbnf_saturation_count)."""
    n = C.c_int64(0)
    _check(lib().gbnf_saturation_count(C.byref(n), 1 if reset else 0))
    return int(n.value)


def training_saturation_count(reset=False):
    """The part of ``saturation_count`` that came from TRAINING launches (forward / backward sweeps of a trainer, its weight re-pack):
    they saturate and are not repaired -- non-zero means steps with wrong gradients (gbnf_training_saturation_count).  Synchronises."""
    n = C.c_int64(0)
    _check(lib().gbnf_training_saturation_count(C.byref(n), 1 if reset else 0))
    return int(n.value)


def activation_pattern(spec):
    """The activations of a component's coupling nets, step by step (a hashable)."""
    if spec["kind"] == "glow":
        return tuple(st["net"]["act"] for st in spec["steps"])
    return tuple((st["t_net"]["act"], st["s_net"]["act"]) for st in spec["steps"])


def needs_per_step_activation(specs):
    """True when the components cannot share one uniform-activation kernel: some component mixes activations, or two
    components use different ones (the reference's `--coupling_network random`, models/glow.py:295-296)."""
    pats = {activation_pattern(s) for s in specs}
    return len(pats) > 1 or any(len(set(p)) > 1 for p in pats)


def flows_for_mixture(specs, math="default"):
    """NativeFlow handles of `specs` that are guaranteed to fit one NativeMixture."""
    per_step = needs_per_step_activation(specs)
    return [NativeFlow(s, math=math, per_step_activation=per_step) for s in specs]


def mixture_from_specs(specs, math="default"):
    """(NativeMixture, [NativeFlow]) of `specs`.  A mixture is one launch on ONE kernel variant; when the components
    would individually end up on different variants (different activations, or a geometry only the per-step-activation
    supersets cover) all of them are created for those supersets."""
    flows = flows_for_mixture(specs, math=math)
    try:
        return NativeMixture(flows), flows
    except GbnfError:
        flows = [NativeFlow(s, math=math, per_step_activation=True) for s in specs]
        return NativeMixture(flows), flows


class NativeMixture:
    """C same-architecture components behind one launch (gbnf_mixture)."""

    def __init__(self, flows):
        self.flows = list(flows)           # keeps the NativeFlow handles alive
        arr = (C.c_void_p * len(self.flows))(*[f.handle for f in self.flows])
        h = C.c_void_p()
        _check(lib().gbnf_mixture_create(arr, len(self.flows), C.byref(h)))
        self.handle = h
        self.d = self.flows[0].d

    @property
    def n_components(self):
        return len(self.flows)

    def numerics(self):
        """gbnf_mixture_numerics: the mixture's effective math mode and the state of its numerics guard; no sync."""
        st = NumericsStatus()
        _check(lib().gbnf_mixture_numerics(self.handle, C.byref(st)))
        return st

    def set_base(self, mean=None, std=None):
        if mean is None:
            _check(lib().gbnf_mixture_set_base(self.handle, None, None))
            return
        m = np.ascontiguousarray(np.asarray(mean, dtype=np.float32))
        s = np.ascontiguousarray(np.asarray(std, dtype=np.float32))
        if m.shape != (self.d,) or s.shape != (self.d,):
            raise GbnfError(f"base mean/std must have shape ({self.d},)")
        _check(lib().gbnf_mixture_set_base(self.handle, _fptr(m), _fptr(s)))

    def component_log_prob(self, x, c_begin=0, c_end=None, out=None):
        """ll (c_end-c_begin, n) for components [c_begin, c_end) in one launch."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 2 or x.shape[1] != self.d:
            raise GbnfError(f"x must be (n,{self.d}), got {tuple(x.shape)}")
        c_end = self.n_components if c_end is None else c_end
        n = x.shape[0]
        if out is None:
            out = torch.empty((c_end - c_begin, n), dtype=torch.float32, device=x.device)
        else:
            _require_device_f32(out, "out")
            if tuple(out.shape) != (c_end - c_begin, n):
                raise GbnfError("out has the wrong shape")
        _check(lib().gbnf_mixture_component_log_prob(
            self.handle, C.c_void_p(x.data_ptr() if n else 0), n, c_begin, c_end,
            C.c_void_p(out.data_ptr() if out.numel() else 0), _stream_ptr()))
        return out

    def component_forward(self, x, c_begin=0, c_end=None, want_z=True, want_ldj=True, want_ll=False):
        """(z (C',n,d), ldj (C',n), ll (C',n)) of components [c_begin, c_end) in ONE launch (gbnf_mixture_component_forward):
        the reference's per-component calls ``model(x=x, components=c)`` of one batch, all at once."""
        import torch
        _require_device_f32(x, "x")
        if x.dim() != 2 or x.shape[1] != self.d:
            raise GbnfError(f"x must be (n,{self.d}), got {tuple(x.shape)}")
        c_end = self.n_components if c_end is None else c_end
        n, k = x.shape[0], c_end - c_begin
        z = torch.empty((k, n, self.d), dtype=torch.float32, device=x.device) if want_z else None
        ldj = torch.empty((k, n), dtype=torch.float32, device=x.device) if want_ldj else None
        ll = torch.empty((k, n), dtype=torch.float32, device=x.device) if want_ll else None
        if n and k:
            ptr = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
            _check(lib().gbnf_mixture_component_forward(self.handle, C.c_void_p(x.data_ptr()), n, c_begin, c_end,
                                                        ptr(z), ptr(ldj), ptr(ll), _stream_ptr()))
        return z, ldj, ll

    def prepared_component_log_prob(self, x, out, c_begin=0, c_end=None, col_offset=0):
        """Bind every argument once and return ``launch(stream_ptr)``: the per-call host cost is then one ctypes
        call (for latency-sensitive loops such as bench.py's pipeline).  ``out`` is a (c_end-c_begin, >= n) table;
        this batch's log-densities go to columns [col_offset, col_offset + n) (rows keep out's row stride)."""
        _require_device_f32(x, "x")
        if not out.is_cuda or out.dtype.is_floating_point is False or out.stride(1) != 1:
            raise GbnfError("out must be a float32 device table with unit column stride")
        c_end = self.n_components if c_end is None else c_end
        n = x.shape[0]
        if out.shape[0] != c_end - c_begin or out.shape[1] < col_offset + n or x.shape[1] != self.d:
            raise GbnfError("prepared_component_log_prob: shape mismatch")
        fn, h = lib().gbnf_mixture_component_log_prob_strided, self.handle
        stride = out.stride(0) if out.shape[0] > 1 else max(out.shape[1], n)
        xp = C.c_void_p(x.data_ptr())
        op = C.c_void_p(out.data_ptr() + 4 * col_offset)
        keep = (x, out)

        def launch(stream_ptr, _keep=keep):
            rc = fn(h, xp, n, c_begin, c_end, op, stride, stream_ptr)
            if rc:
                _check(rc)
        return launch

    def prepared_group_log_prob(self, xs, out, c_begin=0, c_end=None):
        """Bound launch of ONE kernel over a group of batches: xs = list of (n,d) device tensors (same n), ``out`` a
        (c_end-c_begin, len(xs)*n) table; batch b lands in columns [b*n, (b+1)*n).  Returns ``launch(stream_ptr)``."""
        c_end = self.n_components if c_end is None else c_end
        n = xs[0].shape[0]
        for t in xs:
            _require_device_f32(t, "x")
            if tuple(t.shape) != (n, self.d):
                raise GbnfError("all batches of a group must be (n, d) with the same n")
        _require_device_f32(out, "out")
        if tuple(out.shape) != (c_end - c_begin, len(xs) * n):
            raise GbnfError("out must be (components, len(xs) * n)")
        fn, h = lib().gbnf_mixture_component_log_prob_multi, self.handle
        arr = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        nb, stride, op = len(xs), len(xs) * n, C.c_void_p(out.data_ptr())
        keep = (list(xs), out, arr)

        def launch(stream_ptr, _keep=keep):
            rc = fn(h, arr, nb, n, c_begin, c_end, op, stride, stream_ptr)
            if rc:
                _check(rc)
        return launch

    def log_prob(self, x, rho, n_used=None, ll_out=None, out=None):
        """The measured path: G (n,) = mixture log-density over components [0, n_used)."""
        import torch
        _require_device_f32(x, "x")
        _require_device_f32(rho, "rho")
        n_used = self.n_components if n_used is None else int(n_used)
        n = x.shape[0]
        if ll_out is None:
            ll_out = torch.empty((n_used, n), dtype=torch.float32, device=x.device)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=x.device)
        if rho.numel() < n_used:
            raise GbnfError("rho shorter than n_used")
        _check(lib().gbnf_mixture_log_prob(
            self.handle, C.c_void_p(x.data_ptr() if n else 0), n, n_used, C.c_void_p(rho.data_ptr()),
            C.c_void_p(ll_out.data_ptr() if ll_out.numel() else 0),
            C.c_void_p(out.data_ptr() if n else 0), _stream_ptr()))
        return out, ll_out

    def close(self):
        if getattr(self, "handle", None):
            lib().gbnf_mixture_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """An RCCL communicator owned by the library (gbnf_comm): the exchange of the component-sharded group without
    torch.distributed on the data path.  ``Comm.from_torch_distributed()`` builds one for the ranks of an initialised process
    group (the 128-byte id travels through a broadcast of that group -- set-up only); ``Comm(0, 1, Comm.unique_id())`` is a
    one-rank communicator."""

    def __init__(self, rank, world, unique_id):
        if len(unique_id) != 128:
            raise GbnfError("an RCCL unique id has 128 bytes")
        buf = (C.c_uint8 * 128)(*unique_id)
        h = C.c_void_p()
        _check(lib().gbnf_comm_create(buf, int(rank), int(world), C.byref(h)))
        self.handle, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib().gbnf_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def probe():
        """Local, non-collective: can this process build a communicator at all (librccl loadable, its symbols bound, an id
        obtainable)?  -> (ok, reason)."""
        try:
            Comm.unique_id()
            return True, ""
        except Exception as e:                 # GbnfError (dlopen / UNSUPPORTED), OSError (library missing)
            return False, f"{type(e).__name__}: {e}"

    @classmethod
    def from_torch_distributed(cls, group=None):
        """The ranks AGREE before anything that can block: every rank probes locally, the flags meet in one
        all_reduce(MIN), and only then does rank 0 broadcast the id (always: ``None`` if it could not make one) and every rank
        enter ncclCommInitRank.  A rank that cannot load RCCL therefore makes every rank raise GbnfError instead of leaving its
        peers inside a collective."""
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            return cls(0, 1, cls.unique_id())
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        if world == 1:
            return cls(0, 1, cls.unique_id())
        ok, why = cls.probe()
        # the hand-shake tensor lives where the group can reduce it: a group whose backend string mentions nccl (plain "nccl", or a
        # composite like "cpu:gloo,cuda:nccl") takes a device tensor, anything else a host tensor
        use_cuda = "nccl" in str(dist.get_backend(group)).lower() and torch.cuda.is_available()
        dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise GbnfError("no library communicator: " + (why if not ok else "another rank cannot load RCCL"))
        uid = None
        if rank == 0:
            try:
                uid = cls.unique_id()
            except Exception:
                uid = None
        box = [uid]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if box[0] is None:
            raise GbnfError("no library communicator: rank 0 could not obtain an RCCL unique id")
        return cls(rank, world, box[0])

    def close(self):
        if getattr(self, "handle", None) is not None and _lib is not None:
            _lib.gbnf_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GroupLaunch:
    """One group of the component-sharded mixture bound to its buffers (gbnf_mixture_group_log_prob): ``mix`` = this rank's
    components, ``comm`` = a ``Comm`` or None (one rank, no exchange), ``xs`` = the group's resident (n, d) batches,
    ``ll_local`` (C/W, S n), ``ll_full`` (C, S n), ``G`` (S n).  ``launch(stream_ptr)`` enqueues flow launch -> all-gather ->
    recursion on that stream with ONE library call; with ``graph=True`` the sequence is captured once into a HIP graph and
    ``launch`` is a hipGraphLaunch (falls back to the plain call, with ``self.graph_error`` set, when the capture is refused)."""

    def __init__(self, mix, comm, xs, n_components, rho, ll_local, ll_full, G, graph=True):
        n = xs[0].shape[0]
        for t in list(xs) + [rho, ll_local, G] + ([ll_full] if ll_full is not None else []):
            _require_device_f32(t, "tensor")
        world = comm.world if comm is not None else 1
        if n_components % world or tuple(ll_local.shape) != (n_components // world, len(xs) * n) or G.numel() != len(xs) * n:
            raise GbnfError("GroupLaunch: buffer shapes do not match the group")
        if comm is not None and (ll_full is None or tuple(ll_full.shape) != (n_components, len(xs) * n)):
            raise GbnfError("GroupLaunch: ll_full must be (C, len(xs) * n)")
        self._keep = (mix, comm, list(xs), rho, ll_local, ll_full, G)
        self._arr = (C.c_void_p * len(xs))(*[t.data_ptr() for t in xs])
        ptr = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)
        self._args = (mix.handle, comm.handle if comm is not None else C.c_void_p(0), self._arr, len(xs), n, int(n_components),
                      ptr(rho), ptr(ll_local), ptr(ll_full), ptr(G))
        self.graph, self.graph_error = None, None
        if graph:
            h = C.c_void_p()
            rc = lib().gbnf_group_graph_create(*self._args, C.byref(h))
            if rc == 0:
                self.graph = h
            else:
                self.graph_error = lib().gbnf_last_error().decode(errors="replace")

    def launch(self, stream_ptr):
        if self.graph is not None:
            rc = lib().gbnf_group_graph_launch(self.graph, stream_ptr)
        else:
            rc = lib().gbnf_mixture_group_log_prob(*self._args, stream_ptr)
        if rc:
            _check(rc)

    def close(self):
        if getattr(self, "graph", None) is not None and _lib is not None:
            _lib.gbnf_group_graph_destroy(self.graph)
            self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def mixture_lse(ll, rho, out=None):
    """Recursive prefix-normalised log-sum-exp of (C,n) per-component log-densities."""
    import torch
    _require_device_f32(ll, "ll")
    _require_device_f32(rho, "rho")
    if ll.dim() != 2:
        raise GbnfError("ll must be (C, n)")
    c, n = ll.shape
    if rho.numel() < c:
        raise GbnfError("rho shorter than the number of components")
    if out is None:
        out = torch.empty(n, dtype=torch.float32, device=ll.device)
    _check(lib().gbnf_mixture_lse(C.c_void_p(ll.data_ptr() if n else 0), n, C.c_void_p(rho.data_ptr()), c, n,
                                  C.c_void_p(out.data_ptr() if n else 0), _stream_ptr()))
    return out


def prepared_mixture_lse(ll, rho, out):
    """Bound form of mixture_lse: returns ``launch(stream_ptr)``."""
    _require_device_f32(ll, "ll")
    _require_device_f32(rho, "rho")
    _require_device_f32(out, "out")
    c, n = ll.shape
    if rho.numel() < c or out.numel() != n:
        raise GbnfError("prepared_mixture_lse: shape mismatch")
    fn = lib().gbnf_mixture_lse
    lp, rp, op = C.c_void_p(ll.data_ptr()), C.c_void_p(rho.data_ptr()), C.c_void_p(out.data_ptr())
    keep = (ll, rho, out)

    def launch(stream_ptr, _keep=keep):
        rc = fn(lp, n, rp, c, n, op, stream_ptr)
        if rc:
            _check(rc)
    return launch


def actnorm_init(z, scale=1.0):
    """ActNorm data-dependent initialisation statistics of a device batch z (n,d) -> (bias (d,), logs (d,))."""
    import torch
    _require_device_f32(z, "z")
    if z.dim() != 2 or z.shape[0] < 1:
        raise GbnfError("z must be (n,d) with n >= 1")
    n, d = z.shape
    bias = torch.empty(d, dtype=torch.float32, device=z.device)
    logs = torch.empty(d, dtype=torch.float32, device=z.device)
    _check(lib().gbnf_actnorm_init(C.c_void_p(z.data_ptr()), n, d, float(scale), C.c_void_p(bias.data_ptr()),
                                   C.c_void_p(logs.data_ptr()), _stream_ptr()))
    return bias, logs


def boosting_weights(G, beta=1.0):
    """Boosting sample weights from the fixed components' mixture log-density G (n,) on the device."""
    import torch
    _require_device_f32(G, "G")
    if G.dim() != 1 or G.numel() < 1:
        raise GbnfError("G must be (n,) with n >= 1")
    w = torch.empty_like(G)
    _check(lib().gbnf_boosting_weights(C.c_void_p(G.data_ptr()), G.numel(), float(beta), C.c_void_p(w.data_ptr()),
                                       _stream_ptr()))
    return w
