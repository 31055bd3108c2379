"""GPU: the C-ABI entry points that neither the package's own classes nor another test call -- straight through ctypes, as a binding in
another language would (include/gbnf.h): gbnf_flow_validate, gbnf_flow_create_mode, gbnf_image_flow_create, gbnf_comm_info."""
import ctypes as C

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_flow_validate_and_create_mode(golden_case):
    import torch
    from gbnf_amd import native
    dev = torch.device("cuda:0")
    g = golden_case("g2_glow_native_d43_h32_c3")
    L = native.lib()
    desc, keep = native.flow_desc_from_spec(g.specs[0])
    assert L.gbnf_flow_validate(C.byref(desc)) == 0
    # the checks gbnf_flow_create applies, without creating anything: a step count of zero, a permutation that is none
    bad = native.flow_desc_from_spec(g.specs[0])[0]
    bad.n_steps = 0
    assert L.gbnf_flow_validate(C.byref(bad)) != 0 and len(L.gbnf_last_error()) > 0
    spec2 = dict(g.specs[0])
    spec2["steps"] = [dict(st) for st in g.specs[0]["steps"]]
    perm = np.array(spec2["steps"][0]["perm"]).copy()
    perm[0] = perm[1]
    spec2["steps"][0]["perm"] = perm
    bad2, keep2 = native.flow_desc_from_spec(spec2)
    assert L.gbnf_flow_validate(C.byref(bad2)) != 0 and b"perm" in L.gbnf_last_error().lower()
    # an explicit math mode through the three-argument form
    h = C.c_void_p()
    assert L.gbnf_flow_create_mode(C.byref(desc), native.MATH["f32"], C.byref(h)) == 0
    try:
        x = torch.from_numpy(g.x).to(dev)
        n, d = x.shape
        ll = torch.empty(n, dtype=torch.float32, device=dev)
        rc = L.gbnf_flow_forward(h, C.c_void_p(x.data_ptr()), n, C.c_void_p(0), C.c_void_p(0), C.c_void_p(ll.data_ptr()), native._stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        assert rel_err(ll.cpu().numpy(), g.ll[0]) < 1e-5
        ref = native.NativeFlow(g.specs[0], math="f32").forward(x, want_z=False, want_ldj=False, want_ll=True)[2]
        assert np.array_equal(ll.cpu().numpy(), ref.cpu().numpy())
    finally:
        assert L.gbnf_flow_destroy(h) == 0
    h2 = C.c_void_p()
    assert L.gbnf_flow_create_mode(C.byref(desc), 99, C.byref(h2)) != 0 and not h2.value          # no such mode
    del keep, keep2


def test_image_flow_create_default_mode():
    """gbnf_image_flow_create(desc, &h) = gbnf_image_flow_create_mode(desc, GBNF_MATH_DEFAULT, &h): the handle evaluates a fixture the
    reference generated, through the raw entry points only."""
    import torch
    from gbnf_amd import native
    from test_hip_image import load_image_case, IMAGE_CASES, LL_RTOL
    cfg, specs, x, noise, data = load_image_case(IMAGE_CASES[0])
    dev = torch.device("cuda:0")
    L = native.lib()
    desc, keep = native.image_flow_desc_from_spec(specs[0])
    h = C.c_void_p()
    assert L.gbnf_image_flow_create(C.byref(desc), C.byref(h)) == 0 and h.value
    try:
        flow = native.NativeImageFlow.__new__(native.NativeImageFlow)        # the package's forward() around the raw handle
        flow.handle, flow._ws = h, None
        flow.input_size = tuple(int(v) for v in specs[0]["input_size"])
        flow.n_levels = len(specs[0]["levels"])
        zc, zh, zw, macs = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
        assert L.gbnf_image_flow_info(h, C.byref(zc), C.byref(zh), C.byref(zw), C.byref(macs)) == 0
        flow.z_shape, flow.macs_per_image = (zc.value, zh.value, zw.value), macs.value
        z, ldj, ll = flow.forward(torch.from_numpy(x).to(dev), torch.from_numpy(noise).to(dev))
        assert rel_err(ll.cpu().numpy(), data["ll"][0]) < LL_RTOL
        flow.handle = None                                                   # (destroyed below, once)
    finally:
        assert L.gbnf_image_flow_destroy(h) == 0
    del keep


def test_comm_info_of_a_one_rank_communicator():
    from gbnf_amd import native
    ok, why = native.Comm.probe()
    if not ok:
        pytest.skip("no RCCL on this box: " + why)
    import torch
    torch.cuda.set_device(0)
    comm = native.Comm(0, 1, native.Comm.unique_id())
    r, w = C.c_int32(-1), C.c_int32(-1)
    assert native.lib().gbnf_comm_info(comm.handle, C.byref(r), C.byref(w)) == 0 and (r.value, w.value) == (0, 1)
    assert native.lib().gbnf_comm_info(C.c_void_p(0), C.byref(r), C.byref(w)) != 0
    native.lib().gbnf_comm_destroy(comm.handle)
    comm.handle = None
