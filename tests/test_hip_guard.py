"""The library's own safety net for the DEFAULT math mode: the on-device numerics
guard (include/gbnf.h, gbnf_numerics_status) and the repair protocol's launch marks -- through the C ABI, the way
sharded.GroupPipeline and bench.py call it (no BoostedFlow module in between)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture()
def tuning():
    from gbnf_amd import native
    keys = ("force_nt", "wg_pairs", "repair", "check_every", "check_tolerance_e9")
    saved = {k: native.tuning_get(k) for k in keys}
    yield native
    for k, v in saved.items():
        native.tuning_set(k, v)


def _mixture(specs, math):
    from gbnf_amd import native
    flows = [native.NativeFlow(s, math=math) for s in specs]
    return native.NativeMixture(flows), flows


def test_guard_checks_the_first_launch_and_every_nth(dev, tuning):
    import torch
    from gbnf_amd import native, synth
    specs = synth.synth_boosted_specs("glow", 4, 43, 215, 5, seed=1)
    x = torch.from_numpy(synth.synth_batch(1024, 43, seed=0)).to(dev)
    rho = torch.full((4,), 0.25, device=dev)
    mix, flows = _mixture(specs, "default")
    assert native.MATH_NAME[flows[0].info().math_mode] == "f16x3"
    st = mix.numerics()
    assert (st.checks, st.demoted) == (0, 0) and abs(st.tolerance - 2.5e-6) < 1e-9
    G0, ll0 = mix.log_prob(x, rho)
    torch.cuda.synchronize()
    st = mix.numerics()
    assert st.checks == 1 and st.demoted == 0 and 0.0 <= st.worst_rel_err < 2.5e-6
    for _ in range(5):
        mix.log_prob(x, rho)
    torch.cuda.synchronize()
    assert mix.numerics().checks == 1            # check_every = 256: launches 1..5 are not checked
    tuning.tuning_set("check_every", 2)
    for _ in range(6):                           # launches 6..11: the even ones are checked
        G, ll = mix.log_prob(x, rho)
    torch.cuda.synchronize()
    assert mix.numerics().checks == 4
    assert torch.equal(G, G0) and torch.equal(ll, ll0)      # a passed check changes nothing
    # a single flow handle has a guard of its own (gbnf_flow_forward)
    z, ldj, ll1 = flows[0].forward(x, want_ll=True)
    torch.cuda.synchronize()
    assert flows[0].numerics().checks == 1 and flows[0].numerics().demoted == 0
    # an explicit f16x3 handle keeps the caller's choice: no guard
    f = native.NativeFlow(specs[0], math="f16x3")
    f.forward(x)
    torch.cuda.synchronize()
    assert f.numerics().checks == 0


@pytest.mark.parametrize("n", [77, 4096])
def test_failed_check_re_evaluates_the_launch_and_demotes_the_handle(n, dev, tuning):
    """tolerance 0 => the first check fails: the launch that was checked must come back as bf16x6 results in full (no
    result of a failed mode reaches the caller), the flag is visible without a sync, later launches run bf16x6 directly."""
    import torch
    from gbnf_amd import native, synth
    specs = synth.synth_boosted_specs("glow", 3, 43, 64, 3, seed=5)
    x = torch.from_numpy(synth.synth_batch(n, 43, seed=2)).to(dev)
    safe, _ = _mixture(specs, "bf16x6")
    want = safe.component_log_prob(x)
    fast, _ = _mixture(specs, "f16x3")
    assert not torch.equal(fast.component_log_prob(x), want)           # the two modes do differ in the last bits
    tuning.tuning_set("check_tolerance_e9", 0)
    mix, flows = _mixture(specs, "default")
    assert mix.numerics().math_mode == native.MATH["f16x3"]
    got = mix.component_log_prob(x)                                    # launch 0: checked, fails, re-evaluated on the device
    torch.cuda.synchronize()
    st = mix.numerics()
    assert st.checks == 1 and st.demoted == 1 and st.math_mode == native.MATH["bf16x6"]
    assert torch.equal(got, want)
    tuning.tuning_set("check_tolerance_e9", 2500)
    again = mix.component_log_prob(x)                                  # host saw the flag: a plain bf16x6 launch
    assert torch.equal(again, want)
    # the group form (what GroupPipeline / bench.py launch)
    mix2, _ = _mixture(specs, "default")
    tuning.tuning_set("check_tolerance_e9", 0)
    xs = [x, torch.from_numpy(synth.synth_batch(n, 43, seed=3)).to(dev)]
    table = torch.empty((3, 2 * n), device=dev)
    mix2.prepared_group_log_prob(xs, table)(native._stream_ptr())
    torch.cuda.synchronize()
    assert mix2.numerics().demoted == 1
    assert torch.equal(table[:, :n], want) and torch.equal(table[:, n:], safe.component_log_prob(xs[1]))


def test_guard_flag_makes_every_repair_pass_redo_the_launch_before_the_host_notices(dev, tuning):
    """Between the failed check and the host's next look at the flag nothing wrong may come out either: with the flag set
    on the device, the bf16x6 pass behind an f16x3 launch re-evaluates the whole work list."""
    import torch
    from gbnf_amd import native, synth
    specs = synth.synth_boosted_specs("realnvp", 2, 21, 105, 3, seed=7)
    x = torch.from_numpy(synth.synth_batch(2048, 21, seed=4)).to(dev)
    safe, _ = _mixture(specs, "bf16x6")
    want = safe.component_log_prob(x)
    tuning.tuning_set("check_tolerance_e9", 0)
    mix, _ = _mixture(specs, "default")
    outs = [mix.component_log_prob(x) for _ in range(4)]           # queued back to back: the host cannot have seen the flag for all
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)


def test_launch_marks_that_share_a_slot_do_not_hide_each_other(dev, tuning):
    """two f16x3 launches in flight on different streams whose serial numbers are 1024 apart share a mark slot;
    both must still get their out-of-range samples repaired (atomicMax marks, the repair pass runs when the slot holds its
    own OR A LATER serial).  Launch A (serial s) is parked behind a device-side sleep on stream 1; 1023 one-row launches and
    launch B (serial s + 1024, the same slot) run to completion on stream 2 in the meantime, so A finds B's mark in its slot."""
    import torch
    from gbnf_amd import native, synth
    spec = synth.synth_glow_spec(6, 30, 3, seed=1)                   # tanh Glow: finite for any finite input
    f = native.NativeFlow(spec, math="f16x3")
    ref = native.NativeFlow(spec, math="bf16x6")
    a_np, b_np = synth.synth_batch(4096, 6, seed=1), synth.synth_batch(512, 6, seed=2)
    a_np[::97] *= np.float32(1e6)                                    # rows far outside the fp16 range
    b_np[::7] *= np.float32(1e6)
    xa, xb = torch.from_numpy(a_np).to(dev), torch.from_numpy(b_np).to(dev)
    tiny = torch.from_numpy(synth.synth_batch(1, 6, seed=3)).to(dev)
    want_a = ref.forward(xa, want_z=False, want_ll=True)[2]
    want_b = ref.forward(xb, want_z=False, want_ll=True)[2]
    assert torch.isfinite(want_a).all() and torch.isfinite(want_b).all()
    native.saturation_count(reset=True)
    s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        torch.cuda._sleep(int(4e8))                                  # ~0.2 s: A stays queued while stream 2 runs
        got_a = f.forward(xa, want_z=False, want_ll=True)[2]        # serial s
    with torch.cuda.stream(s2):
        for _ in range(1023):                                        # serials s+1 .. s+1023
            f.forward(tiny, want_z=False, want_ldj=False, want_ll=True)
        got_b = f.forward(xb, want_z=False, want_ll=True)[2]        # serial s+1024: the same mark slot as A
    torch.cuda.synchronize()
    assert native.saturation_count() > 0, "the test inputs are meant to leave the fp16 range"
    assert not torch.isnan(got_a).any() and not torch.isnan(got_b).any()
    assert rel_err(got_a.cpu().numpy(), want_a.cpu().numpy()) < 1e-5
    assert rel_err(got_b.cpu().numpy(), want_b.cpu().numpy()) < 1e-5


def test_tuning_keys(tuning):
    from gbnf_amd import native
    for key in ("force_nt", "wg_pairs", "repair", "nt2_min_waves", "check_every", "check_tolerance_e9"):
        v = native.tuning_get(key)
        native.tuning_set(key, v)
    with pytest.raises(native.GbnfError):
        native.tuning_set("no_such_key", 1)
