"""The shipped gfx950 code must not read a v_mfma result early on ANY path (tools/isa_hazard_lint.py).

hipcc pads the wait states a VALU / LDS / VMEM use of a v_mfma result needs only along the fall-through path of a
conditional branch; the round-2 tile corruption was a reader at a branch target (profiles/r3_mfma_hazard_root_cause.txt).
This walks the control-flow graph of every kernel in every object of the shipped build -- CPU only, no GPU needed -- and
is the regression test for that class of bug (a recompile or toolchain bump can re-introduce it silently)."""
import glob
import os
import sys

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))
import isa_hazard_lint as lint  # noqa: E402

OBJ = os.path.join(REPO, "gradient-boosted-normalizing-flows_amd", "csrc", "obj")


def _objects():
    return sorted(glob.glob(os.path.join(OBJ, "*.o")))


def test_lint_recognises_a_reader_behind_a_taken_branch():
    """The checker itself: a v_exp_f32 on the taken side of a conditional branch, 1 wait state behind the v_mfma."""
    mk = lambda addr, text: lint.Inst(addr, text.split(None, 1)[0], [t.strip() for t in text.split(None, 1)[1].split(",")] if " " in text else [], text)
    insts = [mk(0x00, "v_mfma_f32_16x16x32_f16 v[36:39], v[56:59], v[40:43], v[36:39]"),
             mk(0x08, "s_cbranch_scc1 9"),                       # -> 0x08 + 4 + 36 = 0x30
             mk(0x0c, "s_nop 7"), mk(0x10, "s_nop 7"),
             mk(0x14, "v_lshl_add_u64 v[26:27], vcc, 0, v[74:75]"), mk(0x1c, "s_nop 0"), mk(0x20, "s_nop 0"),
             mk(0x24, "s_nop 0"), mk(0x28, "s_nop 0"), mk(0x2c, "s_nop 0"),
             mk(0x30, "v_exp_f32_e32 v26, v36"), mk(0x34, "s_endpgm")]
    hits = lint.check_kernel(insts)
    assert hits == [(0, 10, 1, 7, True)]
    # the same reader behind enough idle wait states on both paths is fine
    insts[1:1] = [mk(0x04, "s_nop 7")]
    assert lint.check_kernel(insts) == []
    # a dependent accumulation (srcC) and a same-shape overwrite are not hazards; a read as srcA is
    chain = [mk(0x00, "v_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]"),
             mk(0x08, "v_mfma_f32_16x16x32_f16 v[12:15], v[0:3], v[4:7], v[8:11]"),
             mk(0x10, "v_mfma_f32_16x16x32_f16 v[10:13], v[0:3], v[4:7], v[20:23]"), mk(0x18, "s_endpgm")]
    assert lint.check_kernel(chain) == []
    chain[2] = mk(0x10, "v_mfma_f32_16x16x32_f16 v[20:23], v[12:15], v[4:7], v[20:23]")
    assert [h[:3] for h in lint.check_kernel(chain)] == [(1, 2, 0)]


def test_lint_recognises_a_counted_wait_that_lets_the_staging_dma_slip():
    """A stage end waits with vmcnt(N) for everything but the N operations the wave issued BEHIND its staging DMA; when the
    compiler drops some of those (loads of values nobody reads) the DMA is no longer covered."""
    mk = lambda addr, text: lint.Inst(addr, text.split(None, 1)[0], [t.strip() for t in text.split(None, 1)[1].split(",")] if " " in text else [], text)
    stage = [mk(0x00, "global_load_lds_dwordx4 v[2:3], off"),
             *[mk(0x08 + 8 * k, f"global_store_dword v[4:5], v{10 + k}, off") for k in range(4)],
             *[mk(0x28 + 8 * k, f"global_load_dword v{20 + k}, v[4:5], off") for k in range(4)],
             mk(0x48, "s_waitcnt vmcnt(8) lgkmcnt(0)"), mk(0x4c, "s_barrier"), mk(0x50, "s_endpgm")]
    assert lint.check_counted_waits(stage) == []
    del stage[5:9]                                  # the four loads were dead code
    assert lint.check_counted_waits(stage) == [(5, 0, 4, 8)]
    stage[5] = mk(0x48, "s_waitcnt vmcnt(4) lgkmcnt(0)")
    assert lint.check_counted_waits(stage) == []


@pytest.mark.skipif(not _objects(), reason="csrc/obj is empty: build the library first (python __graft_entry__.py)")
def test_no_shipped_kernel_reads_a_matrix_result_early():
    bad = []
    n_mfma = 0
    for f in _objects():
        for name, insts in lint.disassemble(f).items():
            n_mfma += sum(1 for x in insts if lint.is_mfma(x.mn))
            for i, j, between, n in lint.check_counted_waits(insts):
                bad.append(f"{os.path.basename(f)} {name[:60]}: {insts[i].text} with {between} operations behind the staging DMA")
            for i, j, ws, need, crossed in lint.check_kernel(insts):
                bad.append(f"{os.path.basename(f)} {name[:60]}: {insts[i].text} -> {insts[j].text} after {ws} of {need} wait states"
                           f"{' (across a branch)' if crossed else ''}")
    assert n_mfma > 100000, "the objects of every kernel variant are expected under csrc/obj"
    assert not bad, "\n".join(bad[:20])


@pytest.mark.skipif(not _objects(), reason="csrc/obj is empty: build the library first (python __graft_entry__.py)")
def test_staged_kernels_count_their_lds_reads():
    """With __builtin_amdgcn_global_load_lds in a kernel hipcc (ROCm 7.2) turns EVERY wait for a ds_read into lgkmcnt(0)
    (tools/ubench/lgkmcnt_dma.hip); the staging DMA is therefore issued from inline asm (lds_dma16).  A kernel that stages
    through LDS DMA and reads fragments with ds_read_b128 must show counted waits -- if it does not, the builtin is back."""
    import re
    checked = 0
    for f in _objects():
        name = os.path.basename(f)
        if not (name.startswith("v_hx3_0_14_3_2_") or name.startswith("v_hx3t_0_14_3_1_") or name.startswith("v_hx3b_0_14_3_0_0")):
            continue
        for kname, insts in lint.disassemble(f).items():
            if not any(x.mn.startswith("global_load_lds") for x in insts):
                continue
            reads = sum(1 for x in insts if x.mn == "ds_read_b128")
            counted = sum(1 for x in insts if x.mn == "s_waitcnt" and re.search(r"lgkmcnt\(([1-9]\d*)\)", x.text))
            assert reads >= 50 and counted >= 10, f"{name} {kname[:60]}: {reads} ds_read_b128, {counted} counted lgkmcnt waits"
            checked += 1
    assert checked >= 4
