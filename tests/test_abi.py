"""CPU: the C-ABI library loads and exports every symbol include/gbnf.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO
from gbnf_amd import native


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "gbnf.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gbnf_[a-z_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(native.LIB_PATH), "build the library first: python __graft_entry__.py"
    lib = ctypes.CDLL(native.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} missing from libgbnf_hip.so"
    lib.gbnf_version.restype = ctypes.c_int
    assert lib.gbnf_version() == 4           # GBNF_ABI_VERSION; touches no device


def test_ctypes_structs_match_header_layout():
    # sizes implied by include/gbnf.h on LP64
    assert ctypes.sizeof(native._Linear) == 24
    assert ctypes.sizeof(native._Net) == 16
    assert ctypes.sizeof(native._GlowStep) == 40
    assert ctypes.sizeof(native._RealNVPStep) == 8 + 4 * 8 + 8 + 16 + 16
    assert ctypes.sizeof(native._FlowDesc) == 32
    assert ctypes.sizeof(native.KernelInfo) == 16 + 8 + 8 + 8 + 8


def test_variant_list_covers_baseline_configs():
    """Every BASELINE.json config geometry has an exact compiled variant (not a padded superset)."""
    lines = [ln.split("#")[0].split() for ln in open(os.path.join(
        REPO, "gradient-boosted-normalizing-flows_amd", "csrc", "variants.list"))]
    keys = {tuple(ln) for ln in lines if ln}
    assert tuple("0 14 2 6 3 1 0 0".split()) in keys      # MINIBOONE d=43 h=215 Glow tanh depth 1, exact f32
    assert tuple("1 7 3 3 1 1 0 0".split()) in keys       # HEPMASS d=21 h=105 RealNVP tanh depth 1, exact f32
    assert tuple("1 4 4 3 1 1 0 0".split()) in keys       # toy / small RealNVP
    assert tuple("hx3 0 14 3 0 0".split()) in keys        # MINIBOONE, split-f16 kernel
    assert tuple("hx3 1 7 1 0 0".split()) in keys         # HEPMASS, split-f16 kernel
