"""CPU: consistency of csrc/variants.list -- every latency-form (`coop`) line reads the blob of an evaluation (`hx3`) line of exactly
the same geometry and activation key, the forms field is well formed, and csrc/build.py turns the lines into the objects the library
registers (the registry key of a cooperative variant carries the form in its `nt` field)."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "gradient-boosted-normalizing-flows_amd", "csrc")


def _lines():
    out = []
    with open(os.path.join(CSRC, "variants.list")) as f:
        for line in f:
            line = line.split("#")[0].strip()
            if line:
                out.append(line.split())
    return out


def test_every_cooperative_line_has_its_evaluation_line():
    hx3 = set()
    for t in _lines():
        if t[0] == "hx3":
            vals = [int(v) for v in t[1:] if v.lstrip("-").isdigit()]
            depth = vals[5] if len(vals) > 5 else 1
            hx3.add(tuple(vals[:5]) + (depth,))
    coops = [t for t in _lines() if t[0] == "coop"]
    assert coops, "no latency-form variants listed"
    for t in coops:
        key = tuple(int(v) for v in t[1:6])
        assert key + (1,) in hx3, f"coop {key}: no `hx3` line of the same geometry (the latency form reads that line's blob)"
        assert key[3] in (0, 1, 3) and key[4] in (0, 1, 3), "TanhNet / ReLUNet (3 = the activation per step)"
        forms = t[6] if len(t) > 6 else "123"
        assert forms and set(forms) <= set("123") and len(set(forms)) == len(forms), forms


def test_build_lists_one_object_per_form():
    sys.path.insert(0, CSRC)
    import build
    variants = build.read_variants()
    coop = [v for v in variants if v[0] == "coop"]
    want = 0
    for t in _lines():
        if t[0] == "coop":
            want += len(t[6]) if len(t) > 6 else 3
    assert len(coop) == want
    # an `eval` line builds the evaluation kernels only: no training sweeps for its geometry
    ev = [t for t in _lines() if t[0] == "hx3" and t[-1] == "eval"]
    for t in ev:
        vals = tuple(int(v) for v in t[1:-1])
        kind, ht, ot, acta, actb = vals[:5]
        depth = vals[5] if len(vals) > 5 else 1
        assert not [v for v in variants if v[0] in ("hx3t", "hx3b") and v[1:4] == (kind, ht, ot) and v[-1] == depth and acta in v and actb in v
                    and (v[0] == "hx3b" and v[4:6] == (acta, actb) or v[0] == "hx3t" and v[5:7] == (acta, actb))], t
        assert [v for v in variants if v[0] == "hx3" and v[1:4] == (kind, ht, ot) and v[5:7] == (acta, actb) and v[-1] == depth]
