"""CPU: the host-side pieces of bench.py that need no GPU -- the PMC traffic look-up (profiles/headline_traffic.json) and
the argument surface the driver relies on."""
import importlib.util
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(REPO, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_lookup_matches_only_the_measured_workloads():
    b = _bench()
    rec = json.load(open(os.path.join(REPO, "profiles", "headline_traffic.json")))
    g = rec["workload"]["group"]
    t = b.measured_traffic("miniboone_glow", 4096, 8, g, "f16x3", 1)
    assert t == float(rec["traffic_bytes_per_launch"]) and t > 0
    # the correction and the units the guide prescribes: 2 x FETCH_SIZE + WRITE_SIZE, both in KB
    assert t == rec["FETCH_SIZE_kb_per_launch"] * 1024 * 2 + rec["WRITE_SIZE_kb_per_launch"] * 1024
    for other in rec.get("other_group_sizes", []):
        w = other["workload"]
        assert b.measured_traffic(w["config"], w["batch"], w["components"], w["group"], w["math"], w["n_gpus"]) == float(other["traffic_bytes_per_launch"])
    assert b.measured_traffic("miniboone_glow", 4096, 8, 7, "f16x3", 1) is None          # never measured
    assert b.measured_traffic("miniboone_glow", 4096, 8, g, "f32", 1) is None            # another kernel
    assert b.measured_traffic("miniboone_glow", 4096, 8, g, "f16x3", 8) is None          # another rank count
    assert b.measured_traffic("hepmass_realnvp", 4096, 8, g, "f16x3", 1) is None         # another batch size (N = 65536 is measured since round 4)


def test_driver_invocation_parses_and_defaults_finish_quickly():
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--help"], stdout=subprocess.PIPE, text=True, check=True).stdout
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out
    b = _bench()
    assert b.GROUP <= 32 and set(b.CONFIGS) >= {"miniboone_glow", "hepmass_realnvp"}


def test_traffic_is_labelled_as_a_lookup_with_its_source_and_date():
    """roofline.traffic is NOT counted in the run; the line says where it comes from and when it was taken."""
    b = _bench()
    src = b.traffic_source()
    assert src["file"] == "profiles/headline_traffic.json" and "not counted in this run" in src["note"]
    assert src["pmc_summary"] and os.path.exists(os.path.join(REPO, src["pmc_summary"]))
    assert src["taken"]                                       # the date of the PMC passes (tools/collect_final_profiles.py)
    text = open(os.path.join(REPO, "bench.py")).read()
    assert '"--pipeline", default="auto"' in text             # real ranks stay on the torch.distributed pipeline
    assert "tail_summary" in text
