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


def _worst_case_record(b):
    """A record with every optional block present, long free-text fields and more legs than bench.py has today."""
    long = "x" * 4000
    roof = {"kernel": "gbnf::flow_kernel_hx3" + long, "bound": "mfma", "achieved": 373.61234567, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": 0.149444444444, "executed_frac": 0.50712345678, "traffic": 198312345.0, "launch_ms": 1.05612345678,
            "flops_per_launch": 394526720000.0, "hbm_algorithmic_bytes_per_launch": 16711680.0, "note": long,
            "traffic_source": {"file": "profiles/headline_traffic.json", "pmc_summary": "profiles/r5_final_miniboone_c8_n4096_group32.txt",
                               "taken": "2026-01-01", "note": long}}
    cpu = {"value": 51412.123456, "unit": "samples/s", "cores": 16, "kind": "port", "host_cores": 256, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
           "one_thread_value": 5123.123456, "thread_probe_samples_per_s": {str(t): 1.0 / 3 for t in (1, 4, 8, 16, 32, 64)}, "sample": long,
           "one_thread_sample": long}
    child = {"metric": long, "value": 151123456.789, "unit": "samples/s", "dtype": "f16x3", "roofline": dict(roof), "cpu_baseline": dict(cpu),
             "config": {"workload": long}, "timing": {"repetitions": 21}, "wall_s": 12.3}
    legs = {k: {"value": 45312345.678, "frac": 0.0971234567, "executed_frac": 0.331234567, "note": long, "module_calls_only_ms": 0.138123,
                "ms_per_step": 1.13, "library_ms_per_step": 0.45}
            for k in ("group1", "f32_exact", "bf16x6", "fresh_batches", "module_evaluate_loop", "boosted_step_batch512")}
    legs["reference_batch_sizes"] = {"512": {"value": 1.0e7 / 3, "us_per_call": 40.123456}, "1024": {"value": 2.0e7 / 3, "us_per_call": 46.123456},
                                     "4096": {"value": 2.0e7 / 3, "us_per_call": 90.123456}, "note": long}
    legs["configs"] = {f"some_other_baseline_configuration_with_a_long_name_{i:02d}_n65536": dict(child) for i in range(16)}
    legs["configs"]["failed_leg"] = {"error": long}
    legs["configs"]["boosted_step_batch512"] = {"value": 4.5e5, "ms_per_step": 1.13, "library_ms_per_step": 0.31, "note": long}
    return {"metric": "density-eval samples/sec, Boosted-Glow C=8 MINIBOONE d=43", "value": 74451234.56789, "unit": "samples/s", "n_gpus": 8,
            "steps": 4096, "warmup": 64, "ms_per_step": 0.0550212345, "prewarm_s": 0.3, "higher_is_better": True,
            "timing": {"repetitions": 21, "value_from": "median repetition", "elapsed_ms_median": 1.1, "elapsed_ms_min": 1.0, "elapsed_ms_max": 1.3,
                       "value_at_min": 8.0e7},
            "numerics_guard": {"checks": 3, "worst_rel_err": 2.4e-7, "demoted": False, "tolerance": 1e-5},
            "math_modes_by_rank": ["f16x3"] * 8, "scaling": "strong", "vs_baseline": None, "dtype": "f16x3", "data": "synthetic",
            "config": {"workload": long, "global_batch": 4096, "components": 8, "math": long, "group": 20, "group_note": long,
                       "parallelism": long, "emulated": False},
            "roofline": roof, "cpu_baseline": cpu, "speedup_vs_cpu": 1448.123456, "max_rel_err_vs_cpu": 2.4e-7,
            "rccl": {"ranks_seen": list(range(8)), "allgather_us": 31.2, "allgather_bytes_per_rank": 524288,
                     "pipeline": {"kind": "torch.distributed", "fallback_reason": long}, "graph_errors": [long, long], "note": long},
            "legs": legs}


def test_the_drivers_line_stays_short_whatever_the_run_measured():
    """BENCH_r05.parsed was null because the line had grown to 20 KB.  The line is built by compact_line() from an
    allow-list, free text is clipped, and a worst-case record (every block present, 4 KB strings everywhere, 17 configuration legs)
    still comes out under LINE_LIMIT with the tier's `roofline` and `cpu_baseline` objects intact."""
    b = _bench()
    assert b.LINE_LIMIT <= 6000
    text = b.compact_line(_worst_case_record(b))
    assert len(text.encode()) <= b.LINE_LIMIT and "\n" not in text
    j = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert set(j["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(j["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert j["config"]["workload"] and "model" not in j["config"]
    assert j["full_record"] == b.FULL_RECORD and "legs" not in j and "note" not in j["roofline"]
    # a record the size of a real N = 1 run keeps every optional block
    rec = _worst_case_record(b)
    rec["legs"]["configs"] = dict(list(rec["legs"]["configs"].items())[:8])
    j = json.loads(b.compact_line(rec))
    assert "dropped_for_length" not in j and len(j["configs"]) == 8 and j["legs_summary"]["group1"][0] > 0
    assert j["legs_summary"]["boosted_step_batch512_library_ms"] is None      # (that leg was cut from this record with the other eight)
    rec["legs"]["configs"]["boosted_step_batch512"] = {"value": 4.5e5, "ms_per_step": 1.13, "library_ms_per_step": 0.31}
    j = json.loads(b.compact_line(rec))
    assert j["legs_summary"]["boosted_step_batch512_library_ms"] == 0.31 and j["legs_summary"]["boosted_step_batch512_ms"] == 1.13
    # a sharded line (no legs, no cpu baseline) is just as parseable
    rec = _worst_case_record(b)
    del rec["legs"]
    rec["cpu_baseline"] = None
    j = json.loads(b.compact_line(rec))
    assert j["cpu_baseline"] is None and j["rccl"]["ranks_seen"] == list(range(8))
    # the source builds its last stdout line through compact_line and nothing else
    src = open(os.path.join(REPO, "bench.py")).read()
    assert "line = compact_line(out" in src and "line = json.dumps(out)" not in src
