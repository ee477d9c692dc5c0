"""The committed counter summaries that bench.py quotes in `roofline.traffic` must be readable by bench.py (round 5 shipped
the C4 file in another schema and the driver's line carried traffic = null)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_c4_traffic_profile_has_the_schema_bench_reads():
    import bench
    pj = json.load(open(os.path.join(ROOT, "profiles", bench.TRAFFIC_PROFILE)))
    assert int(pj["n"]) == 100000
    assert pj["avg_traffic_bytes_per_launch"] is not None and pj["avg_traffic_bytes_per_launch"] > 1e9
    assert isinstance(pj["command"], str) and "rocprofv3" in pj["command"]
    assert abs(pj["avg_traffic_bytes_per_launch"] - pj["fetch_bytes_x2_corrected_per_launch"] - pj["write_bytes_per_launch"]) < 1.0


def test_c5_traffic_profile_has_the_schema_bench_reads():
    import bench
    pj = json.load(open(os.path.join(ROOT, "profiles", bench.C5_TRAFFIC_PROFILE)))
    assert pj["which"] == "localization" and int(pj["batch"]) == 8192
    assert pj["traffic_bytes_per_launch"] > 1e6


def test_raw_counter_summary_converts():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_c4_post
    raw = {"dnlp::gemm_nt_update_fast(double*)": {"FETCH_bytes_x2_corrected_per_dispatch": 3.0e9, "WRITE_bytes_per_dispatch": 1.0e9,
                                                  "dispatches_FETCH_SIZE": 7, "avg_ms_under_FETCH_SIZE": 2.0}}
    out = pmc_c4_post.convert(raw, 100000)
    assert out["n"] == 100000 and out["avg_traffic_bytes_per_launch"] == 4.0e9 and out["dispatches"] == 7
    assert pmc_c4_post.convert(out) is out
