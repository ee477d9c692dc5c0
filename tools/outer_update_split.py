"""Split a rocprofv3 kernel trace of bench.py into the outer Schur-complement updates (the launches of
gemm_nt_update_fast on the update stream's queue) and the panel-internal launches of the same kernel, so that the
trace's average can be compared with the HIP-event figure bench.py prints (roofline.avg_launch_ms).
python tools/outer_update_split.py <kernel_trace.csv> <bench.json> <out.json>"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "gemm_nt_update_fast" in r["Kernel_Name"]]
bench = json.load(open(sys.argv[2]))
big = max(rows, key=lambda r: int(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0)))
queue = big["Queue_Id"]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6  # noqa: E731
outer = [dur(r) for r in rows if r["Queue_Id"] == queue]
inner = [dur(r) for r in rows if r["Queue_Id"] != queue]
out = {"what": "rocprofv3 --kernel-trace of `python3 bench.py --steps 3 --warmup 1 --no-cpu`, gemm_nt_update_fast split by queue",
       "outer_updates": {"launches": len(outer), "avg_ms": sum(outer) / len(outer), "total_ms": sum(outer)},
       "panel_internal": {"launches": len(inner), "avg_ms": sum(inner) / max(len(inner), 1), "total_ms": sum(inner)},
       "bench_hip_events_same_run": {"launches_timed": bench["roofline"]["launches"], "avg_launch_ms": bench["roofline"]["avg_launch_ms"],
                                      "achieved_tflops": bench["roofline"]["achieved"], "value_iters_per_s": bench["value"]},
       "note": "the trace holds the warm-up iteration's launches too (4 factorisations x 96 outer updates), the HIP events only the timed ones"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
