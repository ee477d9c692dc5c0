"""Kernels of the LAST `span_ms` milliseconds of a rocprofv3 kernel trace in start order, with the idle gap
before each (no kernel running on any queue): python tools/kernel_order.py <kernel_trace.csv> [span_ms] [min_us].
Consecutive launches of one kernel are folded into one line.  Shows where a solve's wall time goes that the
per-kernel totals do not: host gaps, small kernels between the big ones."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
span = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 30e6
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].split("(")[0].replace("dnlp::", "").replace("void ", "")[:44]) for r in rows)
tend = max(e[1] for e in ev)
ev = [e for e in ev if e[0] >= tend - span]
t0 = ev[0][0]
busy_end = ev[0][0]
lines = []            # name, count, first start, kernel time, idle gap before
for s, e, name in ev:
    gap = max(0, s - busy_end) / 1e3
    busy_end = max(busy_end, e)
    if lines and lines[-1][0] == name and gap < 20.0:
        lines[-1][1] += 1
        lines[-1][3] += (e - s) / 1e3
        lines[-1][4] += gap
    else:
        lines.append([name, 1, (s - t0) / 1e3, (e - s) / 1e3, gap])
tot_gap = sum(l[4] for l in lines)
for name, cnt, start, dur, gap in lines:
    print(f"{start:10.1f} us  {name:44s} x{cnt:<4d} kernel {dur:9.1f} us   idle before {gap:8.1f} us")
print(f"span {(tend - t0) / 1e3:.1f} us, idle (no kernel on any queue) {tot_gap:.1f} us")
