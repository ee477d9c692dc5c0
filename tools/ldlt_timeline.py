"""Print the kernel timeline of the LAST blocked-LDL^T factorisation found in a rocprofv3 kernel trace:
python tools/ldlt_timeline.py <kernel_trace.csv> [max_rows].  Columns: kernel, start (us from the first
panel kernel), duration (us), gap to the previous kernel's end on the same queue (us)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0].replace("dnlp::", "").replace("void ", ""), int(r["Start_Timestamp"]),
       int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows]
first = [i for i, e in enumerate(ev) if e[0] in ("ldlt_top128_kernel", "ldlt_top128_mfma_kernel", "ldlt_diag_kernel")]
# the last factorisation starts at the last panel kernel with j0 == 0: approximate by the largest gap
starts = [first[0]] + [first[k] for k in range(1, len(first)) if ev[first[k]][1] - ev[first[k - 1]][2] > 2_000_000]
i0 = starts[-1]
t0 = ev[i0][1]
last_end = {}
tot = {}
for e in ev[i0:]:
    tot.setdefault(e[0], [0, 0.0])
    tot[e[0]][0] += 1
    tot[e[0]][1] += (e[2] - e[1]) / 1e3
for e in ev[i0:i0 + limit]:
    gap = (e[1] - last_end[e[3]]) / 1e3 if e[3] in last_end else 0.0
    last_end[e[3]] = e[2]
    print(f"{e[0][:24]:24s} q{e[3]} start {((e[1] - t0) / 1e3):9.1f}  dur {((e[2] - e[1]) / 1e3):8.1f}  gap {gap:6.1f}")
print({k: (v[0], round(v[1], 1)) for k, v in tot.items()})
print("span_us", (max(e[2] for e in ev[i0:]) - t0) / 1e3)
