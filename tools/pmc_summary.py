"""Summarise rocprofv3 --pmc passes (csv output): per kernel, the number of dispatches and the sum / mean
of every counter found under the given directories.

    python tools/pmc_summary.py out.json dir1 [dir2 ...] [--kernel substring] [--min-ms X] [--update-queue]
(--min-ms keeps dispatches that ran at least X ms.  --update-queue keeps, per file, only the dispatches on
the queue of the kernel's largest grid: bench.py's outer Schur-complement updates run on their own stream,
the panel-internal launches of the same kernel on the other one.)

FETCH_SIZE / WRITE_SIZE are reported in bytes with the gfx950 correction of MI355X_MICROARCH.md (HBM
section): rocprofv3 gives KB; FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads, so it is
doubled; WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    out = args.pop(0)
    kernel_filter = None
    if "--kernel" in args:
        i = args.index("--kernel")
        kernel_filter = args[i + 1]
        del args[i:i + 2]
    min_ms = 0.0
    if "--min-ms" in args:
        i = args.index("--min-ms")
        min_ms = float(args[i + 1])
        del args[i:i + 2]
    update_queue = "--update-queue" in args
    if update_queue:
        args.remove("--update-queue")
    acc = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows = list(csv.DictReader(open(f)))
            queue = None
            if update_queue:
                sel = [r for r in rows if not kernel_filter or kernel_filter in r.get("Kernel_Name", "")]
                if sel:
                    queue = max(sel, key=lambda r: int(r.get("Grid_Size", 0)))["Queue_Id"]
            for r in rows:
                k = r.get("Kernel_Name", "")
                if kernel_filter and kernel_filter not in k:
                    continue
                if queue is not None and r.get("Queue_Id") != queue:
                    continue
                c = r.get("Counter_Name")
                ms = (float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0))) * 1e-6
                if ms < min_ms:
                    continue
                acc[k][c] += float(r.get("Counter_Value", 0.0))
                dur[k][c] += ms
                cnt[k][c] += 1
    res = {}
    for k in acc:
        row = {}
        for c, v in acc[k].items():
            n = cnt[k][c]
            if c == "FETCH_SIZE":
                row["FETCH_bytes_x2_corrected_per_dispatch"] = v * 1024.0 * 2.0 / n
            elif c == "WRITE_SIZE":
                row["WRITE_bytes_per_dispatch"] = v * 1024.0 / n
            else:
                row[c + "_per_dispatch"] = v / n
            row["dispatches_" + c] = n
            row["avg_ms_under_" + c] = dur[k][c] / n
        if "SQ_WAVE_CYCLES_per_dispatch" in row:
            wc = row["SQ_WAVE_CYCLES_per_dispatch"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c + "_per_dispatch" in row and wc > 0:
                    row[c + "_frac_of_wave_cycles"] = row[c + "_per_dispatch"] / wc
        res[k[:120]] = row
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:6000])


if __name__ == "__main__":
    main()
