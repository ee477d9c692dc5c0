"""Turn the raw per-kernel summary of tools/pmc_summary.py (the C4 counter passes of tools/gpu_round.sh pmc_c4) into the
profile file bench.py's `roofline.traffic` reads (keys n, command, avg_traffic_bytes_per_launch ...).

    python tools/pmc_c4_post.py raw.json out.json [n]
"""
import json
import sys

COMMAND = ("rocprofv3 --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) / SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES "
           "GRBM_GUI_ACTIVE (pass 3) --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-full-solve  "
           "(tools/gpu_round.sh pmc_c4; tools/pmc_summary.py --update-queue; tools/pmc_c4_post.py)")


def convert(raw, n=100000, kernel="gemm_nt_update_fast"):
    if "avg_traffic_bytes_per_launch" in raw:      # already in the profile schema
        return raw
    names = [k for k in raw if kernel in k]
    if not names:
        raise KeyError("no %s row in the raw summary" % kernel)
    name = names[0]
    r = raw[name]
    fetch, write = r["FETCH_bytes_x2_corrected_per_dispatch"], r["WRITE_bytes_per_dispatch"]
    out = {"n": int(n), "command": COMMAND, "kernel": name, "dispatches": int(r["dispatches_FETCH_SIZE"]),
           "fetch_bytes_x2_corrected_per_launch": fetch, "write_bytes_per_launch": write,
           "avg_traffic_bytes_per_launch": fetch + write, "avg_launch_ms_under_counters": r["avg_ms_under_FETCH_SIZE"]}
    busy, active = r.get("SQ_VALU_MFMA_BUSY_CYCLES_per_dispatch"), r.get("GRBM_GUI_ACTIVE_per_dispatch")
    if busy and active:
        # SQ_VALU_MFMA_BUSY_CYCLES sums over the chip's SIMD quads in units of 4 cycles: 1024 SIMDs / 8 (r02-r04 files: same rule)
        out["mfma_busy_cycles_per_launch"] = busy
        out["grbm_gui_active_per_launch"] = active
        out["mfma_pipe_busy_fraction"] = busy / (active * 128.0)
        ms = r.get("avg_ms_under_GRBM_GUI_ACTIVE")
        if ms:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            out["effective_clock_ghz"] = active / 8.0 / (ms * 1e-3) / 1e9
    out["raw"] = {name: r}
    return out


if __name__ == "__main__":
    raw = json.load(open(sys.argv[1]))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
    out = convert(raw, n)
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "raw"}, indent=1))
