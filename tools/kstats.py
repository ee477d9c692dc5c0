"""Print the top rows of a rocprofv3 `--kernel-trace --stats --output-format csv` kernel table.
    python tools/kstats.py <output dir> [rows]"""
import csv
import glob
import sys

d = sys.argv[1]
rows_n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))
if not f:
    sys.exit("no *kernel_stats.csv under " + d)
rows = list(csv.DictReader(open(f[0])))
print("%-64s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
for r in rows[:rows_n]:
    print("%-64s %8s %12.1f %10.2f %6s" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                           float(r["AverageNs"]) / 1e3, r["Percentage"]))
