"""Per-kernel timeline of one MutualProjectionLoss step from a rocprofv3 --kernel-trace CSV: start / end of every
kernel of the LAST complete step relative to the step's first kernel.  usage: tools/timeline_mvloss.py trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "mutual_project" in n and "fwd" in n or "mv_project" in n]
if len(starts) < 3:
    starts = [i for i, n in enumerate(names) if "mutual_project_fwd" in n or "project_compact" in n]
a, b = starts[-2], starts[-1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0][-60:]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%8.1f -> %8.1f  (%7.1f us)  q%s  %s" % (s / 1e3, e / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), n))
