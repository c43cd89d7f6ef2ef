"""Per-kernel means of the counters in rocprofv3 counter_collection CSVs: python tools/summarize_sq.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if sub in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(os.path.basename(os.path.dirname(f)), k[:70])
        for c, v in cs.items():
            print("   %-24s mean %14.1f  (n=%d)" % (c, sum(v) / len(v), len(v)))
