"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/rNN_pmc_traffic.json.

usage: python tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json> [kernel substring ...]
bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950: FETCH_SIZE counts 64 B per 128-B request,
MI355X_MICROARCH.md section HBM)."""
import csv, glob, json, os, sys

def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            out.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return out

fetch_dir, write_dir, dst = sys.argv[1:4]
wanted = sys.argv[4:] or ["sphere_zbuf_fwd_kernel", "sphere_zbuf_bwd_kernel"]
fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
res = {}
for w in wanted:
    fk = [k for k in fe if w in k]; wk = [k for k in wr if w in k]
    if not fk or not wk:
        continue
    fv = sum((fe[k] for k in fk), []); wv = sum((wr[k] for k in wk), [])
    fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
    res[w] = {"FETCH_SIZE_KB_mean": fm, "FETCH_SIZE_launches": len(fv), "WRITE_SIZE_KB_mean": wm,
              "WRITE_SIZE_launches": len(wv), "hbm_bytes_per_launch": int(round((2 * fm + wm) * 1024))}
res["_note"] = ("rocprofv3 --kernel-trace --pmc <COUNTER> (separate passes) on `python bench.py --steps 50 --warmup 10 "
                "--no-cpu-baseline`; workload = batch 256, 128x128, 41 spheres.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                "(gfx950 FETCH_SIZE half-count correction, MI355X_MICROARCH.md section HBM).")
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1))
