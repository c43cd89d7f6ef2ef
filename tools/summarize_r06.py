"""gpurun_out/r06 (tools/collect_profiles_r06.sh) -> profiles/r06_*: python tools/summarize_r06.py"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r06")
P = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)

def shr_rows(src, dst, keep=lambda name: True):
    rows = list(csv.reader(open(src)))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(rows[0])
        for r in rows[1:]:
            if keep(r[0]):
                w.writerow([r[0].split("(")[0][:120]] + r[1:])

ours = lambda n: "shr::" in n or "group_norm" in n or "d2m_" in n
shr_rows(os.path.join(O, "stats_headline", "bench_kernel_stats.csv"), os.path.join(P, "r06_bench_kernel_stats.csv"))
shr_rows(os.path.join(O, "stats", "bench_kernel_stats.csv"), os.path.join(P, "r06_secondary_kernel_stats.csv"), ours)
for n in ("bench_line.json", "bench_line_graph.json", "bench_line_steps20.json"):
    shutil.copy(os.path.join(O, n), os.path.join(P, "r06_" + n))
# per-launch durations of the headline kernels from the trace: mean, median, quartiles (>= 2000 launches each)
import statistics
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(O, "stats_headline", "*kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "sphere_zbuf" in k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
med = {}
for k, v in dur.items():
    v.sort()
    med[k] = {"launches": len(v), "mean_us": round(statistics.mean(v), 3), "median_us": round(statistics.median(v), 3),
              "p25_us": round(v[len(v) // 4], 3), "p75_us": round(v[3 * len(v) // 4], 3), "min_us": round(v[0], 3)}
med["_note"] = ("rocprofv3 --kernel-trace --stats -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary: "
                "End - Start of every launch of the two headline kernels (the traced process's own HIP-event means are in "
                "its JSON line, gpurun_out/r06/stats_headline.log)")
try:
    line = [l for l in open(os.path.join(O, "stats_headline.log")) if l.startswith("{")][-1]
    med["traced_process_hip_event_us"] = json.loads(line)["roofline"]["launch_us"]
except Exception as e:      # noqa
    med["traced_process_hip_event_us"] = None
json.dump(med, open(os.path.join(P, "r06_headline_launch_durations.json"), "w"), indent=1)
if os.path.exists(os.path.join(O, "fuzz.log")):
    shutil.copy(os.path.join(O, "fuzz.log"), os.path.join(P, "r06_fuzz_summary.txt"))
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), os.path.join(O, "pmc_fetch"),
                       os.path.join(O, "pmc_write"), os.path.join(P, "r06_pmc_traffic.json"), "sphere_zbuf_fwd_kernel",
                       "sphere_zbuf_bwd_kernel"], stdout=subprocess.DEVNULL)
# config 5's loss: HBM traffic per launch of its kernels (tools/prof_mvloss.py under the same two PMC passes)
if os.path.isdir(os.path.join(O, "pmc_fetch_mvloss")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "summarize_pmc.py"), os.path.join(O, "pmc_fetch_mvloss"),
                           os.path.join(O, "pmc_write_mvloss"), os.path.join(P, "r06_pmc_traffic_config5_loss.json"),
                           "sphere_zbuf_mse_box_kernel", "d2m_compact_kernel", "d2m_points_kernel", "mv_loss_combine_kernel",
                           "mutual_project_fwd_kernel"], stdout=subprocess.DEVNULL)
    d = json.load(open(os.path.join(P, "r06_pmc_traffic_config5_loss.json")))
    d["_note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/prof_mvloss.py: "
                  "MutualProjectionLoss forward + backward, 1152 crops @256x256 (384 observed images of 256 KB = 100.7 MB), fresh "
                  "observations every call.  bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 half-count correction).")
    json.dump(d, open(os.path.join(P, "r06_pmc_traffic_config5_loss.json"), "w"), indent=1)
if os.path.exists(os.path.join(O, "mvloss_wall.log")):
    with open(os.path.join(O, "mvloss_wall.log")) as f, open(os.path.join(P, "r06_config5_loss_wall.txt"), "w") as g:
        g.write("# tools/prof_mvloss.py (20 steps each) and tools/ab_mvloss_overlap.py (200 steps each), untraced\n")
        g.writelines(l for l in f if l.startswith(("MutualProjectionLoss", "one stream", "render-and-compare")))
if os.path.isdir(os.path.join(O, "stats_mvloss")):
    shr_rows(os.path.join(O, "stats_mvloss", "mv_kernel_stats.csv"), os.path.join(P, "r06_config5_loss_kernel_stats.csv"), ours)
if os.path.isdir(os.path.join(O, "stats_c5size")):
    shr_rows(os.path.join(O, "stats_c5size", "c5_kernel_stats.csv"), os.path.join(P, "r06_config5_size_kernel_stats.csv"), ours)
if os.path.isdir(os.path.join(O, "stats_fk")):
    shr_rows(os.path.join(O, "stats_fk", "fk_kernel_stats.csv"), os.path.join(P, "r06_pose_kernels_stats.csv"), ours)
for src, dst in (("mvloss_timeline.log", "r06_config5_loss_timeline.txt"), ("tri.log", "r06_triangle_path.txt"),
                 ("mvloss_graph.log", "r06_config5_loss_eager_vs_graph.txt")):
    if os.path.exists(os.path.join(O, src)):
        with open(os.path.join(O, src)) as f, open(os.path.join(P, dst), "w") as g:
            g.writelines(l for l in f if not l.startswith(("W2026", "/opt/amdgpu", "[rocprofv3]", "E2026")) and "amdgpu.ids" not in l)
if os.path.isdir(os.path.join(O, "stats_synth")):
    shr_rows(os.path.join(O, "stats_synth", "synth_kernel_stats.csv"), os.path.join(P, "r06_synth_kernel_stats.csv"), ours)
for src, dst in (("synth.log", "r06_hand_synthesizer.txt"), ("mesh256.log", "r06_depth_render_256.txt"), ("floor_large.log", "r06_launch_floor_large_batch.txt")):
    if os.path.exists(os.path.join(O, src)):
        with open(os.path.join(O, src)) as f, open(os.path.join(P, dst), "w") as g:
            g.writelines(l for l in f if "amdgpu.ids" not in l)
for n in ("r06_mse_timeline.json", "r06_mse_phases.txt"):
    if os.path.exists(os.path.join(ROOT, "gpurun_out", n)):
        shutil.copy(os.path.join(ROOT, "gpurun_out", n), os.path.join(P, n))
if os.path.exists(os.path.join(ROOT, "gpurun_out", "r06_headline_timeline.json")):
    shutil.copy(os.path.join(ROOT, "gpurun_out", "r06_headline_timeline.json"), os.path.join(P, "r06_headline_timeline.json"))
if os.path.isdir(os.path.join(O, "cloop")):
    shr_rows(os.path.join(O, "cloop", "cloop_kernel_stats.csv"), os.path.join(P, "r06_cloop_kernel_stats.csv"))
    with open(os.path.join(P, "r06_cloop_lines.txt"), "w") as f:
        f.write("unprofiled:\n" + open(os.path.join(O, "cloop_plain.log")).read() + "\nunder rocprofv3 --kernel-trace --stats:\n" +
                open(os.path.join(O, "cloop_traced.log")).read())
# SQ counters: per kernel, mean per launch over every launch of every pass that saw it
sq = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in glob.glob(os.path.join(O, "sq_*", "*counter_collection.csv")) + glob.glob(os.path.join(O, "fetch_d2m*", "*counter_collection.csv")):
    tag = os.path.basename(os.path.dirname(f))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if not ours(k) or "group_norm" in k or "soft_argmax" in k or "fk_" in k or "lbs_" in k or "paint" in k or "noise" in k:
            continue
        key = k.replace("void ", "") + " grid=" + r["Grid_Size"] if "Grid_Size" in r else k.replace("void ", "")
        if tag.startswith("fetch") and "data_to_model" not in k:
            continue
        sq[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(sq.items()):
    m = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    d = {"launches_seen": max(len(v) for v in cs.values()), "counters_mean_per_launch": m}
    if "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        wc = m["SQ_WAVE_CYCLES"]
        d["share_of_wave_cycles"] = {"issuing (ACTIVE_INST_ANY)": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3),
                                     "of which VALU (ACTIVE_INST_VALU)": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3),
                                     "parked at s_waitcnt / barrier (WAIT_ANY)": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
                                     "issue-stalled (WAIT_INST_ANY)": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3)}
        if m.get("SQ_WAVES"):
            d["per_wave"] = {"VALU_instructions": round(m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"], 1),
                             "SALU_instructions": round(m.get("SQ_INSTS_SALU", 0) / m["SQ_WAVES"], 1),
                             "LDS_instructions": round(m.get("SQ_INSTS_LDS", 0) / m["SQ_WAVES"], 1),
                             "lifetime_cycles (4 x WAVE_CYCLES / WAVES)": round(4 * wc / m["SQ_WAVES"], 0)}
    if m.get("SQ_ACTIVE_INST_LDS"):
        d["lds_bank_conflict / active_inst_lds"] = round(m.get("SQ_LDS_BANK_CONFLICT", 0) / m["SQ_ACTIVE_INST_LDS"], 3)
        if "SQ_WAIT_INST_LDS" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"]:
            d["wait_inst_lds / wave_cycles"] = round(m["SQ_WAIT_INST_LDS"] / m["SQ_WAVE_CYCLES"], 4)
    if "FETCH_SIZE" in m:
        d["hbm_read_bytes_per_launch (2 x FETCH_SIZE KB, gfx950 half-count)"] = int(2 * m["FETCH_SIZE"] * 1024)
    out[k] = d
out["_note"] = ("rocprofv3 --kernel-trace --pmc <8 SQ counters> (two passes, sets A and B of tools/collect_profiles_r06.sh) on "
                "bench.py (--no-secondary: the headline kernels at batch 256; with the secondary set: every other kernel) and on "
                "tools/prof_d2m.py (1152 crops, S = 128 / 256).  SQ_*_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles "
                "summed over all waves; a kernel seen at several problem sizes is split by grid size.")
json.dump(out, open(os.path.join(P, "r06_sq_counters.json"), "w"), indent=1)
print(json.dumps({k: v.get("share_of_wave_cycles") for k, v in out.items() if isinstance(v, dict)}, indent=1))
