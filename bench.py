#!/usr/bin/env python3
"""Headline benchmark: sphere-set depth rasterizer, forward + backward.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): a batch of 256 crops of 128x128 px, 41
spheres per crop from 256 JointAngleDataset poses (torch seed 0 on rank 0, seed
r on rank r) pushed through forward kinematics; upstream gradient N(0,1).  One
STEP = one forward launch (spheres -> depth[256,128,128] + uint8 owner map) + one backward launch
(grad_depth, owner map -> grad_spheres[256,41,4]) through the C ABI, inputs and outputs
resident in HBM -- the pair ops.SphereDepthRaster issues: the owner map never leaves that pair,
so the forward runs with SHR_RASTER_OWNER_TOUCHED_ROWS (owner bytes on the rows the backward
reads; `roofline.full_owner_map` times the public full-map forward beside it).  Multi-GPU: every rank rasterizes its own 256 crops (weak
scaling, no data-path collective -- crops are independent); value =
crops of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (the driver's contract), with
  roofline      the dominant kernel's algorithmic HBM bytes / its MEAN launch
                duration (HIP events on the launching stream) vs the 8 TB/s peak;
                `large_batch` repeats it at 1152 and 9216 crops per launch (more
                crops than CUs), which separates the launch floor of a 256-crop
                launch from what the kernels sustain;
  secondary     (N=1) the other kernels of the path at the sizes DESIGN.md quotes,
                each with its own roofline fraction: config 5's per-GPU multiview
                loss, the triangle path, the fused render-and-compare, a hipGraph
                replay of the headline step, the reference-sized training step;
  cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm)
                timed on this host on the same batch, rank 0, N=1 only.
  N > 1         `secondary` carries the only communication of the design, max over ranks: the bare
                9.24-MB flat-bucket gradient all-reduce and the DDP training step at the reference's
                per-rank batch (network/engine.py:318-376).

Before the W warm-up steps the step runs for CLOCK_WARMUP_MS: the device's clocks ramp over tens
of milliseconds after an idle period, W = 5 steps are 70 us, and the timed region is meant to
measure the kernels, not the ramp (`config.clock_warmup_ms`).
"""
import argparse
import ctypes
import gc
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S = 128
BATCH = 256
J = 41
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_WARMUP_MS = 50.0
OWNER_TOUCHED_ROWS = 1   # SHR_RASTER_OWNER_TOUCHED_ROWS (include/spherehand_hip.h)
GRAD_BUCKET_FLOATS = 2308946   # the 1-stack hourglass: 9.24 MB of fp32 gradients


def make_inputs(rank, device):
    from spherehand_amd import hand_model
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.render import HandBallPrimitiveRender
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    mesh = hand_model.load_mesh()
    params = sample_poses(BATCH, seed=rank).to(device)
    fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(device)
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(device)
    with torch.no_grad():
        spheres = hbr.spheres(fk(params)).contiguous()          # [256,41,4]: HIP forward kinematics
    g = torch.Generator().manual_seed(1)
    grad = torch.randn(BATCH, S, S, generator=g)
    return spheres, grad.to(device)


def cpu_baseline(spheres_host, grad_host, budget_s=10.0):
    from oracle import oracle
    oracle.build()
    cores = oracle.num_threads()
    oracle.sphere_raster_fwd(spheres_host, S, S, want_argmin=False)   # warm-up
    oracle.sphere_raster_bwd(spheres_host, grad_host)
    t0 = time.perf_counter()
    passes = 0
    while True:
        oracle.sphere_raster_fwd(spheres_host, S, S, want_argmin=False)
        oracle.sphere_raster_bwd(spheres_host, grad_host)
        passes += 1
        el = time.perf_counter() - t0
        if el > budget_s or passes >= 400:
            break
    # one thread, for reference: a few crops of the same batch, ~2 s
    oracle.set_num_threads(1)
    n1, best1 = 16, None
    for _ in range(3):   # best of three (the first call after the thread-count change pays for it)
        t1 = time.perf_counter()
        oracle.sphere_raster_fwd(spheres_host[:n1], S, S, want_argmin=False)
        oracle.sphere_raster_bwd(spheres_host[:n1], grad_host[:n1])
        dt = time.perf_counter() - t1
        best1 = dt if best1 is None else min(best1, dt)
    one = n1 / best1
    oracle.set_num_threads(cores)
    return {"value": round(passes * BATCH / el, 1), "unit": "crops/s", "cores": cores, "kind": "port",
            "sample": "%d fwd+bwd passes over the same 256-crop 128x128 batch (%.1f s, OpenMP over crops)"
                      % (passes, el),
            "one_thread_crops_per_s": round(one, 1)}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the newest committed PMC summary
    (profiles/rNN_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this same command, corrected as MI355X_MICROARCH.md prescribes).  PMC counters
    cannot be read from inside the process, so this is the profiled value, not a
    live one; None if no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return int(d[kernel]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except (KeyError, ValueError):
        return None, None


VALU_ISSUE_PER_S = 1024 * 2.4e9 / 4     # wave64 VALU instructions the device can issue per second: 256 CUs x 4 SIMDs, one per
#                                          4 cycles per SIMD at 2.4 GHz (MI355X_MICROARCH.md) -- the ceiling of an issue-bound kernel


def sq_valu(kernel, grid=None):
    """Wave-level VALU instructions per launch of `kernel` (name substring; `grid` = total work-items of the launch, to
    tell a kernel's problem sizes apart) from the newest committed SQ-counter summary (profiles/rNN_sq_counters.json:
    rocprofv3 --pmc SQ_INSTS_VALU ... passes of the same launches; counters cannot be read from inside the process).
    -> (valu, salu, source) or (None, None, None)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")))
    if not files:
        return None, None, None
    try:
        d = json.load(open(files[-1]))
    except ValueError:
        return None, None, None
    best = None
    for k, v in d.items():
        if not isinstance(v, dict) or kernel not in k:
            continue
        if grid is not None and not k.endswith("grid=%d" % grid):
            continue
        c = v.get("counters_mean_per_launch", {})
        if "SQ_INSTS_VALU" in c and (best is None or v.get("launches_seen", 0) > best[2]):
            best = (c["SQ_INSTS_VALU"], c.get("SQ_INSTS_SALU"), v.get("launches_seen", 0))
    if best is None:
        return None, None, None
    return best[0], best[1], os.path.relpath(files[-1], ROOT)


def valu_roof(kernel, grid, us, hbm_bytes):
    """An ISSUE-bound kernel's line: bound "valu", the profiled wave-instruction count of this launch shape, the fraction
    of the device's VALU issue rate it sustains over the LIVE duration, and the HBM fraction beside it (hbm_frac: the
    same algorithmic bytes / duration / 8 TB/s the older lines called `frac`)."""
    valu, salu, src = sq_valu(kernel, grid)
    h = roof(hbm_bytes, us)
    out = {"bound": "valu", "valu_insts": None if valu is None else int(valu), "salu_insts": None if salu is None else int(salu),
           "valu_issue_peak_per_s": VALU_ISSUE_PER_S,
           "valu_issue_frac": None if valu is None else round(valu / (VALU_ISSUE_PER_S * us * 1e-6), 4),
           "valu_source": src, "hbm_frac": h["frac"], "hbm_achieved_GBs": h["achieved"],
           "algorithmic_bytes_per_launch": h["algorithmic_bytes_per_launch"]}
    return out


def headline_timeline():
    """The in-kernel timeline of the two headline kernels (newest profiles/rNN_headline_timeline.json, tools/
    headline_timeline.py: s_memtime stamps of every wave at the phase boundaries): where a launch's microseconds are."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_headline_timeline.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        out = {"source": os.path.relpath(files[-1], ROOT)}
        for k in ("fwd", "bwd"):
            w, l = d[k]["inside_a_workgroup_us"], d[k]["launch_on_the_device_clock_us"]
            pick = lambda dd, key: next(v for kk, v in dd.items() if key in kk)
            out[k] = {"prologue_us": pick(w, "PROLOGUE"), "scan_or_walk_us": pick(w, "slowest wave done"),
                      "stream_out_or_combine_us": pick(w, "second barrier"), "workgroup_lifetime_us": pick(w, "lifetime, median"),
                      "outside_the_waves_us": pick(l, "outside the stamps"),
                      "launch_us_of_the_instrumented_build": next(v for kk, v in d[k].items() if kk.startswith("hip_event_us"))}
        return out
    except (KeyError, ValueError, StopIteration):
        return None


def rocprof_avg_us():
    """AverageNs of the two headline kernels in the newest committed rocprofv3 --kernel-trace --stats summary of this
    command (profiles/rNN_bench_kernel_stats.csv), beside the live HIP-event means: {fwd, bwd, source} or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats.csv")))
    if not files:
        return None
    out = {"source": os.path.relpath(files[-1], ROOT)}
    try:
        for r in csv.DictReader(open(files[-1])):
            for key, name in (("fwd", "sphere_zbuf_fwd_kernel"), ("bwd", "sphere_zbuf_bwd_kernel")):
                if name in r["Name"] and key not in out:
                    out[key] = round(float(r["AverageNs"]) / 1e3, 3)
                    out[key + "_calls"] = int(r["Calls"])
    except (KeyError, ValueError):
        return None
    # the same two launches from a C loop (tools/prof_cloop.py): the tracer's averages with the host out of the way
    cl = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_cloop_kernel_stats.csv")))
    if cl:
        try:
            c = {"source": os.path.relpath(cl[-1], ROOT)}
            for r in csv.DictReader(open(cl[-1])):
                for key, name in (("fwd", "sphere_zbuf_fwd_kernel"), ("bwd", "sphere_zbuf_bwd_kernel")):
                    if name in r["Name"] and key not in c:
                        c[key] = round(float(r["AverageNs"]) / 1e3, 3)
            if "fwd" in c and "bwd" in c:
                out["launched_from_c"] = c
        except (KeyError, ValueError):
            pass
    return out if "fwd" in out and "bwd" in out else None


def timed_steps(step, steps, warmup, dist, device):
    """`warmup` untimed steps, then exactly `steps` steps; returns the MAX over ranks of the elapsed
    seconds (every rank gets the same number).  The region opens with barrier + device synchronize
    (the driver's contract).  It CLOSES with this rank's device synchronize, the clock stops, and only then
    come the closing barrier and the max-over-ranks all-reduce: a deliberate reading of "barrier +
    synchronize on both sides" -- the job's time is the slowest rank's time to finish its K steps, and
    the closing barrier's own latency (a collective of tens of microseconds against a 300-us region of
    20 steps) is not work of the job.  With one rank the two readings coincide."""
    def sync():
        if device.type == "cuda":
            torch.cuda.synchronize(device)

    def fence():
        sync()
        if dist is not None:
            dist.barrier()
        sync()
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    fence()
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def mean_launch_us(fn, stream, reps=200, batches=5, warm=20, warm_ms=0.0):
    """MEAN duration of one launch of fn(stream_handle): HIP events on the launching stream around `batches`
    runs of `reps` back-to-back launches (all of them averaged: what rocprofv3's AverageNs reports)."""
    sh = stream.cuda_stream
    for _ in range(warm):
        fn(sh)
    if warm_ms:   # a launch shape first seen after host-side work: let the clocks come back up before anything is timed
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < warm_ms:
            for _ in range(reps):
                fn(sh)
            stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total = 0.0
    for _ in range(batches):
        e0.record(stream)
        for _ in range(reps):
            fn(sh)
        e1.record(stream)
        e1.synchronize()
        total += e0.elapsed_time(e1) * 1e3
    return total / (reps * batches)


def roof(bytes_per_launch, us):
    gbs = bytes_per_launch / (us * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(bytes_per_launch)}


def large_batch(lib, _lib, dev, stream):
    """The headline kernels at more crops than CUs (1152 = config 5's per-GPU share, 9216 = its node-wide count):
    us per launch, us per 256 crops, roofline fraction on the same algorithmic bytes per crop."""
    from spherehand_amd import hand_model
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import HandBallPrimitiveRender
    mesh = hand_model.load_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
    out = {}
    for n in (1152, 9216):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev)
        owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
        grad = torch.randn(n, S, S, device=dev)
        gs = torch.empty(n, J, 4, device=dev)
        p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
        reps = 40 if n == 1152 else 8
        owner.fill_(254)
        _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], OWNER_TOUCHED_ROWS, stream.cuda_stream), "fwd")
        stream.synchronize()
        written = float((owner != 254).float().mean())
        f = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], OWNER_TOUCHED_ROWS, s),
                                                "fwd"), stream, reps, 5, 3, warm_ms=40.0)
        b = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s), "bwd"),
                           stream, reps, 5, 3, warm_ms=40.0)
        # one owner-byte convention for both directions: the owner bytes of the touched rows (what the pair moves)
        bf, bb = int(n * (4 * S * S + written * S * S + 16 * J)), int(n * (4 * S * S + written * S * S + 32 * J))
        out[str(n)] = {"fwd_us": round(f, 2), "bwd_us": round(b, 2), "fwd_us_per_256": round(f * 256 / n, 3),
                       "bwd_us_per_256": round(b * 256 / n, 3), "fwd_frac": roof(bf, f)["frac"],
                       "bwd_frac": roof(bb, b)["frac"], "crops_per_s_fwd_bwd": round(n / ((f + b) * 1e-6), 1)}
        # what these launches are bound by: the profiled wave-level VALU instructions of the same launch shapes (forward:
        # 1024 work-items per crop, backward: 512) against the device's issue rate over the live durations -- beside the HBM
        # fractions above, which stay the contract's `frac`
        for tag, kern, grid, us, hfrac in (("fwd", "sphere_zbuf_fwd_kernel<true, ", n * 1024, f, out[str(n)]["fwd_frac"]),
                                           ("bwd", "sphere_zbuf_bwd_kernel<true, true, false, 8, false, false>", n * 512, b, out[str(n)]["bwd_frac"])):
            valu, _, src = sq_valu(kern, grid)
            if valu is not None:
                vf = round(valu / (VALU_ISSUE_PER_S * us * 1e-6), 4)
                out[str(n)][tag + "_valu_issue_frac"] = vf
                out[str(n)][tag + "_bound"] = "valu" if vf > hfrac else "hbm"
                out[str(n)]["valu_source"] = src
        del sph, depth, owner, grad, gs
    out["1152@256"] = config5_size_kernels(lib, _lib, dev, stream, mesh)
    return out


def config5_size_kernels(lib, _lib, dev, stream, mesh, reps=25, batches=4):
    """The rasterizer kernels at config 5's OWN size: 1152 crops @256x256 (128 samples x 3 x 3 view pairs, the
    projections MutualProjectionLoss renders).  Plain forward without an owner map (what the loss's same-view mode and
    MutualProjection's no-grad call launch), forward + owner bytes on the touched rows, backward; HIP events over
    reps x batches back-to-back launches (tools/collect_profiles_r05.sh runs the same launches >= 1000 times under
    rocprofv3: profiles/r05_config5_size_kernel_stats.csv)."""
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    B5, S5 = 128, 256
    ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
    crit = MutualProjectionLoss(S5, mesh).to(dev)
    n = B5 * 9
    with torch.no_grad():
        _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + 1.0)
    rad = crit.data_to_model_criterion.radiuses.view(-1)
    sph = torch.cat([pts.squeeze(-1).reshape(n, J, 3), rad.view(1, J, 1).expand(n, J, 1)], -1).contiguous()
    del ds, crit, pts
    depth = torch.empty(n, S5, S5, device=dev)
    owner = torch.empty(n, S5, S5, device=dev, dtype=torch.uint8)
    grad = torch.randn(n, S5, S5, device=dev)
    gs = torch.empty(n, J, 4, device=dev)
    p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
    owner.fill_(254)
    _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S5, S5, p[1], p[2], OWNER_TOUCHED_ROWS, stream.cuda_stream), "fwd")
    stream.synchronize()
    written = float((owner != 254).float().mean())
    f0 = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S5, S5, p[1], None, 0, s), "fwd"),
                        stream, reps, batches, 3, warm_ms=40.0)
    f1 = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_fwd_ex(p[0], n, J, S5, S5, p[1], p[2], OWNER_TOUCHED_ROWS, s),
                                             "fwd"), stream, reps, batches, 3, warm_ms=40.0)
    b = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S5, S5, p[4], s), "bwd"),
                       stream, reps, batches, 3, warm_ms=40.0)
    b0, b1 = n * (4 * S5 * S5 + 16 * J), int(n * (4 * S5 * S5 + written * S5 * S5 + 16 * J))
    bb = int(n * (4 * S5 * S5 + written * S5 * S5 + 32 * J))
    return {"crops": n, "image": [S5, S5], "launches_timed": reps * batches,
            "fwd_depth_only": dict(us=round(f0, 1), **roof(b0, f0)),
            "fwd_with_owner_bytes_on_touched_rows": dict(us=round(f1, 1), owner_rows_written=round(written, 4), **roof(b1, f1)),
            "bwd": dict(us=round(b, 1), **roof(bb, b)),
            "crops_per_s_fwd_bwd": round(n / ((f1 + b) * 1e-6), 1)}


def secondary(lib, _lib, dev, stream, graph_step_us):
    """The other kernels of the path (rank 0, N=1), inputs resident in HBM, HIP events on the launching stream.
    Algorithmic bytes per crop are DESIGN.md section 4's (what must cross HBM once)."""
    import numpy as np
    from types import SimpleNamespace
    import depth_rasterization
    from spherehand_amd import hand_model, ops
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.multiview_utility import MutualProjectionLoss
    from spherehand_amd.render import DepthRender
    mesh = hand_model.load_mesh()
    sec = {"headline_step_graph_replay_us": round(graph_step_us, 3)}

    def training_step_ms():
        # ---- reference-sized training step: 25 x 3 real + 48 synthetic crops @64x64, every loss term on ------------
        # (on the DEFAULT stream, as a training script runs it: under a side stream autograd's backward pays extra event
        # synchronisation -- 9.9-10.7 ms against 9.1-9.6 for the same step)
        stream.synchronize()
        with torch.cuda.stream(torch.cuda.default_stream(dev)):
            import tempfile
            from spherehand_amd.engine import Engine
            o = SimpleNamespace(synthesize=True, mv_projection=True, mv_consistency=True, temporal=False, prior=False,
                                collision=True, bone_length=True, mode="Train", model_dir=tempfile.mkdtemp(), initial_model=None,
                                restore_from_model=None, restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None,
                                depth_resample=0, lr=1e-3, tag="b", image_size=64, log_every=10 ** 9, real_batch=25, synt_batch=48)
            ds = SyntheticMultiviewDataset(mesh, 50, 64, seed=0, device=dev)
            eng = Engine(o, mesh=mesh, real_train_dataset=ds, real_eval_dataset=ds, device=dev)
            eng.network.train()
            realb = [torch.stack([ds[i][k] for i in range(25)]) for k in range(4)]
            pose = sample_poses(48, seed=1)
            for _ in range(8):
                eng.step(realb, pose, True, True)
            # five batches of ten steps: the step is host-sensitive (~530 launches from Python), so the median batch is
            # reported, with the fastest and the slowest beside it
            batches = []
            for _ in range(5):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    eng.step(realb, pose, True, True)
                torch.cuda.synchronize(dev)
                batches.append((time.perf_counter() - t0) / 10 * 1e3)
            batches.sort()
            return round(batches[2], 3), [round(batches[0], 3), round(batches[-1], 3)]

    if os.environ.get("SHR_BENCH_TRAIN_FIRST"):   # experiment: the step before the heavy kernels of this function
        sec["training_step_first_ms"] = training_step_ms()[0]

    def torch_us(fn, reps, batches=3, warm=3):       # torch-launched work on the current (= launching) stream
        return mean_launch_us(lambda _s: fn(), stream, reps, batches, warm)

    # ---- BASELINE config 5, per-GPU share: 128 samples x 3 x 3 view pairs = 1152 crops @256x256 ----------
    B5, S5 = 128, 256
    ds = SyntheticMultiviewDataset(mesh, B5, S5, seed=0, device=dev)
    crit = MutualProjectionLoss(S5, mesh).to(dev)
    real, cam, inv = ds.dms.to(dev), ds.cam.to(dev), ds.inv_cam.to(dev)
    joints = (ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev)).requires_grad_(True)

    def mv_step():
        joints.grad = None
        loss, _ = crit(cam, inv, joints, real, True)
        loss.backward()
    n5 = B5 * 9
    # the data->model term compacts every observed image into a point list first; MutualProjectionLoss keeps the lists
    # while it is handed the same observations again (a second hourglass stack, a fitting loop).  Training feeds fresh
    # observations every step: the headline number is measured with the cache OFF, the cached one beside it.
    # (50 steps per timed batch: the events bracket the host's way to the first launch of a batch too -- one step's worth
    # of Python before the GPU has anything -- which 10-step batches charged at a tenth per step)
    crit.cache_points = False
    t_mv = torch_us(mv_step, 50, 3, 10)
    crit.cache_points = True
    t_mv_cached = torch_us(mv_step, 50, 3, 10)
    crit.cache_points = False

    def same_view_step():    # is_mv = False: the V same-view pairs only -- what the reference trains with after its
        joints.grad = None   # first 1500 iterations (network/engine.py:361); all V*V projections are still returned
        loss, _ = crit(cam, inv, joints, real, False)
        loss.backward()
    t_sv = torch_us(same_view_step, 50, 3, 10)
    # the same two steps as Engine's epoch loops run them: projected depth maps not materialised
    # (MutualProjectionLoss.return_projections = False -- the module's default returns the reference's pair)
    crit.return_projections = False
    t_mv_lean = torch_us(mv_step, 50, 3, 10)
    t_sv_lean = torch_us(same_view_step, 50, 3, 10)
    crit.return_projections = True
    crit.cache_points = False
    with torch.no_grad():
        _, pts = crit.mutual_projection(cam, inv, joints.detach())
    obs = real.view(B5 * 3, S5, S5).contiguous()
    index = (torch.arange(B5, device=dev, dtype=torch.int32).view(B5, 1, 1) * 3 +
             torch.arange(3, device=dev, dtype=torch.int32).view(1, 1, 3)).expand(B5, 3, 3).reshape(-1).contiguous()
    cen = pts.squeeze(-1).reshape(n5, J, 3).contiguous()
    rad = crit.data_to_model_criterion.radiuses.view(-1).contiguous()
    sph = torch.cat([cen, rad.view(1, J, 1).expand(n5, J, 1)], -1).contiguous()
    R = lib.shr_data_to_model_parts(n5, S5, S5)
    ls = torch.empty(n5 * R, device=dev); gr = torch.empty(n5 * R, J, 3, device=dev)
    a = [t.data_ptr() for t in (obs, index, cen, rad, ls, gr)]
    t_d2m = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_partial(a[0], a[1], a[2], 3, a[3], n5, J, S5, S5, R,
                                                                               a[4], a[5], s), "d2m"), stream, 20, 3, 3)
    # two-step data->model: compaction per observed IMAGE (384), search per crop (1152)
    M5 = B5 * 3
    ws = ops.d2m_compact(obs)
    Pp = ops.d2m_points_parts(n5)
    ls2 = torch.empty(n5 * Pp, device=dev); gr2 = torch.empty(n5 * Pp, J, 3, device=dev)
    t_cmp = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_compact(a[0], M5, S5, S5, ws.data_ptr(), s), "compact"),
                           stream, 20, 3, 3)
    t_pts = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_from_points(ws.data_ptr(), M5, a[1], a[2], 3, a[3], n5, J, S5, S5, Pp,
                                                                                   ls2.data_ptr(), gr2.data_ptr(), s), "points"), stream, 20, 3, 3)
    fg_px = int((obs <= 99).sum().item())                       # foreground pixels of the 384 observed images
    Rm = lib.shr_sphere_raster_mse_regions(S5, S5)
    dep = torch.empty(n5, S5, S5, device=dev); sse = torch.empty(n5 * Rm, device=dev)
    gsp = torch.empty(n5 * Rm, J, 4, device=dev)
    m = [t.data_ptr() for t in (sph, obs, index, dep, sse, gsp)]
    t_mse = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_mse(m[0], n5, J, S5, S5, m[1], m[2], m[3], m[4],
                                                                           m[5], s), "mse"), stream, 20, 3, 3)
    sec["config5_per_gpu_1152_crops_256x256"] = {
        "mutual_projection_loss_fwd_bwd_us": round(t_mv, 1),
        "mutual_projection_loss_fwd_bwd_same_observations_us": round(t_mv_cached, 1),
        "mutual_projection_loss_is": "forward + backward of MutualProjectionLoss on fresh observations every call (point-list "
                                     "cache off: what a training step pays); _same_observations_us = the same with the cache "
                                     "on and the observed images unchanged between calls (second hourglass stack, fitting loop)",
        "mutual_projection_loss_same_view_pairs_only_fwd_bwd_us": round(t_sv, 1),
        "without_materialised_projections_us": {"all_pairs": round(t_mv_lean, 1), "same_view_pairs_only": round(t_sv_lean, 1),
                                                "is": "return_projections = False, as Engine's epoch loops set it: the loss "
                                                      "returns (loss, None); the numbers above return the reference's pair"},
        "crops_per_s": round(n5 / (t_mv * 1e-6), 1),
        # two-step data->model (the path the loss takes): images read once (4 S^2 each) and their foreground written as
        # 8-byte points; the search reads every crop's image list (8 B per point) + the 41 records
        "d2m_compact_kernel": dict(us=round(t_cmp, 1), images=M5, **roof(M5 * 4 * S5 * S5 + 8 * fg_px, t_cmp)),
        # (issue-bound: DESIGN 4.3 -- 12 VALU instructions per (point, sphere) evaluated; 8-wave workgroups, one per crop)
        "d2m_points_kernel": dict(us=round(t_pts, 1), parts=Pp, **valu_roof("d2m_points_kernel<true, 8>", n5 * Pp * 512, t_pts,
                                                                             3 * 8 * fg_px + n5 * 16 * J)),
        # the streaming kernel (round 2; still behind shr_data_to_model_partial): every pair reads its image (4 S^2)
        "data_to_model_kernel": dict(us=round(t_d2m, 1), workgroups_per_crop=R,
                                     **valu_roof("data_to_model_kernel<true, 4, true>", n5 * R * 256, t_d2m, n5 * (4 * S5 * S5 + 16 * J))),
        # fused render-and-compare: reads the observed image, writes the projection (returned by the loss)
        # (issue-bound: DESIGN 4.2b; the box variant, Rm row regions per crop, 16 waves per workgroup)
        "sphere_zbuf_mse_kernel": dict(us=round(t_mse, 1), **valu_roof("sphere_zbuf_mse_box_kernel", n5 * Rm * 1024, t_mse,
                                                                       n5 * (8 * S5 * S5 + 32 * J))),
    }
    del ws, ls2, gr2
    del ds, crit, real, obs, dep, gsp, gr

    # the same two kernels at 128x128 (1152 crops)
    ds = SyntheticMultiviewDataset(mesh, B5, S, seed=0, device=dev)
    crit = MutualProjectionLoss(S, mesh).to(dev)
    with torch.no_grad():
        _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + 1.0)
    obs = ds.dms.to(dev).view(B5 * 3, S, S).contiguous()
    cen = pts.squeeze(-1).reshape(n5, J, 3).contiguous()
    R = lib.shr_data_to_model_parts(n5, S, S)
    ls = torch.empty(n5 * R, device=dev); gr = torch.empty(n5 * R, J, 3, device=dev)
    a = [t.data_ptr() for t in (obs, index, cen, rad, ls, gr)]
    t_d2m = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_partial(a[0], a[1], a[2], 3, a[3], n5, J, S, S, R,
                                                                               a[4], a[5], s), "d2m"), stream, 40, 3, 3)
    sec["data_to_model_kernel_1152_crops_128x128"] = dict(us=round(t_d2m, 1), **valu_roof("data_to_model_kernel<true, 4, false>", n5 * R * 256,
                                                                                         t_d2m, n5 * (4 * S * S + 16 * J)))
    # the same loss through the two-step path (384 images compacted once, 1152 point-list searches): what
    # ops.data_to_model takes from 2^23 observed pixels on -- at this size the fused loss gains nothing from it
    # (111 us either way), the stand-alone term does
    ws = ops.d2m_compact(obs)
    ls1 = torch.empty(n5, device=dev); gr1 = torch.empty(n5, J, 3, device=dev)
    t_c = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_compact(a[0], B5 * 3, S, S, ws.data_ptr(), s), "compact"),
                         stream, 40, 3, 3)
    t_p = mean_launch_us(lambda s: _lib.check(lib.shr_data_to_model_from_points(ws.data_ptr(), B5 * 3, a[1], a[2], 3, a[3], n5, J, S, S, 1,
                                                                                ls1.data_ptr(), gr1.data_ptr(), s), "points"),
                         stream, 40, 3, 3)
    sec["data_to_model_two_step_1152_crops_128x128"] = {"compact_us": round(t_c, 1), "search_us": round(t_p, 1),
                                                        "us": round(t_c + t_p, 1)}
    del ds, crit, obs, ws

    # ---- fused render-and-compare at the headline batch (256 crops @128x128, with the depth output) -------
    spheres, _ = make_inputs(0, dev)
    tgt = torch.full((BATCH, S, S), 100.0, device=dev); tgt[:, 32:96, 32:96] = 0.0
    dep = torch.empty(BATCH, S, S, device=dev); sse = torch.empty(BATCH, device=dev); gsp = torch.empty(BATCH, J, 4, device=dev)
    m = [t.data_ptr() for t in (spheres, tgt, dep, sse, gsp)]
    t = mean_launch_us(lambda s: _lib.check(lib.shr_sphere_raster_mse(m[0], BATCH, J, S, S, m[1], None, m[2], m[3], m[4], s),
                                            "mse"), stream, 200, 3, 20)
    sec["fused_render_and_compare_256_crops_128x128"] = dict(us=round(t, 2), **valu_roof("sphere_zbuf_mse_kernel<true, false>", BATCH * 1024, t,
                                                                                        BATCH * (8 * S * S + 32 * J)))

    # ---- triangle path: DepthRender (skinning + fused raster/clamp/resize) and the literal 640x640 drop-in ----
    fkm = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
    dr = DepthRender(mesh, S).to(dev)
    with torch.no_grad():
        T = fkm(sample_poses(BATCH, seed=1).to(dev))
        verts = dr.lbs(T, dr.camera, None).contiguous()
        fv = verts[:, dr.rasterizer.faces, 0:3].reshape(BATCH, -1, 3, 3).contiguous()
    nv, nf = verts.shape[1], dr.rasterizer.num_faces
    with torch.no_grad():
        t_dr = torch_us(lambda: dr(T), 20)
    outd = torch.empty(BATCH, S, S, device=dev)
    v = [verts.data_ptr(), dr.rasterizer.faces_i32.data_ptr(), outd.data_ptr()]
    t_md = mean_launch_us(lambda s: _lib.check(lib.shr_mesh_depth_fwd(v[0], v[1], BATCH, nv, nf, 640, S, 100.0, v[2], s),
                                               "mesh_depth"), stream, 20, 3, 3)
    # vertices (16 NV B) + shared indices + the S x S output per crop
    # ... and the module's own call: skinning + camera + raster + clamp + resize as ONE launch (shr_mesh_render_fwd)
    l = dr.lbs
    rf = [T.contiguous().data_ptr(), l.skin_vertex_start.data_ptr(), l.skin_bone.data_ptr(), l.skin_wv.data_ptr()]
    cxy = dr.camera
    t_fr = mean_launch_us(lambda s: _lib.check(lib.shr_mesh_render_fwd(rf[0], BATCH, 17, nv, rf[1], rf[2], rf[3], 1, cxy[0], cxy[1], cxy[2],
                                                                       cxy[3], None, v[1], nf, 640, S, 100.0, v[0], v[2], s),
                                               "mesh_render"), stream, 20, 3, 3)
    sec["depth_render_256_crops_128x128"] = {"module_us": round(t_dr, 1),
                                             "one_launch_mesh_lattice_kernel_with_skinning_us": round(t_fr, 1),
                                             # (issue-bound: the reference's seven IEEE divisions per sampled pixel)
                                             "mesh_lattice_kernel": dict(us=round(t_md, 1), **valu_roof("mesh_lattice_kernel<1, 16, 32, false", BATCH * 1024, t_md,
                                                                                                      BATCH * (16 * nv + 4 * S * S) + 12 * nf))}
    # ---- DepthRender at config 5's resolution (S = 256 from 640: no lattice; the band kernel with the resize epilogue) and
    # HandSynthesizer (network/util_modules.py:104-122: FK + RandScale + DepthRender + x depth_scale + DepthNoise + heat-maps)
    # as ONE launch, eager and as a hipGraph
    from spherehand_amd.util_modules import HandSynthesizer
    dr256 = DepthRender(mesh, 256).to(dev)
    with torch.no_grad():
        T64 = T[:64].contiguous()
        t_dr256 = torch_us(lambda: dr256(T64), 20)
        v256 = dr256.lbs(T64, dr256.camera, None).contiguous()
    out256 = torch.empty(64, 256, 256, device=dev)
    t_md256 = mean_launch_us(lambda s: _lib.check(lib.shr_mesh_depth_fwd(v256.data_ptr(), v[1], 64, nv, nf, 640, 256, 100.0, out256.data_ptr(), s),
                                                  "mesh_depth"), stream, 20, 3, 3)
    sec["depth_render_64_crops_256x256_us"] = {"module_us": round(t_dr256, 1), "tri_band_kernel_with_resize_epilogue_us": round(t_md256, 1),
                                               "is": "DepthRender(mesh, 256): skinning launch + the triangle band kernel at 640 x 640 whose "
                                                     "stream-out is clamp + bilinear resize (round 5: the tile kernel, 304 us)"}
    del out256, v256
    syn = HandSynthesizer(mesh, S, 16, 1.0, 0.01).to(dev)
    pose_syn = sample_poses(BATCH, seed=2).to(dev)
    syn_eager = sorted(mean_launch_us(lambda _s: syn(pose_syn), stream, 50, 1, 20 if i == 0 else 0, warm_ms=300.0 if i == 0 else 0.0)
                       for i in range(5))
    gsyn = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gsyn, stream=stream):
        syn(pose_syn)
    t_syn_graph = mean_launch_us(lambda _s: gsyn.replay(), stream, 100, 3, 10)
    del gsyn
    syn.one_launch = False
    t_syn3 = mean_launch_us(lambda _s: syn(pose_syn), stream, 50, 3, 20)
    syn.fused = False
    t_syn_modules = mean_launch_us(lambda _s: syn(pose_syn), stream, 30, 3, 10)
    sec["hand_synthesizer_256_crops_128x128_us"] = {
        "eager_us": round(syn_eager[2], 1), "eager_fastest_slowest_batch_us": [round(syn_eager[0], 1), round(syn_eager[-1], 1)],
        "one_hipgraph_us": round(t_syn_graph, 1), "three_launches_eager_us": round(t_syn3, 1),
        "module_by_module_us": round(t_syn_modules, 1), "heatmap_size": 16,
        "is": "256 poses -> (noised scaled depth [256,128,128], uv / depth heat-maps [256,41,16,16], key-points): ONE launch "
              "(shr_hand_synth_fwd: every workgroup runs its crop's FK + draws, skins, rasterizes, scales, noises, paints); "
              "three_launches = shr_synth_pose_fwd + shr_mesh_render_post_fwd + shr_heatmap_render_fwd; module_by_module = "
              "round 5's chain (fused = False: ~15 launches, torch RNG)"}
    del syn
    raw = torch.empty(BATCH, 640, 640, device=dev)
    r = [fv.data_ptr(), raw.data_ptr()]
    t_tri = mean_launch_us(lambda s: _lib.check(lib.shr_tri_raster_fwd(r[0], BATCH, nf, 640, 640, r[1], s), "tri"),
                           stream, 5, 3, 2)
    # (bound by the reference's seven IEEE divisions per covered pixel: latency / issue; HBM fraction beside it)
    sec["depth_rasterization_forward_640x640_256_crops"] = dict(us=round(t_tri, 1), **valu_roof("tri_band_kernel<false", None, t_tri,
                                                                                                BATCH * (36 * nf + 4 * 640 * 640)))
    del raw, fv

    # ---- the whole north-star chain once: pose[256,26] -> FK -> key-point skinning -> raster forward -> backward ->
    # skinning / FK backward -> d/d pose (mesh/kinematicsTransformation.py:157-177, mesh/render.py:81-90), captured
    # as ONE hipGraph (no Python between the launches) and replayed
    from spherehand_amd.render import HandBallPrimitiveRender
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
    pose = sample_poses(BATCH, seed=0).to(dev).requires_grad_(True)
    gdepth = torch.randn(BATCH, S, S, device=dev)

    def chain():                     # four launches: pose -> records, raster forward, backward, records -> d/d pose
        pose.grad = None
        hbr.pose_depth(fkm, pose).backward(gdepth)

    def chain_modules():             # the reference's module boundaries: T visits HBM, six launches
        pose.grad = None
        depth = ops.SphereDepthRaster.apply(hbr.spheres(fkm(pose)).contiguous(), S, S)
        depth.backward(gdepth)

    def graph_us(fn):
        for _ in range(3):
            fn()
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            fn()
        t = mean_launch_us(lambda _s: g.replay(), stream, 100, 3, 10)
        del g
        return t
    t_chain, t_chain_modules = graph_us(chain), graph_us(chain_modules)
    # (host-bound.  The host cores idle at 1.2 GHz while the graph replays above only wait for the GPU, and a process can
    # sit for half a second on a slow or shared core before the scheduler moves it -- 161 us per iteration for the first
    # 3000 iterations, 103 after, in one of three consecutive runs of the same loop on one box: timed after 1.2 s of
    # the loop itself, seven 50-iteration batches, the MEDIAN batch reported with the fastest and slowest beside it)
    eager_batches = sorted(mean_launch_us(lambda _s: chain(), stream, 50, 1, 30 if i == 0 else 0, warm_ms=1200.0 if i == 0 else 0.0)
                           for i in range(7))
    t_eager = eager_batches[3]
    t_eager_modules = sorted(mean_launch_us(lambda _s: chain_modules(), stream, 50, 1, 30 if i == 0 else 0,
                                            warm_ms=300.0 if i == 0 else 0.0) for i in range(5))[2]
    sec["pose_to_depth_to_pose_us"] = {"graph_replay_us": round(t_chain, 2), "eager_autograd_us": round(t_eager, 1),
                                       "eager_fastest_slowest_batch_us": [round(eager_batches[0], 1), round(eager_batches[-1], 1)],
                                       "crops_per_s_graph": round(BATCH / (t_chain * 1e-6), 1),
                                       "chain": "pose[256,26] -> pose_fwd (FK + key-point skinning, T in LDS) -> sphere raster fwd "
                                                "(+ owner map) -> bwd -> pose_bwd -> grad pose[256,26]: 4 launches",
                                       "module_by_module": {"graph_replay_us": round(t_chain_modules, 2),
                                                            "eager_autograd_us": round(t_eager_modules, 1),
                                                            "chain": "fk_fwd -> key-point skinning -> raster fwd -> bwd -> "
                                                                     "skinning bwd -> fk_bwd: 6 launches, same bits"}}

    if os.environ.get("SHR_BENCH_SKIP_TRAIN"):      # counter passes: the step's ~700 launches only bloat the trace
        return sec
    sec["training_step_25x3_real_48_synt_64x64_ms"], sec["training_step_fastest_slowest_batch_ms"] = training_step_ms()
    return sec


def collective_secondary(dist, rank, world, dev):
    """N > 1: the design's only communication (network/engine.py:318-376), max over ranks.
    (i) the bare all-reduce of the flat 9.24-MB gradient bucket; (ii) the DDP training step at the reference's
    per-rank batch (25 x 3 real + 48 synthetic crops @64x64, every loss term on) with that bucket in its backward."""
    import tempfile
    from types import SimpleNamespace
    from spherehand_amd import hand_model
    from spherehand_amd.datasets import SyntheticMultiviewDataset
    from spherehand_amd.engine import Engine
    from spherehand_amd.joint_angle import sample_poses

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def max_over_ranks(v):
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    bucket = torch.randn(GRAD_BUCKET_FLOATS, device=dev)
    for _ in range(5):
        dist.all_reduce(bucket)
    sync(); dist.barrier(); sync()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        dist.all_reduce(bucket)
    sync()
    ar_us = max_over_ranks((time.perf_counter() - t0) / reps * 1e6)
    out = {"grad_bucket_allreduce_us": round(ar_us, 1), "grad_bucket_bytes": GRAD_BUCKET_FLOATS * 4,
           "grad_bucket_busbw_GBs": round(2 * (world - 1) / world * GRAD_BUCKET_FLOATS * 4 / (ar_us * 1e-6) / 1e9, 2)}
    mesh = hand_model.load_mesh()
    o = SimpleNamespace(synthesize=True, mv_projection=True, mv_consistency=True, temporal=False, prior=False,
                        collision=True, bone_length=True, mode="Train", model_dir=tempfile.mkdtemp(), initial_model=None,
                        restore_from_model=None, restore_from_epoch=-1, num_stacks=1, epoch=3, dataset_dir=None,
                        depth_resample=0, lr=1e-3, tag="b", image_size=64, log_every=10 ** 9, real_batch=25, synt_batch=48)
    ds = SyntheticMultiviewDataset(mesh, 50, 64, seed=rank, device=dev)
    eng = Engine(o, mesh=mesh, real_train_dataset=ds, real_eval_dataset=ds, device=dev)
    assert eng.env.world == world and type(eng.ddp_network).__name__ == "DistributedDataParallel"
    eng.network.train()
    realb = [torch.stack([ds[i][k] for i in range(25)]) for k in range(4)]
    pose = sample_poses(48, seed=1 + rank)
    nsteps = int(os.environ.get("SHR_BENCH_DDP_STEPS", "10"))
    for _ in range(3):
        eng.step(realb, pose, True, True)
    sync(); dist.barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(nsteps):
        eng.step(realb, pose, True, True)
    sync()
    out["ddp_training_step_25x3_real_48_synt_64x64_ms"] = round(max_over_ranks((time.perf_counter() - t0) / nsteps * 1e3), 3)
    out["ddp_samples_per_s"] = round(world * (25 * 3 + 48) / (out["ddp_training_step_25x3_real_48_synt_64x64_ms"] * 1e-3), 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--launch", choices=["direct", "graph"], default="direct",
                    help="direct: two C-ABI calls per step; graph: one hipGraph replay per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the large-batch and secondary measurements")
    args = ap.parse_args()

    # `python bench.py --gpus N` started plainly (no launcher in front of it): start the N ranks here, exactly as the
    # driver's own multi-GPU command does -- one process per GPU under torch.distributed.run on 127.0.0.1 -- and hand
    # their output through (rank 0 prints the one JSON line).  --gpus 1 never takes this branch.
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this driver
        env.setdefault("OMP_NUM_THREADS", "4")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    # test hooks (tests/test_bench_gpu.py runs two ranks on ONE GPU, where RCCL refuses duplicate devices):
    # SHR_BENCH_DEVICE pins every rank to one device, SHR_BENCH_BACKEND=gloo swaps the process-group backend
    if os.environ.get("SHR_BENCH_DEVICE"):
        local_rank = int(os.environ["SHR_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # SHR_BENCH_FORCE_DIST=1 (tests/test_bench_gpu.py): the multi-rank branch with however many ranks were launched --
    # one rank under `torchrun --nproc-per-node 1` forms a real RCCL group on the one GPU of a test box
    force_dist = os.environ.get("SHR_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ["SHR_FORCE_DIST"] = "1"      # engine.DistEnv: DDP-wrap even at world 1 (collective_secondary)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SHR_BENCH_BACKEND", "nccl")
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get("SHR_BENCH_PG_TIMEOUT_S", "300")))   # a lost rank must not hang the node for the default 10 min
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        # the launcher's world size, --gpus and what RCCL actually connected must agree
        assert dist.get_world_size() == world == args.gpus, \
            "launched %d ranks (RCCL sees %d) but --gpus %d" % (world, dist.get_world_size(), args.gpus)
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                               # one RCCL all-reduce over xGMI: counts the ranks
        assert int(seen.item()) == world, "RCCL all-reduce saw %d ranks, expected %d" % (int(seen.item()), world)
        rccl_ranks = int(seen.item())
    else:
        assert args.gpus == 1, "launched as one rank (WORLD_SIZE=1) but --gpus %d: torch.distributed.run --nproc-per-node %d" \
            % (args.gpus, args.gpus)
        rccl_ranks = 1

    from spherehand_amd import _lib
    lib = _lib.lib()
    # Everything imported so far (torch: about a million tracked objects) goes to the permanent generation: a full
    # collection of it in the middle of a timed loop is a 40-ms host stall with the GPU idle (tools' A/B runs: one
    # 200-step run in six read 470 us per step instead of 276).  Collections stay ON; they only stop re-walking those.
    gc.collect()
    gc.freeze()
    spheres, grad = make_inputs(rank, dev)
    depth = torch.empty(BATCH, S, S, device=dev)
    owner = torch.empty(BATCH, S, S, device=dev, dtype=torch.uint8)
    gsph = torch.empty(BATCH, J, 4, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp, gp, dp, op, ap = spheres.data_ptr(), grad.data_ptr(), depth.data_ptr(), gsph.data_ptr(), owner.data_ptr()

    def fwd(s):
        _lib.check(lib.shr_sphere_raster_fwd_ex(sp, BATCH, J, S, S, dp, ap, OWNER_TOUCHED_ROWS, s), "fwd")

    def fwd_full(s):     # the public contract: owner map complete
        _lib.check(lib.shr_sphere_raster_fwd(sp, BATCH, J, S, S, dp, ap, s), "fwd")

    def bwd(s):
        _lib.check(lib.shr_sphere_raster_bwd(sp, gp, ap, BATCH, J, S, S, op, s), "bwd")

    with torch.cuda.stream(stream):
        sh = stream.cuda_stream
        graph = None
        if args.launch == "graph":
            fwd(sh); bwd(sh)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                fwd(sh); bwd(sh)

        def step():
            if graph is not None:
                graph.replay()
            else:
                fwd(sh); bwd(sh)

        # owner bytes the flagged forward writes (its algorithmic bytes: the rows some sphere's box touches)
        owner.fill_(254)
        fwd(sh)
        stream.synchronize()
        owner_written = float((owner != 254).float().mean())
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < CLOCK_WARMUP_MS:
            for _ in range(50):
                step()
            stream.synchronize()
        elapsed = timed_steps(step, args.steps, args.warmup, dist, dev)

        def kernel_us(fn, reps=200, batches=5, warm=20):
            return mean_launch_us(fn, stream, reps, batches, warm)
        fwd_us = kernel_us(fwd)
        bwd_us = kernel_us(bwd)
        fwd_full_us = kernel_us(fwd_full)
        # context for the roofline: what ONE plain fill launch of the depth output (16.8 of the
        # forward's 21.1 MB, no arithmetic, same stream) takes at this batch size -- the practical
        # ceiling of any kernel that has to write a 256-crop batch per launch
        fill_us = kernel_us(lambda _s: depth.fill_(100.0))
        # the launch floor of the forward's shape (one 1024-thread workgroup with the whole CU's LDS per crop) over
        # the forward's bytes: records in, depth + the touched rows' owner bytes out, nothing else (capi.hip)
        rows_t = int(round(owner_written * S))
        floor_lds = 160 * 1024
        r0f = (S - rows_t) // 2

        def floor(s):
            _lib.check(lib.shr_selftest_launch_floor(sp, BATCH, J, S, S, r0f, r0f + rows_t, dp, ap, floor_lds, s), "floor")
        floor_us = kernel_us(floor)
        # the same K steps with the PUBLIC forward (complete owner map), timed like the headline
        def step_full():
            fwd_full(sh); bwd(sh)
        # (the same clock warm-up as the headline region: this one follows a run of single-kernel loops and read 13.5 M on
        # the driver's box against 16.7 M here in round 4 -- a cold region, not a slower step)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < CLOCK_WARMUP_MS:
            for _ in range(50):
                step_full()
            stream.synchronize()
        elapsed_full = timed_steps(step_full, args.steps, args.warmup, dist, dev)
        # STRONG-scaling leg: BASELINE's batch of 256 crops split over the ranks (256 / N each, the same step, max-over-ranks
        # time) -- the metric as BASELINE.json words it ("batch 256 ... at 1/2/4/8 MI355X"); at N = 1 it IS the headline.
        strong = None
        if world > 1 or dist is not None:
            n_strong = max(1, BATCH // world)
            sp_s, gp_s = spheres[:n_strong].contiguous(), grad[:n_strong].contiguous()
            d_s, o_s, g_s = depth[:n_strong], owner[:n_strong], gsph[:n_strong]
            ps = [t.data_ptr() for t in (sp_s, gp_s, d_s, g_s, o_s)]

            def step_strong():
                _lib.check(lib.shr_sphere_raster_fwd_ex(ps[0], n_strong, J, S, S, ps[2], ps[4], OWNER_TOUCHED_ROWS, sh), "fwd")
                _lib.check(lib.shr_sphere_raster_bwd(ps[0], ps[1], ps[4], n_strong, J, S, S, ps[3], sh), "bwd")
            t0 = time.perf_counter()
            while (time.perf_counter() - t0) * 1e3 < CLOCK_WARMUP_MS:
                for _ in range(50):
                    step_strong()
                stream.synchronize()
            el_s = timed_steps(step_strong, args.steps, args.warmup, dist, dev)
            strong = {"value": round(world * n_strong * args.steps / el_s, 1), "unit": "crops/s", "crops_per_rank": n_strong,
                      "global_batch": world * n_strong, "ms_per_step": round(el_s / args.steps * 1e3, 6)}
        big = sec = None
        if rank == 0 and world == 1 and dist is None and not args.no_secondary:
            # one hipGraph replay of the headline step (forward + backward as one graph launch)
            g2 = torch.cuda.CUDAGraph()
            fwd(sh); bwd(sh); stream.synchronize()
            with torch.cuda.graph(g2, stream=stream):
                fwd(sh); bwd(sh)
            graph_us = mean_launch_us(lambda _s: g2.replay(), stream, 200, 3, 20)
            # what a graph launch costs by itself: EIGHT steps in one graph, per step -- the difference to the one-step
            # graph is the fixed cost of a hipGraphLaunch on this runtime (a submission of its own with its barrier
            # packets, where direct launches pipeline in the queue), not the kernels
            g8 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g8, stream=stream):
                for _ in range(8):
                    fwd(sh); bwd(sh)
            graph8_us = mean_launch_us(lambda _s: g8.replay(), stream, 50, 3, 5) / 8.0
            del g8
            big = large_batch(lib, _lib, dev, stream)
            sec = secondary(lib, _lib, dev, stream, graph_us)
            sec["headline_step_in_a_graph_of_8_steps_us"] = round(graph8_us, 3)

    coll = None
    if dist is not None and not args.no_secondary:
        try:                                                  # (the headline line is printed whatever happens here)
            coll = collective_secondary(dist, rank, world, dev)   # every rank takes part; rank 0 prints
        except Exception as e:                                # noqa: BLE001
            coll = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank == 0:
        # algorithmic bytes ("u8 argmin saved" variant: 165 808 B/crop fwd+bwd @128):
        #   fwd writes depth f32 + owner u8, reads the spheres; bwd reads grad f32 + owner u8 +
        #   spheres, writes grad_spheres
        #   (forward with SHR_RASTER_OWNER_TOUCHED_ROWS: the owner bytes of the touched rows only -- the measured
        #   fraction; the backward is priced at the whole images although it reads the touched rows only)
        # Three byte conventions, each applied to BOTH directions and named in the line:
        #   touched   depth / grad image + the owner bytes of the touched rows (what this pair has to move) + records
        #   full      ... + the complete owner map (the public argmin contract; SURVEY 8d "u8 argmin saved")
        #   survey8d  SURVEY 8d's primary figure: no owner map at all (66 192 + 66 848 B per crop)
        bytes_fwd = int(BATCH * (4 * S * S + owner_written * S * S + 16 * J))
        bytes_bwd = int(BATCH * (4 * S * S + owner_written * S * S + 16 * J + 16 * J))
        bytes_fwd_full = BATCH * (4 * S * S + S * S + 16 * J)
        bytes_bwd_full = BATCH * (4 * S * S + S * S + 16 * J + 16 * J)
        bytes_fwd_8d = BATCH * (4 * S * S + 16 * J)
        bytes_bwd_8d = BATCH * (4 * S * S + 16 * J + 16 * J)
        dom, dom_us, dom_bytes = ("sphere_zbuf_bwd_kernel", bwd_us, bytes_bwd) if bwd_us >= fwd_us else \
            ("sphere_zbuf_fwd_kernel", fwd_us, bytes_fwd)
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(dom)
        rp = rocprof_avg_us()
        frac_rocprof = None
        if rp is not None and "launched_from_c" in rp:      # the committed rocprofv3 summary of the C-loop launches
            key = "bwd" if dom.endswith("bwd_kernel") else "fwd"
            frac_rocprof = {"frac": roof(dom_bytes, rp["launched_from_c"][key])["frac"],
                            "frac_fwd": roof(bytes_fwd, rp["launched_from_c"]["fwd"])["frac"],
                            "frac_bwd": roof(bytes_bwd, rp["launched_from_c"]["bwd"])["frac"],
                            "avg_us": {"fwd": rp["launched_from_c"]["fwd"], "bwd": rp["launched_from_c"]["bwd"]},
                            "source": rp["launched_from_c"]["source"],
                            "is": "same algorithmic bytes / AverageNs of the committed rocprofv3 --kernel-trace --stats "
                                  "run of the C-loop launches (tools/collect_profiles_r06.sh); a different process "
                                  "than this line's live HIP-event means"}
        out = {
            "metric": "depth crops/s (raster fwd+bwd, 128x128, batch 256)",
            "value": round(world * BATCH * args.steps / elapsed, 1),
            "value_full_owner_map": round(world * BATCH * args.steps / elapsed_full, 1),
            "value_is": "forward with SHR_RASTER_OWNER_TOUCHED_ROWS + backward: the pair ops.SphereDepthRaster issues "
                        "(the owner map never leaves it); value_full_owner_map = the same steps with the public "
                        "forward that completes the uint8 owner map (round 2's definition of the step)",
            "unit": "crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            # the same metric with BASELINE's batch of 256 SPLIT over the ranks (256 / N crops each; N = 1: the headline itself)
            "strong": strong if strong is not None else {"value": round(world * BATCH * args.steps / elapsed, 1), "unit": "crops/s",
                                                         "crops_per_rank": BATCH, "global_batch": BATCH,
                                                         "ms_per_step": round(elapsed / args.steps * 1e3, 6)},
            "strong_is": "strong scaling: a global batch of 256 crops, 256 / N per rank, max-over-ranks time of the same K steps; "
                         "`value` is weak scaling (256 crops per rank)",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch-256 128x128 sphere raster fwd+bwd, 41 spheres/crop, "
                                   "JointAngleDataset poses (seed 0), grad N(0,1)",
                       "crops_per_gpu": BATCH, "image": [S, S], "spheres_per_crop": J,
                       "launch": args.launch, "clock_warmup_ms": CLOCK_WARMUP_MS,
                       "parallelism": "batch-sharded x%d, no data-path collective" % world,
                       "rccl_ranks": rccl_ranks, "backend": dist.get_backend() if dist is not None else None},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "launch_us": {"fwd": round(fwd_us, 3), "bwd": round(bwd_us, 3)},
                         "launch_us_is": "mean of 1000 back-to-back launches (HIP events on the launching stream, this "
                                         "process); `frac` uses it.  rocprof_avg_us = AverageNs of the committed "
                                         "rocprofv3 --kernel-trace --stats run of this command (the tracer "
                                         "serialises the dispatches and, from Python, the host then costs more per "
                                         "launch than the kernel takes; launched_from_c = the same launches from a C "
                                         "loop under the tracer: profiles/README.md)",
                         "rocprof_avg_us": rp,
                         "frac_rocprof": frac_rocprof,
                         "owner_map": "touched rows only (SHR_RASTER_OWNER_TOUCHED_ROWS): %.1f %% of the owner bytes"
                                      % (100 * owner_written),
                         "full_owner_map": dict(fwd_us=round(fwd_full_us, 3), **roof(bytes_fwd_full, fwd_full_us)),
                         "frac_fwd": roof(bytes_fwd, fwd_us)["frac"], "frac_bwd": roof(bytes_bwd, bwd_us)["frac"],
                         "byte_conventions": {
                             "touched_rows": {"fwd_bytes": bytes_fwd, "bwd_bytes": bytes_bwd,
                                              "frac_fwd": roof(bytes_fwd, fwd_us)["frac"],
                                              "frac_bwd": roof(bytes_bwd, bwd_us)["frac"],
                                              "is": "depth / grad image + owner bytes of the touched rows + records; "
                                                    "`frac`, `frac_fwd`, `frac_bwd` use it"},
                             "full_owner_map": {"fwd_bytes": bytes_fwd_full, "bwd_bytes": bytes_bwd_full,
                                                "frac_fwd": roof(bytes_fwd_full, fwd_full_us)["frac"],
                                                "frac_bwd": roof(bytes_bwd_full, bwd_us)["frac"],
                                                "is": "SURVEY 8d 'u8 argmin saved' (165 808 B per crop fwd + bwd); the "
                                                      "forward timed is the public one that writes the whole map"},
                             "survey8d_no_owner_map": {"fwd_bytes": bytes_fwd_8d, "bwd_bytes": bytes_bwd_8d,
                                                       "frac_fwd": roof(bytes_fwd_8d, fwd_us)["frac"],
                                                       "frac_bwd": roof(bytes_bwd_8d, bwd_us)["frac"],
                                                       "is": "SURVEY 8d's primary 133 040 B per crop (argmin recomputed): "
                                                             "the owner bytes these kernels do move are not counted"}},
                         "launch_floor": {"us": round(floor_us, 3), "fwd_over_floor": round(fwd_us / floor_us, 3),
                                          "bytes": bytes_fwd,
                                          "shape": "%d workgroups x 1024 threads, %d B of LDS each (one per CU), 16-byte "
                                                   "sc1 stores of depth + the owner bytes of %d rows, the records read: "
                                                   "no arithmetic (shr_selftest_launch_floor)" % (BATCH, floor_lds, rows_t),
                                          **{k: v for k, v in roof(bytes_fwd, floor_us).items() if k in ("achieved", "frac")}},
                         "plain_fill_of_the_depth_output_us": round(fill_us, 3),
                         "timeline": headline_timeline()},
        }
        if big is not None:
            out["roofline"]["large_batch"] = big
        if sec is not None:
            out["secondary"] = sec
        if coll is not None:
            out["secondary"] = coll
        if world == 1 and dist is None and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spheres.cpu().numpy(), grad.cpu().numpy())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
