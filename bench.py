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
resident in HBM.  Multi-GPU: every rank rasterizes its own 256 crops (weak
scaling, no data-path collective -- crops are independent); value =
crops of all ranks / max-over-ranks time.

Prints ONE JSON line on rank 0 (the driver's contract), with
  roofline      the dominant kernel's algorithmic HBM bytes / its mean launch
                duration (HIP events on the launching stream) vs the 8 TB/s peak
  cpu_baseline  the CPU oracle (oracle/, a port of the reference algorithm)
                timed on this host on the same batch, rank 0, N=1 only.
"""
import argparse
import ctypes
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


def timed_steps(step, steps, warmup, dist, device):
    """The driver's timing contract: `warmup` untimed steps, then exactly `steps` steps
    bracketed by a barrier + device synchronize on both sides; returns the MAX over
    ranks of the elapsed seconds (every rank gets the same number)."""
    def fence():
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        if device.type == "cuda":
            torch.cuda.synchronize(device)
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--launch", choices=["direct", "graph"], default="direct",
                    help="direct: two C-ABI calls per step; graph: one hipGraph replay per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        # the launcher's world size, --gpus and what RCCL actually connected must agree
        assert dist.get_world_size() == world == args.gpus, \
            "launched %d ranks (RCCL sees %d) but --gpus %d" % (world, dist.get_world_size(), args.gpus)
        seen = torch.ones(1, device=dev)
        dist.all_reduce(seen)                               # one RCCL all-reduce over xGMI: counts the ranks
        assert int(seen.item()) == world, "RCCL all-reduce saw %d ranks, expected %d" % (int(seen.item()), world)
        rccl_ranks = int(seen.item())
    else:
        assert args.gpus == 1, "--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus)
        rccl_ranks = 1

    from spherehand_amd import _lib
    lib = _lib.lib()
    spheres, grad = make_inputs(rank, dev)
    depth = torch.empty(BATCH, S, S, device=dev)
    owner = torch.empty(BATCH, S, S, device=dev, dtype=torch.uint8)
    gsph = torch.empty(BATCH, J, 4, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp, gp, dp, op, ap = spheres.data_ptr(), grad.data_ptr(), depth.data_ptr(), gsph.data_ptr(), owner.data_ptr()

    def fwd(s):
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

        elapsed = timed_steps(step, args.steps, args.warmup, dist, dev)

        # per-kernel mean launch duration: HIP events on the launching stream
        # around R back-to-back launches of one kernel
        def kernel_us(fn, reps=200):
            for _ in range(20):
                fn(sh)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = None
            for _ in range(5):
                e0.record(stream)
                for _ in range(reps):
                    fn(sh)
                e1.record(stream)
                e1.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / reps
                best = us if best is None else min(best, us)
            return best
        fwd_us = kernel_us(fwd)
        bwd_us = kernel_us(bwd)
        # context for the roofline: what ONE plain fill launch of the depth output (16.8 of the
        # forward's 21.1 MB, no arithmetic, same stream) takes at this batch size -- the practical
        # ceiling of any kernel that has to write a 256-crop batch per launch
        fill_us = kernel_us(lambda _s: depth.fill_(100.0))

    if rank == 0:
        # algorithmic bytes ("u8 argmin saved" variant: 165 808 B/crop fwd+bwd @128):
        #   fwd writes depth f32 + owner u8, reads the spheres; bwd reads grad f32 + owner u8 +
        #   spheres, writes grad_spheres
        bytes_fwd = BATCH * (4 * S * S + S * S + 16 * J)
        bytes_bwd = BATCH * (4 * S * S + S * S + 16 * J + 16 * J)
        dom, dom_us, dom_bytes = ("sphere_zbuf_bwd_kernel", bwd_us, bytes_bwd) if bwd_us >= fwd_us else \
            ("sphere_zbuf_fwd_kernel", fwd_us, bytes_fwd)
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic, traffic_src = pmc_traffic(dom)
        out = {
            "metric": "depth crops/s (raster fwd+bwd, 128x128, batch 256)",
            "value": round(world * BATCH * args.steps / elapsed, 1),
            "unit": "crops/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 6),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch-256 128x128 sphere raster fwd+bwd, 41 spheres/crop, "
                                   "JointAngleDataset poses (seed 0), grad N(0,1)",
                       "crops_per_gpu": BATCH, "image": [S, S], "spheres_per_crop": J,
                       "launch": args.launch, "parallelism": "batch-sharded x%d, no data-path collective" % world,
                       "rccl_ranks": rccl_ranks},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes,
                         "launch_us": {"fwd": round(fwd_us, 3), "bwd": round(bwd_us, 3)},
                         "plain_fill_of_the_depth_output_us": round(fill_us, 3)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spheres.cpu().numpy(), grad.cpu().numpy())
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
