"""GPU: bench.py's multi-rank path end to end -- two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`) on ONE MI355X (backend gloo and a
pinned device through bench.py's test hooks: RCCL refuses two ranks on one device).  Checks the JSON contract:
whole-job value = crops of all ranks / max-over-ranks time, weak scaling, the rank count the collective saw, and
the two timings of the design's only communication (bare gradient-bucket all-reduce, DDP training step)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from test_ddp_cpu import free_port

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_one_gpu():
    env = dict(os.environ, SHR_BENCH_BACKEND="gloo", SHR_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               SHR_BENCH_DDP_STEPS="3")
    from conftest import run_torchrun
    out = run_torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10"], env=env,
                       timeout=600, capture=True)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out                               # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 50 and d["warmup"] == 10 and d["scaling"] == "weak"
    assert d["config"]["rccl_ranks"] == 2 and d["config"]["crops_per_gpu"] == 256
    assert abs(d["value"] - 2 * 256 * 50 / (d["ms_per_step"] * 1e-3 * 50)) <= 1e-3 * d["value"]
    assert "cpu_baseline" not in d                            # N = 1 only
    sec = d["secondary"]                                      # N > 1: the collective, max over ranks
    assert sec["grad_bucket_bytes"] == 2308946 * 4 and sec["grad_bucket_allreduce_us"] > 0
    assert sec["ddp_training_step_25x3_real_48_synt_64x64_ms"] > 0 and sec["ddp_samples_per_s"] > 0
    assert "config5_per_gpu_1152_crops_256x256" not in sec   # the single-GPU kernel table stays with N = 1
    assert 0 < d["roofline"]["frac"] < 1


def test_bench_one_rank_over_rccl():
    """bench.py's `world > 1` branch on RCCL itself: one rank under torchrun with SHR_BENCH_FORCE_DIST=1 --
    init_process_group("nccl", device_id=...), the rank-count all-reduce, the closing barrier of the timed region, the
    gradient-bucket all-reduce and the DDP-wrapped training step all run through RCCL on the MI355X."""
    env = dict(os.environ, SHR_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", SHR_BENCH_DDP_STEPS="3")
    from conftest import run_torchrun
    out = run_torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "10"], env=env,
                       timeout=600, capture=True)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["config"]["backend"] == "nccl" and d["config"]["rccl_ranks"] == 1 and d["n_gpus"] == 1
    sec = d["secondary"]
    assert "error" not in sec, sec
    assert sec["grad_bucket_bytes"] == 2308946 * 4 and sec["grad_bucket_allreduce_us"] > 0
    assert sec["ddp_training_step_25x3_real_48_synt_64x64_ms"] > 0
    assert d["value"] > 1e6 and 0 < d["roofline"]["frac"] < 1
    assert d["strong"]["crops_per_rank"] == 256 and d["strong"]["value"] > 1e6       # (one rank: the whole batch)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher in front of it (how the driver starts its N = 1 run): bench.py itself
    re-executes under torch.distributed.run with two ranks and still prints ONE JSON line from rank 0 (gloo + a pinned
    device through the test hooks: this box has one GPU)."""
    env = dict(os.environ, SHR_BENCH_BACKEND="gloo", SHR_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2",
               SHR_BENCH_DDP_STEPS="2")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["rccl_ranks"] == 2
    assert abs(d["value"] - 2 * 256 * 20 / (d["ms_per_step"] * 1e-3 * 20)) <= 1e-3 * d["value"]
    # both scaling legs: weak (256 crops per rank: `value`) and strong (BASELINE's batch of 256 split over the ranks)
    assert d["scaling"] == "weak" and d["strong"]["crops_per_rank"] == 128 and d["strong"]["global_batch"] == 256
    assert abs(d["strong"]["value"] - 256 / (d["strong"]["ms_per_step"] * 1e-3)) <= 1e-3 * d["strong"]["value"]
    assert d["strong"]["value"] > 1e5


def test_bench_rejects_a_world_size_mismatch():
    """one rank launched (WORLD_SIZE=1 in the environment) but --gpus 2 asked for: refused, not silently a 1-GPU number"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "nproc-per-node" in (r.stderr + r.stdout)
