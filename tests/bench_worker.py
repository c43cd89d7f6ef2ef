"""Worker for tests/test_bench_cpu.py (torch.distributed.run, gloo, CPU): exercises
bench.timed_steps -- barrier, max-over-ranks -- with a sleep standing in for the step."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if __name__ == "__main__":
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    calls = {"n": 0}

    def step():                      # rank 1 is the slow one: 3 ms per step vs 1 ms
        calls["n"] += 1
        time.sleep(0.003 if rank == 1 else 0.001)

    elapsed = bench.timed_steps(step, 20, 3, dist, torch.device("cpu"))
    out = [None] * world
    dist.all_gather_object(out, {"rank": rank, "elapsed": elapsed, "calls": calls["n"]})
    if rank == 0:
        json.dump(out, open(sys.argv[1], "w"))
    dist.destroy_process_group()
