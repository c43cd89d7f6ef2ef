"""Where the driver's 20-step region loses 1 us per step against the 2000-step one.

Times K = 20, 40, 80, 200, 2000 headline steps (forward + backward on one stream) with the host clock, closing the
region three ways: torch.cuda.synchronize (bench.py), an event polled with hipEventQuery, and the stream's
hipStreamSynchronize -- and prints the intercept of time over K (what a region pays once) beside the slope (the step).
Run it with ROC_ACTIVE_WAIT_TIMEOUT / HIP_FORCE_SPIN... set to see what the runtime's wait policy does to it.
"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from spherehand_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.lib()
    gc.collect(); gc.freeze()
    B, S, J = bench.BATCH, bench.S, bench.J
    spheres, grad = bench.make_inputs(0, dev)
    depth = torch.empty(B, S, S, device=dev)
    owner = torch.empty(B, S, S, device=dev, dtype=torch.uint8)
    gsph = torch.empty(B, J, 4, device=dev)
    stream = torch.cuda.Stream(device=dev)
    sp, gp, dp, op, ap = spheres.data_ptr(), grad.data_ptr(), depth.data_ptr(), gsph.data_ptr(), owner.data_ptr()
    sh = stream.cuda_stream

    def step():
        lib.shr_sphere_raster_fwd_ex(sp, B, J, S, S, dp, ap, bench.OWNER_TOUCHED_ROWS, sh)
        lib.shr_sphere_raster_bwd(sp, gp, ap, B, J, S, S, op, sh)

    ev = torch.cuda.Event()

    def close_device():
        torch.cuda.synchronize(dev)

    def close_stream():
        stream.synchronize()

    def close_poll():
        ev.record(stream)
        while not ev.query():
            pass
        torch.cuda.synchronize(dev)

    with torch.cuda.stream(stream):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.1:
            for _ in range(50):
                step()
            stream.synchronize()
        for name, close in (("device_synchronize", close_device), ("stream_synchronize", close_stream),
                            ("event_query_poll", close_poll)):
            rows = []
            for K in (20, 40, 80, 200, 2000):
                best = []
                for _ in range(15 if K <= 200 else 3):
                    for _ in range(5):
                        step()
                    torch.cuda.synchronize(dev)
                    t0 = time.perf_counter()
                    for _ in range(K):
                        step()
                    close()
                    best.append((time.perf_counter() - t0) * 1e6)
                best.sort()
                rows.append((K, best[len(best) // 2], best[0]))
            slope = (rows[-1][1] - rows[0][1]) / (rows[-1][0] - rows[0][0])
            icpt = rows[0][1] - slope * rows[0][0]
            print(name, " ".join("K=%d: %.2f us/step (min %.2f)" % (k, m / k, mn / k) for k, m, mn in rows),
                  "| step %.2f us, once per region %.1f us" % (slope, icpt), flush=True)
        # the host's side alone: how long 20 steps take to ENQUEUE
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(20):
            step()
        enq = (time.perf_counter() - t0) * 1e6
        torch.cuda.synchronize(dev)
        print("enqueue of 20 steps: %.1f us" % enq)


if __name__ == "__main__":
    main()
