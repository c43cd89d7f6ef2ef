"""In-kernel timeline of the headline kernels (sphere raster forward + backward, 256 crops @128x128): the library built
with -DSHR_TIMELINE (sphere_zbuf.h SHR_TL: every wave stamps s_memtime at its phase boundaries), launched as bench.py
launches it, the stamps of the LAST launch of each kernel read back and reduced to a decomposition of the launch:
ramp (first wave's entry of the last workgroup to start) / prologue (to the first barrier) / scan or walk / stream-out
/ drain -- against the HIP-event duration of the same launches and the launch floor.

    python tools/headline_timeline.py build     (anywhere: cross-compiles tools/libspherehand_tl.so)
    python tools/headline_timeline.py           (on the GPU box: writes gpurun_out/r06_headline_timeline.json)"""
import ctypes
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libspherehand_tl.so")


def build():
    from spherehand_amd import build as b
    b.build()
    obj = os.path.join("/tmp", "sphere_raster_tl.o")
    flags = [f for f in b.FLAGS if f != "-shared"]
    subprocess.check_call([b.HIPCC] + flags + ["-DSHR_TIMELINE", "-c", "-I", os.path.join(ROOT, "include"), "-I",
                                               os.path.join(b.PKG, "csrc"), "-o", obj, os.path.join(b.PKG, "csrc", "sphere_raster.hip")])
    objs = [o for o in glob.glob(os.path.join(b.OBJ_DIR, "*.o")) if not o.endswith("sphere_raster.o")]
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO, obj] + objs)
    print(SO)


def main():
    import numpy as np
    import torch
    import bench
    lib = ctypes.CDLL(SO)
    vp, i = ctypes.c_void_p, ctypes.c_int
    lib.shr_sphere_raster_fwd_ex.argtypes = [vp, i, i, i, i, vp, vp, i, vp]
    lib.shr_sphere_raster_bwd.argtypes = [vp, vp, vp, i, i, i, i, vp, vp]
    lib.shr_debug_timeline.argtypes = [i, vp, ctypes.c_size_t]
    lib.shr_debug_timeline_rt.argtypes = [i, vp, ctypes.c_size_t]
    dev = torch.device("cuda", 0)
    N, J, S = bench.BATCH, bench.J, bench.S
    spheres, grad = bench.make_inputs(0, dev)
    depth = torch.empty(N, S, S, device=dev)
    owner = torch.empty(N, S, S, device=dev, dtype=torch.uint8)
    gs = torch.empty(N, J, 4, device=dev)
    stream = torch.cuda.Stream(device=dev)
    p = [t.data_ptr() for t in (spheres, depth, owner, grad, gs)]
    fwd = lambda s: lib.shr_sphere_raster_fwd_ex(p[0], N, J, S, S, p[1], p[2], 1, s)
    bwd = lambda s: lib.shr_sphere_raster_bwd(p[0], p[3], p[2], N, J, S, S, p[4], s)
    REPS = 60
    out = {"workload": "256 crops @128x128, 41 spheres (bench.make_inputs): forward with owner bytes on the touched rows, backward; "
                       "the instrumented build (-DSHR_TIMELINE), launches back to back as in bench.py's timed loop, %d launches each" % REPS,
           "clocks": "phases inside a workgroup: s_memtime (shader-clock ticks; NOT synchronised between CUs, only differences "
                     "inside one workgroup are used).  Ramp and tail of a launch: s_memrealtime (100 MHz, device-wide)."}
    with torch.cuda.stream(stream):
        assert fwd(stream.cuda_stream) == 0 and bwd(stream.cuda_stream) == 0
        t_us = {"fwd": bench.mean_launch_us(fwd, stream, 200, 5, 50, warm_ms=300.0), "bwd": bench.mean_launch_us(bwd, stream, 200, 5, 50)}
        res, rts = {}, {}
        for which, name, fn in ((0, "fwd", fwd), (1, "bwd", bwd)):
            rows, rrt = [], []
            for rep in range(REPS):
                for _ in range(3):                     # (back to back, as in the timed loop; every launch rewrites the stamps)
                    (bwd if which == 0 else fwd)(stream.cuda_stream)
                    fn(stream.cuda_stream)
                buf = np.zeros(256 * 16 * 8, np.uint64)
                rt = np.zeros(256 * 16 * 2, np.uint64)
                assert lib.shr_debug_timeline(which, buf.ctypes.data, buf.nbytes) == 0
                assert lib.shr_debug_timeline_rt(which, rt.ctypes.data, rt.nbytes) == 0
                rows.append(buf.reshape(256, 16, 8).astype(np.int64))
                rrt.append(rt.reshape(256, 16, 2).astype(np.int64))
            res[name], rts[name] = np.stack(rows), np.stack(rrt)   # [rep, workgroup, wave, slot]
    med = lambda v: float(np.median(v))
    for name in ("fwd", "bwd"):
        a, rt = res[name], rts[name]
        R = a.shape[0]
        e = a[..., 0].min(2)                                      # a workgroup's first wave enters (its own counter)
        rel = a - e[:, :, None, None]
        b1 = rel[..., 2].max(2)                                   # the first barrier is passed
        sc_fast, sc_slow = rel[..., 3].min(2), rel[..., 3].max(2)
        b2 = rel[..., 4].max(2)
        end = rel[..., 5].max(2)
        life = end                                                # (entry = 0)
        # shader clock: a workgroup's lifetime in ticks against the same lifetime on the 100-MHz counter
        life_rt = (rt[..., 1].max(2) - rt[..., 0].min(2)) * 10.0  # ns
        ghz = med(life.reshape(-1)) / med(life_rt.reshape(-1))
        us = lambda ticks: round(float(ticks) / (ghz * 1e3), 3)
        # the launch on the device-wide clock
        first = rt[..., 0].min((1, 2))
        ramp_med = med(np.median((rt[..., 0].min(2) - first[:, None]), 1)) * 0.01
        ramp_last = med((rt[..., 0].min(2) - first[:, None]).max(1)) * 0.01
        span = med(rt[..., 1].max((1, 2)) - first) * 0.01
        ph = {
            "entry -> records arrived (forward: list wave; backward: lead wave 0)": med(np.median(a[:, :, 0, 6] - a[:, :, 0, 0], 1)),
            "entry -> work list stands (wave 0)": med(np.median(a[:, :, 0, 7] - a[:, :, 0, 0], 1)),
            "entry -> first barrier passed = the PROLOGUE (records, list, run table / staging, background rows)": med(np.median(b1, 1)),
            "first barrier -> fastest wave done with its slice of the scan / walk": med(np.median(sc_fast - b1, 1)),
            "first barrier -> slowest wave done = the SCAN / WALK": med(np.median(sc_slow - b1, 1)),
            "second barrier -> end = STREAM-OUT (forward) / combine + store (backward)": med(np.median(end - b2, 1)),
            "workgroup lifetime, median": med(np.median(life, 1)),
            "workgroup lifetime, slowest workgroup of the launch": med(life.max(1)),
        }
        per_wave = {"reaches the first barrier (ticks after the workgroup's entry), waves 0..15": [int(np.median(rel[:, :, w, 1])) for w in range(16)],
                    "done with its scan / walk slice, waves 0..15": [int(np.median(rel[:, :, w, 3])) for w in range(16)],
                    "end, waves 0..15": [int(np.median(rel[:, :, w, 5])) for w in range(16)],
                    "entry, waves 0..15": [int(np.median(rel[:, :, w, 0])) for w in range(16)]}
        out[name + "_per_wave_ticks"] = per_wave
        out[name] = {
            "hip_event_us_per_launch (this instrumented build; the product build is bench.py's launch_us)": round(t_us[name], 3),
            "shader_clock_GHz (workgroup lifetimes: s_memtime ticks / s_memrealtime ns)": round(ghz, 3),
            "inside_a_workgroup_ticks": ph,
            "inside_a_workgroup_us": {k: us(v) for k, v in ph.items()},
            "launch_on_the_device_clock_us": {
                "ramp: first wave of the MEDIAN workgroup enters after the launch's first": round(ramp_med, 2),
                "ramp: first wave of the LAST workgroup enters after the launch's first": round(ramp_last, 2),
                "first entry -> last end (all 256 workgroups)": round(span, 2),
                "outside the stamps = HIP-event mean - that span (dispatch in front of the first wave, end-of-kernel "
                "write-back and release, the next launch's start)": round(t_us[name] - span, 2),
            },
        }
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_headline_timeline.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
