"""Build libspherehand_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m spherehand_amd.build [--force]

No torch headers, no JIT cache: the .so lands next to this file so it travels
with the source tree.  hipcc cross-compiles gfx950 without a GPU present.
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SO = os.path.join(PKG, "libspherehand_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",                           # parity: one rounding per written op
    "-fhip-fp32-correctly-rounded-divide-sqrt",   # IEEE sqrt/div (hipcc default, stated)
    "-fno-fast-math", "-Wall", "-Wno-unused-function",
    # the first 16 kernel-argument dwords arrive in SGPRs with the wave instead of through an
    # s_load in front of the first global load (sphere backward: 7.90 -> 7.80 us per launch)
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
] + os.environ.get("SHR_HIPCC_EXTRA", "").split()       # experiments only (tools/)


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = sources() + glob.glob(os.path.join(PKG, "csrc", "*.h")) + \
        glob.glob(os.path.join(ROOT, "include", "*.h")) + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


OBJ_DIR = os.path.join(PKG, "csrc", "build")     # per-file objects (git-ignored: `build/`)


def _headers():
    return glob.glob(os.path.join(PKG, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.abspath(__file__)]


def _compile(src, force, verbose):
    """One translation unit -> object (skipped when the object is newer than the source and every header)."""
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj):
        t = os.path.getmtime(obj)
        if all(os.path.getmtime(d) <= t for d in [src] + _headers()):
            return obj
    cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + \
        ["-c", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(PKG, "csrc"), "-o", obj, src]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return obj


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 (in parallel, one object per file) and link the shared library."""
    if not (force or stale()):
        return SO
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ_DIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(lambda src: _compile(src, force, verbose), sources()))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
