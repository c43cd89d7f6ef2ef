"""Register budgets the launch shapes depend on (no GPU: hipcc cross-compiles gfx950 and the kernel
descriptors are read from the assembly).

Two workgroups of 16 waves share a CU only at 8 waves per SIMD: <= 64 VGPRs AND, on gfx950, few enough
SGPRs (measured on the descriptors' sgpr_count: kernels with 84, 93 and 98-106 lost the second workgroup,
kernels with 71-78 kept it).  The backward keeps its in-flight loads in v[96:113], above the
96 registers the compiler may allocate."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ASM = None


def _bodies(text, pattern):
    """{mangled name: instruction lines} of the kernels whose name matches `pattern`."""
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    out = {}
    for n, i in enumerate(starts):
        name = lines[i].split(":")[0]
        if re.search(pattern, name):
            out[name] = lines[i:starts[n + 1] if n + 1 < len(starts) else len(lines)]
    return out


def _compile_asm(unit, tmp):
    from spherehand_amd import build
    out = str(tmp / (unit + ".s"))
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                                                   "-I", os.path.join(build.PKG, "csrc"), "-o", out,
                                                   os.path.join(build.PKG, "csrc", unit + ".hip")],
                          stderr=subprocess.DEVNULL)
    return open(out).read()


@pytest.fixture(scope="module")
def descriptors(tmp_path_factory):
    from spherehand_amd import build
    out = str(tmp_path_factory.mktemp("asm") / "sphere_raster.s")
    flags = [f for f in build.FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call([build.HIPCC] + flags + ["-S", "--cuda-device-only", "-I", os.path.join(ROOT, "include"),
                                                   "-I", os.path.join(build.PKG, "csrc"), "-o", out,
                                                   os.path.join(build.PKG, "csrc", "sphere_raster.hip")],
                          stderr=subprocess.DEVNULL)
    text = open(out).read()
    global _ASM
    _ASM = text
    meta = text[text.index("amdhsa.kernels:"):]
    kernels = {}
    for block in meta.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        kernels[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                         for k in ("sgpr_count", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
    return kernels


def test_two_forward_workgroups_fit_a_cu(descriptors):
    flags = lambda n: re.findall(r"Lb([01])E", re.search(r"kernelI((?:L[bi]\d+E)+)", n).group(1))
    box = {n: d for n, d in descriptors.items()
           if "sphere_zbuf_fwd_kernel" in n and flags(n)[2:5] == ["1", "0", "1"]}
    assert len(box) == 6          # OWNER x VEC4; power-of-two images, one workgroup per crop (PERSIST = false), BOX = true;
                                  # + OWNER x the two-segment instantiation for images from 192 pixels on (SEG2, round 5)
    for name, d in box.items():
        assert d["vgpr_count"] <= 64, (name, d)
        assert d["sgpr_count"] <= 80, (name, d)
        assert d["vgpr_spill_count"] == 0 and d["sgpr_spill_count"] == 0, (name, d)


def test_two_fused_box_workgroups_fit_a_cu(descriptors):
    box = {n: d for n, d in descriptors.items() if "sphere_zbuf_mse_box_kernel" in n}
    assert len(box) == 2          # round 4's packing, and two-segment boxes for images from 192 pixels on (SEG2)
    for name, d in box.items():
        assert d["vgpr_count"] <= 64 and d["sgpr_count"] <= 80 and d["vgpr_spill_count"] == 0, (name, d)


def test_backward_leaves_its_load_registers_alone(descriptors):
    bwd = {n: d for n, d in descriptors.items() if "sphere_zbuf_bwd_kernel" in n}
    assert len(bwd) == 25         # VEC4 x POW2 x PERSIST x (8 waves, 16 waves, 16 waves whole crop) + the 8-wave SEG2 one
    for name, d in bwd.items():
        assert d["vgpr_count"] <= 114, (name, d)      # 96 for the compiler + v[96:113] named in the asm statements
        assert d["vgpr_spill_count"] == 0, (name, d)


def test_no_wave_waits_for_its_own_stores(descriptors):
    """vmcnt is ONE in-order queue for loads and stores, and the stream stores are asm statements the compiler's wait
    insertion does not count: a `s_waitcnt vmcnt` behind a store waits for that store (docs/EXPERIMENTS.md S5).
    Forward, one workgroup per crop: none behind the first store (the records are waited for by every wave in front of
    it).  Fused box kernel: in the convert pass -- between the second and the third barrier -- every wait sits in front of
    the pass's first store, except the one directly behind the load of a unit beyond the four requested at the entry."""
    fwd = {n: b for n, b in _bodies(_ASM, "sphere_zbuf_fwd_kernel").items()
           if re.findall(r"Lb([01])E", re.search(r"kernelI((?:L[bi]\d+E)+)", n).group(1))[3] == "0"}     # PERSIST = false
    assert len(fwd) >= 20
    for name, body in fwd.items():
        first = next(i for i, l in enumerate(body) if "global_store" in l)
        late = [l.strip() for l in body[first:] if "vmcnt" in l]
        assert not late, (name, late)
    box = _bodies(_ASM, "sphere_zbuf_mse_box_kernel")
    assert len(box) == 2
    for name, body in box.items():
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        conv = body[bars[1]:bars[2]]
        first = next(i for i, l in enumerate(conv) if "global_store" in l)
        for i in range(first, len(conv)):
            if "vmcnt" in conv[i]:
                assert any("global_load" in l for l in conv[max(0, i - 3):i]), (name, i, conv[i])
        # ... and the walk that follows starts without one (its first 300 instructions: the slice set-up and the first runs)
        assert not [l for l in body[bars[2]:bars[2] + 300] if "vmcnt" in l], name


@pytest.mark.parametrize("unit", ["tri_raster", "mesh_depth"])
def test_face_setup_stays_in_registers(unit, tmp_path):
    """The sort of a face's vertices by x is three selects per coordinate; the compiler once turned them into loads from a
    select of addresses, which pinned the nine coordinates to a 48-byte scratch array in every kernel that sets faces up
    (DESIGN.md section 4.4): no kernel of these units uses scratch memory."""
    text = _compile_asm(unit, tmp_path)
    sizes = [int(v) for v in re.findall(r"; ScratchSize: (\d+)", text)]
    assert len(sizes) >= 8 and max(sizes) == 0, sizes
