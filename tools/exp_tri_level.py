"""Triangle rasterizer: raster_batch's level walk against its pixel-order walk in BOTH kernels (tri_raster.hip built alone
with -DSHR_TRI_ATOMIC_LEVELS / -DSHR_TRI_BAND_LEVELS; the defaults are levels in the band kernel only).  EXPERIMENTS
R3c's first table also swept a threshold (columns a level must hold to be walked as a level) that is no longer in the code.
    python tools/exp_tri_level.py build     (anywhere: tools/libtri_lv{0,1}.so -- 0: pixel order everywhere, 1: levels everywhere)
    python tools/exp_tri_level.py           (GPU box: shr_tri_raster_fwd at B = 256 / 48 / 1 with the launcher's own choice of
                                             kernel, same-bits check)"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LEVELS = (0, 1)


def so(n):
    return os.path.join(ROOT, "tools", "libtri_lv%d.so" % n)


def build():
    from spherehand_amd import build as b
    for n in LEVELS:
        subprocess.check_call([b.HIPCC] + list(b.FLAGS) + ["-DSHR_TRI_ATOMIC_LEVELS=%d" % n, "-DSHR_TRI_BAND_LEVELS=%d" % n, "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(b.PKG, "csrc"), "-o", so(n), os.path.join(b.PKG, "csrc", "tri_raster.hip")])
        print(so(n))


def main():
    import torch
    import bench
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRender
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.joint_angle import sample_poses
    mesh = hand_model.load_mesh()
    dev = torch.device("cuda", 0)
    fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
    dr = DepthRender(mesh, 128).to(dev)
    vp, i = ctypes.c_void_p, ctypes.c_int
    libs = {n: ctypes.CDLL(so(n)) for n in LEVELS}
    for l in libs.values():
        l.shr_tri_raster_fwd.argtypes = [vp, i, i, i, i, vp, vp]
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for B in (256, 48, 1):
            with torch.no_grad():
                verts = dr.lbs(fk(sample_poses(B, seed=1).to(dev)), dr.camera, None)
                fv = verts[:, dr.rasterizer.faces, 0:3].reshape(B, -1, 3, 3).contiguous()
            F = fv.shape[1]
            out = torch.empty(B, 640, 640, device=dev)
            ref = None
            for rnd in range(2):
                for n in LEVELS:
                    fn = lambda s, l=libs[n]: l.shr_tri_raster_fwd(fv.data_ptr(), B, F, 640, 640, out.data_ptr(), s)
                    assert fn(stream.cuda_stream) == 0
                    stream.synchronize()
                    if ref is None:
                        ref = out.clone()
                    same = torch.equal(out, ref)
                    t = bench.mean_launch_us(fn, stream, 10 if B > 1 else 50, 3, 3, warm_ms=20.0)
                    print("B=%3d %s: %8.1f us  same bits: %s" % (B, "levels everywhere" if n else "pixel order everywhere", t, same), flush=True)


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
