"""A/B of the sphere rasterizer forward / backward between the product library and a variant built with extra -D flags:
    python tools/ab_variant.py build -DEXP_FLAG ...     (anywhere: tools/libspherehand_exp.so)
    python tools/ab_variant.py build:NAME -DEXP_FLAG ...   (several variants side by side: tools/libspherehand_exp_NAME.so)
    python tools/ab_variant.py                          (GPU box: alternating timings at 256 / 1152 / 9216 crops, same-bits check,
                                                         the product library against every variant library found)"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
EXP = os.path.join(ROOT, "tools", "libspherehand_exp.so")


def build(flags, name=""):
    """(SRC=data_to_model in the environment: the variant flags go to that translation unit instead of sphere_raster)"""
    global EXP
    if name:
        EXP = os.path.join(ROOT, "tools", "libspherehand_exp_%s.so" % name)
    from spherehand_amd import build as b
    b.build()
    unit = os.environ.get("SRC", "sphere_raster")
    obj = "/tmp/%s_exp%s.o" % (unit, name)
    subprocess.check_call([b.HIPCC] + [f for f in b.FLAGS if f != "-shared"] + flags +
                          ["-c", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(b.PKG, "csrc"), "-o", obj,
                           os.path.join(b.PKG, "csrc", unit + ".hip")])
    objs = [o for o in glob.glob(os.path.join(b.OBJ_DIR, "*.o")) if not o.endswith(unit + ".o")]
    subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", EXP, obj] + objs)
    print(EXP, flags)


def main():
    import torch
    import bench
    from spherehand_amd import _lib, hand_model
    from spherehand_amd.joint_angle import sample_poses
    from spherehand_amd.kinematicsTransformation import HandTransformationMat
    from spherehand_amd.render import HandBallPrimitiveRender
    vp, i = ctypes.c_void_p, ctypes.c_int
    libs = {"product": _lib.lib()}
    for f in sorted(glob.glob(os.path.join(ROOT, "tools", "libspherehand_exp*.so"))):
        libs[os.path.basename(f)[len("libspherehand_exp"):-3].lstrip("_") or "variant"] = ctypes.CDLL(f)
    for l in libs.values():
        l.shr_sphere_raster_fwd_ex.argtypes = [vp, i, i, i, i, vp, vp, i, vp]
        l.shr_sphere_raster_bwd.argtypes = [vp, vp, vp, i, i, i, i, vp, vp]
        l.shr_sphere_raster_mse.argtypes = [vp, i, i, i, i, vp, vp, vp, vp, vp, vp]
        l.shr_sphere_raster_mse_regions.argtypes = [i, i]
    dev = torch.device("cuda", 0)
    S, J = int(os.environ.get("S", 128)), 41
    mesh = hand_model.load_mesh()
    fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
    hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        for n in [int(v) for v in os.environ.get("NS", "256,1152,9216").split(",")]:
            with torch.no_grad():
                sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
            obs = index = None
            if os.environ.get("C5"):    # config 5's own projections (WIDE boxes) and observed images (three crops share one): n = 9 B
                from spherehand_amd.datasets import SyntheticMultiviewDataset
                from spherehand_amd.multiview_utility import MutualProjectionLoss
                ds = SyntheticMultiviewDataset(mesh, n // 9, S, seed=0, device=dev)
                crit = MutualProjectionLoss(S, mesh).to(dev)
                with torch.no_grad():
                    _, pts = crit.mutual_projection(ds.cam.to(dev), ds.inv_cam.to(dev), ds.joints.to(dev) + torch.randn(ds.joints.shape, device=dev))
                rad = crit.data_to_model_criterion.radiuses.view(-1)
                sph = torch.cat([pts.squeeze(-1).reshape(n, J, 3), rad.view(1, J, 1).expand(n, J, 1)], -1).contiguous()
                obs = ds.dms.to(dev).view(n // 3, S, S).contiguous()
                index = crit._indices(n // 9, 3, dev)[0]
            depth = torch.empty(n, S, S, device=dev); owner = torch.empty(n, S, S, device=dev, dtype=torch.uint8)
            grad = torch.randn(n, S, S, device=dev); gs = torch.empty(n, J, 4, device=dev)
            p = [t.data_ptr() for t in (sph, depth, owner, grad, gs)]
            reps = 200 if n == 256 else (40 if n == 1152 else 8)
            ref = {}
            for rnd in range(3):
                for name, l in libs.items():
                    f = lambda s: l.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], p[2], 1, s)
                    f0 = lambda s: l.shr_sphere_raster_fwd_ex(p[0], n, J, S, S, p[1], None, 0, s)
                    b = lambda s: l.shr_sphere_raster_bwd(p[0], p[3], p[2], n, J, S, S, p[4], s)
                    assert f(stream.cuda_stream) == 0
                    tf = bench.mean_launch_us(f, stream, reps, 3, 3, warm_ms=30.0)
                    tb = bench.mean_launch_us(b, stream, reps, 3, 3, warm_ms=30.0)
                    t0 = bench.mean_launch_us(f0, stream, reps, 3, 3, warm_ms=30.0)
                    f(stream.cuda_stream); b(stream.cuda_stream); stream.synchronize()
                    key = (depth.clone(), gs.clone())
                    same = "" if name == "product" else "  same bits: %s" % (torch.equal(key[0], ref["d"]) and torch.equal(key[1], ref["g"]))
                    if name == "product": ref = {"d": key[0], "g": key[1]}
                    if os.environ.get("MSE"):   # the fused render-and-compare kernel on the same crops (target: the depth map + noise)
                        R = l.shr_sphere_raster_mse_regions(S, S)
                        tgt = obs if obs is not None else (ref.get("d", depth) + 3.0 * torch.randn_like(depth)).contiguous()
                        sse = torch.empty(n * R, device=dev); gsp = torch.empty(n * R * J * 4, device=dev)
                        m = lambda s: l.shr_sphere_raster_mse(p[0], n, J, S, S, tgt.data_ptr(), None if index is None else index.data_ptr(),
                                                              None if os.environ.get("NODEPTH") else p[1], sse.data_ptr(), gsp.data_ptr(), s)
                        assert m(stream.cuda_stream) == 0
                        print("n %5d %-10s render-and-compare %7.2f us" % (n, name, bench.mean_launch_us(m, stream, reps, 3, 3, warm_ms=30.0)), flush=True)
                    print("n %5d %-10s fwd+owner %7.2f  bwd %7.2f  fwd depth-only %7.2f us%s" % (n, name, tf, tb, t0, same), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1].startswith("build"):
        build(sys.argv[2:], sys.argv[1][6:])
    else:
        main()
