"""Where `mesh_depth_kernel` / `mesh_lattice_kernel` (SHR_MESH_LATTICE=0 for the former) spend their time: s_memtime stamps of every wave at the phase boundaries (mesh_depth.hip
built alone with -DMESH_TL into tools/libmesh_tl.so; the product build carries no stamps).
    python tools/exp_mesh_phases.py build     (anywhere)
    python tools/exp_mesh_phases.py           (GPU box: prints medians over the 256 workgroups of one launch, in us)
Stamps: 0 entry | 1 culls + counts done | 2 scans through | 3 face rows stand | 4 queue written | 5 phase B starts |
6 queue empty | 7 epilogue stored."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "tools", "libmesh_tl%s.so" % os.environ.get("VARIANT", ""))


def build():
    from spherehand_amd import build as b
    subprocess.check_call([b.HIPCC] + list(b.FLAGS) + ["-DMESH_TL"] + os.environ.get("XFLAGS", "").split() + ["-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(b.PKG, "csrc"), "-o", SO, os.path.join(b.PKG, "csrc", "mesh_depth.hip"),
                           os.path.join(b.PKG, "csrc", "tri_raster.hip")])      # (tri_raster.hip: shr_lbs_project, the fused entry's fallback)
    print(SO)


def main():
    import numpy as np
    import torch
    import bench
    from spherehand_amd import hand_model
    from spherehand_amd.render import DepthRender
    from spherehand_amd.util_modules import HandSynthesizer
    from spherehand_amd.joint_angle import sample_poses
    mesh = hand_model.load_mesh()
    vp, i, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    lib = ctypes.CDLL(SO)
    lib.shr_mesh_depth_fwd.argtypes = [vp, vp, i, i, i, i, i, f, vp, vp]
    lib.shr_mesh_debug_timeline.argtypes = [vp]
    ghz = float(os.environ.get("SHADER_GHZ", "2.4"))
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for B, S in ((256, 128), (256, 64)):
            syn = HandSynthesizer(mesh, S, 16, 1.0, 0.01).cuda()
            T = syn.hand_skeleton_transform(sample_poses(B, seed=1).cuda())
            dr = DepthRender(mesh, S).cuda()
            with torch.no_grad():
                verts = dr.lbs(T, dr.camera, None).contiguous()
            faces = dr.rasterizer.faces_i32
            NV, F = verts.shape[1], faces.shape[0]
            out = torch.empty(B, S, S, device="cuda")
            fn = lambda s: lib.shr_mesh_depth_fwd(verts.data_ptr(), faces.data_ptr(), B, NV, F, 640, S, 100.0, out.data_ptr(), s)
            if os.environ.get("FUSED"):      # shr_mesh_render_fwd: skinning inside the lattice kernel
                l = dr.lbs
                cx, cy, fx, fy = dr.camera
                Tc = T.contiguous()
                lib.shr_mesh_render_fwd.argtypes = [vp, i, i, i, vp, vp, vp, i, f, f, f, f, vp, vp, i, i, i, f, vp, vp, vp]
                fn = lambda s: lib.shr_mesh_render_fwd(Tc.data_ptr(), B, 17, NV, l.skin_vertex_start.data_ptr(), l.skin_bone.data_ptr(),
                                                       l.skin_wv.data_ptr(), 1, cx, cy, fx, fy, None, faces.data_ptr(), F, 640, S, 100.0,
                                                       verts.data_ptr(), out.data_ptr(), s)
            assert fn(stream.cuda_stream) == 0
            us = bench.mean_launch_us(fn, stream, 100, 3, 5, warm_ms=30.0)
            stream.synchronize()
            fn(stream.cuda_stream); stream.synchronize()
            tl = np.zeros(256 * 16 * 16, dtype=np.uint64)
            assert lib.shr_mesh_debug_timeline(tl.ctypes.data) == 0
            tl = tl.reshape(256, 16, 16)[:min(B, 256)].astype(np.int64)
            items, rows = tl[:, 0, 8], tl[:, 0, 9]
            tl = tl[:, :, :8]
            t0 = tl[:, :, 0].min(axis=1, keepdims=True)              # the workgroup's first wave to start
            rel = (tl - t0[:, :, None]) / (ghz * 1e3)                 # us since then
            med = lambda a: float(np.median(a))
            names = ["entry", "culls done", "scans through", "rows stand", "queue written (round 1)", "B starts (round 1)", "queue empty (last round)", "stored"]
            lattice = int(items.max()) == 0        # (the lattice kernel leaves the item count alone)
            if lattice:
                names = ["entry", "culls done", "survivors listed", "first batch set up", "lattice initialised (fused: vertices skinned)", "second batch starts", "last batch done", "stored"]
            print("B=%d S=%d: launch %.1f us (stamped build); per workgroup, us since its first wave's entry "
                  "(median over workgroups of the FIRST / LAST wave to get there):" % (B, S, us))
            if lattice:
                print("   mesh_lattice_kernel; surviving faces per crop: median %d, max %d" % (np.median(rows), rows.max()))
            else:
                print("   work items per crop: median %d, max %d (a round of the queue holds 3584); surviving faces: median %d, max %d (768 rows)"
                      % (np.median(items), items.max(), np.median(rows), rows.max()))
            for k in range(8):
                if names[k] is None:
                    continue
                print("   %d %-24s first %6.2f  last %6.2f" % (k, names[k], med(rel[:, :, k].min(axis=1)), med(rel[:, :, k].max(axis=1))))
            sys.stdout.flush()


if __name__ == "__main__":
    build() if len(sys.argv) > 1 and sys.argv[1] == "build" else main()
