import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from spherehand_amd import _lib, ops, hand_model
from spherehand_amd.joint_angle import sample_poses
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.render import HandBallPrimitiveRender
dev = torch.device("cuda", 0); S, J = 128, 41
mesh = hand_model.load_mesh()
fk = HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
hbr = HandBallPrimitiveRender(mesh["bones"], S, S).to(dev)
lib = _lib.lib(); st = torch.cuda.Stream(device=dev)
with torch.cuda.stream(st):
    for n in (512, 1152, 9216):
        with torch.no_grad():
            sph = hbr.spheres(fk(sample_poses(n, seed=7).to(dev))).contiguous()
        depth = torch.empty(n, S, S, device=dev)
        f = lambda s: lib.shr_sphere_raster_fwd_ex(sph.data_ptr(), n, J, S, S, depth.data_ptr(), None, 0, s)
        for rnd in range(2):
            for tb in (-1, 1):
                ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, tb)
                f(st.cuda_stream)
                print("n %5d depth-only run_table %2d: %7.2f us" % (n, tb, bench.mean_launch_us(f, st, 40 if n < 5000 else 8, 3, 3, warm_ms=30.0)), flush=True)
        ops.set_tuning(ops.TUNE_FWD_RUN_TABLE, -1)
