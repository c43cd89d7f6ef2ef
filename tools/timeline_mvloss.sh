cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for MV in 1 0; do
ISMV=$MV rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl_mv$MV -o tl -- python - <<'PY' > gpurun_out/tl_mv$MV.log 2>&1
import os, sys, torch, gc
sys.path.insert(0, os.getcwd())
from spherehand_amd import hand_model
from spherehand_amd.datasets import SyntheticMultiviewDataset
from spherehand_amd.multiview_utility import MutualProjectionLoss
mesh = hand_model.load_mesh()
ds = SyntheticMultiviewDataset(mesh, 128, 256, seed=0, device="cuda")
crit = MutualProjectionLoss(256, mesh).cuda(); crit.cache_points = False
real, cam, inv = ds.dms.cuda(), ds.cam.cuda(), ds.inv_cam.cuda()
joints = (ds.joints.cuda() + torch.randn_like(ds.joints.cuda())).requires_grad_(True)
is_mv = os.environ["ISMV"] == "1"
gc.collect(); gc.freeze()
for _ in range(12):
    joints.grad = None
    loss, _ = crit(cam, inv, joints, real, is_mv)
    loss.backward()
torch.cuda.synchronize()
PY
python tools/timeline_mvloss.py gpurun_out/tl_mv$MV/tl_kernel_trace.csv
done
