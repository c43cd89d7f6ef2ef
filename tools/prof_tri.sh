cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tri -o tri -- python - > gpurun_out/prof_tri.log 2>&1 <<'PY'
import os,sys,torch
sys.path.insert(0,os.getcwd())
import depth_rasterization
from spherehand_amd import hand_model
from spherehand_amd.render import DepthRender
from spherehand_amd.kinematicsTransformation import HandTransformationMat
from spherehand_amd.joint_angle import sample_poses
mesh=hand_model.load_mesh(); dev=torch.device("cuda",0)
fk=HandTransformationMat([b["offset_matrix"].astype("float32") for b in mesh["bones"]]).to(dev)
dr=DepthRender(mesh,128).to(dev)
with torch.no_grad():
    verts=dr.lbs(fk(sample_poses(256,seed=1).to(dev)),dr.camera,None)
    fv=verts[:,dr.rasterizer.faces,0:3].reshape(256,-1,3,3).contiguous()
for _ in range(60): depth_rasterization.forward(640,640,fv)
torch.cuda.synchronize()
for _ in range(20): torch.full((256,640,640),1000.0,device=dev)
torch.cuda.synchronize()
PY
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/prof_tri/tri_kernel_stats.csv')):
    n=r['Name'].split('(')[0][-60:]
    if int(r['Calls'])>=20: print("%-62s %6s avg %9.0f min %8s"%(n,r['Calls'],float(r['AverageNs']),r['MinNs']))
PY
