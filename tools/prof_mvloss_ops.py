import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["x"]
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/prof_mvloss.py")).read().split("for _ in range(5): step()")[0])
for _ in range(3): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
