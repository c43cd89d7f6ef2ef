#!/bin/bash
# rocprofv3 --kernel-trace --stats of the headline command only (no secondary set); prints the two kernels' rows
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02h; rm -rf "$O"; mkdir -p "$O"
for i in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/s$i" -o bench -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary > "$O/s$i.log" 2>&1
grep -E "sphere_zbuf" "$O/s$i/bench_kernel_stats.csv" | cut -d, -f1-8 | sed 's/(HIP_vector.*)"/"/' | cut -c1-160
grep -o '"launch_us": {[^}]*}' "$O/s$i.log"
rm -f "$O/s$i/bench_kernel_trace.csv"
done
python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-secondary | grep -o '"launch_us": {[^}]*}'
