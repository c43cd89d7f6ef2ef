#!/bin/bash
# A/B of two builds of the library on ONE box, alternating, each once under rocprofv3 --kernel-trace --stats and once
# unprofiled (-> profiles/r03_ab_rocprof.txt).  tools/libshr_round_start.so = the library of the commit to compare with:
#   git worktree add /tmp/old <commit> && (cd /tmp/old && python -m spherehand_amd.build) &&
#   cp /tmp/old/spherehand_amd/libspherehand_hip.so tools/libshr_round_start.so && git worktree remove --force /tmp/old
# (tools/*.so is git-ignored and travels with gpurun).  Run on the GPU box from the repo root: bash tools/ab_rocprof.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
cp spherehand_amd/libspherehand_hip.so /tmp/new.so
for v in new old new old; do
  if [ $v = old ]; then cp tools/libshr_round_start.so spherehand_amd/libspherehand_hip.so; else cp /tmp/new.so spherehand_amd/libspherehand_hip.so; fi
  n=$(ls -d $O/st_${v}* 2>/dev/null | wc -l)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_${v}_$n -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $O/st_${v}_$n.log 2>&1
  timeout 200 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > $O/plain_${v}_$n.log 2>&1
done
cp /tmp/new.so spherehand_amd/libspherehand_hip.so
for d in $O/st_*/; do echo "== $d"; f=$(find $d -name "*kernel_stats.csv" | head -1); grep -E "sphere_zbuf_(fwd|bwd)" $f | cut -d, -f1-5 | sed 's/(.*)//' | cut -c1-150; done > $O/summary.txt
for f in $O/plain_*.log $O/st_*.log; do echo "== $f"; grep -o '"value": [0-9.]*\|"launch_us": {[^}]*}' $f | tr '\n' ' '; echo; done >> $O/summary.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
cat $O/summary.txt
