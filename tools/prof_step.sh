cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_step; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o step -- python tools/prof_engine.py > $O/log.txt 2>&1
find $O -name "*kernel_stats.csv" | head -2
