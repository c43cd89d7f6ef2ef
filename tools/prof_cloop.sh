cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/cloop; rm -rf $O; mkdir -p $O
timeout 200 python tools/prof_cloop.py > $O/plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o cloop -- python tools/prof_cloop.py > $O/traced.log 2>&1
find $O -name "*kernel_trace.csv" -delete
cat $O/plain.log | tail -3; cat $O/traced.log | tail -3
