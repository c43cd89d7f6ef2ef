#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repo root).
# usage: tools/collect_profiles.sh rNN
set -u
R=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
rm -rf "$O"; mkdir -p "$O"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline > "$O/stats.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$O/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > "$O/pmc_write.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/secondary" -o losses -- python tools/bench_losses.py > "$O/secondary.log" 2>&1
timeout 300 python bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"
find "$O" -name "*kernel_stats.csv" | head
tail -1 "$O/bench_line.json" | cut -c1-300
