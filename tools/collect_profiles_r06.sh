#!/bin/bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root):  bash tools/collect_profiles_r06.sh
# PMC passes carry --kernel-trace only (no --stats, no other trace domain), one counter set per pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; rm -rf "$O"; mkdir -p "$O"
# (a) the headline kernels over >= 2000 steps: stats summary + per-launch trace (medians)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_headline" -o bench -- python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-secondary > "$O/stats_headline.log" 2>&1
# (b) the same tracing over the full bench (secondary set)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o bench -- python bench.py --steps 300 --warmup 30 --no-cpu-baseline > "$O/stats.log" 2>&1
BS="python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-secondary"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o bench -- $BS > "$O/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o bench -- $BS > "$O/pmc_write.log" 2>&1
SQA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU"
SQB="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_WR"
timeout 300 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d "$O/sq_a_raster" -o bench -- $BS > "$O/sq_a_raster.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d "$O/sq_b_raster" -o bench -- $BS > "$O/sq_b_raster.log" 2>&1
export SHR_BENCH_SKIP_TRAIN=1
timeout 400 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d "$O/sq_a_secondary" -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/sq_a_secondary.log" 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d "$O/sq_b_secondary" -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/sq_b_secondary.log" 2>&1
unset SHR_BENCH_SKIP_TRAIN
for S in 128 256; do
  S=$S REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d "$O/sq_a_d2m$S" -o d2m -- python tools/prof_d2m.py > "$O/sq_a_d2m$S.log" 2>&1
  S=$S REPS=10 timeout 300 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d "$O/sq_b_d2m$S" -o d2m -- python tools/prof_d2m.py > "$O/sq_b_d2m$S.log" 2>&1
done
# (b2) config 5's loss (tools/prof_mvloss.py: 1152 crops @256x256, fresh observations every call): kernel stats, HBM
# traffic and SQ counters of its kernels (render-and-compare, compaction, point search, assembly)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_mvloss" -o mv -- python tools/prof_mvloss.py > "$O/stats_mvloss.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_mvloss" -o mv -- python tools/prof_mvloss.py > "$O/pmc_fetch_mvloss.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_mvloss" -o mv -- python tools/prof_mvloss.py > "$O/pmc_write_mvloss.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQA --output-format csv -d "$O/sq_a_mvloss" -o mv -- python tools/prof_mvloss.py > "$O/sq_a_mvloss.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $SQB --output-format csv -d "$O/sq_b_mvloss" -o mv -- python tools/prof_mvloss.py > "$O/sq_b_mvloss.log" 2>&1
# (b2') the same step untraced: wall per step with the two terms on one stream / side by side (the module's default) / on
# unchanged observations; and the A/B of the two orders in one process
{ timeout 120 python tools/prof_mvloss.py; OVERLAP=1 timeout 120 python tools/prof_mvloss.py; OVERLAP=1 CACHE=1 timeout 120 python tools/prof_mvloss.py;
  timeout 200 python tools/ab_mvloss_overlap.py; } > "$O/mvloss_wall.log" 2>&1
# (b2c) the same-view mode (is_mv = False) of the same loss: per-kernel timeline of one step in both modes
bash tools/timeline_mvloss.sh > "$O/mvloss_timeline.log" 2>&1
# (b2d) the rasterizer kernels at config 5's own size, 1000 launches each under the tracer
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_c5size" -o c5 -- python tools/prof_config5_size.py > "$O/stats_c5size.log" 2>&1
# (b2e) pose <-> sphere records kernels and the pose -> depth -> pose chain; the triangle path; the in-kernel timeline
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_fk" -o fk -- python tools/bench_fk.py 256 > "$O/stats_fk.log" 2>&1
timeout 200 python tools/bench_tri.py > "$O/tri.log" 2>&1
timeout 200 python tools/bench_mesh.py >> "$O/tri.log" 2>&1
timeout 200 python tools/headline_timeline.py > "$O/headline_timeline.log" 2>&1
# (b2f) round 6: HandSynthesizer (one launch / three launches / module chain) untraced and under the tracer; DepthRender at
# S = 256 (band kernel with the resize epilogue against the tile kernel); the pure-mover floor at 1152 / 9216 crops
{ timeout 200 python tools/bench_synth.py 256 128 16; timeout 200 python tools/bench_synth.py 256 128 32 | head -3; timeout 200 python tools/bench_synth.py 48 64 16 | head -3; } > "$O/synth.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_synth" -o synth -- python tools/bench_synth.py 256 128 16 > "$O/stats_synth.log" 2>&1
{ for b in 16 64 256 1152; do timeout 200 python tools/bench_mesh256.py $b 256; done; } > "$O/mesh256.log" 2>&1
timeout 200 python tools/exp_floor_large.py > "$O/floor_large.log" 2>&1
# (b2g) the fused render-and-compare kernel: in-kernel timeline (instrumented build) and the phase ablations (variant libraries
# built beforehand: `python tools/headline_timeline.py build`, `bash tools/exp_mse_phases.sh build`)
timeout 200 python tools/mse_timeline.py > "$O/mse_timeline.log" 2>&1
ls tools/libspherehand_exp_*.so > /dev/null 2>&1 && timeout 600 bash tools/exp_mse_phases.sh > "$O/mse_phases.log" 2>&1
timeout 200 python tools/bench_mvloss_graph.py > "$O/mvloss_graph.log" 2>&1
# (b3) the headline launches from a C loop (tools/cloop.c), unprofiled and under the tracer: the durations `frac_rocprof` uses
timeout 200 python tools/prof_cloop.py > "$O/cloop_plain.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/cloop" -o cloop -- python tools/prof_cloop.py > "$O/cloop_traced.log" 2>&1
# (c) the unprofiled lines: the default run, the driver's short run, the graph variant
timeout 500 python bench.py > "$O/bench_line.json" 2> "$O/bench_line.err"
timeout 300 python bench.py --steps 20 --warmup 5 > "$O/bench_line_steps20.json" 2>> "$O/bench_line.err"
timeout 300 python bench.py --launch graph --no-secondary --no-cpu-baseline > "$O/bench_line_graph.json" 2>> "$O/bench_line.err"
# (d) the fuzzer on these kernels: FUZZ_SECONDS per family (default 45)
timeout 1200 python tools/fuzz.py ${FUZZ_SECONDS:-45} 3 > "$O/fuzz.log" 2>&1
python tools/summarize_r06.py gpurun_out/r06_profiles > "$O/summarize.log" 2>&1
find "$O" -name "*counter_collection.csv" -delete; find "$O" -name "*kernel_trace.csv" -delete
tail -3 "$O/summarize.log"; tail -2 "$O/fuzz.log"; tail -1 "$O/bench_line_steps20.json" | cut -c1-160
du -sh "$O"
