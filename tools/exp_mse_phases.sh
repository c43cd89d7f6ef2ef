#!/bin/bash
# Round 6: what the fused render-and-compare kernel's time consists of, at config 5's size (1152 crops @256x256), by
# TIMING-ONLY ablations (wrong results) and perturbations of sphere_zbuf_mse_body, A/B against the product library in one
# process (tools/ab_variant.py):
#   noSCAN / noCONVERT / noWALK / noCW / noSCW   a phase compiled out (EXP_MSE_SKIP_*)
#   junk4 / junk8 / junk16                       that many independent VALU instructions more per chunk pair of the scan
#   stag2 / stag5                                the second workgroup of every CU starts 2 / 5 x 3.4 us late
#   notile                                       the rows a wide box leaves to the tile code are dropped
#   oldwaits / bgpro / nolfence / norot / nobgst   round 6's store / wait changes undone (all / background rows stored in the prologue /
#                                                no wait for the observed image in front of the convert pass / regions in image order);
#                                                nobgst: the background rows never stored (wrong output: what those stores cost);
#                                                box7: seven waves evaluate the touched box, as in the forward, instead of one
#   noSCWT / noSCWTt / empty1T / empty3          the skeleton without the observed-image loads / and without the tile code / only the
#                                                prologue up to the first barrier / only the launch
# Build here (no GPU needed):   bash tools/exp_mse_phases.sh build
# Run on the GPU box:           bash tools/exp_mse_phases.sh        -> gpurun_out/r06_mse_phases.txt
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = "build" ]; then
  rm -f tools/libspherehand_exp*.so
  for v in SCAN CONVERT WALK; do python tools/ab_variant.py build:no$v -DEXP_MSE_SKIP_$v | tail -1; done
  python tools/ab_variant.py build:noCW -DEXP_MSE_SKIP_CONVERT -DEXP_MSE_SKIP_WALK | tail -1
  python tools/ab_variant.py build:noSCW -DEXP_MSE_SKIP_SCAN -DEXP_MSE_SKIP_CONVERT -DEXP_MSE_SKIP_WALK | tail -1
  for v in 4 8 16; do python tools/ab_variant.py build:junk$v -DEXP_MSE_SCAN_JUNK=$v | tail -1; done
  for v in 2 5; do python tools/ab_variant.py build:stag$v -DEXP_MSE_STAGGER=$v | tail -1; done
  python tools/ab_variant.py build:notile -DEXP_MSE_SKIP_TILE | tail -1
  python tools/ab_variant.py build:noSCWT -DEXP_MSE_SKIP_SCAN -DEXP_MSE_SKIP_CONVERT -DEXP_MSE_SKIP_WALK -DEXP_MSE_SKIP_TARGET | tail -1
  python tools/ab_variant.py build:noSCWTt -DEXP_MSE_SKIP_SCAN -DEXP_MSE_SKIP_CONVERT -DEXP_MSE_SKIP_WALK -DEXP_MSE_SKIP_TARGET -DEXP_MSE_SKIP_TILE | tail -1
  python tools/ab_variant.py build:empty1T -DEXP_MSE_EMPTY=1 -DEXP_MSE_SKIP_TARGET | tail -1
  python tools/ab_variant.py build:empty3 -DEXP_MSE_EMPTY=3 | tail -1
  # the store / wait findings of round 6 (docs/EXPERIMENTS.md S5), each undone on its own and all together
  python tools/ab_variant.py build:oldwaits -DEXP_MSE_NO_LOAD_FENCE -DEXP_MSE_BG_PROLOGUE -DEXP_NO_RECORD_FENCE -DEXP_MSE_NO_ROTATE | tail -1
  python tools/ab_variant.py build:bgpro -DEXP_MSE_BG_PROLOGUE | tail -1
  python tools/ab_variant.py build:nolfence -DEXP_MSE_NO_LOAD_FENCE | tail -1
  python tools/ab_variant.py build:norot -DEXP_MSE_NO_ROTATE | tail -1
  python tools/ab_variant.py build:nobgst -DEXP_MSE_NO_BG_STORE | tail -1
  python tools/ab_variant.py build:box7 -DEXP_MSE_SEVEN_BOXES | tail -1
  exit 0
fi
mkdir -p gpurun_out
{
  echo "# (config 5's own projections and observed images: C5=1 -- wide boxes; centred hands read ~8 us less)"
  echo "# fused render-and-compare, 1152 crops @256x256, us per launch (mean of 3 rounds x 3 batches of 40 launches), tools/exp_mse_phases.sh"
  for nd in 0 1; do
    if [ $nd = 1 ]; then export NODEPTH=1; echo "## without the depth output (return_projections = False)"; else unset NODEPTH; echo "## with the depth output"; fi
    C5=1 S=256 NS=1152 MSE=1 python tools/ab_variant.py 2>&1 | grep render | awk '{a[$3]+=$5; c[$3]++} END {for (k in a) printf "%-10s %.1f\n", k, a[k]/c[k]}' | sort
  done
} > gpurun_out/r06_mse_phases.txt
cat gpurun_out/r06_mse_phases.txt
