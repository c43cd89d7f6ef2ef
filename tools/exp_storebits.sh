#!/bin/bash
# Forward kernel with explicit cache-policy bits on its streaming stores (run through gpurun from the repo root).
# mode = 10 * depth bits + owner bits; bits: 0 "", 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt, 4 sc1, 5 sc0, 6 sc1 nt
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['launch_us'])"; }
for mode in ${MODES:-11 44 41 45 14}; do for bm in 0 4 1; do
  SHR_HIPCC_EXTRA="-DSHR_STORE_MODE=$mode -DSHR_BWD_STORE_MODE=$bm" python -m spherehand_amd.build --force > /dev/null || exit 1
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "mode $mode bwd $bm"
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "mode $mode bwd $bm"
done; done
python -m spherehand_amd.build --force > /dev/null
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "default (44, 4)"
