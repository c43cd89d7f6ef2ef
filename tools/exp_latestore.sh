#!/bin/bash
# Forward: cache policy of the touched-row stores (after the scan conversion) against the background's sc1.
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['roofline']['launch_us'])"; }
for mode in ${MODES:-44 00 11 40 04 41 14}; do
  SHR_HIPCC_EXTRA="-DSHR_LATE_STORE_MODE=$mode" python -m spherehand_amd.build --force > /dev/null || exit 1
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "late mode $mode"
  timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | line "late mode $mode"
done
python -m spherehand_amd.build --force > /dev/null
