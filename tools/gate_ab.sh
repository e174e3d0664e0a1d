#!/bin/bash
set -uo pipefail
mkdir -p gpurun_out
# same-box A/B of the mask-gated launches: CRFP_MASK_GATE=0 (dense) vs default, BASELINE configs 2-5, two runs each
for c in 2 3 4 5; do for g in 0 1 0 1; do
  CRFP_MASK_GATE=$g python bench.py --config $c --no-extras --no-cpu-baseline --no-kernel-profile --no-other-configs 2>>gpurun_out/gate_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c gate $g', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms per step')"
done; done
