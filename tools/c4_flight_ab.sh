#!/bin/bash
# same-box A/B of config 4's call patterns (clips per call x calls in flight); stderr of every leg is kept in gpurun_out/c4_flight_ab.err
set -uo pipefail
mkdir -p gpurun_out
for v in "--batch-clips 4 --in-flight 1" "--batch-clips 2 --in-flight 2" "--batch-clips 1 --in-flight 4" "--batch-clips 1 --in-flight 2" "--clips-per-gpu 8 --batch-clips 4 --in-flight 2"; do
  for rep in 1 2; do
  python bench.py --config 4 $v --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-profile 2>>gpurun_out/c4_flight_ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
