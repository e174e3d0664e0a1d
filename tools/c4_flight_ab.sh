for v in "--batch-clips 4 --in-flight 1" "--batch-clips 2 --in-flight 2" "--batch-clips 1 --in-flight 4" "--batch-clips 1 --in-flight 2" "--clips-per-gpu 8 --batch-clips 4 --in-flight 2"; do
  for rep in 1 2; do
  python bench.py --config 4 $v --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
