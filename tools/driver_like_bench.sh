#!/bin/bash
# Runs bench.py with no flags, the way the round-end driver does, and prints
# the fields the judge reads plus the wall clock around the whole command.
set -u
mkdir -p gpurun_out
t0=$(date +%s.%N)
python bench.py > gpurun_out/driver_like_bench.json 2> gpurun_out/driver_like_bench.err
rc=$?
t1=$(date +%s.%N)
echo "bench.py rc=$rc wall=$(python -c "print(round($t1-$t0,1))") s"
python - <<'PY'
import json
lines = open("gpurun_out/driver_like_bench.json").read().strip().splitlines()
d = json.loads(lines[-1])
print({k: d[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup",
                         "ms_per_step", "dtype", "scaling", "vs_baseline")})
print(d["config"])
r = d["roofline"]
print("roofline", {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac")}, "traffic", r["traffic"])
c = d["cpu_baseline"]
print("cpu_baseline", c["value"], c["unit"], "cores", c["cores"], c["kind"])
for k in ("parity", "dcn_fused"):
    if k in d:
        print(k, d[k] if not isinstance(d[k], dict) else {a: b for a, b in d[k].items() if not isinstance(b, (dict, list))})
PY
