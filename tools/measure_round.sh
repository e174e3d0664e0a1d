#!/bin/bash
# One measurement set on the GPU box (through gpurun): bench lines of BASELINE configs 2-5, rocprofv3 kernel stats + PMC traffic
# and MFMA / LDS utilisation for both storage modes, per-site tables.  Usage: bash tools/measure_round.sh <tag>
TAG=${1:-r02_vX}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out
# the default run (what the driver runs: config 2 + short legs of configs 3 / 4 / 5), then the full lines of the bf16 configs
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err
for c in 3 4 5; do python bench.py --config $c > gpurun_out/${TAG}_bench_c$c.json 2> gpurun_out/${TAG}_bench_c$c.err; done
python tools/site_table.py > gpurun_out/${TAG}_site_f32.txt 2>&1
python tools/site_table.py --storage bf16 > gpurun_out/${TAG}_site_bf16.txt 2>&1
bash tools/collect_profiles.sh ${TAG}_f32 > gpurun_out/${TAG}_collect_f32.log 2>&1
BENCH_ARGS="--storage bf16" bash tools/collect_profiles.sh ${TAG}_bf16 > gpurun_out/${TAG}_collect_bf16.log 2>&1
# BASELINE config 4 as round 4 runs it: 4 clips per rank in ONE crfp_dsv_forward_batch call (lock-step launches over all clips)
BENCH_ARGS="--config 4" bash tools/collect_profiles.sh ${TAG}_c4 > gpurun_out/${TAG}_collect_c4.log 2>&1
python tools/prof_batch.py f32 4 > gpurun_out/${TAG}_lockstep_vs_loop_f32.txt 2>&1
python tools/prof_batch.py bf16 4 > gpurun_out/${TAG}_lockstep_vs_loop_bf16.txt 2>&1
python tools/bench_batch.py both > gpurun_out/${TAG}_batch_scaling.txt 2>&1
bash tools/collect_util.sh ${TAG}_f32_util > gpurun_out/${TAG}_util_f32.log 2>&1
BENCH_ARGS="--storage bf16" bash tools/collect_util.sh ${TAG}_bf16_util > gpurun_out/${TAG}_util_bf16.log 2>&1
for c in 2 3 4 5; do python - <<P
import json
d=json.loads(open("gpurun_out/${TAG}_bench_c$c.json").read().strip().splitlines()[-1])
print($c, round(d["value"],1), round(d["ms_per_step"],2), d["roofline"]["kernel"], round(d["roofline"]["frac"],3), round(d.get("warp_dcn",{}).get("frac",0),3), d.get("cpu_baseline",{}).get("value"), d.get("parity"))
P
done
