#!/bin/bash
# Copy one measurement set (tools/measure_round.sh <tag>, merged back into gpurun_out/) into profiles/ under its tag and make its PMC
# summaries the ones bench.py reads (profiles/pmc_summary_latest*).  Usage: bash tools/install_profiles.sh <tag>
# The set is staged in a temporary directory and moved into profiles/ only when every file of it was found: a missing file aborts the
# install and leaves profiles/ (and the pmc_summary_latest* triple bench.py reads) untouched.
set -euo pipefail
if [ $# -ne 1 ] || [ -z "$1" ]; then echo "usage: bash tools/install_profiles.sh <tag>" >&2; exit 2; fi
T=$1
cd "$(dirname "$0")/.."
STAGE=$(mktemp -d profiles/.install_XXXXXX)
trap 'rm -rf "$STAGE"' EXIT
for c in 2 3 4 5; do cp gpurun_out/${T}_bench_c$c.json $STAGE/${T}_bench_config$c.json; done
for s in f32 bf16; do
  cp gpurun_out/${T}_site_$s.txt $STAGE/${T}_${s}_per_site_table.txt
  cp gpurun_out/${T}_lockstep_vs_loop_$s.txt $STAGE/
  cp gpurun_out/${T}_${s}_util/util_summary.txt $STAGE/${T}_${s}_mfma_lds_util.txt
done
for s in f32 bf16 c4; do
  cp gpurun_out/${T}_$s/summary.txt $STAGE/${T}_${s}_pmc_summary.txt
  cp gpurun_out/${T}_$s/pmc_summary.json $STAGE/${T}_${s}_pmc_summary.json
  cp gpurun_out/${T}_$s/pmc_summary.meta.json $STAGE/${T}_${s}_pmc_summary.meta.json
  cp gpurun_out/${T}_$s/stats/runc/*_kernel_stats.csv $STAGE/${T}_${s}_rocprof_kernel_stats.csv
done
cp gpurun_out/${T}_batch_scaling.txt $STAGE/
if [ -f gpurun_out/${T}_cra_engine.txt ]; then cp gpurun_out/${T}_cra_engine.txt $STAGE/; fi
cp $STAGE/${T}_f32_pmc_summary.json $STAGE/pmc_summary_latest.json; cp $STAGE/${T}_f32_pmc_summary.meta.json $STAGE/pmc_summary_latest.meta.json
cp $STAGE/${T}_bf16_pmc_summary.json $STAGE/pmc_summary_latest_bf16.json; cp $STAGE/${T}_bf16_pmc_summary.meta.json $STAGE/pmc_summary_latest_bf16.meta.json
cp $STAGE/${T}_c4_pmc_summary.json $STAGE/pmc_summary_latest_c4.json; cp $STAGE/${T}_c4_pmc_summary.meta.json $STAGE/pmc_summary_latest_c4.meta.json
mv $STAGE/* profiles/
cat profiles/pmc_summary_latest.meta.json; echo
