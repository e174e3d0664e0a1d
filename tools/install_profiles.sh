#!/bin/bash
# Copy one measurement set (tools/measure_round.sh <tag>, merged back into gpurun_out/) into profiles/ under its tag and make its PMC
# summaries the ones bench.py reads (profiles/pmc_summary_latest*).  Usage: bash tools/install_profiles.sh <tag>
T=$1
for c in 2 3 4 5; do cp gpurun_out/${T}_bench_c$c.json profiles/${T}_bench_config$c.json; done
for s in f32 bf16; do
  cp gpurun_out/${T}_site_$s.txt profiles/${T}_${s}_per_site_table.txt
  cp gpurun_out/${T}_lockstep_vs_loop_$s.txt profiles/
  cp gpurun_out/${T}_${s}_util/util_summary.txt profiles/${T}_${s}_mfma_lds_util.txt
done
for s in f32 bf16 c4; do
  cp gpurun_out/${T}_$s/summary.txt profiles/${T}_${s}_pmc_summary.txt
  cp gpurun_out/${T}_$s/pmc_summary.json profiles/${T}_${s}_pmc_summary.json
  cp gpurun_out/${T}_$s/pmc_summary.meta.json profiles/${T}_${s}_pmc_summary.meta.json
  cp gpurun_out/${T}_$s/stats/runc/*_kernel_stats.csv profiles/${T}_${s}_rocprof_kernel_stats.csv
done
cp gpurun_out/${T}_batch_scaling.txt profiles/
[ -f gpurun_out/${T}_cra_engine.txt ] && cp gpurun_out/${T}_cra_engine.txt profiles/
cp profiles/${T}_f32_pmc_summary.json profiles/pmc_summary_latest.json; cp profiles/${T}_f32_pmc_summary.meta.json profiles/pmc_summary_latest.meta.json
cp profiles/${T}_bf16_pmc_summary.json profiles/pmc_summary_latest_bf16.json; cp profiles/${T}_bf16_pmc_summary.meta.json profiles/pmc_summary_latest_bf16.meta.json
cp profiles/${T}_c4_pmc_summary.json profiles/pmc_summary_latest_c4.json; cp profiles/${T}_c4_pmc_summary.meta.json profiles/pmc_summary_latest_c4.meta.json
cat profiles/pmc_summary_latest.meta.json; echo
