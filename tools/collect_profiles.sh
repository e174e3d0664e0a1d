#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + HBM traffic counters of the bench command.
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; never combined
# with sys/hip/hsa tracing).  Summaries land in gpurun_out/<tag>/ ; copy what should be judged to profiles/.
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-extras ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/bench_stats.json 2> $OUT/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- $CMD > /dev/null 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- $CMD > /dev/null 2> $OUT/write.err
cd $ROOT
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/fetch/*/*kernel_trace.csv $OUT/write/*/*kernel_trace.csv $OUT/stats/*/*kernel_trace.csv
tail -40 $OUT/summary.txt
