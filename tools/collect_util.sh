#!/bin/bash
# Run on the GPU box: MFMA utilisation and LDS activity counters of the bench command, one rocprofv3 --pmc pass each
# (never combined with other tracing).  Output: gpurun_out/<tag>/util_summary.txt
set -u
TAG=${1:-util}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-extras ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- $CMD > /dev/null 2> $OUT/mfma.err
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -- $CMD > /dev/null 2> $OUT/lds.err
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- $CMD > /dev/null 2> $OUT/sq.err
# co-execution of vector and matrix instructions, wave-parked cycles (MI355X_MICROARCH.md, two waves per SIMD, item 9)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/coexec -- $CMD > /dev/null 2> $OUT/coexec.err
cd $ROOT
python3 tools/summarize_util.py $OUT > $OUT/util_summary.txt 2>&1
rm -f $OUT/*/*/*kernel_trace.csv
cat $OUT/util_summary.txt
