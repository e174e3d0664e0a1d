#!/bin/bash
# Run on the GPU box: dynamic instruction mix per kernel (VALU / SALU / MFMA / LDS / VMEM instructions per wave), one rocprofv3 --pmc pass.
set -u
TAG=${1:-insts}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-extras ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $OUT/a -- $CMD > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/b -- $CMD > /dev/null 2> $OUT/b.err
cd $ROOT
python3 - <<P
import csv, glob, collections
def load(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("crfp::", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc
A, B = load("a"), load("b")
print(f"{'kernel':52s} {'VALU/wave':>10s} {'SALU/wave':>10s} {'MFMA/wave':>10s} {'LDS/wave':>9s} {'VMEMrd/wave':>11s}")
for k in sorted(A, key=lambda k: -A[k]["SQ_INSTS_VALU"]):
    w = max(A[k]["SQ_WAVES"], 1); wb = max(B[k]["SQ_WAVES"], 1)
    print(f"{k[:52]:52s} {A[k]['SQ_INSTS_VALU']/w:10.0f} {A[k]['SQ_INSTS_SALU']/w:10.0f} {B[k]['SQ_INSTS_MFMA']/wb:10.0f} {B[k]['SQ_INSTS_LDS']/wb:9.0f} {B[k]['SQ_INSTS_VMEM_RD']/wb:11.0f}")
P
