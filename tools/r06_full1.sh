#!/bin/bash
# full GPU suite + one measurement set
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_v1_pytest_gpu.txt 2>&1
tail -5 gpurun_out/r06_v1_pytest_gpu.txt
bash tools/measure_round.sh r06_v1 2>&1 | tail -8
