#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
O=gpurun_out/r06_ab7
V="seqslow=lib=_ab/libcrfp_seqslow.so nochain=lib=_ab/libcrfp_nochain.so chain3=lib=_ab/libcrfp_chain3.so new="
timeout 1200 python tools/ab_sites.py --sites conv_narrow --rounds 2 --steps 8 $V > ${O}_f32.txt 2>&1
cat ${O}_f32.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or odd_geom or seven or config_b" > ${O}_tests.txt 2>&1
tail -5 ${O}_tests.txt
