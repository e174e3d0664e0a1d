#!/bin/bash
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
O=gpurun_out/r06_ab4
V="r05=lib=_ab/libcrfp_r05.so new= noseq=lib=_ab/libcrfp_noseq.so nost2=lib=_ab/libcrfp_nost2.so"
timeout 1200 python tools/ab_sites.py --sites conv_narrow,state_lrelu --rounds 2 --steps 8 $V > ${O}_f32.txt 2>&1
cat ${O}_f32.txt
timeout 900 python tools/ab_sites.py --storage bf16 --sites conv_narrow:res3,state_lrelu --rounds 2 --steps 8 r05=lib=_ab/libcrfp_r05.so new= nost2=lib=_ab/libcrfp_nost2.so > ${O}_bf16.txt 2>&1
cat ${O}_bf16.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gate.py tests/test_gpu_round2.py -m gpu -x -q > ${O}_tests.txt 2>&1
tail -5 ${O}_tests.txt
