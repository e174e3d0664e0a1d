cd ${GRAFT_REPO_ROOT:-$(pwd)}
python tools/ab_sites.py --storage bf16 --rounds 3 --steps 10 --sites conv_narrow:res3,conv_narrow_chain nochain=lib=_ab/libcrfp_nochainb.so chain= 2>&1
for v in nochain chain; do lib=crfp_amd/libcrfp_hip.so; [ $v = nochain ] && lib=_ab/libcrfp_nochainb.so; echo "== n=4 $v"; CRFP_HIP_LIB=$PWD/$lib python tools/prof_batch.py bf16 4 2>&1 | grep -E "digest|res3|total"; done
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_round5.py tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -3
