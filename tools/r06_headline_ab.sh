#!/bin/bash
# same-box A/B of the round-6 library against the round-5 library (_ab/libcrfp_r05.so = the library of commit 391cfcb): VERDICT r5 item 5
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
O=gpurun_out/r06_headline_ab.txt
{
echo "# same-box A/B, round-5 library (r05 = commit 391cfcb's libcrfp_hip.so) vs the round-6 library (r06); tools/ab_sites.py --rounds 3 --steps 10 (min over rounds)"
echo "## BASELINE configs[1]: fp32, 7 x 180 x 320, one clip per call"
timeout 1200 python tools/ab_sites.py --sites fused,conv_narrow,state_lrelu --rounds 3 --steps 10 r05=lib=_ab/libcrfp_r05.so r06= 2>&1
echo "## bf16 storage, 7 x 180 x 320, one clip per call"
timeout 1200 python tools/ab_sites.py --storage bf16 --sites fused,state_lrelu --rounds 3 --steps 10 r05=lib=_ab/libcrfp_r05.so r06= 2>&1
echo "## bf16, one frame per call (30 calls per step, resident inputs)"
timeout 900 python tools/ab_sites.py --storage bf16 --mode stream --t 30 --sites fused --rounds 2 r05=lib=_ab/libcrfp_r05.so,AB_RESIDENT=1 r06=AB_RESIDENT=1 2>&1
echo "## bf16 lock-step batch of 4 clips (tools/prof_batch.py bf16 4: ms of kernels per clip, one-clip calls | lock-step)"
for v in r05 r06; do
  lib=crfp_amd/libcrfp_hip.so; [ $v = r05 ] && lib=_ab/libcrfp_r05.so
  echo "# $v"; CRFP_HIP_LIB=$PWD/$lib timeout 600 python tools/prof_batch.py bf16 4 2>&1 | grep -E "digest|fused|total"
done
} > $O 2>&1
cat $O
