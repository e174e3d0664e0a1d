#!/bin/bash
# same-box A/B of the LDS-DMA ring form of the bf16 32-cout convs (-DCRFP_BF16_RING build in _ab/libcrfp_ring.so) against the shipped kernels:
# 4-clip lock-step batch (per-kernel us per clip) and one clip (per-site us).  usage: bash tools/ring_ab.sh > gpurun_out/<tag>_ring_ab.txt
set -uo pipefail
cd "$(dirname "$0")/.."
echo "== lock-step batch of 4 clips, shipped"; python tools/prof_batch.py bf16 4 2>&1 | grep -E "digest|conv_mfma|total"
for wgs in 0 64 128; do
  echo "== lock-step batch of 4 clips, ring (CRFP_BF16_RING=1, CRFP_BF16_RING_WGS=$wgs)"
  CRFP_HIP_LIB=$PWD/_ab/libcrfp_ring.so CRFP_BF16_RING=1 CRFP_BF16_RING_WGS=$wgs python tools/prof_batch.py bf16 4 2>&1 | grep -E "digest|conv_mfma|total"
done
echo "== one clip (n = 1)"
python tools/ab_sites.py --storage bf16 --rounds 1 --sites conv_mfma shipped= ring=lib=_ab/libcrfp_ring.so,CRFP_BF16_RING=1 ring256=lib=_ab/libcrfp_ring.so,CRFP_BF16_RING=1,CRFP_BF16_RING_WGS=256 2>&1
