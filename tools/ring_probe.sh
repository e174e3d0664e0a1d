#!/bin/bash
# destructive timing probes of the ring conv (results wrong by design): CRFP_BF16_RING_PROBE bits 1 = no DMA issue, 2 = no operand reads / MFMAs,
# 4 = no epilogue, 8 = return at once, 16 = preamble only.  us per clip of the 32 -> 32 convs in a 4-clip lock-step batch.
set -uo pipefail
cd "$(dirname "$0")/.."
for wgs in ${RING_WGS:-64 128}; do for p in 0 1 2 4 6 3 5 7 16 8; do
  echo -n "wgs/clip $wgs probe $p: "
  CRFP_HIP_LIB=$PWD/_ab/libcrfp_ring.so CRFP_BF16_RING=1 CRFP_BF16_RING_WGS=$wgs CRFP_BF16_RING_PROBE=$p python tools/prof_batch.py bf16 4 2>&1 | grep -E "conv_mfma:res.conv1 |conv_mfma:res.main0 " | awk '{printf "%s %s us/clip   ", $1, $6}'; echo
done; done
