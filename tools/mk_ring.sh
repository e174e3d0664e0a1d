#!/bin/bash
# Build _ab/libcrfp_ring.so: the PRODUCT objects with the bf16 conv object recompiled with -DCRFP_BF16_RING (the LDS-DMA ring conv; enabled
# per process with CRFP_BF16_RING=<min workgroups>).  usage: bash tools/mk_ring.sh ["extra flags"]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/crfp_amd/csrc
mkdir -p $ROOT/_ab/obj
make -C $C -j8 >/dev/null 2>&1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops \
  -DCRFP_BF16_RING -DCRFP_ACT_BF16 -mllvm -amdgpu-sched-strategy=max-ilp ${1:-} -Rpass-analysis=kernel-resource-usage -c $C/conv_mfma.hip -o $ROOT/_ab/obj/conv_mfma.ring.bf16.o 2>&1 \
  | grep -A12 "ring_kernel" | grep -E "error|VGPRs:|Occupancy|Scratch" || true
objs=""
for f in runtime conv_mfma conv_narrow gather resample metrics engine engine_rt api spynet; do objs="$objs $C/build/$f.o"; done
for f in conv_narrow gather resample engine; do objs="$objs $C/build/$f.bf16.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/_ab/libcrfp_ring.so $objs $ROOT/_ab/obj/conv_mfma.ring.bf16.o
echo built _ab/libcrfp_ring.so
