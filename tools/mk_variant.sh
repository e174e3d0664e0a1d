#!/bin/bash
# A/B builds of the lab library: recompile ONE source with extra flags and link it with the other lab objects.
#   tools/mk_variant.sh <name> <file.hip> "<extra flags>"   ->  _ab/libcrfp_<name>.so   (use: CRFP_HIP_LIB=_ab/libcrfp_<name>.so)
set -e
name=$1; src=$2; extra=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/crfp_amd/csrc
mkdir -p $ROOT/_ab/obj
base=$(basename $src .hip)
SCHED="-mllvm -amdgpu-sched-strategy=max-memory-clause"
[ "$base" = conv_mfma ] && SCHED="-mllvm -amdgpu-sched-strategy=max-ilp"
[ "$base" = gather ] && SCHED=""
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=off -Xclang -target-feature -Xclang -packed-fp32-ops"
LABDEF="-DCRFP_LAB"; OBJDIR=build_lab
if [ "${LAB:-1}" = 0 ]; then LABDEF=""; OBJDIR=build; fi   # LAB=0: a variant of the PRODUCT build (no lab switches compiled in)
/opt/rocm/bin/hipcc $FLAGS $LABDEF $SCHED $extra -c $C/$base.hip -o $ROOT/_ab/obj/$base.$name.o 2>&1 | grep -v "not a recognized feature" | grep -E "error|warning: fail" || true
objs=""
for f in runtime conv_mfma conv_narrow gather resample metrics engine engine_rt api spynet; do   # = SRCS of csrc/Makefile
  if [ $f = $base ]; then objs="$objs $ROOT/_ab/obj/$base.$name.o"; else objs="$objs $C/$OBJDIR/$f.o"; fi
done
# BF16=1: the bf16-storage object of the same source is recompiled with the same flags too (default: the product's bf16 objects)
for f in conv_mfma conv_narrow gather resample engine; do
  if [ "${BF16:-0}" = 1 ] && [ $f = $base ]; then
    /opt/rocm/bin/hipcc $FLAGS -DCRFP_ACT_BF16 $SCHED $extra -c $C/$base.hip -o $ROOT/_ab/obj/$base.$name.bf16.o 2>&1 | grep -v "not a recognized feature" | grep -E "error|warning: fail" || true
    objs="$objs $ROOT/_ab/obj/$base.$name.bf16.o"
  else objs="$objs $C/build/$f.bf16.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/_ab/libcrfp_$name.so $objs
echo built _ab/libcrfp_$name.so
