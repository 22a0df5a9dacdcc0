#!/bin/bash
# usage: build_variant.sh <name> <extra hipcc flags...>  -> climate2weather_amd/build/alt/libc2w_<name>.so
# A LABORATORY build: conv_patch3 / conv_patch / wgrad_patch are compiled from csrc/experimental/*_lab.hip (the kernels with their
# compile-time schedule switches, ablation bits -DC2W_EXP=... and timestamp hooks; some of those builds give WRONG results by design)
# instead of the product sources; every other object comes from the product build.  Load it with C2W_LIB=<path> (tools/ab_*.py,
# tools/stamp_*.py).  Without extra flags the lab kernels are the shipped schedules: the A/B partner of a product-source change.
set -e
cd "$(dirname "$0")/../climate2weather_amd"
name=$1; shift
mkdir -p build/alt/$name
for f in conv_patch3 conv_patch wgrad_patch; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Icsrc -Icsrc/experimental "$@" -c csrc/experimental/${f}_lab.hip -o build/alt/$name/$f.o &
done
wait
objs=""
for f in conv_igemm conv_patch conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest; do
  if [ -f build/alt/$name/$f.o ]; then objs="$objs build/alt/$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/alt/libc2w_$name.so $objs
echo built build/alt/libc2w_$name.so
