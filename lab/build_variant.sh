#!/bin/bash
# usage: lab/build_variant.sh <name> <extra hipcc flags...>  -> climate2weather_amd/build/alt/libc2w_<name>.so
# A LABORATORY build: conv_patch3 / conv_patch / wgrad_patch are compiled from lab/csrc/*_lab.hip (the kernels with their compile-time
# schedule switches, ablation bits -DC2W_EXP=... and timestamp hooks; some of those builds give WRONG results by design) instead of
# the product sources; every other object comes from the product build.  The lab kernels include the PRODUCT headers (csrc/*.h).
# hipcc cross-compiles here; the built .so travels to the GPU box with the snapshot (lab/ itself does not: .gpurunignore).  Load it
# with C2W_LIB=<path> (tools/ab_*.py).  Without extra flags the lab kernels are the shipped schedules.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/climate2weather_amd"
name=$1; shift
mkdir -p build/alt/$name
for f in conv_patch3 conv_patch wgrad_patch; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Icsrc "$@" -c "$root/lab/csrc/${f}_lab.hip" -o build/alt/$name/$f.o &
done
wait
objs=""
for f in conv_igemm conv_patch conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest; do
  if [ -f build/alt/$name/$f.o ]; then objs="$objs build/alt/$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/alt/libc2w_$name.so $objs
echo built build/alt/libc2w_$name.so
