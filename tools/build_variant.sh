#!/bin/bash
# usage: build_variant.sh <name> <extra hipcc flags...>  -> climate2weather_amd/build/alt/libc2w_<name>.so
set -e
cd /root/repo/climate2weather_amd
name=$1; shift
mkdir -p build/alt/$name
for f in conv_igemm conv_patch conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler; do
  if [ $f = conv_patch3 ] || [ $f = wgrad_patch ] || [ $f = conv_patch ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Icsrc "$@" -c csrc/$f.hip -o build/alt/$name/$f.o &
  fi
done
wait
objs=""
for f in conv_igemm conv_patch conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler; do
  if [ -f build/alt/$name/$f.o ]; then objs="$objs build/alt/$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/alt/libc2w_$name.so $objs
echo built build/alt/libc2w_$name.so
