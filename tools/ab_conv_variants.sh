#!/bin/bash
# A/B of kernel-variant builds on one box: lab/build_variant.sh <name> -D... first; then (on the GPU box)
#   tools/ab_conv_variants.sh default t3v8 t3v9 ...        -> per-variant conv time at the three big levels, twice (noise check)
cd "$(dirname "$0")/.."
for rep in 1 2; do
for name in "$@"; do
  if [ "$name" = default ]; then lib=climate2weather_amd/libc2w_hip.so; else lib=climate2weather_amd/build/alt/libc2w_$name.so; fi
  echo "=== $name (rep $rep)"
  for shape in ${SHAPES:-0 1 2}; do
    C2W_LIB=$PWD/$lib python tools/bench_kernels.py --batch 128 --kind ${KIND:-conv} --only $shape --dtypes bf16 --iters 20 --act ${ACT:-0} 2>&1 | grep -v amdgpu.ids
  done
done
done
