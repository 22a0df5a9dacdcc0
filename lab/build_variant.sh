#!/bin/bash
# usage: lab/build_variant.sh <name> <extra hipcc flags...>  -> climate2weather_amd/build/alt/libc2w_<name>.so
# A LABORATORY build: conv_patch3 / conv_patch / wgrad_patch are compiled from lab/csrc/*_lab.hip (the kernels with their compile-time
# schedule switches, ablation bits -DC2W_EXP=... and timestamp hooks; some of those builds give WRONG results by design) instead of
# the product sources; every other object comes from the product build.  The lab kernels include the PRODUCT headers (csrc/*.h).
# hipcc cross-compiles here; the built .so travels to the GPU box with the snapshot (lab/ itself does not: .gpurunignore).  Load it
# with C2W_LIB=<path> (tools/ab_*.py).  Without extra flags the lab kernels are the shipped schedules.
#
# STATE (round 6): the lab sources are the ablation RECORD of rounds 1-4, not maintained copies.  wgrad_patch_lab.hip predates the grouped
# entry points of round 5 (c2w_wgrad_patch_group*), conv_patch3_lab.hip / conv_patch_lab.hip predate the round-6 epilogues (fused loss,
# chain form).  LAB_UNITS names the units to take from lab/csrc (default: none of them -- say which one you have ported); the link uses
# --no-undefined, so a unit that lacks an entry point the product objects reference fails HERE and not at dlopen on the GPU box.
set -e
: "${LAB_UNITS:?set LAB_UNITS to the lab units to compile, e.g. LAB_UNITS=conv_patch3 (see the STATE note in this script)}"
root="$(cd "$(dirname "$0")/.." && pwd)"
cd "$root/climate2weather_amd"
name=$1; shift
mkdir -p build/alt/$name
for f in $LAB_UNITS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I../include -Icsrc "$@" -c "$root/lab/csrc/${f}_lab.hip" -o build/alt/$name/$f.o &
done
wait
objs=""
for f in conv_igemm conv_patch conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest; do
  if [ -f build/alt/$name/$f.o ]; then objs="$objs build/alt/$name/$f.o"; else objs="$objs build/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--no-undefined -o build/alt/libc2w_$name.so $objs
echo built build/alt/libc2w_$name.so
