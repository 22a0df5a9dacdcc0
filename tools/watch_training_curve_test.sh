#!/bin/bash
# flake watch of tests/test_gpu_host.py::test_bf16_and_fp16_training_track_fp32_training (60 optimizer steps of a small network in fp32 / bf16 /
# fp16; the bf16 tail must stay within 10 % of the fp32 one): N fresh-process runs with the default kernel selection and N with round 5's
# (C2W_CONV_S2_PATCH=0 C2W_NO_HALF8=1 C2W_TS2_PAIRS=0), interleaved.  (Run when the stride-2 forward kernel was still ON by default; the host now
# keeps it off, _lib.HOST_KNOB_DEFAULTS: set C2W_CONV_S2_PATCH=1 in the environment to repeat the watch with it.)   N=30 bash tools/watch_training_curve_test.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/curve_watch
mkdir -p $O
T=tests/test_gpu_host.py::test_bf16_and_fp16_training_track_fp32_training
fa=0; fb=0
for i in $(seq 1 ${N:-30}); do
  timeout 120 python -m pytest $T -q -p no:cacheprovider > $O/a.txt 2>&1 || { fa=$((fa+1)); echo "default run $i: $(grep -m1 AssertionError $O/a.txt)"; }
  C2W_CONV_S2_PATCH=0 C2W_NO_HALF8=1 C2W_TS2_PAIRS=0 timeout 120 python -m pytest $T -q -p no:cacheprovider > $O/b.txt 2>&1 || { fb=$((fb+1)); echo "round-5 selection run $i: $(grep -m1 AssertionError $O/b.txt)"; }
done
echo "default kernel selection: $fa failures in ${N:-30} runs; round 5's kernel selection: $fb failures in ${N:-30} runs"
