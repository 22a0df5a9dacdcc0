#!/bin/bash
# flake watch: the full-size gradient parity test N times in fresh processes, and N times inside one process
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k
mkdir -p $O
fail=0
for i in $(seq 1 ${N:-24}); do
  timeout 300 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider > $O/e2e_$i.txt 2>&1 || { fail=$((fail+1)); cp $O/e2e_$i.txt $O/FAILED_e2e_$i.txt; }
  rm -f $O/e2e_$i.txt
done
echo "whole tests/test_gpu_e2e.py, fresh process each time: $fail failures in ${N:-24} runs" | tee $O/flake_summary.txt
ls gpurun_out/parity_fail* 2>/dev/null | tee -a $O/flake_summary.txt
cp gpurun_out/parity_fail* $O/ 2>/dev/null
