#!/bin/bash
# round 6: one sampler step of one member at L = 49, split-K on / off: per-kernel tables
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h
mkdir -p $O
for m in on off; do
  [ $m = off ] && export C2W_NO_SPLITK=1 || unset C2W_NO_SPLITK
  timeout 600 rocprofv3 --kernel-trace -d $O/prof_$m -o s --output-format csv -- python3 tools/bench_sampler_configs3.py --lengths 49 --corrections 0 --steps 12 --members 1 > $O/prof_$m.log 2>&1
  python3 tools/sampler_step_from_trace.py $(find $O/prof_$m -name '*kernel_trace.csv' | head -1) > $O/sampler_l49_step_table_splitk_$m.txt 2>&1
  head -16 $O/sampler_l49_step_table_splitk_$m.txt | cut -c1-160
  rm -rf $O/prof_$m
done
