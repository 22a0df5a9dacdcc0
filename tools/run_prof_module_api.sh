#!/bin/bash
# kernel-trace stats of the reference-shaped loop (drop-in path) and of the Trainer on the same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for leg in bf16_autocast trainer_bf16 fp16_autocast_gradscaler trainer_fp16; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mapi_$leg -o $leg --output-format csv -- python3 tools/bench_module_api.py --legs $leg --steps 5 --warmup 2 > gpurun_out/prof_mapi_$leg.log 2>&1
  echo "$leg rc=$?"; tail -3 gpurun_out/prof_mapi_$leg.log
done
ls gpurun_out/prof_mapi_*/*/ | head
