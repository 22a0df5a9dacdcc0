#!/bin/bash
# round-4 profiles: kernel-trace stats of the training step (two-stream default and serialised), of the reference-shaped loop on the
# drop-in path (bf16 autocast; fp16 autocast + GradScaler) and of the deep variant's training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04 -o step --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r04_step.log 2>&1
C2W_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_ser -o ser --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r04_ser.log 2>&1
for leg in bf16_autocast fp16_autocast_gradscaler; do
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_mapi_$leg -o $leg --output-format csv -- python3 tools/bench_module_api.py --legs $leg --steps 5 --warmup 2 > gpurun_out/prof_r04_mapi_$leg.log 2>&1
done
C2W_WGRAD_STREAM=0 PREC=fp16 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r04_deep -o deep --output-format csv -- python3 tools/bench_deep_variant.py > gpurun_out/prof_r04_deep.log 2>&1
ls gpurun_out/prof_r04*/ | head -40
