#!/bin/bash
# round-3 profiles: kernel-trace stats of the training step (two-stream default and serialised) and of one conditioned L = 8737 sampler step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03 -o step --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r03_step.log 2>&1
C2W_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_ser -o ser --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r03_ser.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r03_sampler -o sampler --output-format csv -- python3 tools/bench_sampler_configs3.py --lengths 8737 --corrections 0 --steps 2 --members 1 > gpurun_out/prof_r03_sampler.log 2>&1
ls gpurun_out/prof_r03*/*/ 2>/dev/null | head -30
