#!/bin/bash
# one conditioned sampler step at L = 1037 (8 batches of 128 windows + 1 of 1), window batches on ONE stream so that every kernel runs
# alone: kernel trace and the per-step table (launches between two predictor kernels)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export C2W_SCORE_STREAMS=1 C2W_WINDOW_BATCH_FLOOR=0
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b128 -o s --output-format csv -- python3 tools/bench_sampler_configs3.py --lengths 1037 --corrections 0 --steps 4 --members 1 > gpurun_out/prof_b128.log 2>&1
python tools/sampler_step_from_trace.py gpurun_out/prof_b128/s_kernel_trace.csv > gpurun_out/r04_sampler_b128_step_table.txt && rm -rf gpurun_out/prof_b128
head -48 gpurun_out/r04_sampler_b128_step_table.txt
