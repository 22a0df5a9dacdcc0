#!/bin/bash
# one conditioned sampler step of ONE member at L = 49 (37 windows per score evaluation: the reference's unchanged member loop,
# exp/downscaling.py:248-265): kernel trace, per-step table (launches between two predictor kernels) and the idle time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_sampler_configs3.py --lengths 49 --corrections 0 --steps 12 --members 1 | head -1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_l49 -o s --output-format csv -- python3 tools/bench_sampler_configs3.py --lengths 49 --corrections 0 --steps 12 --members 1 > gpurun_out/prof_l49.log 2>&1
python tools/trace_idle.py gpurun_out/prof_l49/s_kernel_trace.csv predict
