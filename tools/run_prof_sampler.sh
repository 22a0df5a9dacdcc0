# sampler leg: steps/s at the shipped trajectory lengths, eager and hipGraph-replayed, and the GPU idle time per sampler step from a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_sampler.py --lengths 49,140 --steps 16
python tools/bench_sampler.py --lengths 49 --steps 16 --graph 1
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_samp -o s --output-format csv -- python3 tools/bench_sampler.py --lengths 49 --steps 12 > gpurun_out/prof_samp.log 2>&1
python tools/trace_idle.py gpurun_out/prof_samp/s_kernel_trace.csv predict
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/prof_samp2 -o s --output-format csv -- python3 tools/bench_sampler.py --lengths 140 --steps 12 > gpurun_out/prof_samp2.log 2>&1
python tools/trace_idle.py gpurun_out/prof_samp2/s_kernel_trace.csv predict
