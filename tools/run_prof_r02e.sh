# final profiles of round 2: the driver-style bench line, then rocprofv3 --kernel-trace --stats of the same command (two-stream backward)
# and with the backward on one stream (every kernel alone)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02e_bench.json 2> gpurun_out/r02e_bench.err; tail -c 300 gpurun_out/r02e_bench.err
C2W_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02e_ser -o ser --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r02e_ser.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02e -o two --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r02e.log 2>&1
python tools/trace_idle.py gpurun_out/prof_r02e/two_kernel_trace.csv | tee gpurun_out/r02e_idle.txt
