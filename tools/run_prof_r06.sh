#!/bin/bash
# round-6 closing evidence with the final library: the GPU suite in both stream modes, kernel-trace stats + per-step table of the training
# step, PMC passes over the same step (tools/pmc_step.sh -> profiles/r06_pmc_step.json, which bench.py's roofline.traffic reads when the
# sources digest matches), the reference-shaped loop plain / under DDP, and the full default bench line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06p
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider > $O/gpu_tests_tail_full.txt 2>&1; tail -6 $O/gpu_tests_tail_full.txt > $O/gpu_tests_tail.txt; tail -1 $O/gpu_tests_tail.txt
C2W_WGRAD_STREAM=1 timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider > $O/gpu_tests_tail_two_streams_full.txt 2>&1; tail -6 $O/gpu_tests_tail_two_streams_full.txt > $O/gpu_tests_tail_two_streams.txt; tail -1 $O/gpu_tests_tail_two_streams.txt
cp gpurun_out/full_grad_parity.txt $O/full_grad_parity.txt 2>/dev/null; cp gpurun_out/bench_step_parity.txt $O/bench_step_parity.txt 2>/dev/null; cp gpurun_out/chain_vs_written_parity.txt $O/ 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r06 -o step --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $O/prof_step.log 2>&1
f=$(find gpurun_out/prof_r06 -name '*kernel_trace.csv' | head -1)
python3 tools/step_from_trace.py $f > $O/step_table_step.txt 2>&1
cp $(find gpurun_out/prof_r06 -name '*kernel_stats.csv' | head -1) $O/kernel_stats_step.csv
rm -rf gpurun_out/prof_r06
TAG=r06 bash tools/pmc_step.sh
python3 tools/pmc_step_summary.py gpurun_out/pmc_step_r06 conv_patch_t3 wgrad_patch > $O/pmc_step.json 2> $O/pmc_step.err
rm -rf gpurun_out/pmc_step_r06
for args in "" "--ddp" "--ddp --bucket-view"; do
  tag=$(echo "plain$args" | tr -d ' -')
  python3 tools/bench_module_api.py --legs bf16_autocast --steps 20 --warmup 4 $args > $O/mapi_$tag.json 2> $O/mapi_$tag.err
done
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench_line.json 2> $O/final_bench.err
cp gpurun_out/bench_extras.json $O/final_bench_extras.json
tail -c 600 $O/final_bench_line.json
ls $O
