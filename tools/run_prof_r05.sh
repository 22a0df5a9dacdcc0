#!/bin/bash
# round-5 profiles: kernel-trace stats of the training step (one stream: the default since round 5; and with the weight gradients on a
# second stream, C2W_WGRAD_STREAM=1: the default of rounds 1-4), PMC passes over the same step
# (tools/pmc_step.sh), and the reference-shaped loop under torch DDP with and without gradient_as_bucket_view.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05p
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05 -o step --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r05_step.log 2>&1
C2W_WGRAD_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r05_two -o two --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_r05_two.log 2>&1
for d in prof_r05 prof_r05_two; do
  f=$(find gpurun_out/$d -name '*kernel_trace.csv' | head -1)
  python3 tools/step_from_trace.py $f > gpurun_out/r05p/step_table_${d#prof_r05}.txt 2>&1
  cp $(find gpurun_out/$d -name '*kernel_stats.csv' | head -1) gpurun_out/r05p/kernel_stats${d#prof_r05}.csv
done
TAG=r05 bash tools/pmc_step.sh
python3 tools/pmc_step_summary.py gpurun_out/pmc_step_r05 conv_patch_t3 wgrad_patch > gpurun_out/r05p/pmc_step.json 2> gpurun_out/r05p/pmc_step.err
for args in "" "--ddp" "--ddp --bucket-view"; do
  tag=$(echo "plain$args" | tr -d ' -')
  python3 tools/bench_module_api.py --legs bf16_autocast --steps 20 --warmup 4 $args > gpurun_out/r05p/mapi_$tag.json 2> gpurun_out/r05p/mapi_$tag.err
done
ls gpurun_out/r05p
