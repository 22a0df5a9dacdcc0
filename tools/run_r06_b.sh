#!/bin/bash
# round 6, second GPU call: (1) is the fp16 full-size gradient mismatch of call 1 reproducible?  (2) the whole GPU suite;
# (3) loss tail fused into the output conv: A/B + step table; (4) FORCE_DIST traces by stream.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b
mkdir -p $O
for i in 1 2 3 4; do
  timeout 600 python -m pytest tests/test_gpu_e2e.py -q -k full_size_backward -p no:cacheprovider > $O/flake_$i.txt 2>&1
  tail -3 $O/flake_$i.txt | head -2
done
cp gpurun_out/full_grad_parity.txt $O/full_grad_parity_4runs.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q -rxX > $O/gpu_tests.txt 2>&1
tail -6 $O/gpu_tests.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  timeout 300 $B > $O/bench_fused_$rep.json 2> $O/bench_fused_$rep.err
  C2W_NO_LOSS_FUSION=1 timeout 300 $B > $O/bench_unfused_$rep.json 2> $O/bench_unfused_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])" 2>&1 | tail -1)"; done | tee $O/ab_loss_fusion.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_step -o step --output-format csv -- $T > $O/prof_step.log 2>&1
python3 tools/step_from_trace.py $(find $O/prof_step -name '*kernel_trace.csv' | head -1) > $O/step_table_step.txt 2>&1
head -30 $O/step_table_step.txt
C2W_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_fp32 -o t --output-format csv -- $T > $O/prof_fp32.log 2>&1
C2W_FORCE_DIST=1 C2W_COMM_STREAM=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_fp32_comm -o t --output-format csv -- $T > $O/prof_fp32_comm.log 2>&1
C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16 -o t --output-format csv -- $T > $O/prof_bf16.log 2>&1
C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 C2W_COMM_ON_COMPUTE=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16_oncompute -o t --output-format csv -- $T > $O/prof_bf16_oncompute.log 2>&1
for m in fp32 fp32_comm bf16 bf16_oncompute; do
  f=$(find $O/prof_$m -name '*kernel_trace.csv' | head -1)
  python3 tools/comm_overlap_from_trace.py $f > $O/step_table_force_dist_$m.txt 2>&1
  tail -1 $O/step_table_force_dist_$m.txt
  [ $m = bf16 ] && cp $f $O/kernel_trace_force_dist_bf16.csv
  rm -rf $O/prof_$m
done
find $O/prof_step -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_step.csv \;
rm -rf $O/prof_step
ls $O
