#!/bin/bash
# round 6, first GPU call: the GPU suite, the step with and without a one-rank RCCL communicator (fp32 / bf16 wire; C2W_COMM_ON_COMPUTE=1 =
# round 5's issue order), and kernel traces of the FORCE_DIST steps for tools/comm_overlap_from_trace.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -rxX > $O/gpu_tests.txt 2>&1
tail -4 $O/gpu_tests.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  timeout 300 $B > $O/bench_plain_$rep.json 2> $O/bench_plain_$rep.err
  C2W_FORCE_DIST=1 timeout 300 $B > $O/bench_dist_fp32_$rep.json 2> $O/bench_dist_fp32_$rep.err
  C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 timeout 300 $B > $O/bench_dist_bf16_$rep.json 2> $O/bench_dist_bf16_$rep.err
  C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_bf16_oncompute_$rep.json 2> $O/bench_dist_bf16_oncompute_$rep.err
  C2W_FORCE_DIST=1 C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_fp32_oncompute_$rep.json 2> $O/bench_dist_fp32_oncompute_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'])" 2>&1 | tail -1)"; done | tee $O/ab_comm_stream.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
C2W_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_fp32 -o t --output-format csv -- $T > $O/prof_fp32.log 2>&1
C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16 -o t --output-format csv -- $T > $O/prof_bf16.log 2>&1
C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 C2W_COMM_ON_COMPUTE=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16_oncompute -o t --output-format csv -- $T > $O/prof_bf16_oncompute.log 2>&1
for m in fp32 bf16 bf16_oncompute; do
  f=$(find $O/prof_$m -name '*kernel_trace.csv' | head -1)
  python3 tools/comm_overlap_from_trace.py $f > $O/step_table_force_dist_$m.txt 2>&1
  tail -1 $O/step_table_force_dist_$m.txt
  rm -rf $O/prof_$m
done
cp gpurun_out/full_grad_parity.txt $O/ 2>/dev/null; cp gpurun_out/bench_step_parity.txt $O/ 2>/dev/null
ls $O
