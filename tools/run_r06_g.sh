#!/bin/bash
# round 6: the bucket sequence on a communication stream picked by the PATTERN probe (streams.py): does an emulated 200-us collective hide?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
D="C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16"
for rep in 1 2; do
  timeout 300 $B > $O/bench_plain_$rep.json 2> $O/bench_plain_$rep.err
  env $D C2W_STREAM_DEBUG=1 timeout 300 $B > $O/bench_dist_bf16_comm_$rep.json 2> $O/bench_dist_bf16_comm_$rep.err
  env $D C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_bf16_oncompute_$rep.json 2> $O/bench_dist_bf16_oncompute_$rep.err
  env $D C2W_EMULATE_COLLECTIVE_US=200 timeout 300 $B > $O/bench_dist_bf16_comm_emul200_$rep.json 2> $O/bench_dist_bf16_comm_emul200_$rep.err
  env $D C2W_EMULATE_COLLECTIVE_US=200 C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_bf16_oncompute_emul200_$rep.json 2> $O/bench_dist_bf16_oncompute_emul200_$rep.err
  C2W_FORCE_DIST=1 timeout 300 $B > $O/bench_dist_fp32_$rep.json 2> $O/bench_dist_fp32_$rep.err
  C2W_FORCE_DIST=1 C2W_CHASE_OPT=1 timeout 300 $B > $O/bench_dist_fp32_chase_$rep.json 2> $O/bench_dist_fp32_chase_$rep.err
done
grep -h independent_stream $O/*.err | sort | uniq -c
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['final_loss'])" 2>&1 | tail -1)"; done | tee $O/ab_comm_stream.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
env $D C2W_EMULATE_COLLECTIVE_US=200 timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16_emul -o t --output-format csv -- $T > $O/prof_bf16_emul.log 2>&1
env $D timeout 600 rocprofv3 --kernel-trace -d $O/prof_bf16 -o t --output-format csv -- $T > $O/prof_bf16.log 2>&1
C2W_FORCE_DIST=1 C2W_COMM_STREAM=1 timeout 600 rocprofv3 --kernel-trace -d $O/prof_fp32 -o t --output-format csv -- $T > $O/prof_fp32.log 2>&1
for m in bf16_emul bf16 fp32; do
  f=$(find $O/prof_$m -name '*kernel_trace.csv' | head -1)
  python3 tools/comm_overlap_from_trace.py $f > $O/step_table_force_dist_$m.txt 2>&1
  tail -1 $O/step_table_force_dist_$m.txt
  rm -rf $O/prof_$m
done
