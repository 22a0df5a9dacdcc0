#!/bin/bash
# round 6, third GPU call: new kernels' tests (noise rows, fused loss, chain form), flake watch on the full-size gradient test,
# step A/Bs of the two knobs, step table, B = 64 dispatch threshold
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dispatch.py -q -k "loss or chain or noise_rows or fused or bench_size" -p no:cacheprovider > $O/new_tests.txt 2>&1
tail -12 $O/new_tests.txt | cut -c1-300
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider > $O/e2e_$i.txt 2>&1
  tail -1 $O/e2e_$i.txt
done
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  timeout 300 $B > $O/bench_default_$rep.json 2> $O/bench_default_$rep.err
  C2W_NO_LOSS_FUSION=1 timeout 300 $B > $O/bench_nolossfusion_$rep.json 2> $O/bench_nolossfusion_$rep.err
  C2W_NO_LN_CHAIN=1 timeout 300 $B > $O/bench_nochain_$rep.json 2> $O/bench_nochain_$rep.err
  C2W_NO_LN_CHAIN=1 C2W_NO_LOSS_FUSION=1 timeout 300 $B > $O/bench_neither_$rep.json 2> $O/bench_neither_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['final_loss'])" 2>&1 | tail -1)"; done | tee $O/ab_fusions.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_step -o step --output-format csv -- $T > $O/prof_step.log 2>&1
python3 tools/step_from_trace.py $(find $O/prof_step -name '*kernel_trace.csv' | head -1) > $O/step_table_step.txt 2>&1
head -24 $O/step_table_step.txt
find $O/prof_step -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_step.csv \;
rm -rf $O/prof_step
for w in 1024 512 256; do
  for rep in 1 2; do
    C2W_CONV_T3_MIN_WGS=$w timeout 300 python3 tools/bench_module_api.py --legs trainer_bf16_b64 --steps 30 --warmup 5 > $O/b64_t3min${w}_$rep.json 2> $O/b64_t3min${w}_$rep.err
    echo "B=64 T3_MIN_WGS=$w rep $rep: $(python3 -c "import json; d=json.loads(open('$O/b64_t3min${w}_$rep.json').read().strip().splitlines()[-1]); print(d['trainer_bf16_b64']['ms_per_step'], d['trainer_bf16_b64']['step_ms'])" 2>&1 | tail -1)"
  done
done | tee $O/ab_b64.txt
cp gpurun_out/full_grad_parity.txt gpurun_out/bench_step_parity.txt gpurun_out/chain_vs_written_parity.txt $O/ 2>/dev/null
ls gpurun_out/parity_fail* 2>/dev/null
ls $O | head -50
