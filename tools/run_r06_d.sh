#!/bin/bash
# round 6, fourth GPU call: loss tail with one atomic per workgroup; chain form opt-in; B = 64 levers; L = 49 under the 16x16-tile threshold;
# the bucket sequence's overlap with an emulated 200-us collective
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dispatch.py -q -k "loss or chain or noise_rows or fused" -p no:cacheprovider > $O/new_tests.txt 2>&1
tail -4 $O/new_tests.txt | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider > $O/e2e.txt 2>&1; tail -1 $O/e2e.txt
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  timeout 300 $B > $O/bench_default_$rep.json 2> $O/bench_default_$rep.err
  C2W_NO_LOSS_FUSION=1 timeout 300 $B > $O/bench_nolossfusion_$rep.json 2> $O/bench_nolossfusion_$rep.err
  C2W_LN_CHAIN=1 timeout 300 $B > $O/bench_chain_$rep.json 2> $O/bench_chain_$rep.err
done
D="C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16"
for rep in 1 2; do
  env $D timeout 300 $B > $O/bench_dist_bf16_comm_$rep.json 2> $O/bench_dist_bf16_comm_$rep.err
  env $D C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_bf16_oncompute_$rep.json 2> $O/bench_dist_bf16_oncompute_$rep.err
  env $D C2W_EMULATE_COLLECTIVE_US=200 timeout 300 $B > $O/bench_dist_bf16_comm_emul200_$rep.json 2> $O/bench_dist_bf16_comm_emul200_$rep.err
  env $D C2W_EMULATE_COLLECTIVE_US=200 C2W_COMM_ON_COMPUTE=1 timeout 300 $B > $O/bench_dist_bf16_oncompute_emul200_$rep.json 2> $O/bench_dist_bf16_oncompute_emul200_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['final_loss'])" 2>&1 | tail -1)"; done | tee $O/ab_step.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_step -o step --output-format csv -- $T > $O/prof_step.log 2>&1
python3 tools/step_from_trace.py $(find $O/prof_step -name '*kernel_trace.csv' | head -1) > $O/step_table_step.txt 2>&1
head -24 $O/step_table_step.txt
find $O/prof_step -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_step.csv \;
rm -rf $O/prof_step
for cfg in "C2W_CONV_T3_MIN_WGS=1024" "C2W_CONV_T3_MIN_WGS=512" "C2W_CONV_T3_MIN_WGS=512 C2W_WGRAD_GROUP_TOP=2" "C2W_CONV_T3_MIN_WGS=512 C2W_WGRAD_GROUP_TOP=3" "C2W_CONV_T3_MIN_WGS=512 C2W_WGRAD_GROUP=1"; do
  for rep in 1 2; do
    tag=$(echo "$cfg" | tr -d ' =' | tr 'A-Z' 'a-z')
    env $cfg timeout 300 python3 tools/bench_module_api.py --legs trainer_bf16_b64 --steps 30 --warmup 5 > $O/b64_${tag}_$rep.json 2> $O/b64_${tag}_$rep.err
    echo "B=64 $cfg rep $rep: $(python3 -c "import json; d=json.load(open('$O/b64_${tag}_$rep.json'))['trainer_bf16_b64']; print(d['ms_per_step'], d['step_ms']['median'], d['windows_per_s'], d.get('mfma_frac_whole_step'))" 2>&1 | tail -1)"
  done
done | tee $O/ab_b64.txt
for w in 1024 512; do
  for rep in 1 2; do
    echo "T3_MIN_WGS=$w rep $rep: $(C2W_CONV_T3_MIN_WGS=$w timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  done
done | tee $O/ab_sampler_t3min.txt
cp gpurun_out/full_grad_parity.txt gpurun_out/chain_vs_written_parity.txt $O/ 2>/dev/null
ls $O | wc -l
