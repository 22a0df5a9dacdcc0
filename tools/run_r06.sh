#!/bin/bash
# Round 6: every GPU call of the round as one stage of this script (one gpurun call each; output under gpurun_out/r06<stage>/, the files kept
# for the record are copies under profiles/ -- profiles/README.md and profiles/r06_experiments.md say which).  The closing evidence of the
# round (suite in both stream modes, step table, PMC passes, module-API legs, full bench line) is tools/run_prof_r06.sh.
#   usage: bash tools/run_r06.sh <stage>     stages: a b c d e g h i m n o r t flake z h8x h8y v t3p s2 s2p
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

stage_a() {
# round 6, first GPU call: the GPU suite, the step with and without a one-rank RCCL communicator (fp32 / bf16 wire; C2W_COMM_ON_COMPUTE=1 =
# round 5's issue order), and kernel traces of the FORCE_DIST steps for tools/comm_overlap_from_trace.py
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
}

stage_b() {
# round 6, second GPU call: (1) is the fp16 full-size gradient mismatch of call 1 reproducible?  (2) the whole GPU suite;
# (3) loss tail fused into the output conv: A/B + step table; (4) FORCE_DIST traces by stream.
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
}

stage_c() {
# round 6, third GPU call: new kernels' tests (noise rows, fused loss, chain form), flake watch on the full-size gradient test,
# step A/Bs of the two knobs, step table, B = 64 dispatch threshold
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
}

stage_d() {
# round 6, fourth GPU call: loss tail with one atomic per workgroup; chain form opt-in; B = 64 levers; L = 49 under the 16x16-tile threshold;
# the bucket sequence's overlap with an emulated 200-us collective
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
}

stage_e() {
# round 6, fifth GPU call: split-K tests + the whole suite; does the communication stream overlap (probe + debug prints); defaults A/B; sampler with
# split-K; what the DDP bucket-view path launches
O=gpurun_out/r06e
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dispatch.py -q -k "split_k" -p no:cacheprovider > $O/splitk_tests.txt 2>&1
tail -3 $O/splitk_tests.txt | cut -c1-300
timeout 300 python3 tools/probe_comm_stream.py > $O/probe_comm_stream.txt 2>&1; grep -v "^\[" $O/probe_comm_stream.txt | tail -6
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
D="C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 C2W_EMULATE_COLLECTIVE_US=200"
env $D C2W_STREAM_DEBUG=1 timeout 300 $B > $O/bench_dist_debug.json 2> $O/bench_dist_debug.err; grep -i "independent_stream" $O/bench_dist_debug.err $O/bench_dist_debug.json | head
env $D C2W_WGRAD_STREAM=1 timeout 300 $B > $O/bench_dist_twostream_emul200.json 2> $O/bench_dist_twostream_emul200.err
C2W_FORCE_DIST=1 C2W_ALLREDUCE_DTYPE=bf16 C2W_WGRAD_STREAM=1 timeout 300 $B > $O/bench_dist_twostream.json 2> $O/bench_dist_twostream.err
for rep in 1 2; do
  timeout 300 $B > $O/bench_default_$rep.json 2> $O/bench_default_$rep.err
  C2W_CONV_T3_MIN_WGS=1024 timeout 300 $B > $O/bench_t3min1024_$rep.json 2> $O/bench_t3min1024_$rep.err
done
for f in $O/bench_*.json; do echo "$f $(python3 -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms'], d['final_loss'])" 2>&1 | tail -1)"; done | tee $O/ab_step.txt
for rep in 1 2; do
  echo "default rep $rep: $(timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "C2W_NO_SPLITK=1 rep $rep: $(C2W_NO_SPLITK=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler_splitk.txt
echo "B=64 default: $(timeout 300 python3 tools/bench_module_api.py --legs trainer_bf16_b64 --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin)['trainer_bf16_b64']; print(d['ms_per_step'], d['windows_per_s'], d.get('mfma_frac_whole_step'))")" | tee $O/b64_default.txt
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_ddp -o ddp --output-format csv -- python3 tools/bench_module_api.py --legs bf16_autocast --ddp --bucket-view --steps 8 --warmup 3 > $O/prof_ddp.log 2>&1
python3 - <<'PY' > $O/ddp_bucketview_step_kernels.txt 2>&1
import csv, glob
f = glob.glob("gpurun_out/r06e/prof_ddp/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r["Kernel_Name"]]
i0, i1 = marks[-3], marks[-2]
agg = {}
for r in rows[i0 + 1:i1 + 1]:
    n = r["Kernel_Name"].replace("void ", "")[:100]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
span = (int(rows[i1]["End_Timestamp"]) - int(rows[i0]["End_Timestamp"])) / 1e6
print(f"one step of the five-strings loop under DDP with gradient_as_bucket_view=True: {i1 - i0} launches, {span:.3f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{v[0]:6d} launches {v[1] / 1e3:8.3f} ms  {k}")
PY
head -16 $O/ddp_bucketview_step_kernels.txt
rm -rf $O/prof_ddp
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt | cut -c1-250
}

stage_g() {
# round 6: the bucket sequence on a communication stream picked by the PATTERN probe (streams.py): does an emulated 200-us collective hide?
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
}

stage_h() {
# round 6: one sampler step of one member at L = 49, split-K on / off: per-kernel tables
O=gpurun_out/r06h
mkdir -p $O
for m in on off; do
  [ $m = off ] && export C2W_NO_SPLITK=1 || unset C2W_NO_SPLITK
  timeout 600 rocprofv3 --kernel-trace -d $O/prof_$m -o s --output-format csv -- python3 tools/bench_sampler_configs3.py --lengths 49 --corrections 0 --steps 12 --members 1 > $O/prof_$m.log 2>&1
  python3 tools/sampler_step_from_trace.py $(find $O/prof_$m -name '*kernel_trace.csv' | head -1) > $O/sampler_l49_step_table_splitk_$m.txt 2>&1
  head -16 $O/sampler_l49_step_table_splitk_$m.txt | cut -c1-160
  rm -rf $O/prof_$m
done
}

stage_i() {
O=gpurun_out/r06i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dispatch.py -q -k "split_k" -p no:cacheprovider > $O/splitk_tests.txt 2>&1
tail -3 $O/splitk_tests.txt | cut -c1-300
for rep in 1 2 3; do
  echo "default rep $rep: $(timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "C2W_NO_SPLITK=1 rep $rep: $(C2W_NO_SPLITK=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler_splitk.txt
stage_h
cp gpurun_out/r06h/sampler_l49_step_table_splitk_*.txt $O/
timeout 600 python -m pytest tests/test_gpu_host.py -q -k "sampler or ensemble or score or guidance or window" -p no:cacheprovider > $O/sampler_tests.txt 2>&1; tail -2 $O/sampler_tests.txt
}

stage_m() {
O=gpurun_out/r06m
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_dispatch.py -q -k "ts2 or TS2 or stride or conv_case or forward_case or dispatch or bench_size" -p no:cacheprovider > $O/ts2_tests.txt 2>&1
tail -3 $O/ts2_tests.txt | cut -c1-300
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  echo "pairs rep $rep: $(timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
  echo "C2W_TS2_PAIRS=0 rep $rep: $(C2W_TS2_PAIRS=0 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
done | tee $O/ab_ts2_pairs.txt
T="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o step --output-format csv -- $T > $O/prof.log 2>&1
python3 tools/step_from_trace.py $(find $O/prof -name '*kernel_trace.csv' | head -1) 2>&1 | grep -i "ts2\|launches," | head
rm -rf $O/prof
}

stage_n() {
O=gpurun_out/r06n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_dispatch.py -q -k "ts2 or TS2 or stride or conv_case or forward_case or bench_size" -p no:cacheprovider > $O/ts2_tests.txt 2>&1
tail -2 $O/ts2_tests.txt | cut -c1-300
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision fp16"
for rep in 1 2 3; do
  echo "fp16 pairs (no deferred K half) rep $rep: $(timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
  echo "fp16 C2W_TS2_PAIRS=0 rep $rep: $(C2W_TS2_PAIRS=0 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
done | tee $O/ab_ts2_pairs_fp16.txt
}

stage_o() {
O=gpurun_out/r06o
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/gpu_tests.txt 2>&1
tail -3 $O/gpu_tests.txt | cut -c1-300
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2 3; do
  echo "half8 rep $rep: $(timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
  echo "C2W_NO_HALF8=1 rep $rep: $(C2W_NO_HALF8=1 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
done | tee $O/ab_half8.txt
for rep in 1 2; do
  echo "half8 rep $rep: $(timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "C2W_NO_HALF8=1 rep $rep: $(C2W_NO_HALF8=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler_half8.txt
}

stage_r() {
# PMC passes on the 512 -> 512 @8x8 conv (B = 128: 256 workgroups, paired images) for the 4-wave and the 8-wave 8x16-tile kernel
O=gpurun_out/r06r
mkdir -p $O
groups=("SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA")
for m in half8 half4; do
  [ $m = half4 ] && export C2W_NO_HALF8=1 || unset C2W_NO_HALF8
  for i in 0 1 2 3 4; do
    timeout -s KILL 150 rocprofv3 --kernel-trace --pmc ${groups[$i]} --output-format csv -d $O/$m/g$i -- python3 tools/bench_kernels.py --batch 128 --dtypes bf16 --only 4 --kind conv --iters 3 --act 0 > $O/${m}_g$i.log 2>&1
    echo "$m group $i rc=$?"
  done
done
python3 - <<'PY' | tee gpurun_out/r06r/pmc_8x8_summary.txt
import csv, glob, collections
for m in ("half8", "half4"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r06r/{m}/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_patch_half" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = []
    for f in glob.glob(f"gpurun_out/r06r/{m}/g0/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_patch_half" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(m, "launches", len(dur), "mean us %.1f" % (sum(dur) / max(len(dur), 1)))
    for k, v in sorted(agg.items()):
        print(f"   {k:28s} {sum(v) / len(v):16.0f}")
PY
rm -rf $O/half8 $O/half4
}

stage_t() {
O=gpurun_out/r06t
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
for rep in 1 2; do
  for w in 256 512 1024 100000; do
    echo "B=128 C2W_HALF8_MAX_WGS=$w rep $rep: $(C2W_HALF8_MAX_WGS=$w timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'])")"
  done
done | tee $O/ab_half8_max_wgs_b128.txt
for rep in 1 2; do
  for w in 256 512 100000; do
    echo "B=64 C2W_HALF8_MAX_WGS=$w rep $rep: $(C2W_HALF8_MAX_WGS=$w timeout 300 python3 tools/bench_module_api.py --legs trainer_bf16_b64 --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; s=sys.stdin.read(); d=json.loads(s[:s.index(chr(10)+'}'+chr(10))+3])['trainer_bf16_b64']; print(d['ms_per_step'], d['step_ms']['median'], d.get('mfma_frac_whole_step'))")"
  done
done | tee $O/ab_half8_max_wgs_b64.txt
for w in 256 512 100000; do
  echo "sampler C2W_HALF8_MAX_WGS=$w: $(C2W_HALF8_MAX_WGS=$w timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_half8_max_wgs_sampler.txt
}

stage_flake() {
# flake watch: the full-size gradient parity test N times in fresh processes, and N times inside one process
O=gpurun_out/r06k
mkdir -p $O
fail=0
for i in $(seq 1 ${N:-24}); do
  timeout 300 python -m pytest tests/test_gpu_e2e.py -q -p no:cacheprovider > $O/e2e_$i.txt 2>&1 || { fail=$((fail+1)); cp $O/e2e_$i.txt $O/FAILED_e2e_$i.txt; }
  rm -f $O/e2e_$i.txt
done
echo "whole tests/test_gpu_e2e.py, fresh process each time: $fail failures in ${N:-24} runs" | tee $O/flake_summary.txt
ls gpurun_out/parity_fail* 2>/dev/null | tee -a $O/flake_summary.txt
cp gpurun_out/parity_fail* $O/ 2>/dev/null
}

stage_h8x() {
# ablation builds of the eight-wave 8x16-tile kernel's stage (lab/h8/make_variants.py; libraries under climate2weather_amd/build/alt/):
# which of weight LDS-DMA / fragment reads / MFMAs / barriers the ~1200 cycles of a stage are made of, and four other places for the DMA issue
O=gpurun_out/r06h8x
mkdir -p $O
L=""
for v in ${H8_VARIANTS:-0 1 2 4 8 3 5 6 7 11 16 32 64 128}; do L="$L climate2weather_amd/build/alt/libc2w_h8_$v.so"; done
for b in 128 16 38; do
  echo "== B=$b" >> $O/ab_h8_variants.txt
  B=$b SHAPES=4 ROUNDS=9 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants.txt 2>&1
done
echo "== B=128, 16x16 levels" >> $O/ab_h8_variants.txt
B=128 SHAPES=3,7 ROUNDS=7 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants.txt 2>&1
tail -80 $O/ab_h8_variants.txt
}

stage_h8y() {
# second lab A/B of the eight-wave kernel: weight pieces between the MFMA groups (p32), plus a second patch buffer at <= 256 workgroups (db)
O=gpurun_out/r06h8y
mkdir -p $O
L=""
for v in ${H8_VARIANTS:-0 p32 db}; do L="$L climate2weather_amd/build/alt/libc2w_h8_$v.so"; done
for b in 128 64 38 16; do
  echo "== B=$b" >> $O/ab_h8_variants2.txt
  B=$b SHAPES=4 ROUNDS=9 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants2.txt 2>&1
  B=$b SHAPE=1,8,1024,512 ROUNDS=9 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants2.txt 2>&1
done
echo "== B=128, 16x16 levels; B=38 16x16" >> $O/ab_h8_variants2.txt
B=128 SHAPES=3,7 ROUNDS=7 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants2.txt 2>&1
B=38 SHAPES=3 ROUNDS=7 timeout 600 python tools/ab_conv.py $L >> $O/ab_h8_variants2.txt 2>&1
grep -v amdgpu.ids $O/ab_h8_variants2.txt | tail -80
}

stage_v() {
# the eight-wave kernel with its weight pieces between the MFMA groups and the second patch buffer (<= 256 workgroups), in the product
# library: conv + dispatch tests, then same-call A/Bs against the previous kernels (C2W_LIB = the lab build of the old conv_patch.hip) and
# against one patch buffer (C2W_HALF8_DB=0): training step B = 128 / B = 64, one member at L = 49 / 121
O=gpurun_out/r06v
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_dispatch.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider > $O/gpu_tests_conv.txt 2>&1
tail -3 $O/gpu_tests_conv.txt | cut -c1-300
OLD=climate2weather_amd/build/alt/libc2w_h8_0.so
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["step_ms"]["median"], d["final_loss"])'
for bs in 128 64; do
  B="python3 bench.py --steps 20 --warmup 5 --batch $bs --no-cpu-baseline --no-extras"
  for rep in 1 2 3; do
    echo "B=$bs new rep $rep: $(timeout 300 $B 2>/dev/null | python3 -c "$J")"
    echo "B=$bs one patch buffer rep $rep: $(C2W_HALF8_DB=0 timeout 300 $B 2>/dev/null | python3 -c "$J")"
    echo "B=$bs old kernels rep $rep: $(C2W_LIB=$OLD timeout 300 $B 2>/dev/null | python3 -c "$J")"
  done
done | tee $O/ab_step.txt
for rep in 1 2 3; do
  echo "new rep $rep: $(timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "one patch buffer rep $rep: $(C2W_HALF8_DB=0 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "old kernels rep $rep: $(C2W_LIB=$OLD timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler.txt
}

stage_t3p() {
# lab A/B of the 16x16-tile kernel: the stage's weight piece behind the first 2 / 4 / 6 pixel columns (8 / 16 / 24 of 32 MFMAs) of its MFMA
# group instead of right behind the barrier (lab/h8/make_t3_variants.py)
O=gpurun_out/r06t3p
mkdir -p $O
L=""
for v in 0 2 4 6; do L="$L climate2weather_amd/build/alt/libc2w_t3p_$v.so"; done
for act in 0 1; do
  echo "== ACT=$act" >> $O/ab_t3_place.txt
  ACT=$act B=128 SHAPES=0,1,2 ROUNDS=7 timeout 900 python tools/ab_conv.py $L >> $O/ab_t3_place.txt 2>&1
done
echo "== B=37 (one member at L = 49)" >> $O/ab_t3_place.txt
ACT=1 B=37 SHAPES=0,1 ROUNDS=7 timeout 900 python tools/ab_conv.py $L >> $O/ab_t3_place.txt 2>&1
grep -v amdgpu.ids $O/ab_t3_place.txt | tail -60
}

stage_s2() {
# stride-2 forward on the parity planes of the halo patch: its tests + the conv parity suites, per-launch A/B against the gather kernel
# (C2W_CONV_S2_PATCH=0) on the four down-convs at B = 128 and B = 37, then the step and one member at L = 49 / 121
O=gpurun_out/r06s2
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_dispatch.py tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "stride2 or S2 or down or conv" > $O/gpu_tests_conv.txt 2>&1
tail -3 $O/gpu_tests_conv.txt | cut -c1-300
for b in 128 37; do
  for knob in 1 0; do
    echo "== B=$b C2W_CONV_S2_PATCH=$knob" >> $O/ab_s2_launches.txt
    C2W_CONV_S2_PATCH=$knob B=$b SHAPES=8,9,10,11 ROUNDS=7 timeout 600 python tools/ab_conv.py climate2weather_amd/libc2w_hip.so 2>&1 | grep -v amdgpu.ids >> $O/ab_s2_launches.txt
  done
done
cat $O/ab_s2_launches.txt
J='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["step_ms"]["median"], d["final_loss"])'
for bs in 128 64; do
  B="python3 bench.py --steps 20 --warmup 5 --batch $bs --no-cpu-baseline --no-extras"
  for rep in 1 2 3; do
    echo "B=$bs parity planes rep $rep: $(C2W_CONV_S2_PATCH=1 timeout 300 $B 2>/dev/null | python3 -c "$J")"
    echo "B=$bs gather kernel rep $rep: $(C2W_CONV_S2_PATCH=0 timeout 300 $B 2>/dev/null | python3 -c "$J")"
  done
done | tee $O/ab_step.txt
for rep in 1 2 3; do
  echo "parity planes rep $rep: $(C2W_CONV_S2_PATCH=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "gather kernel rep $rep: $(C2W_CONV_S2_PATCH=0 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler.txt
}

stage_s2p() {
# the stride-2 forward kernel's two-images-per-tile form (8-pixel-wide output: the 16x16 -> 8x8 down-conv): tests, per-launch A/B against the
# gather kernel at B = 128 / 37, one member at L = 49 / 121
O=gpurun_out/r06s2p
mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_bench_dispatch.py tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -x -p no:cacheprovider -k "stride2 or down or conv or full_size or bench_size" > $O/gpu_tests_conv.txt 2>&1
tail -3 $O/gpu_tests_conv.txt | cut -c1-300
for b in 128 37; do
  for knob in 1 0; do
    echo "== B=$b C2W_CONV_S2_PATCH=$knob" >> $O/ab_s2_pair.txt
    C2W_CONV_S2_PATCH=$knob B=$b SHAPES=11 ROUNDS=9 timeout 600 python tools/ab_conv.py climate2weather_amd/libc2w_hip.so 2>&1 | grep -v amdgpu.ids >> $O/ab_s2_pair.txt
  done
done
cat $O/ab_s2_pair.txt
for rep in 1 2 3; do
  echo "parity planes rep $rep: $(C2W_CONV_S2_PATCH=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "gather kernel rep $rep: $(C2W_CONV_S2_PATCH=0 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler.txt
}

stage_z() {
# last call of the round: the GPU suite at HEAD in both stream modes (Python-side changes after the closing evidence: the stream probe's
# per-device calibration), smoke() and the default bench line as the driver runs them
O=gpurun_out/r06z
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider > $O/gpu_tests_full.txt 2>&1; tail -4 $O/gpu_tests_full.txt > $O/gpu_tests_tail.txt; tail -1 $O/gpu_tests_tail.txt
C2W_WGRAD_STREAM=1 timeout 1500 python -m pytest tests -m gpu -q -rxX -p no:cacheprovider > $O/gpu_tests_two_streams_full.txt 2>&1; tail -4 $O/gpu_tests_two_streams_full.txt > $O/gpu_tests_tail_two_streams.txt; tail -1 $O/gpu_tests_tail_two_streams.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_line.json 2> $O/bench.err; tail -c 400 $O/bench_line.json
}

st=${1:?stage}; shift
"stage_$st" "$@"
