#!/bin/bash
# round 6, fifth GPU call: split-K tests + the whole suite; does the communication stream overlap (probe + debug prints); defaults A/B; sampler with
# split-K; what the DDP bucket-view path launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
