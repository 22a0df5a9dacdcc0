#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
