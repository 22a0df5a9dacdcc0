#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
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
