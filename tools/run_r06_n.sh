#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bench_dispatch.py -q -k "ts2 or TS2 or stride or conv_case or forward_case or bench_size" -p no:cacheprovider > $O/ts2_tests.txt 2>&1
tail -2 $O/ts2_tests.txt | cut -c1-300
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision fp16"
for rep in 1 2 3; do
  echo "fp16 pairs (no deferred K half) rep $rep: $(timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
  echo "fp16 C2W_TS2_PAIRS=0 rep $rep: $(C2W_TS2_PAIRS=0 timeout 300 $B 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['step_ms']['median'], d['final_loss'])")"
done | tee $O/ab_ts2_pairs_fp16.txt
