#!/bin/bash
# Same-box A/B of the training step under an environment knob: alternates baseline / variant runs of bench.py's headline region
# (boxes differ by +-1.5 % in the clock they hold, so only same-box pairs are compared).
#   usage: bash tools/ab_env.sh C2W_NO_DG_PREFETCH=1 [rounds] [steps]
VAR=$1; ROUNDS=${2:-3}; STEPS=${3:-20}
run() { env "$@" python bench.py --no-cpu-baseline --no-extras --steps $STEPS --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  median %.3f  min %.3f  dominant %.4f ms' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['roofline']['avg_launch_ms']))"; }
for i in $(seq $ROUNDS); do
  echo -n "baseline     : "; run C2W_AB_DUMMY=1
  echo -n "$VAR : "; run $VAR
done
