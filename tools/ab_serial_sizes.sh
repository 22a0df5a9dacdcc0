run() { env "$@" python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3 $ARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms/step  median %.3f  min %.3f  %.1f windows/s' % (d['ms_per_step'], d['step_ms']['median'], d['step_ms']['min'], d['value']))"; }
for ARGS in "--batch 64" "--batch 32" "--batch 16" "--vars 4" "--precision fp16" "--batch 32 --size 256 --precision fp16"; do
  for i in 1 2; do
    echo -n "[$ARGS] two streams : "; run C2W_WGRAD_STREAM=1
    echo -n "[$ARGS] one stream  : "; run C2W_WGRAD_STREAM=0
  done
done
