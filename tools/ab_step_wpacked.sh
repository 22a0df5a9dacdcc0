for i in 1 2 3; do
  C2W_NO_WPACKED=1 timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('plain ', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['final_loss'])"
  timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 12 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('packed', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['final_loss'])"
done
