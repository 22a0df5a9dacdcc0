#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_dispatch.py -q -k "split_k" -p no:cacheprovider > $O/splitk_tests.txt 2>&1
tail -3 $O/splitk_tests.txt | cut -c1-300
for rep in 1 2 3; do
  echo "default rep $rep: $(timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
  echo "C2W_NO_SPLITK=1 rep $rep: $(C2W_NO_SPLITK=1 timeout 300 python3 tools/bench_sampler.py --lengths 49,121 --steps 32 2>&1 | grep window-forwards | tr '\n' '|')"
done | tee $O/ab_sampler_splitk.txt
bash tools/run_r06_h.sh
cp gpurun_out/r06h/sampler_l49_step_table_splitk_*.txt $O/
timeout 600 python -m pytest tests/test_gpu_host.py -q -k "sampler or ensemble or score or guidance or window" -p no:cacheprovider > $O/sampler_tests.txt 2>&1; tail -2 $O/sampler_tests.txt
