"""BASELINE configs[3] as the reference runs it -- the conditioned sliding-window sampler at the shipped trajectory lengths
(bench.py::sampler_configs3), stand-alone so that one leg can be profiled:

    python tools/bench_sampler_configs3.py --lengths 8737 --corrections 0 --steps 2 --members 1
    rocprofv3 --kernel-trace --stats ... -- python3 tools/bench_sampler_configs3.py --lengths 8737 --corrections 0 --steps 2 --members 1
"""
import argparse, json, os, sys
sys.path.insert(0, os.getcwd())
import torch
import bench

p = argparse.ArgumentParser()
p.add_argument("--lengths", default="49,121,8737")
p.add_argument("--corrections", default="0,2")
p.add_argument("--steps", type=int, default=3)
p.add_argument("--members", type=int, default=8)
p.add_argument("--precision", default="bf16")
a = p.parse_args()
res = bench.sampler_configs3(torch.device("cuda:0"), a.precision, tuple(int(v) for v in a.lengths.split(",")),
                             tuple(int(v) for v in a.corrections.split(",")), a.steps, a.members, log=lambda r: print(json.dumps(r), flush=True))
print(json.dumps(res))
