"""bench.py::module_api legs one after the other in one process, with the allocator's figures after each."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
p = argparse.ArgumentParser()
p.add_argument("--legs", default="bf16_autocast,fp16_autocast_gradscaler,trainer_bf16")
p.add_argument("--empty", type=int, default=0)
a0 = p.parse_args()
a = argparse.Namespace(steps=20, warmup=3, batch=128, vars=5, markov_order=6, size=128)
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
for leg in a0.legs.split(","):
    r = bench.module_api(dev, a, 1.0, legs=(leg,))
    v = r[leg]
    print(leg, v["windows_per_s"], "median", v["step_ms"]["median"], "reserved %.1f GB allocated %.1f GB" % (torch.cuda.memory_reserved() / 2**30, torch.cuda.memory_allocated() / 2**30), flush=True)
    if a0.empty:
        import gc; gc.collect(); torch.cuda.empty_cache()
