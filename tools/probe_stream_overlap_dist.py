"""tools/probe_stream_overlap.py pattern B (spin kernels, side.wait_stream(compute) per round) at successive stages of a process that becomes
a one-rank RCCL trainer: where does the overlap go?"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch, torch.distributed as dist
from climate2weather_amd import streams
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cur = torch.cuda.current_stream()
US = 2350


def pattern(side, tag):
    def body():
        for _ in range(12):
            torch.cuda._sleep(500 * US)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                torch.cuda._sleep(200 * US)
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); body(); cur.wait_stream(side); e1.record(); torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1):.2f} ms (overlapped ~6.2, serial ~8.4); overtakes: {streams.overtakes(side)}", flush=True)


side0 = streams.independent_stream(dev)
pattern(side0, "1. clean process, independent_stream")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29543")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pattern(side0, "2. after init_process_group(nccl, device_id) -- same side stream")
t = torch.ones(1 << 20, device=dev)
dist.all_reduce(t); torch.cuda.synchronize()
pattern(side0, "3. after the first all_reduce -- same side stream")
pattern(streams.independent_stream(dev), "4. a NEW independent_stream made now")
w = dist.all_reduce(t, async_op=True); w.wait(); torch.cuda.synchronize()
with torch.cuda.stream(side0):
    w = dist.all_reduce(t, async_op=True); w.wait()
torch.cuda.synchronize()
pattern(side0, "5. after an all_reduce issued FROM the side stream")
from climate2weather_amd.score import ScoreUNet
from climate2weather_amd.training import Trainer
os.environ["C2W_FORCE_DIST"] = "1"
cfg = dict(embedding_dim=64, hidden_channels=[64, 64], hidden_blocks=[1, 1], attention_levels=[1], kernel_size=3, padding_mode="zeros")
net = ScoreUNet(channels=6, spatial=2, activation=torch.nn.SiLU, **cfg).to(dev)
tr = Trainer(net, precision="bf16", ema_rates=(), allreduce_dtype="bf16", bucket_mb=0.05)
pattern(side0, "6. after building a Trainer (broadcast of the weights)")
x = torch.randn(4, 6, 32, 32, device=dev)
for _ in range(3):
    tr.step(x)
torch.cuda.synchronize()
pattern(side0, "7. after three trainer steps -- first side stream")
pattern(tr.eng.side_stream(), "8. after three trainer steps -- the ENGINE's side stream")
dist.destroy_process_group()
