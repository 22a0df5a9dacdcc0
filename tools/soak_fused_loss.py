"""Full-size network, bf16 and fp16, N optimizer steps on the synthetic feed, three ways on the same seeds: the default (loss tail fused into the
output conv, the step's noise = the Philox stream rounded to half precision and generated once), C2W_NO_LOSS_FUSION=1 (fp32 stream
regenerated twice, rounds 1-5) and -- opt-in -- the chain form of the residual blocks.  Loss every 25 steps: do the curves agree?"""
import os, subprocess, sys
sys.path.insert(0, os.getcwd())
N = int(os.environ.get("STEPS", "150"))
if len(sys.argv) > 1:
    import torch
    from climate2weather_amd.data import DeviceWindowFeed, SyntheticWindowDataset
    from climate2weather_amd.pipelines import SDAPipeline
    from climate2weather_amd.score import ScoreUNet
    from climate2weather_amd.training import Trainer
    prec = sys.argv[1]
    dev = torch.device("cuda:0")
    CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])
    torch.manual_seed(0)
    net = ScoreUNet(channels=65, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev)
    tr = Trainer(net, SDAPipeline(), lr=2e-4, precision=prec, ema_rates=[0.9999], growth_interval=50, seed=7)
    feed = DeviceWindowFeed(SyntheticWindowDataset(n_frames=76, n_vars=5, height=128, width=128, window=13, seed=0), dev, seed=0)
    losses, out = [], []
    for s in range(N):
        losses.append(tr.step(feed.next_batch(64)))
        if (s + 1) % 25 == 0:
            out.append(f"{float(torch.stack(losses[-25:]).mean()):.4f}")
    print(" ".join(out), "| steps taken", tr.optimizer_steps_taken(), "finite", bool(torch.isfinite(tr.eng.flat).all()), flush=True)
    sys.exit(0)
for prec in ("bf16", "fp16"):
    for tag, env in (("fused loss tail (default)", {}), ("C2W_NO_LOSS_FUSION=1", {"C2W_NO_LOSS_FUSION": "1"}), ("C2W_LN_CHAIN=1", {"C2W_LN_CHAIN": "1"}),
                     ("C2W_CONV_S2_PATCH=0 C2W_NO_HALF8=1 C2W_TS2_PAIRS=0 (round 5's kernel selection)", {"C2W_CONV_S2_PATCH": "0", "C2W_NO_HALF8": "1", "C2W_TS2_PAIRS": "0"})):
        r = subprocess.run([sys.executable, __file__, prec], env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "steps taken" in l]
        print(f"{prec:5s} {tag:28s}: mean loss per 25 steps {line[-1] if line else 'FAILED ' + r.stderr[-300:]}", flush=True)
