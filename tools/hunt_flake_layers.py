"""Localise the rare multi-stream mismatch: the same B = 128 batch through the default network (C = 52, bf16, inference) on four HIP
streams at once, many rounds; every launch's output is kept (Engine.debug_trace) and compared, launch by launch and image by image,
with a single-stream reference.  Prints the FIRST launch whose output differs."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd.score import ScoreUNet
dev = torch.device("cuda:0")
CFG = dict(embedding_dim=512, hidden_blocks=[3] * 5, hidden_channels=[128, 128, 256, 384, 512], kernel_size=3, padding_mode="zeros", attention_levels=[4])


def explain_stages(info, y2, r2, rows, cols, B):
    """Which of the kernel's stages (9 taps x 32-channel halves of the input) explains `got - ref` of the wrong block, cout by cout:
    stage s computed with the weights of stage s2 (same ring slot when (s - s2) % 3 == 0 in the kernel's stage order)."""
    g = info["g"]
    H, W, Cin = g["Hin"], g["Win"], g["Cin"]
    x = info["x"].view(B, H, W, -1)[..., :Cin]
    w = info["w"].reshape(-1)[:g["wrows"] * 9 * Cin].view(g["wrows"], 9, Cin).float()
    b = rows[0].item() // (H * W)
    pix = rows - b * H * W
    oh, ow = pix // W, pix % W
    xi = torch.zeros(H + 2, W + 2, Cin, device=x.device)
    xi[1:-1, 1:-1] = x[b].float()
    nh = Cin // 32
    # the kernel's stage order: s = hc * 9 + IDX, IDX = kw * 3 + kh  ->  tap = kh * 3 + kw, channels hc * 32 ..
    stages = [(hc, idx) for hc in range(nh) for idx in range(9)]
    xs = [xi[oh + (idx % 3), ow + (idx // 3), hc * 32:(hc + 1) * 32] for hc, idx in stages]
    d = (y2[rows][:, cols].float() - r2[rows][:, cols].float())
    for j, co in enumerate(cols.tolist()):
        dj = d[:, j]
        if dj.abs().max().item() == 0:
            continue
        ws = [w[co, (idx % 3) * 3 + idx // 3, hc * 32:(hc + 1) * 32] for hc, idx in stages]
        best = (1e9, None)
        for s in range(len(stages)):
            base = xs[s] @ ws[s]
            for s2 in range(len(stages)):
                if s2 == s:
                    r = (dj + base).norm().item()              # stage missing
                else:
                    r = (dj + base - xs[s] @ ws[s2]).norm().item()
                if r < best[0]:
                    best = (r, (s, s2))
        s, s2 = best[1]
        print(f"      cout {co}: |d| {dj.norm().item():.3e}; best: stage {s} {stages[s]} " + ("missing" if s == s2 else f"used the weights of stage {s2} {stages[s2]} (s2 - s = {s2 - s})")
              + f", residual {best[0]:.3e}")


ROUNDS, NS = int(os.environ.get("ROUNDS", "40")), int(os.environ.get("NSTREAMS", "4"))
torch.manual_seed(0)
net = ScoreUNet(channels=52, spatial=2, activation=torch.nn.SiLU, **CFG).to(dev).eval()
net.precision = os.environ.get("PRECISION", "bf16")
eng = net._get_engine()
x = torch.randn(128, 52, 128, 128, device=dev)
t = torch.tensor(0.7, device=dev)
with torch.no_grad():
    net(x, t)
    eng.debug_trace = []
    net(x, t)
    torch.cuda.synchronize()
    ref = [(e[0], e[1].clone()) for e in eng.debug_trace]
    streams = [torch.cuda.Stream() for _ in range(NS)]
    found = 0
    for rnd in range(ROUNDS):
        traces = []
        for st in streams:
            st.wait_stream(torch.cuda.current_stream())
        for st in streams:
            eng.debug_trace = []
            with torch.cuda.stream(st):
                net(x, t)
            traces.append(eng.debug_trace)
        torch.cuda.synchronize()
        for si, tr in enumerate(traces):
            for (n, y, *info), (nr, yr) in zip(tr, ref):
                if not torch.equal(y, yr):
                    B = 128
                    d = (y.float() - yr.float()).view(B, -1)
                    imgs = d.abs().amax(1).nonzero().flatten().tolist()
                    rows = (y.float() - yr.float()).abs().amax(1).nonzero().flatten()
                    print(f"round {rnd} stream {si}: FIRST differing launch '{n}' shape {tuple(y.shape)}: images {imgs[:8]}, {rows.numel()} pixel rows "
                          f"(first {rows[:6].tolist()}), max |d| {d.abs().max().item():.3e}", flush=True)
                    if os.environ.get("DUMP", "0") == "1":
                        y2, r2 = y.view(-1, y.shape[-1]), yr.view(-1, yr.shape[-1])
                        for rr in rows[:3].tolist() + rows[-1:].tolist():
                            cols = (y2[rr] != r2[rr]).nonzero().flatten().tolist()
                            print(f"   row {rr} (image {rr // (y2.shape[0] // B)}, pixel {rr % (y2.shape[0] // B)}): {len(cols)} channels differ, first {cols[:16]}")
                            for c in cols[:6]:
                                print(f"      ch {c}: got {y2[rr, c].item():+.6e} ({y2[rr, c].view(torch.int16).item() & 0xffff:04x})  "
                                      f"ref {r2[rr, c].item():+.6e} ({r2[rr, c].view(torch.int16).item() & 0xffff:04x})")
                        cols = (y2[rows] != r2[rows]).any(0).nonzero().flatten()
                        Wimg = int(round((y2.shape[0] // B) ** 0.5))
                        for o in [k * Wimg for k in range(-8, 9) if k] + [-2, -1, 1, 2]:
                            rs = (rows + o).clamp(0, y2.shape[0] - 1)
                            same = (y2[rows][:, cols] == r2[rs][:, cols]).float().mean().item()
                            if same > 0.5:
                                print(f"   {same * 100:.1f} % of the wrong block equals the REFERENCE output {o} pixel rows away ({o // Wimg if o % Wimg == 0 else o} image rows)")
                        for (n2, yo) in ref:
                            if yo.shape == y.shape and n2 != n:
                                same = (y2[rows][:, cols] == yo.view(-1, yo.shape[-1])[rows][:, cols]).float().mean().item()
                                if same > 0.5:
                                    print(f"   {same * 100:.1f} % of the wrong block equals the reference output of launch '{n2}' at the same place")
                        if info and info[0]["g"]["mode"] == 1 and info[0]["act"] == 0 and info[0]["g"]["Hin"] == info[0]["g"]["Hout"]:
                            explain_stages(info[0], y2, r2, rows, cols, B)
                        dr = rows[1:] - rows[:-1]
                        print(f"   row span {rows[0].item()}..{rows[-1].item()}, gaps {sorted(set(dr.tolist()))[:8]}; per-row differing channel counts "
                              f"{[(y2[q] != r2[q]).sum().item() for q in rows[:12].tolist()]}")
                    found += 1
                    break
    eng.debug_trace = None
print("mismatching forwards:", found, "of", ROUNDS * NS)
