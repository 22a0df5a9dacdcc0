"""One conv launch (3x3 stride 1, 128 -> 128 @128x128, B = 128, bf16) on several HIP streams at once, many rounds, every output compared
with a single-stream reference.  On a mismatch the difference is explained stage by stage: the kernel accumulates 36 stages (9 taps x
4 halves of 32 input channels); the tool tests whether `got - ref` is one stage's contribution missing, or one stage computed with
ANOTHER stage's weights (a ring slot read before its refill landed / after it was overwritten).  Environment: C2W_LIB (the library under
test), ROUNDS, NSTREAMS, ACT (0 none / 1 SiLU), REPS (launches per stream per round)."""
import os, sys, math
sys.path.insert(0, os.getcwd())
import torch
from climate2weather_amd import ops

dev = torch.device("cuda:0")
ROUNDS, NS, ACT, REPS = (int(os.environ.get(k, d)) for k, d in (("ROUNDS", "200"), ("NSTREAMS", "4"), ("ACT", "0"), ("REPS", "4")))
B, H, C = 128, 128, 128
torch.manual_seed(0)
x = torch.randn(B * H * H, C, device=dev).to(torch.bfloat16)
w = (torch.randn(C, 9, C, device=dev) / math.sqrt(9 * C)).to(torch.bfloat16)
bias = torch.randn(C, device=dev) * 0.1
g = dict(B=B, Hin=H, Win=H, Cin=C, Hout=H, Wout=H, Cout=C, ldy=C, wrows=C, mode=ops.CONV_S1)
act = ops.ACT_SILU if ACT else ops.ACT_NONE


def run(y):
    ops.conv(x, w, bias, y, g, ops.DTYPE_BF16, act=act)


def explain(got, ref):
    d = (got.float() - ref.float())
    rows = d.abs().amax(1).nonzero().flatten()
    cols = d[rows].abs().amax(0).nonzero().flatten()
    print(f"   {rows.numel()} pixel rows {rows[0].item()}..{rows[-1].item()}, channels {cols.tolist()}")
    b = rows[0].item() // (H * H)
    pix = rows - b * H * H
    oh, ow = pix // H, pix % H
    print(f"   image {b}, tile rows {sorted(set((oh % 16).tolist()))} cols {sorted(set((ow % 16).tolist()))}, tile origin ({(oh[0] // 16 * 16).item()}, {(ow[0] // 16 * 16).item()})")
    if ACT:
        return
    xi = torch.zeros(H + 2, H + 2, C, device=dev)
    xi[1:-1, 1:-1] = x.view(B, H, H, C)[b].float()
    wf = w.float()
    dd = d[rows][:, cols]                                                     # (P, K)
    xs, ws = [], []
    for t in range(9):
        kh, kw = t // 3, t % 3
        for h in range(4):
            xs.append(xi[oh + kh, ow + kw, h * 32:(h + 1) * 32])                # (P, 32)
            ws.append(wf[cols, t, h * 32:(h + 1) * 32])                          # (K, 32)
    contrib = [xs[s] @ ws[s].t() for s in range(36)]
    nd = dd.norm().item()
    best = []
    for s in range(36):
        best.append(((dd + contrib[s]).norm().item() / nd, f"stage (tap {s // 4}, half {s % 4}) missing"))
        for s2 in range(36):
            if s2 != s:
                best.append(((dd + contrib[s] - xs[s] @ ws[s2].t()).norm().item() / nd, f"stage (tap {s // 4}, half {s % 4}) used the weights of (tap {s2 // 4}, half {s2 % 4})"))
                best.append(((dd + contrib[s] - xs[s2] @ ws[s].t()).norm().item() / nd, f"stage (tap {s // 4}, half {s % 4}) used the pixels of (tap {s2 // 4}, half {s2 % 4})"))
    best.sort()
    for r, what in best[:4]:
        print(f"   residual {r:.4f} of |d| if {what}")


ref = torch.empty_like(x)
run(ref); run(ref)
torch.cuda.synchronize()
ys = [torch.empty_like(x) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
found = 0
for rnd in range(ROUNDS):
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for rep in range(REPS):
        for st, y in zip(streams, ys):
            with torch.cuda.stream(st):
                run(y)
    torch.cuda.synchronize()
    for si, y in enumerate(ys):
        if not torch.equal(y, ref):
            found += 1
            print(f"round {rnd} stream {si}: output differs", flush=True)
            if found <= 6:
                explain(y, ref)
print(f"mismatching outputs: {found} of {ROUNDS * NS} checked ({ROUNDS * NS * REPS} launches)")
