"""GPU idle time per training step from a rocprofv3 --kernel-trace CSV: wall time between optimizer launches minus the union of the
kernel intervals, the gaps by size and the kernels on both sides of the largest ones.

    python tools/trace_idle.py gpurun_out/prof/xyz_kernel_trace.csv [marker kernel substring]
"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:80]) for r in rows)
marker = sys.argv[2] if len(sys.argv) > 2 else "adamw_ema"  # a kernel that runs once per step
ends = [e for s, e, n in ev if marker in n]
if len(ends) >= 2:
    n = min(4, len(ends) - 1)
    t0, t1 = ends[-1 - n], ends[-1]
else:  # no per-step marker: the second half of the trace as one window
    n = 1
    t0, t1 = ev[len(ev) // 2][0], ev[-1][1]
sel = [(s, e, k) for s, e, k in ev if s >= t0 and e <= t1]
busy, (cs, ce, ck) = 0, sel[0]
gaps = []
for s, e, k in sel[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((s - ce, ck, k))
        cs, ce, ck = s, e, k
    elif e > ce:
        ce, ck = e, k
busy += ce - cs
print(f"last {n} steps: wall {(t1 - t0) / n / 1e6:.3f} ms/step, busy {busy / n / 1e6:.3f}, idle {(t1 - t0 - busy) / n / 1e6:.3f} in {len(gaps) / n:.1f} gaps/step")
hist = collections.Counter(min(int(g[0] / 1e3) // 4 * 4, 40) for g in gaps)
print("gap histogram (us bucket: count over the window):", sorted(hist.items()))
for g in sorted(gaps, reverse=True)[:8]:
    print(f"  {g[0] / 1e3:7.1f} us  {g[1][:60]}  ->  {g[2][:60]}")
