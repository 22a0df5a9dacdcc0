"""Per-kernel statistics from a rocprofv3 --kernel-trace results.db (sqlite), plus -- for one kernel and grid size -- the
launches split into those that had the chip to themselves and those that overlapped a kernel of another queue/stream
(the weight-gradient stream of engine.grad_stream).

    python tools/rocprof_db_stats.py gpurun_out/prof_r01j/r01j_results.db [--kernel conv_patch_t3 --grid 2097152] > profiles/....csv
"""
import argparse
import sqlite3

p = argparse.ArgumentParser()
p.add_argument("db")
p.add_argument("--kernel", default="conv_patch_t3")
p.add_argument("--grid", type=int, default=8192 * 256, help="grid_x (threads) of the launches to split: 8192 tiles x 256 threads")
a = p.parse_args()
cur = sqlite3.connect(a.db).cursor()
rows = cur.execute("select name, start, end, grid_x, queue_id from kernels order by start").fetchall()
agg = {}
for name, s, e, gx, q in rows:
    d = agg.setdefault(name, [0, 0])
    d[0] += 1
    d[1] += e - s
total = sum(v[1] for v in agg.values())
print("name,calls,total_ms,avg_us,pct")
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{name}\",{n},{t / 1e6:.3f},{t / n / 1e3:.1f},{100.0 * t / total:.2f}")
sel = [(s, e, q) for name, s, e, gx, q in rows if a.kernel in name and gx == a.grid]
if sel:
    alone, shared = [], []
    for s, e, q in sel:
        ov = any(q2 != q and s2 < e and e2 > s for _, s2, e2, _, q2 in rows)
        (shared if ov else alone).append((e - s) / 1e3)
    print(f"# {a.kernel} grid_x={a.grid}: {len(sel)} launches")
    for label, v in (("exclusive (no kernel of another queue in flight)", alone), ("overlapping another queue's kernel", shared)):
        if v:
            print(f"#   {label}: {len(v)} launches, avg {sum(v) / len(v):.1f} us, min {min(v):.1f}, max {max(v):.1f}")
    allv = alone + shared
    print(f"#   all: avg {sum(allv) / len(allv):.1f} us")
