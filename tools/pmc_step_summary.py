#!/usr/bin/env python3
"""Summarise tools/pmc_step.sh: per (kernel, grid size) mean counter values per launch + mean duration, for the kernels matching a
substring; FETCH_SIZE is doubled (gfx950: the counter tallies 128-B requests at 64 B, MI355X_MICROARCH.md "HBM") and both sizes are
converted from KB to bytes.  Writes JSON to stdout.
    python tools/pmc_step_summary.py gpurun_out/pmc_step_r03 conv_patch_t3 wgrad_patch"""
import collections, csv, glob, json, sys
root, pats = sys.argv[1], sys.argv[2:]


def short(name):
    n = name.replace("void (anonymous namespace)::", "")
    return n.split("(C2wConvArgs")[0].split("((anonymous")[0]


out = collections.defaultdict(dict)
for g in ("fetch", "write", "mfma"):
    for f in sorted(glob.glob(f"{root}/{g}/**/*counter_collection.csv", recursive=True)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            key = (short(name), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            agg[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (key, cname), v in agg.items():
            out[key][cname] = sum(v) / len(v)
            out[key]["launches_" + g] = len(v)
    for f in sorted(glob.glob(f"{root}/{g}/**/*kernel_trace.csv", recursive=True))[:1]:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if not any(p in name for p in pats):
                continue
            wg = int(r["Workgroup_Size"]) if "Workgroup_Size" in r else int(r.get("Workgroup_Size_X", 1))
            grid = int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"])
            key = (short(name), grid // max(wg, 1))
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for key, v in dur.items():
            out[key]["duration_us_" + g] = sum(v) / len(v) / 1e3
res = []
for (name, wgs), d in sorted(out.items(), key=lambda kv: -kv[1].get("duration_us_fetch", 0) * kv[1].get("launches_fetch", 0)):
    e = dict(kernel=name, workgroups=wgs, **{k: round(v, 3) for k, v in d.items()})
    if "FETCH_SIZE" in d:
        e["hbm_read_bytes"] = round(2 * d["FETCH_SIZE"] * 1024)
    if "WRITE_SIZE" in d:
        e["hbm_write_bytes"] = round(d["WRITE_SIZE"] * 1024)
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        e["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE"] / 8), 4)
        e["clock_ghz"] = round(d["GRBM_GUI_ACTIVE"] / 8 / (d["duration_us_mfma"] * 1e3), 3)
    res.append(e)


def forward_dominant(g, counter):
    """mean counter value over the FORWARD launches of the dominant layers: conv_patch_t3 launches of 8192 workgroups (128 -> 128
    @128^2 at B = 128, the padded network-input / output convs included: same geometry) between an optimizer kernel (step start) and the loss
    kernel (end of the forward pass)"""
    vals, per_kernel = [], collections.defaultdict(list)
    for f in sorted(glob.glob(f"{root}/{g}/**/*counter_collection.csv", recursive=True)):
        rows = sorted((r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter), key=lambda r: int(r["Dispatch_Id"]))
        fwd = False
        for r in rows:
            n = r["Kernel_Name"]
            if "adamw_ema_kernel" in n:
                fwd = True
            elif "mse_loss_grad" in n:
                fwd = False
            elif fwd and "conv_patch_t3" in n and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) == 8192:
                vals.append(float(r["Counter_Value"]))
                per_kernel[short(n)].append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals) if vals else None), len(vals), {k: round(sum(v) / len(v), 1) for k, v in per_kernel.items()}


fr, nf, kf = forward_dominant("fetch", "FETCH_SIZE")
wr, nw, kw_ = forward_dominant("write", "WRITE_SIZE")
summary = dict(kernels=res)
try:  # provenance: the digest of the sources the profiled library was built from (bench.py refuses the figure for another build)
    sys.path.insert(0, ".")
    from climate2weather_amd import build as _b
    summary["c2w_sources_sha256"] = _b.embedded_digest()
except Exception:  # pragma: no cover
    summary["c2w_sources_sha256"] = None
if fr is not None and wr is not None:
    summary["forward_dominant_launches"] = dict(
        note="conv_patch_t3 launches of 8192 workgroups inside the forward pass of the profiled steps (14 per step: 12 residual-block convs + the padded network-input / output convs); FETCH_SIZE x 2 (gfx950) + WRITE_SIZE, KB -> bytes",
        launches=nf, hbm_read_bytes=round(2 * fr * 1024), hbm_write_bytes=round(wr * 1024), hbm_bytes=round(2 * fr * 1024 + wr * 1024),
        fetch_kb_by_kernel=kf, write_kb_by_kernel=kw_)
print(json.dumps(summary, indent=1))
