"""Invariants of the shipped gfx950 ISA that the C++ sources cannot express -- checked by build.py on the objects it has just
compiled (hipcc's resource-usage remarks + the device assembly of the SAME compilation, -save-temps), BEFORE it links them: a
library that violates one is never produced.  tests/test_kernel_resources.py exercises the checkers on seeded violations.

1. no scratch in a kernel that counts its own loads.  The LDS-DMA kernels wait with hand-counted `s_waitcnt vmcnt(N)` ("all but my
   N youngest loads have landed").  A register spill adds scratch loads/stores to the same counter and scratch (flat-family)
   accesses return out of order with buffer loads -- seen on the GPU as wrong weight rows at chunk boundaries when conv_patch3 once
   spilled 40 VGPRs.  Rule: a kernel with a counted (N > 0) vmcnt wait inside an inline-asm block uses no scratch at all.
2. no LDS read outstanding at a barrier of an LDS-DMA kernel.  The rings refill a slot right behind the barrier that follows its
   last read; a fragment read still queued when its wave arrives there can be overtaken by the refill (round 3: one wrong forward in
   ~200 under four streams, hipcc had sunk the first use of a ds_read_b128 below the barrier).
3. no inline-asm conversion on a fresh MFMA result.  The hazard recognizer pads MFMA -> VALU reads of the compiler's own
   instructions, not of asm statements (common.h: pack_bf16x2); seen once as NaNs out of the attention kernels.
4. register budgets.  The two-workgroups-per-CU design of the 16x16-tile conv kernel needs its 8-wave instantiations at <= 128
   VGPRs (four waves per SIMD); above that the second workgroup no longer fits and the epilogue is uncovered -- a performance
   cliff, not an error, but one a toolchain bump can cause silently.
"""
from __future__ import annotations

import re
from typing import Dict, List, Tuple

# (regex on the demangled-ish mangled name, max VGPRs): instantiations that must keep their occupancy
VGPR_BUDGETS = [(re.compile(r"conv_patch_t3_kernel"), 128)]


def parse_remarks(log: str) -> Dict[str, dict]:
    """kernel (mangled name) -> dict(scratch, spill, vgprs, sgprs, lds) out of -Rpass-analysis=kernel-resource-usage output."""
    out = {}
    for blk in re.split(r"Function Name: ", log)[1:]:
        name = blk.split()[0]

        def num(pat, blk=blk):
            m = re.search(pat, blk)
            return int(m.group(1)) if m else 0
        out[name] = dict(scratch=num(r"ScratchSize \[bytes/lane\]: (\d+)"), spill=num(r"VGPRs Spill: (\d+)"), sgpr_spill=num(r"SGPRs Spill: (\d+)"),
                         vgprs=num(r"\sVGPRs: (\d+)"), agprs=num(r"AGPRs: (\d+)"), lds=num(r"LDS Size \[bytes/block\]: (\d+)"))
    return out


def kernel_bodies(asm: str) -> List[Tuple[str, List[str]]]:
    """[(mangled kernel name, raw lines)] of every function in a device assembly file."""
    lines = asm.splitlines()
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l)] + [len(lines)]
    return [(lines[starts[k]].split(":")[0], lines[starts[k]:starts[k + 1]]) for k in range(len(starts) - 1)]


def kernels_with_counted_vmcnt(asm: str) -> List[str]:
    """Kernels that contain `s_waitcnt vmcnt(N)`, N > 0, INSIDE an inline-asm block: hand-counted waits."""
    found = []
    for name, body in kernel_bodies(asm):
        in_asm = False
        for raw in body:
            t = raw.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
            elif t.startswith(";;#ASMEND"):
                in_asm = False
            elif in_asm:
                m = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", t)
                if m and int(m.group(1)) > 0:
                    found.append(name)
                    break
    return found


def scratch_in_counted_vmcnt_kernels(remarks: str, asm: str) -> List[str]:
    res = parse_remarks(remarks)
    bad = []
    for name in kernels_with_counted_vmcnt(asm):
        r = res.get(name)
        if r is None:
            bad.append(f"{name}: no resource-usage remark for a kernel with counted vmcnt waits")
        elif r["scratch"] or r["spill"]:
            bad.append(f"{name}: counted vmcnt waits next to scratch ({r['scratch']} B/lane, {r['spill']} spilled VGPRs)")
    return bad


def lds_reads_outstanding_at_barriers(asm: str):
    """[(kernel, line, [reads])] for every s_barrier of every kernel that issues LDS-DMA at which LDS reads are still outstanding
    (the wave's LDS queue is followed in layout order: ds_* and s_load push, `s_waitcnt lgkmcnt(N)` retires all but the N youngest)."""
    found, kernels, barriers = [], 0, 0
    for name, raw in kernel_bodies(asm):
        body = [l.split(";")[0].strip() for l in raw]
        if not any(re.match(r"(buffer_load|global_load_lds).*\blds\b|global_load_lds", l) for l in body):
            continue
        kernels += 1
        q = []
        for i, l in enumerate(body):
            if re.match(r"ds_(read|load)", l):
                q.append(l)
            elif l.startswith(("ds_", "s_load", "s_buffer_load")):
                q.append("")  # counts on lgkmcnt, not a read of staged data
            elif l.startswith("s_waitcnt"):
                m = re.search(r"lgkmcnt\((\d+)\)", l)
                if m:
                    q = q[len(q) - int(m.group(1)):] if int(m.group(1)) else []
            elif l.startswith("s_barrier"):
                barriers += 1
                if any(q):
                    found.append((name, i, [x for x in q if x]))
    return found, kernels, barriers


def asm_conversions_fed_by_mfma(asm: str, lookback: int = 24):
    """[(line, conversion, mfma)]: a v_cvt_pk_* inside an inline-asm block (;;#ASMSTART ... ;;#ASMEND) one of whose source registers
    is the destination of a v_mfma among the `lookback` instructions in front of it.  The compiler pads MFMA -> VALU reads of its own
    instructions with wait states; it does not look into asm statements."""
    ins, in_asm = [], False
    for ln, raw in enumerate(asm.splitlines()):
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        t = t.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((ln, t, in_asm))
    found, seen = [], 0
    for i, (ln, t, in_asm) in enumerate(ins):
        m = re.match(r"v_cvt_pk_\w+ v(\d+), v(\d+), v(\d+)", t)
        if not (in_asm and m):
            continue
        seen += 1
        # reads of an MFMA's destination (result not written yet) and writes into any of its operand ranges (still being read)
        dst, srcs = int(m.group(1)), {int(m.group(2)), int(m.group(3))}
        for ln2, t2, _ in ins[max(0, i - lookback):i]:
            if not t2.startswith("v_mfma"):
                continue
            rng = [(int(a), int(b)) for a, b in re.findall(r"v\[(\d+):(\d+)\]", t2)]
            if rng and (any(rng[0][0] <= r <= rng[0][1] for r in srcs) or any(a <= dst <= b for a, b in rng)):
                found.append((ln, t, t2))
    return found, seen


def over_budget(remarks: str) -> List[str]:
    bad = []
    for name, r in parse_remarks(remarks).items():
        for pat, cap in VGPR_BUDGETS:
            if pat.search(name) and r["vgprs"] > cap:
                bad.append(f"{name}: {r['vgprs']} VGPRs > budget {cap} (occupancy the launch plan relies on)")
    return bad


def violations(src: str, remarks: str, asm: str) -> List[str]:
    """Every violated invariant of one translation unit, as printable strings (empty list: the object may be linked)."""
    out = [f"{src}: {v}" for v in scratch_in_counted_vmcnt_kernels(remarks, asm)]
    found, _, _ = lds_reads_outstanding_at_barriers(asm)
    out += [f"{src}: {n[-64:]}: LDS read outstanding at the s_barrier on line {i}: {r[:2]}" for n, i, r in found]
    conv, _ = asm_conversions_fed_by_mfma(asm)
    out += [f"{src}: line {ln}: asm conversion `{t}` reads/overwrites registers of `{t2}` still in flight" for ln, t, t2 in conv]
    out += [f"{src}: {v}" for v in over_budget(remarks)]
    return out


def summary(remarks: str, asm: str) -> dict:
    """Counts for the build log: how much each checker looked at."""
    _, lk, lb = lds_reads_outstanding_at_barriers(asm)
    _, cv = asm_conversions_fed_by_mfma(asm)
    return dict(kernels=len(parse_remarks(remarks)), counted_vmcnt=len(kernels_with_counted_vmcnt(asm)), lds_dma_kernels=lk, barriers=lb,
                asm_conversions=cv)
