"""The ISA invariants live in the BUILD (climate2weather_amd/isa_checks.py, run by build.py on the objects it just compiled, before
linking): a library that violates one is never produced.  Here: the checkers see seeded violations, the build refuses a seeded
spill, and the product library on disk was checked (its build directory holds the evidence of the same compilation)."""
import os
import shutil
import subprocess

import pytest

from climate2weather_amd import build as c2w_build
from climate2weather_amd import isa_checks as ic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or ("/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else None)
needs_hipcc = pytest.mark.skipif(HIPCC is None, reason="needs hipcc")


def test_the_barrier_checker_sees_an_outstanding_read():
    asm = "\n".join(["_Zk:", " buffer_load_dwordx4 v1, s[0:3], 0 offen lds", " ds_read_b128 v[2:5], v9", " ds_read_b128 v[6:9], v9 offset:16",
                     " s_waitcnt lgkmcnt(1)", " s_barrier", " s_waitcnt lgkmcnt(0)", " s_barrier", "_Zplain:", " ds_read_b32 v1, v2", " s_barrier"])
    found, kernels, barriers = ic.lds_reads_outstanding_at_barriers(asm)
    assert kernels == 1 and barriers == 2 and len(found) == 1 and found[0][2] == ["ds_read_b128 v[6:9], v9 offset:16"]
    assert any("LDS read outstanding" in v for v in ic.violations("x.hip", "", asm))


def test_the_mfma_hazard_checker_sees_one():
    asm = "\n".join(["k:", " v_mfma_f32_16x16x32_bf16 v[4:7], v[0:3], v[8:11], 0", " v_add_f32 v20, v21, v22", " ;;#ASMSTART",
                     " v_cvt_pk_bf16_f32 v30, v5, v20", " ;;#ASMEND", " v_mfma_f32_16x16x32_bf16 v[12:15], v[0:3], v[8:11], 0",
                     " v_cvt_pk_bf16_f32 v31, v12, v13"])
    found, seen = ic.asm_conversions_fed_by_mfma(asm)
    assert seen == 1 and len(found) == 1 and found[0][1].startswith("v_cvt_pk_bf16_f32 v30")


def _remark(name, scratch=0, spill=0, vgprs=64):
    return (f"remark: f.hip:1:0: Function Name: {name} [-Rpass-analysis=kernel-resource-usage]\n"
            f"remark: f.hip:1:0:     VGPRs: {vgprs} [-R]\nremark: f.hip:1:0:     AGPRs: 0 [-R]\n"
            f"remark: f.hip:1:0:     ScratchSize [bytes/lane]: {scratch} [-R]\nremark: f.hip:1:0:     SGPRs Spill: 0 [-R]\n"
            f"remark: f.hip:1:0:     VGPRs Spill: {spill} [-R]\nremark: f.hip:1:0:     LDS Size [bytes/block]: 0 [-R]\n")


def test_the_scratch_checker_only_cares_about_hand_counted_waits():
    counted = "\n".join(["_Zcounted:", " ;;#ASMSTART", " s_waitcnt vmcnt(2)", " ;;#ASMEND", " s_endpgm"])
    compiler = "\n".join(["_Zplain:", " s_waitcnt vmcnt(2)", " ;;#ASMSTART", " s_waitcnt vmcnt(0)", " ;;#ASMEND", " s_endpgm"])
    assert ic.kernels_with_counted_vmcnt(counted + "\n" + compiler) == ["_Zcounted"]
    assert ic.scratch_in_counted_vmcnt_kernels(_remark("_Zcounted") + _remark("_Zplain", 16, 4), counted + "\n" + compiler) == []
    bad = ic.scratch_in_counted_vmcnt_kernels(_remark("_Zcounted", 8, 2), counted)
    assert len(bad) == 1 and "2 spilled VGPRs" in bad[0]
    assert ic.scratch_in_counted_vmcnt_kernels("", counted)  # a counted-wait kernel nobody reported on is a finding too
    assert ic.over_budget(_remark("_ZN1x20conv_patch_t3_kernelILi16EtLi8EEEv", vgprs=130)) and not ic.over_budget(_remark("_Zother", vgprs=500))


SEEDED = r"""
#include <hip/hip_runtime.h>
// 1024 threads per block leave 64 VGPRs per lane on gfx950: eighty live accumulators across a hand-counted wait must spill
extern "C" __global__ __launch_bounds__(1024) void seeded_spill_kernel(const float* __restrict__ a, float* __restrict__ out, int n) {
    float acc[%d];
#pragma unroll
    for (int i = 0; i < %d; ++i) acc[i] = a[threadIdx.x + i * 1024];
    asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    for (int k = 0; k < n; ++k) {
#pragma unroll
        for (int i = 0; i < %d; ++i) acc[i] = acc[i] * acc[(i + 1) %% %d] + (float)k;
    }
#pragma unroll
    for (int i = 0; i < %d; ++i) out[threadIdx.x + i * 1024] = acc[i];
}
"""


@needs_hipcc
def test_build_refuses_to_link_a_spilling_counted_vmcnt_kernel(tmp_path):
    """The same build() that makes the product library, pointed at a translation unit whose kernel waits with a hand-counted
    vmcnt AND spills: it must raise before linking and leave no library behind; the un-spilling twin builds."""
    csrc = tmp_path / "csrc"
    csrc.mkdir()
    lib = str(tmp_path / "libseeded.so")
    (csrc / "seeded.hip").write_text(SEEDED.replace("extern \"C\" __global__", "__global__") % ((8,) * 5))
    out = c2w_build.build(force=True, verbose=False, sources=["seeded.hip"], csrc=str(csrc), lib=lib, out_dir=str(tmp_path / "ok"))
    assert out == lib and os.path.exists(lib)
    os.remove(lib)
    (csrc / "seeded.hip").write_text(SEEDED.replace("extern \"C\" __global__", "__global__") % ((80,) * 5))
    with pytest.raises(c2w_build.IsaViolation, match="counted vmcnt waits next to scratch"):
        c2w_build.build(force=True, verbose=False, sources=["seeded.hip"], csrc=str(csrc), lib=lib, out_dir=str(tmp_path / "bad"))
    assert not os.path.exists(lib)


@needs_hipcc
def test_the_product_library_was_built_through_the_checks():
    """build.py keeps, next to every object, the device assembly and the resource-usage remarks of the compilation that produced it.
    Re-run the checkers on that evidence for the library on disk (the build already did, or there would be no library): every LDS-DMA
    kernel instantiation was looked at, none spills, none reaches a barrier with an LDS read outstanding, no asm conversion sits on
    a fresh MFMA result, and the 16x16-tile conv kernel keeps its 128-register occupancy."""
    bdir = os.path.join(ROOT, "climate2weather_amd", "build")
    have = all(os.path.exists(os.path.join(bdir, s.replace(".hip", ".tmp"), "resource_usage.txt")) for s in c2w_build.SOURCES)
    c2w_build.build(verbose=False, force=not have)  # compiles only if the library is stale (or was copied without its build directory)
    tot = dict(kernels=0, counted_vmcnt=0, lds_dma_kernels=0, barriers=0, asm_conversions=0)
    for src in c2w_build.SOURCES:
        stem = src.replace(".hip", "")
        tmp = os.path.join(bdir, stem + ".tmp")
        asm = open(os.path.join(tmp, f"{stem}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
        remarks = open(os.path.join(tmp, "resource_usage.txt")).read()
        assert ic.violations(src, remarks, asm) == []
        for k, v in ic.summary(remarks, asm).items():
            tot[k] += v
    assert tot["counted_vmcnt"] >= 40 and tot["lds_dma_kernels"] >= 40 and tot["barriers"] >= 300 and tot["asm_conversions"] >= 100, tot
