#!/usr/bin/env python3
"""Ablation builds of conv_patch_half8_kernel's stage (round 6, after the closing evidence): where do the ~1200 cycles of a stage go when
its MFMA work is 512?  Writes lab/csrc/conv_patch_h8_lab.hip = the PRODUCT conv_patch.hip with the eight-wave kernel's stage body
replaced by one that takes -DC2W_H8_EXP=<bits>:
    1   no weight LDS-DMA inside the loop          (WRONG results: timing only)
    2   no fragment reads inside the loop          (WRONG results)
    4   no MFMAs inside the loop                   (WRONG results)
    8   no workgroup barriers inside the loop      (WRONG results / racy)
    16  the weight pieces issued behind the first 4 MFMAs of the stage instead of right behind the barrier   (correct)
    32  ... between the two MFMA groups (behind the deferred half's MFMAs, in front of the next deferred reads)  (correct)
    64  ... behind the first 4 MFMAs of the second group                                                        (correct)
    128 one piece behind the barrier, the other behind the first MFMA group                                     (correct)
and compiles + links one library per variant into climate2weather_amd/build/alt/libc2w_h8_<bits>.so (other objects: the product build).
    python lab/h8/make_variants.py 0 1 2 4 8 16 32 64 128
STATE: a record. The script patches the PRODUCT source by text; it applies to csrc/ as of commit d0ee3af (`git worktree add /tmp/w d0ee3af`, run it there).
The product kernel has since taken over variants 32 / db, so the text it looks for is gone from HEAD; the generated lab sources are not tracked."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "climate2weather_amd/csrc/conv_patch.hip")).read()
k0 = src.index("void conv_patch_half8_kernel(const C2wConvArgs p)")
a = src.index("    auto stage = [&](auto TAPc, int c) {", k0)
b = src.index("#pragma unroll 1", a)
NEW = r'''    auto stage = [&](auto TAPc, int c) {
        constexpr int TAP = decltype(TAPc)::value;
        constexpr int KH = TAP / 3, KW = TAP % 3, WS = TAP % 3, T2 = (TAP + 2) % 9;
        constexpr int X = C2W_H8_EXP;
        const int s = (c - c_lo) * 9 + TAP;
        wait_vm2(np);
        if constexpr (!(X & 8)) __builtin_amdgcn_s_barrier();
        if (TAP == 0 && c > c_lo) {
            issue_patch(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (!(X & 8)) __builtin_amdgcn_s_barrier();
        }
        np = 0;
        const bool ahead = s + 2 < NS;
        auto issue_piece = [&](int i) {
            if constexpr (!(X & 1)) {
                const uint32_t so = (uint32_t)(T2 * p.Cin + (TAP + 2 >= 9 ? c + 1 : c) * CK) * ESZ;
                glds16(rw, smem + H_PBYTES + ((TAP + 2) % 3) * WBYTES + wid * 1024 + i * 8192, wvo[i], so);
            }
        };
        auto issue_now = [&](int lo, int hi) {
            if (ahead) {
                __builtin_amdgcn_sched_barrier(0);
                for (int i = lo; i < hi; ++i) issue_piece(i);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!(X & 1)) np = 2;
            }
        };
        constexpr int PLACE = X & (16 | 32 | 64 | 128);
        if (PLACE == 0 || s == 0) issue_now(0, 2);
        else if (PLACE == 128) issue_now(0, 1);
        u32x4_t a0[4], b0[2];
        if constexpr (!(X & 2)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA0 + WS * WBYTES + m * 2048);
#pragma unroll
            for (int n = 0; n < 2; ++n) b0[n] = *(const u32x4_t*)(smem + preB0[KW] + n * (PW * 128) + KH * PROW);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[m] = da[m];
#pragma unroll
            for (int n = 0; n < 2; ++n) b0[n] = db[n];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (PLACE == 16 && m == 2) issue_now(0, 2);
                if constexpr (!(X & 4)) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);
                } else {
                    asm volatile("" ::"v"(da[m]), "v"(db[0]), "v"(db[1]));
                }
            }
            if (PLACE == 32) issue_now(0, 2);
            if (PLACE == 128) issue_now(1, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(X & 2)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + WS * WBYTES + m * 2048);
#pragma unroll
            for (int n = 0; n < 2; ++n) db[n] = *(const u32x4_t*)(smem + (preB0[KW] ^ 64u) + n * (PW * 128) + KH * PROW);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (PLACE == 64 && m == 2 && s > 0) issue_now(0, 2);
            if constexpr (!(X & 4)) {
#pragma unroll
                for (int n = 0; n < 2; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
            } else {
                asm volatile("" ::"v"(a0[m]), "v"(b0[0]), "v"(b0[1]));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
'''
out = src[:a] + NEW + src[b:]
os.makedirs(os.path.join(root, "lab/csrc"), exist_ok=True)
lab = os.path.join(root, "lab/csrc/conv_patch_h8_lab.hip")
open(lab, "w").write("#ifndef C2W_H8_EXP\n#define C2W_H8_EXP 0\n#endif\n" + out)
pkg = os.path.join(root, "climate2weather_amd")
procs = []
for v in sys.argv[1:]:
    d = os.path.join(pkg, "build/alt/h8_" + v)
    os.makedirs(d, exist_ok=True)
    procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(root, "include"),
                                       "-I" + os.path.join(pkg, "csrc"), "-DC2W_H8_EXP=" + v, "-Rpass-analysis=kernel-resource-usage", "-c", lab, "-o", d + "/conv_patch.o"],
                                      stderr=subprocess.PIPE, text=True)))
for v, pr in procs:
    err = pr.communicate()[1]
    if pr.returncode:
        print(err[-3000:]); sys.exit(1)
    lines = err.splitlines()
    for i, l in enumerate(lines):
        if "Function Name: " in l and "half8" in l and "ItLb1ELb0" in l:  # bf16 PAIR, no split
            print(v, " ".join(x.split("remark: ")[-1] for x in lines[i + 1:i + 9] if "VGPRs:" in x or "Spill" in x or "ScratchSize" in x))
    objs = [os.path.join(pkg, "build", f + ".o") for f in "conv_igemm conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest".split()]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o",
                           os.path.join(pkg, "build/alt/libc2w_h8_%s.so" % v), os.path.join(pkg, "build/alt/h8_" + v, "conv_patch.o")] + objs)
    print("built libc2w_h8_%s.so" % v)
