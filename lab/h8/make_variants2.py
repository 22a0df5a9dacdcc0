#!/usr/bin/env python3
"""Second set of lab builds of conv_patch_half8_kernel (round 6): the weight pieces issued between the two MFMA groups of a stage (variant
32 of make_variants.py: -11 % at 512 -> 512 @8x8) plus, with -DC2W_H8_DB=1, a SECOND patch buffer for launches of at most 256 workgroups
(one per CU anyway: 110,592 B of LDS): the next chunk's patch is issued behind stage 0's weight pieces and may stay in flight until the
wait of stage 3 (counted vmcnt(6) at stages 1 and 2), instead of one exposed vmcnt(0) + barrier per chunk.
    python lab/h8/make_variants2.py p32 db      -> climate2weather_amd/build/alt/libc2w_h8_{p32,db}.so
STATE: a record. The script patches the PRODUCT source by text; it applies to csrc/ as of commit d0ee3af (`git worktree add /tmp/w d0ee3af`, run it there).
The product kernel has since taken over variants 32 / db, so the text it looks for is gone from HEAD; the generated lab sources are not tracked."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(root, "climate2weather_amd/csrc/conv_patch.hip")).read()
k0 = src.index("void conv_patch_half8_kernel(const C2wConvArgs p)")
a = src.index("    auto issue_patch = [&](int chunk) {", k0)
b = src.index("    // fragment addresses as in conv_patch_ts2_pair", a)
src = src[:a] + r'''    const bool db = C2W_H8_DB && gridDim.x <= 256;  // the launcher gave a second patch buffer behind the weight ring
    auto issue_patch = [&](int chunk, uint32_t pbase) {
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(rx, smem + pbase + pdst[r], pvo[r], (uint32_t)chunk * 128u);
    };

''' + src[b:]
a = src.index("    issue_patch(c_lo);", k0)
src = src[:a] + "    issue_patch(c_lo, 0);" + src[a + len("    issue_patch(c_lo);"):]
a = src.index("    auto stage = [&](auto TAPc, int c) {", k0)
b = src.index("#pragma unroll 1", a)
NEW = r'''    auto wait_allow = [&](int n) {  // wave-uniform n in {0, 2, 4, 6}
        if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int c_hi = c_lo + nchunk;
    auto stage = [&](auto TAPc, int c) {
        constexpr int TAP = decltype(TAPc)::value;
        constexpr int KH = TAP / 3, KW = TAP % 3, WS = TAP % 3, T2 = (TAP + 2) % 9;
        const int s = (c - c_lo) * 9 + TAP;
        const bool pnext = db && c + 1 < c_hi;                      // this chunk prefetches the next one's patch (stage 0, behind the weights)
        const uint32_t pb = db && ((c - c_lo) & 1) ? (uint32_t)H_LDS : 0u;
        wait_allow(np + ((TAP == 1 || TAP == 2) && pnext ? 4 : 0));
        __builtin_amdgcn_s_barrier();
        if (!db && TAP == 0 && c > c_lo) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        np = 0;
        u32x4_t a0[4], b0[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA0 + WS * WBYTES + m * 2048);
#pragma unroll
        for (int n = 0; n < 2; ++n) b0[n] = *(const u32x4_t*)(smem + pb + preB0[KW] + n * (PW * 128) + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], db_[n], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < NS) {  // the weight pieces of stage s + 2: behind the first MFMA group, where their issue cost runs beside the matrix pipe
            issue_w(TAP + 2 >= 9 ? c + 1 : c, T2, (TAP + 2) % 3);
            np = 2;
        }
        if (TAP == 0 && pnext) issue_patch(c + 1, pb ? 0u : (uint32_t)H_LDS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + WS * WBYTES + m * 2048);
#pragma unroll
        for (int n = 0; n < 2; ++n) db_[n] = *(const u32x4_t*)(smem + pb + (preB0[KW] ^ 64u) + n * (PW * 128) + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // deferred fragments are in registers before their slot may be refilled
    };
'''
src = src[:a] + NEW + src[b:]
# the deferred pixel fragments are called db in the product kernel: rename (db is the double-buffer flag here)
k1 = src.index("    u32x4_t da[4] = {}, db[2] = {};", k0)
src = src[:k1] + "    u32x4_t da[4] = {}, db_[2] = {};" + src[k1 + len("    u32x4_t da[4] = {}, db[2] = {};"):]
k2 = src.index("        for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);", src.index("#pragma unroll 1", k0))
src = src[:k2] + "        for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], db_[n], acc[m][n]);" + src[k2 + len("        for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);"):]
old_attr = "hipFuncSetAttribute((const void*)conv_patch_half8_kernel<T, PAIR, SPLITK>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024)"
assert old_attr in src
src = src.replace(old_attr, old_attr.replace("100 * 1024", "112 * 1024"))
old_l = "conv_patch_half8_kernel<T, PAIR, SPLITK><<<nwg, H8_NTHR, H_LDS, st>>>(a);"
assert old_l in src
src = src.replace(old_l, "conv_patch_half8_kernel<T, PAIR, SPLITK><<<nwg, H8_NTHR, (C2W_H8_DB && nwg <= 256) ? H_LDS + H_PBYTES : H_LDS, st>>>(a);")
lab = os.path.join(root, "lab/csrc/conv_patch_h8b_lab.hip")
open(lab, "w").write("#ifndef C2W_H8_DB\n#define C2W_H8_DB 0\n#endif\n" + src)
pkg = os.path.join(root, "climate2weather_amd")
procs = []
FL = {"p32": ["-DC2W_H8_DB=0"], "db": ["-DC2W_H8_DB=1"]}
for v in sys.argv[1:]:
    d = os.path.join(pkg, "build/alt/h8_" + v)
    os.makedirs(d, exist_ok=True)
    procs.append((v, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + os.path.join(root, "include"),
                                       "-I" + os.path.join(pkg, "csrc")] + FL[v] + ["-Rpass-analysis=kernel-resource-usage", "-c", lab, "-o", d + "/conv_patch.o"],
                                      stderr=subprocess.PIPE, text=True)))
for v, pr in procs:
    err = pr.communicate()[1]
    if pr.returncode:
        print(err[-3000:]); sys.exit(1)
    lines = err.splitlines()
    for i, l in enumerate(lines):
        if "Function Name: " in l and "half8" in l:
            print(v, l.split("half8_kernel")[1][:12], " ".join(x.split("remark: ")[-1].split(" [-R")[0] for x in lines[i + 1:i + 9] if " VGPRs:" in x or "ScratchSize" in x))
    objs = [os.path.join(pkg, "build", f + ".o") for f in "conv_igemm conv_patch3 wgrad wgrad_patch pointwise attention attention_mfma sampler conv_center sources_digest".split()]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o",
                           os.path.join(pkg, "build/alt/libc2w_h8_%s.so" % v), os.path.join(pkg, "build/alt/h8_" + v, "conv_patch.o")] + objs)
    print("built libc2w_h8_%s.so" % v)
