// EXPERIMENT, NOT BUILT INTO libc2w_hip.so (measured slower; kept as the record of the design and as a starting point).
// Result on one MI355X (bf16, B = 128; same box, conv_patch_t3_kernel beside it): 128->128 @128^2 0.855 vs 0.615 ms, @64^2 0.220 vs
// 0.139 ms, 256->256 @32^2 0.164 vs 0.123 ms -- with every conv parity test green when forced.  The main loop compiles to what was
// intended (after a barrier: two LDS-DMA issues, then 64 MFMAs with the next stage's fragment reads between them; 256 AGPRs of
// accumulators + 256 VGPRs), but with ONE wave per SIMD nothing runs under that wave's own LDS-DMA issue (60-185 cycles per
// piece, during which the wave's instruction stream -- its MFMAs included -- is held), nor under the first patch's latency, the
// mid-tile patch reload and the four-pass epilogue: ~96 k cycles per 16x32 tile against 36.9 k cycles of MFMA issue.  At two
// waves per SIMD the partner wave covers exactly those holes, which is why the 256-register kernel wins.  A 512-register design
// needs loader waves that are not compute waves (and a register file that is not split evenly), i.e. another programming model.
// To build it again: add the file to build.py's SOURCES and a dispatch hook in conv_patch.hip::c2w_conv_patch_s1
// (c2w_conv_patch4_wanted / c2w_conv_patch4 below), and prefetch_rows32 stays in conv_epilogue.h.
//
// Fourth-generation halo-patch kernel for the 16-bit 3x3 stride-1 convolutions: ONE wave per SIMD with the full 512-entry
// register budget, 16x32-pixel tiles, every LDS fragment read issued one stage ahead of the MFMAs that consume it.
//
// Why (profiles/r01_experiments.md, ablations of conv_patch_t3_kernel): at two waves per SIMD a wave has 256 registers, the
// accumulator tile is 64 co x 128 px and every variant that adds live state (a second set of weight fragments, an early bias,
// another LDS-DMA placement) spills.  With that tile a wave issues 2 LDS-DMA pieces of weights per 32 MFMAs (21 % of the kernel
// time goes away when they are removed) and reads its weight fragments between the stage's barrier and its first MFMA.  Here:
//   * a workgroup (4 waves, one per SIMD, 512 VGPRs each) owns 16x32 pixels x 128 output channels; wave w owns tile rows
//     4w .. 4w+3: 128 co x 128 px = 256 accumulator registers; the weight stage (one tap x 32 channels x 128 co = 8 KiB) now
//     feeds 64 MFMAs per wave -- half the LDS-DMA pieces and 0.19 instead of 0.26 fragment reads per MFMA;
//   * the weight ring has 4 slots with three stages in flight; after a stage's barrier the wave issues the LDS-DMA of stage
//     s + 3 and then, BETWEEN the stage's MFMAs, reads stage s + 1's eight weight fragments into a second register set and the
//     pixel rows stage s + 1 needs (column-major stage order: one new row per tap, the next kernel column's rows into the
//     registers of rows that died) -- nothing is read between a barrier and the MFMAs that follow it;
//   * LDS: patch 18 x 36 pixels x 128 B = 81 KiB + ring 32 KiB = 113 KiB, one workgroup per CU (the registers allow no more).
// The price: nothing overlaps the prologue (first patch) and the epilogue of the CU's only workgroup.
// Layouts, swizzles, MFMA shape and the EpiStore epilogue are those of conv_patch_t3_kernel.
#include <cstdlib>

#include "conv_epilogue.h"

namespace {

constexpr int T4_NTHR = 256;
constexpr int T4_TR = 16, T4_TC = 32;                      // tile rows x columns
constexpr int T4_PW = 36;                                  // patch row pitch in pixels (34 used)
constexpr int T4_NPIECE = (T4_TR + 2) * T4_PW / 8;         // 81 LDS-DMA pieces of 8 pixels
constexpr int T4_PBYTES = T4_NPIECE * 1024;                // 82,944
constexpr int T4_ROUNDS = (T4_NPIECE + 3) / 4;             // 21 pieces per wave
constexpr int T4_WBYTES = 128 * 64;                        // one stage of weights: 128 co x 32 ci
constexpr int T4_NSLOT = 4;
constexpr int T4_OS = 128 * 2 + 16;                        // epilogue row stride
constexpr int T4_LDS_LOOP = T4_PBYTES + T4_NSLOT * T4_WBYTES;  // 115,712
constexpr int T4_LDS_EPI = 128 * T4_OS + 512;              // one pass of 128 output rows + LayerNorm column sums
constexpr int T4_LDS = T4_LDS_LOOP > T4_LDS_EPI ? T4_LDS_LOOP : T4_LDS_EPI;

template <int N> struct IC4 { static constexpr int value = N; };

__device__ __forceinline__ uint32_t t4_pswz(int col) { return (uint32_t)(col & 7); }
__device__ __forceinline__ uint32_t t4_wswz(int row) { return (uint32_t)((4 - ((row >> 2) & 3)) & 3); }

__device__ __forceinline__ void t4_wait(int outstanding) {
    if (outstanding >= 2) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <typename T>
__global__ __launch_bounds__(T4_NTHR, 1) void conv_patch_t4_kernel(const C2wConvArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int ESZ = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [patch | W0 | W1 | W2 | W3]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;

    const int nN = (p.Cout + 127) / 128;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * 128;
    const int H = p.Hin, W = p.Win;
    const int tw = W >> 5, tpi = (H >> 4) * tw;
    const int b = tm / tpi, tt = tm - b * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 4, ow0 = tx << 5;

    const size_t img_bytes = (size_t)H * W * p.Cin * ESZ;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)img_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    auto issue_patch = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < T4_ROUNDS; ++r) {
            int pc = r * 4 + wid;
            pc = pc < T4_NPIECE ? pc : T4_NPIECE - 1;
            const int f = pc * 8 + (lane >> 3);  // flattened patch pixel
            const int pr = f / T4_PW, px = f - pr * T4_PW;
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < T4_TC + 2 && pr < T4_TR + 2;
            const uint32_t cg = (uint32_t)(lane & 7) ^ t4_pswz(px);
            const uint32_t voff = ok ? (uint32_t)((ih * W + iw) * p.Cin) * ESZ + (cg << 4) : C2W_OOB;
            glds16(rx, smem + pc * 1024, voff, (uint32_t)chunk * 128u);
        }
    };
    // weight stage: 128 rows x 4 slots of 16 B = 2 pieces per wave; lane -> row = (round * 4 + wave) * 16 + lane / 4, slot = lane & 3
    uint32_t wvo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (i * 4 + wid) * 16 + (lane >> 2);
        const uint32_t cg = (uint32_t)(lane & 3) ^ t4_wswz(row);
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (cg << 4);
    }
    const int nchunk = p.Cin / 64;
    const int NS = nchunk * 18;
    // global stage st = chunk * 18 + i, i = half * 9 + kw * 3 + kh (kernel-column-major); ring slot st & 3
    auto issue_stage = [&](int st) {
        const int c2 = st / 18, i2 = st - c2 * 18;
        const int half = i2 / 9, tap = (i2 % 3) * 3 + (i2 % 9) / 3;
        const uint32_t so = (uint32_t)(tap * p.Cin + c2 * 64 + half * 32) * ESZ;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(rw, smem + T4_PBYTES + (st & 3) * T4_WBYTES + (i * 4 + wid) * 1024, wvo[i], so);
    };

    // fragment read offsets.  A[m]: weight row m*16 + li -> offA + m * 1024 (+ slot); B(row r, column half ch, tap column kw):
    // pixel (4 wid + r, li + kw + 16 ch) -> offB[kw] + r * pitch + ch * 16 pixels; k-half 1 flips slot bit 2
    const uint32_t offA = (uint32_t)(T4_PBYTES + li * 64 + (((uint32_t)lg ^ t4_wswz(li)) << 4));
    uint32_t offB[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        offB[kw] = (uint32_t)((wid * 4 * T4_PW + px) * 128 + (((uint32_t)lg ^ t4_pswz(px)) << 4));
    }
    auto rowp = [&](int kw, int half, int r, int ch) {
        return (const u32x4_t*)(smem + (offB[kw] ^ (half * 64)) + r * (T4_PW * 128) + ch * (16 * 128));
    };

    f32x4_t acc[8][8];  // [pixel tile n = 2 r + ch][co tile m]
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[n][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_patch(0);
    issue_stage(0);
    issue_stage(1);
    issue_stage(2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // patch and stage 0 landed; stages 1, 2 (two pieces each) may be in flight
    __builtin_amdgcn_s_barrier();
    u32x4_t a_cur[8], a_nxt[8], bq[6][2];
#pragma unroll
    for (int m = 0; m < 8; ++m) a_cur[m] = *(const u32x4_t*)(smem + offA + m * 1024);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ch = 0; ch < 2; ++ch) bq[r][ch] = *rowp(0, 0, r, ch);

    auto stage = [&](auto IDXc, int c) {
        constexpr int IDX = decltype(IDXc)::value;
        constexpr int HALF = IDX / 9, KW = (IDX % 9) / 3, KH = IDX % 3;
        constexpr int NXT = (IDX + 1) % 18, HALF_N = NXT / 9, KW_N = (NXT % 9) / 3;
        const int s = c * 18 + IDX;
        if (s + 1 < NS) {
            t4_wait(s + 2 < NS ? 2 : 0);   // my pieces of stage s + 1 have landed (in flight at most: s + 1, s + 2)
            __builtin_amdgcn_s_barrier();  // everyone's have; everyone has read stage s (during stage s - 1) and is done with slot (s - 1) & 3
        }
        if (IDX == 0 && c > 0) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            if (s + 3 < NS) issue_stage(s + 3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const uint32_t ac = offA + (uint32_t)(s & 3) * T4_WBYTES;
#pragma unroll
            for (int m = 0; m < 8; ++m) a_cur[m] = *(const u32x4_t*)(smem + ac + m * 1024);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) bq[r][ch] = *rowp(0, 0, r, ch);
        } else if (s + 3 < NS) {
            issue_stage(s + 3);
        }
        constexpr bool LASTI = IDX == 17;  // the next stage opens a chunk (or nothing): it reads its own fragments
        constexpr bool PREF = KH == 2 && !LASTI;  // next stage = first tap of the next kernel column, same patch chunk
        const uint32_t ao = offA + (uint32_t)((s + 1) & 3) * T4_WBYTES;
        if constexpr (KH < 2) {  // the one new pixel row of stage kh + 1
            bq[4 + KH][0] = *rowp(KW, HALF, 4 + KH, 0);
            bq[4 + KH][1] = *rowp(KW, HALF, 4 + KH, 1);
        }
        if constexpr (PREF) {  // rows 0 and 1 died with stage kh = 1: their registers take the next kernel column's rows 0, 1
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                bq[0][ch] = *rowp(KW_N, HALF_N, 0, ch);
                bq[1][ch] = *rowp(KW_N, HALF_N, 1, ch);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int n = 2 * r + ch;
                if constexpr (!LASTI) a_nxt[n] = *(const u32x4_t*)(smem + ao + n * 1024);  // next stage's weights: 8 reads under 64 MFMAs
#pragma unroll
                for (int m = 0; m < 8; ++m) acc[n][m] = mfma16<T>(a_cur[m], bq[r + KH][ch], acc[n][m]);
            }
            if constexpr (PREF) {  // row r + 2 of this column was used last by the MFMAs above: its registers take the next column's row
                if (r + 2 < 4) {
                    bq[r + 2][0] = *rowp(KW_N, HALF_N, r + 2, 0);
                    bq[r + 2][1] = *rowp(KW_N, HALF_N, r + 2, 1);
                }
            }
        }
        if constexpr (!LASTI) {
#pragma unroll
            for (int m = 0; m < 8; ++m) a_cur[m] = a_nxt[m];
        }
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        stage(IC4<0>{}, c); stage(IC4<1>{}, c); stage(IC4<2>{}, c); stage(IC4<3>{}, c); stage(IC4<4>{}, c); stage(IC4<5>{}, c);
        stage(IC4<6>{}, c); stage(IC4<7>{}, c); stage(IC4<8>{}, c); stage(IC4<9>{}, c); stage(IC4<10>{}, c); stage(IC4<11>{}, c);
        stage(IC4<12>{}, c); stage(IC4<13>{}, c); stage(IC4<14>{}, c); stage(IC4<15>{}, c); stage(IC4<16>{}, c); stage(IC4<17>{}, c);
    }

    // ---- epilogue: four passes; pass j stages tile rows {j, 4 + j, 8 + j, 12 + j} (row j of every wave, 32 pixels x 128 co each)
    // as LDS rows R = 32 wave + column, then one EpiStore pass over the 128 rows (bias / SiLU / pair / multiplier / residual /
    // fused LayerNorm forward and backward: conv_epilogue.h).
    float bv[8][4];
    {
        const bool has = p.bias != nullptr;
        const float* bp = has ? p.bias : (const float*)p.w;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + m * 16 + lg * 4 + r;
                const int idx = (has && co < p.wrows) ? co : 0;
                const float v = has ? bp[idx] : 0.f;
                bv[m][r] = (has && co < p.wrows) ? v : 0.f;
            }
    }
    __syncthreads();
    char* const O = smem;
    float* const red = (float*)(smem + 128 * T4_OS);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (p.ln_x != nullptr && tid < 128) red[tid] = 0.f;
        EpiStore<T, 128, T4_NTHR> est;
        est.prefetch_rows32(p, tid, co0, ((long long)b * H + oh0 + j) * W + ow0, W);
        auto stage_out = [&](auto SILUc) {  // compile-time activation: a run-time select would evaluate exp/rcp for every conv
            constexpr bool SILU = decltype(SILUc)::value != 0;
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                const int row = wid * 32 + ch * 16 + li;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = acc[2 * j + ch][m][r] + bv[m][r];
                        if constexpr (SILU) v[r] = silu_f(v[r]);
                    }
                    *(u32x2_t*)(O + row * T4_OS + (m * 16 + lg * 4) * 2) = (u32x2_t){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
                }
            }
        };
        if (p.act == C2W_ACT_SILU) stage_out(IC4<1>{});  // wave-uniform branch
        else stage_out(IC4<0>{});
        __syncthreads();
        if (p.ln_x != nullptr) est.finish_ln(p, O, T4_OS, tid, b, red);
        else if (p.lnf_y != nullptr) est.finish_lnf(p, O, T4_OS, tid, b);
        else est.finish(p, O, T4_OS, tid);
        if (j + 1 < 4) __syncthreads();
    }
}

template <typename T>
int t4_launch(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_t4_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, T4_LDS));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hin >> 4) * (a.Win >> 5);
    conv_patch_t4_kernel<T><<<nM * nN, T4_NTHR, T4_LDS, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

// Opt-in while it is being measured (C2W_CONV_T4=1; =2 forces it wherever the image is tiled by 16x32).
bool c2w_conv_patch4_wanted(const C2wConvArgs& a, int dtype) {
    static const int mode = getenv("C2W_CONV_T4") ? atoi(getenv("C2W_CONV_T4")) : 0;
    if (mode == 0 || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16) || (a.Hin & 15) != 0 || (a.Win & 31) != 0) return false;
    const long long wgs = (long long)a.B * (a.Hin >> 4) * (a.Win >> 5) * ((a.Cout + 127) / 128);
    return mode == 2 || wgs >= 512;
}

int c2w_conv_patch4(const C2wConvArgs& a, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? t4_launch<f16_t>(a, st) : t4_launch<bf16_t>(a, st);
}
