// Halo-patch implicit GEMM for the 3x3 stride-1 convolutions (the 60 res-block convs and their input gradients:
// 98.8 of the 116 GFLOP forward, model/nn.py:155,157) on gfx950 -- the 8x16-pixel-tile kernels: fp32 mode, 16-bit launches below
// 512 workgroups of the 16x16-tile kernel (conv_patch3.hip; C2W_CONV_T3_MIN_WGS), 8-pixel-wide images (two per tile), the stride-2 input
// gradient per output-parity class (conv_patch_ts2_*) and the stride-2 forward on the parity planes of its patch (conv_patch_s2_kernel).
// Four-wave form conv_patch_half_kernel (rounds 1-5; fp32 above 256 workgroups), eight-wave form conv_patch_half8_kernel (round 6).
//
// Why a halo patch: PMC on the gather kernel (profiles/r01_pmc_conv_gather_b32.md) shows ~290 non-MFMA instructions per 32 MFMAs
// per wave -- per-tap gather address arithmetic and scalar loop overhead -- so the wave's in-order issue stream, not the matrix
// pipe / LDS / HBM, set the time.  Here
//   * a workgroup owns an 8x16 output tile of one image and stages the (8+2)x(16+2) input halo patch ONCE per K-chunk
//     (128 B of channels per pixel); all 9 taps read it from LDS -> each input pixel is fetched once, not 9 times;
//   * the 9 taps are unrolled: the patch row pitch is 24 pixels (a multiple of 8), so the XOR swizzle of a fragment
//     read depends only on (lane, kw) and every LDS address is  register + immediate  -- no per-stage VALU;
//   * per stage (one tap of one chunk) a wave issues its weight pieces by LDS-DMA one tap ahead, counted vmcnt, one raw s_barrier;
//   * weights ring: 3 slots of [128 co][128 B]; slot = tap % 3 is a compile-time constant.
// MFMA tiling: 128 px x 128 co per workgroup, 4 waves x (4x4) 16x16 tiles, A = weights, B = pixels.
// (The 16x16-tile, one-workgroup-per-CU form of this kernel -- 156 KB of LDS -- measured 5 % behind two half-tile workgroups per CU
// and was removed in round 3; the cycle-stamp / ablation builds live in lab/csrc/conv_patch_lab.hip.)
#include <cstdlib>
#include <type_traits>

#include "conv_epilogue.h"
#include "knobs.h"

namespace {

constexpr int PW = 24;                    // patch row pitch in pixels (18 used)
constexpr int PROW = PW * 128;            // bytes per patch row
constexpr int WBYTES = 128 * 128;         // one tap's weight tile: 128 co x 128 B

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) { c = mfma16<bf16_t>(a, b, c); }
};
template <> struct Mma<f16_t> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) { c = mfma16<f16_t>(a, b, c); }
};
template <> struct Mma<float> {
    __device__ static __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]), c, 0, 0, 0);
    }
};

template <int N> struct IC { static constexpr int value = N; };

// ------------------------------------------------------------------------------------------------------------------
// Two-workgroups-per-CU variant.  Cycle stamps (profiles/r01_stamps_conv_patch.md) show a 256-pixel tile spending 40-48 %
// of its cycles in its prologue / epilogue (plus the gap between workgroups), and the 156-KiB kernel above admits one
// workgroup per CU, so nothing runs meanwhile.  This variant is the same pipeline on HALF the tile -- 8x16 output pixels,
// 4 waves (2 over channels x 2 over pixels, each still a 64x64 MFMA sub-tile) -- with one patch buffer and the 3-slot
// weight ring: 30,720 + 49,152 = 79,872 B, so TWO workgroups share a CU (8 waves, 2 per SIMD, independent barriers):
// one's prologue, chunk-boundary patch load and epilogue overlap the other's MFMAs.
constexpr int H_NTHR = 256;
constexpr int H_NPIECE = 10 * 3;              // (8+2) patch rows x 3 pieces
constexpr int H_PBYTES = H_NPIECE * 1024;     // 30,720
constexpr int H_LDS = H_PBYTES + 3 * WBYTES;  // 79,872

__device__ __forceinline__ void wait_vm4(int n) {  // wave-uniform n in {0, 4}
    if (n == 4) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

// PAIR: 8-pixel-wide images (the 8x8 bottleneck level): the 8x16 tile is TWO images side by side (b, b + 1), each with its own
// zero halo -- patch columns 0..9 belong to image b, 10..19 to image b + 1; tile column c >= 8 reads patch column c + 2 + kw.
// Replaces the per-tap gather kernel at that level (585 -> ~1000 TFLOP/s class); the fused LayerNorm epilogues (one image per
// tile) are not available in this mode.
// SPLITK (round 6; C2wConvArgs.splitk / splitk_ws): launches that leave most of the chip idle -- the deep levels of a sampler step on a
// short trajectory: 512 -> 512 @8x8 at 37 windows is 76 workgroups walking 72 stages each -- split the K chunks over `splitk`
// workgroups per tile; each stores its fp32 accumulators as a partial tile, conv_splitk_epilogue_kernel adds them in a fixed order
// and applies bias / activation / multiplier / residual (the chain a launch costs is its stage count whatever it carries:
// profiles/r04_experiments.md section 15).  A second launch, not an in-kernel fix-up: results stay bit-reproducible.
template <typename T, bool PAIR = false, bool SPLITK = false>
__global__ __launch_bounds__(H_NTHR, 2) void conv_patch_half_kernel(const C2wConvArgs p) {
    constexpr int ESZ = sizeof(T);
    constexpr int CK = 128 / ESZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [patch | W0 | W1 | W2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;

    const int nN = (p.Cout + 127) / 128;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nsplit = SPLITK ? p.splitk : 1;
    const int sp = SPLITK ? L % nsplit : 0;  // the splits of a tile are neighbours in the launch order
    const int Lt = SPLITK ? L / nsplit : L;
    const int tn = Lt % nN, tm = Lt / nN;
    const int co0 = tn * 128;
    // H x W = the grid the tiles and the taps live on (= the output); with C2W_CONV_UP it is the nearest-neighbour x2 upsampling of the
    // Hs x Ws source map, which is never materialised: patch pixel (ih, iw) is fetched from source pixel (ih >> 1, iw >> 1)
    // (model/nn.py:184-189; the zero padding is that of the upsampled map).
    const bool up = !PAIR && p.mode == C2W_CONV_UP;
    const int H = p.Hout, W = p.Wout, Ws = p.Win;
    const int tw = PAIR ? 1 : W >> 4, tpi = (H >> 3) * tw;
    const int b = PAIR ? 2 * (tm / tpi) : tm / tpi, tt = tm - (tm / tpi) * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;

    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESZ;
    const int nimg = PAIR ? (b + 1 < p.B ? 2 : 1) : 1;  // images under the descriptor: a missing partner reads as zeros (out of range)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)(img_bytes * nimg));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    // patch pieces: 30 pieces over 4 waves = 8 rounds (pieces past the end repeat the last one)
    uint32_t pvo[8];
    int pdst[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int pc = r * 4 + wid;
        pc = pc < H_NPIECE ? pc : H_NPIECE - 1;
        const int pr = pc / 3, pg = pc - pr * 3;
        const int px = pg * 8 + (lane >> 3);
        const int pimg = PAIR && px >= 10 ? 1 : 0;  // PAIR: patch columns 10..19 = image b + 1
        const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px - 10 * pimg;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < (PAIR ? 20 : 18);
        const uint32_t lc = (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));
        const int spix = up ? (ih >> 1) * Ws + (iw >> 1) : ih * Ws + iw;
        pvo[r] = ok ? (uint32_t)(spix * p.Cin) * ESZ + (uint32_t)pimg * (uint32_t)img_bytes + (lc << 4) : C2W_OOB;
        pdst[r] = pc * 1024;
    }
    uint32_t wvo[4];  // weight tile: 128 rows x 8 chunks = 4 rounds of 256 threads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (uint32_t)(((tid & 7) ^ (row & 7)) << 4);
    }
    auto issue_w = [&](int chunk, int tap, int wslot) {
        const uint32_t so = (uint32_t)(tap * p.Cin + chunk * CK) * ESZ;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rw, smem + H_PBYTES + wslot * WBYTES + wid * 1024 + i * 4096, wvo[i], so);
    };
    auto issue_patch = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < 8; ++r) glds16(rx, smem + pdst[r], pvo[r], (uint32_t)chunk * 128u);
    };

    uint32_t offA[2][4], preB[2][3][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = wm * 64 + m * 16 + li;
            offA[ks][m] = (uint32_t)(H_PBYTES + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int px = li + kw + (PAIR && li >= 8 ? 2 : 0);
                preB[ks][kw][n] = (uint32_t)(((wn * 4 + n) * PW + px) * 128 + (((ks * 4 + lg) ^ (px & 7)) << 4));
            }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nchunk_all = p.Cin / CK;
    const int c_lo = SPLITK ? sp * nchunk_all / nsplit : 0;  // near-equal chunk ranges (8 chunks over 3 workgroups: 2, 3, 3)
    const int nchunk = SPLITK ? (sp + 1) * nchunk_all / nsplit - c_lo : nchunk_all;
    const int NS = nchunk * 9;

    issue_patch(c_lo);
    issue_w(c_lo, 0, 0);
    issue_w(c_lo, 1, 1);
    float bv[4][4];
    if constexpr (!SPLITK) epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);
    int np = 4;  // LDS-DMA pieces of the previous stage that may still be in flight
    u32x4_t da[4] = {}, db[4] = {};

    auto stage = [&](auto TAPc, int c) {
        constexpr int TAP = decltype(TAPc)::value;
        constexpr int KH = TAP / 3, KW = TAP % 3, WS = TAP % 3, T2 = (TAP + 2) % 9;
        const int s = (c - c_lo) * 9 + TAP;
        wait_vm4(np);
        __builtin_amdgcn_s_barrier();
        if (TAP == 0 && c > c_lo) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        np = 0;
        if (s + 2 < NS) {
            issue_w(TAP + 2 >= 9 ? c + 1 : c, T2, (TAP + 2) % 3);
            np = 4;
        }
        u32x4_t a0[4], b0[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA[0][m] + WS * WBYTES);
#pragma unroll
        for (int n = 0; n < 4; ++n) b0[n] = *(const u32x4_t*)(smem + preB[0][KW][n] + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + offA[1][m] + WS * WBYTES);
#pragma unroll
        for (int n = 0; n < 4; ++n) db[n] = *(const u32x4_t*)(smem + preB[1][KW][n] + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // deferred fragments are in registers before their slot may be refilled
    };
#pragma unroll 1
    for (int c = c_lo; c < c_lo + nchunk; ++c) {
        stage(IC<0>{}, c); stage(IC<1>{}, c); stage(IC<2>{}, c); stage(IC<3>{}, c); stage(IC<4>{}, c);
        stage(IC<5>{}, c); stage(IC<6>{}, c); stage(IC<7>{}, c); stage(IC<8>{}, c);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);

    if constexpr (SPLITK) {  // partial tile [tile row R = 16 * row + column][128 co] fp32, straight from the accumulators (64 KB per workgroup)
        float* const dst = p.splitk_ws + ((size_t)sp * (gridDim.x / nsplit) + Lt) * (128 * 128);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                *(f32x4_t*)(dst + (wn * 64 + n * 16 + li) * 128 + wm * 64 + m * 16 + lg * 4) = acc[m][n];
        return;
    }

    constexpr int OS = 128 * ESZ + 16;
    EpiStore<T, 128, H_NTHR> est;
    const bool pool2 = !PAIR && (p.flags & C2W_CONV_POOL2) != 0;
    if constexpr (PAIR) est.prefetch_pair8(p, tid, co0, ((long long)b * H + oh0) * W, H * W, nimg);
    else if (!pool2) est.prefetch_tile16(p, tid, co0, ((long long)b * H + oh0) * W + ow0, W);
    __syncthreads();
    char* const O = smem;
    float* const red = (float*)(smem + 128 * OS);  // 128 floats behind O: column sums of the fused LN backward
    if constexpr (ESZ == 2) {
        if (p.ln_x != nullptr && tid < 128) red[tid] = 0.f;
    }
    epi_acc_to_lds<T>(O, OS, acc, bv, p.act, wm * 64, wn * 64, li, lg);
    __syncthreads();
    if (pool2) {
        est.finish_pool2(p, O, OS, tid, co0, ((long long)b * (H >> 1) + (oh0 >> 1)) * (W >> 1) + (ow0 >> 1), W >> 1);
    } else if constexpr (ESZ == 2 && !PAIR) {
        if (p.ln_x != nullptr) est.finish_ln(p, O, OS, tid, b, red);
        else if (p.lnf_y != nullptr) est.finish_lnf(p, O, OS, tid, b);
        else est.finish(p, O, OS, tid);
    } else {
        est.finish(p, O, OS, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// EIGHT waves on the same 8x16-pixel x 128-channel tile (round 6): for launches of at most one workgroup per CU.  There the 4-wave
// kernel leaves ONE wave on every SIMD, and a lone wave issues its LDS-DMA, its fragment reads, its barrier waits and its 32 MFMAs per
// stage one after the other: 0.55 us per stage where the matrix work is 0.25 (512 -> 512 @8x8 at B = 128: 256 workgroups, 0.31 of peak;
// the deep levels of a sampler step on a short trajectory: 76-228).  Eight waves = 2 (channel halves) x 4 (pixel-row pairs), wave tile
// 64 co x 32 px = 32 accumulator registers, two waves per SIMD: one's waits are the other's MFMAs.  Same LDS plan, same weight ring
// (two 1-KiB pieces per wave and stage: counted vmcnt), same patch, same epilogues (EpiStore<T, 128, 512>).
//
// What a stage is made of (ablation builds, profiles/r06u_ab_h8_stage_ablation.txt, 512 -> 512 @8x8, B = 128: 39.5 us; launch + prologue +
// epilogue + the eight exposed patch loads 11): MFMAs + barriers alone 27.3 us, fragment reads + barriers alone 27.7, weight LDS-DMA +
// barriers alone 27.1 -- three chains of about the same length, 16 us each, that overlapped to 28.  Two changes came out of it:
//   * the stage's two weight pieces are issued BETWEEN its two MFMA groups, not right behind the barrier: an LDS-DMA piece takes the wave
//     ~100 issue cycles, and behind the barrier both waves of a SIMD pay them at the same time with the matrix pipe idle; behind the first
//     group they run beside its eight MFMAs.  -12 ... -15 % at the 8x8 level, -3 ... -5 % at 16x16 (r06v_ab_h8_p32_db.txt; results
//     bit-identical; in the middle of a group or behind the second one it is worth half of that or nothing);
//   * DB (launches of at most 256 workgroups: one per CU whatever the LDS size): a SECOND patch buffer behind the weight ring (110,592 B).
//     The next chunk's patch is issued behind stage 0's weight pieces and may stay in flight until the wait of stage 3 (vmcnt(6) at stages
//     1 and 2: the pieces older than it are the ones those waits are for), instead of one exposed vmcnt(0) + barrier per chunk: a further
//     -5 ... -7 %.  (Round 6's earlier attempt at this -- two patch buffers AND a five-slot weight ring, ten loads in flight -- was 6 us
//     slower than the plain kernel; section 10 of profiles/r06_experiments.md.)
constexpr int H8_NTHR = 512;
constexpr int H8_LDS_DB = H_LDS + H_PBYTES;  // 110,592
__device__ __forceinline__ void wait_vm_h8(int n) {  // wave-uniform n in {0, 2, 4, 6}
    if (n == 6) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (n == 4) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else if (n == 2) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <typename T, bool PAIR = false, bool SPLITK = false, bool DB = false>
__global__ __launch_bounds__(H8_NTHR, DB ? 1 : 2) void conv_patch_half8_kernel(const C2wConvArgs p) {
    constexpr int ESZ = sizeof(T);
    constexpr int CK = 128 / ESZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [patch | W0 | W1 | W2 | DB: second patch]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;  // wn: pixel rows 2 wn, 2 wn + 1 of the tile

    const int nN = (p.Cout + 127) / 128;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int nsplit = SPLITK ? p.splitk : 1;
    const int sp = SPLITK ? L % nsplit : 0;
    const int Lt = SPLITK ? L / nsplit : L;
    const int tn = Lt % nN, tm = Lt / nN;
    const int co0 = tn * 128;
    const bool up = !PAIR && p.mode == C2W_CONV_UP;
    const int H = p.Hout, W = p.Wout, Ws = p.Win;
    const int tw = PAIR ? 1 : W >> 4, tpi = (H >> 3) * tw;
    const int b = PAIR ? 2 * (tm / tpi) : tm / tpi, tt = tm - (tm / tpi) * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;

    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESZ;
    const int nimg = PAIR ? (b + 1 < p.B ? 2 : 1) : 1;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)(img_bytes * nimg));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    const int nchunk_all = p.Cin / CK;
    const int c_lo = SPLITK ? sp * nchunk_all / nsplit : 0;
    const int nchunk = SPLITK ? (sp + 1) * nchunk_all / nsplit - c_lo : nchunk_all;
    const int NS = nchunk * 9;

    // patch pieces: 30 pieces over 8 waves = 4 rounds (pieces past the end repeat the last one)
    uint32_t pvo[4];
    int pdst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int pc = r * 8 + wid;
        pc = pc < H_NPIECE ? pc : H_NPIECE - 1;
        const int pr = pc / 3, pg = pc - pr * 3;
        const int px = pg * 8 + (lane >> 3);
        const int pimg = PAIR && px >= 10 ? 1 : 0;
        const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px - 10 * pimg;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < (PAIR ? 20 : 18);
        const uint32_t lc = (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));
        const int spix = up ? (ih >> 1) * Ws + (iw >> 1) : ih * Ws + iw;
        pvo[r] = ok ? (uint32_t)(spix * p.Cin) * ESZ + (uint32_t)pimg * (uint32_t)img_bytes + (lc << 4) : C2W_OOB;
        pdst[r] = pc * 1024;
    }
    uint32_t wvo[2];  // weight tile: 128 rows x 8 chunks = 2 rounds of 512 threads
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 64 * i;
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (uint32_t)(((tid & 7) ^ (row & 7)) << 4);
    }
    auto issue_w = [&](int chunk, int tap, int wslot) {
        const uint32_t so = (uint32_t)(tap * p.Cin + chunk * CK) * ESZ;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(rw, smem + H_PBYTES + wslot * WBYTES + wid * 1024 + i * 8192, wvo[i], so);
    };
    auto issue_patch = [&](int chunk, uint32_t pbase) {
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(rx, smem + pbase + pdst[r], pvo[r], (uint32_t)chunk * 128u);
    };

    // fragment addresses as in conv_patch_ts2_pair: A row m = + m * 2048, patch row n = + n * PW * 128 (immediates), second K half = ^ 64
    const uint32_t offA0 = (uint32_t)(H_PBYTES + (wm * 64 + li) * 128 + ((lg ^ (li & 7)) << 4));
    uint32_t preB0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw + (PAIR && li >= 8 ? 2 : 0);
        preB0[kw] = (uint32_t)((wn * 2 * PW + px) * 128 + ((lg ^ (px & 7)) << 4));
    }

    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    issue_patch(c_lo, 0);
    issue_w(c_lo, 0, 0);
    issue_w(c_lo, 1, 1);
    int np = 2;  // LDS-DMA pieces of the next stage that may still be in flight
    u32x4_t da[4] = {}, dq[2] = {};
    const int c_hi = c_lo + nchunk;

    auto stage = [&](auto TAPc, int c) {
        constexpr int TAP = decltype(TAPc)::value;
        constexpr int KH = TAP / 3, KW = TAP % 3, WS = TAP % 3, T2 = (TAP + 2) % 9;
        const int s = (c - c_lo) * 9 + TAP;
        const bool pnext = DB && c + 1 < c_hi;  // this chunk fetches the next one's patch into the other buffer (stage 0, behind the weights)
        const uint32_t pb = DB && ((c - c_lo) & 1) ? (uint32_t)H_LDS : 0u;
        // the weights of this stage were issued two stages ago; younger than them: the previous stage's pieces (np) and, at stages 1 and 2
        // of a chunk, the four patch pieces issued behind stage 0's
        wait_vm_h8(np + ((TAP == 1 || TAP == 2) && pnext ? 4 : 0));
        __builtin_amdgcn_s_barrier();
        if (!DB && TAP == 0 && c > c_lo) {  // single patch buffer: every wave is past the previous chunk only now
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
                for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], dq[n], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < NS) {  // the weight pieces of stage s + 2 (the slot read in stage s - 1: every wave is behind this stage's barrier)
            issue_w(TAP + 2 >= 9 ? c + 1 : c, T2, (TAP + 2) % 3);
            np = 2;
        }
        if (TAP == 0 && pnext) issue_patch(c + 1, pb ? 0u : (uint32_t)H_LDS);  // the buffer chunk c - 1 was read from
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + WS * WBYTES + m * 2048);
#pragma unroll
        for (int n = 0; n < 2; ++n) dq[n] = *(const u32x4_t*)(smem + pb + (preB0[KW] ^ 64u) + n * (PW * 128) + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // deferred fragments are in registers before their slot may be refilled
    };
#pragma unroll 1
    for (int c = c_lo; c < c_lo + nchunk; ++c) {
        stage(IC<0>{}, c); stage(IC<1>{}, c); stage(IC<2>{}, c); stage(IC<3>{}, c); stage(IC<4>{}, c);
        stage(IC<5>{}, c); stage(IC<6>{}, c); stage(IC<7>{}, c); stage(IC<8>{}, c);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], dq[n], acc[m][n]);

    if constexpr (SPLITK) {
        float* const dst = p.splitk_ws + ((size_t)sp * (gridDim.x / nsplit) + Lt) * (128 * 128);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                *(f32x4_t*)(dst + (wn * 32 + n * 16 + li) * 128 + wm * 64 + m * 16 + lg * 4) = acc[m][n];
        return;
    }

    float bv[4][4];
    epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);
    constexpr int OS = 128 * ESZ + 16;
    EpiStore<T, 128, H8_NTHR> est;
    const bool pool2 = !PAIR && (p.flags & C2W_CONV_POOL2) != 0;
    if constexpr (PAIR) est.prefetch_pair8(p, tid, co0, ((long long)b * H + oh0) * W, H * W, nimg);
    else if (!pool2) est.prefetch_tile16(p, tid, co0, ((long long)b * H + oh0) * W + ow0, W);
    __syncthreads();
    char* const O = smem;
    float* const red = (float*)(smem + 128 * OS);
    if constexpr (ESZ == 2) {
        if (p.ln_x != nullptr && tid < 128) red[tid] = 0.f;
    }
    epi_acc_to_lds_n<T, 2>(O, OS, acc, bv, p.act, wm * 64, wn * 32, li, lg);
    __syncthreads();
    if (pool2) {
        est.finish_pool2(p, O, OS, tid, co0, ((long long)b * (H >> 1) + (oh0 >> 1)) * (W >> 1) + (ow0 >> 1), W >> 1);
    } else if constexpr (ESZ == 2 && !PAIR) {
        if (p.ln_x != nullptr) est.finish_ln(p, O, OS, tid, b, red);
        else if (p.lnf_y != nullptr) est.finish_lnf(p, O, OS, tid, b);
        else est.finish(p, O, OS, tid);
    } else {
        est.finish(p, O, OS, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// FORWARD of the stride-2 convolutions (C2W_CONV_S2: the four down-convs, model/nn.py:169-174) on the halo patch (round 6; 16-bit; outputs
// 16 pixels wide or more, or exactly 8: S2Plan<PAIR>).  Output pixel (n, j) of an 8x16 output tile reads, for tap (kh, kw), input pixel (2n + kh, 2j + kw) of
// the (17 x 33)-pixel patch whose origin is (2 oh0 - 1, 2 ow0 - 1).  The patch is staged as its four PARITY PLANES -- plane (p, q) holds
// patch pixels (2a + p, 2b + q) -- so that tap (kh, kw) is a stride-1 view of plane (kh & 1, kw & 1) shifted by (kh >> 1, kw >> 1): every
// fragment address is register + immediate as in the stride-1 kernels, and the LDS-DMA does the de-interleaving (a piece = 8 consecutive
// plane pixels = 8 source pixels two apart; per-lane source offsets anyway).  Planes are packed (pitch 17 / 16 pixels, padded to whole
// pieces): 160 + 136 + 144 + 128 pixels x 128 B = 72,704 B per 64-channel chunk, + the 3-slot weight ring = 121,856 B: one workgroup of
// EIGHT waves per CU (the tile, the wave tiling, the stage body and the epilogue are conv_patch_half8_kernel's).
// There is no room for a second patch, but the two ROW parities are not live at the same time: the taps run kh = 1, 0, 2; the odd patch
// rows (planes (1, .): 8 rows, 34 KB) are read by the three kh = 1 stages only and the even ones (planes (0, .): 9 rows, 38 KB) by the six
// others.  So the even rows of chunk c are fetched during ITS OWN kh = 1 stages (issued behind stage 0's weight pieces, needed at stage 3)
// and the odd rows of chunk c + 1 during stages 3-5 (issued behind stage 3's weights, needed six stages later): five pieces per wave each
// time, allowed in flight by counted vmcnt(7) at the two following stages -- no exposed patch load after the prologue.
// Replaces the gather kernel (0.15-0.26 of peak on these launches: ~290 non-MFMA instructions per 32 MFMAs, each input pixel fetched 2.25
// times).  fp32 stays there.
// PAIR: 8-pixel-wide OUTPUT (the 16x16 -> 8x8 down-conv): the tile is two images side by side as in conv_patch_half_kernel<T, PAIR>; every plane
// holds the two images' columns one after the other (9 + 9 / 8 + 8 per row), each image with its own zero border.
template <bool PAIR>
struct S2Plan {
    static constexpr int W0 = 9, W1 = 8;                                   // plane columns of ONE 8-wide image, column parity 0 / 1
    static constexpr int PITCH0 = PAIR ? 2 * W0 : 17, PITCH1 = 16;          // plane row pitch in pixels
    // plane pixels, padded to whole 8-pixel pieces: 9 / 8 rows x pitch
    static constexpr int NP00 = (9 * PITCH0 + 7) / 8 * 8, NP10 = 8 * PITCH0, NP01 = 9 * PITCH1, NP11 = 8 * PITCH1;  // 160 136 144 128 | 168 144 144 128
    static constexpr int B00 = 0, B10 = NP00 * 128;                        // column parity 0: [even rows | odd rows]
    static constexpr int B01 = (NP00 + NP10) * 128, B11 = B01 + NP01 * 128;  // column parity 1
    static constexpr int PBYTES = B11 + NP11 * 128;                        // 72,704 | 74,752
    static constexpr int LDS = PBYTES + 3 * WBYTES;                        // 121,856 | 123,904
    static_assert((NP00 + NP01) / 8 <= 40 && (NP10 + NP11) / 8 <= 40, "five rounds of eight waves cover a row parity's pieces");
    static_assert(128 * (128 * 2 + 16) + 512 <= LDS, "output staging fits");
};

__device__ __forceinline__ void wait_vm_s2(int n) {  // wave-uniform n in {0, 2, 5, 7}
    if (n == 7) {
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    } else if (n == 5) {
        asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    } else if (n == 2) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

template <typename T, bool PAIR = false>
__global__ __launch_bounds__(H8_NTHR, 1) void conv_patch_s2_kernel(const C2wConvArgs p) {
    constexpr int ESZ = sizeof(T);
    static_assert(ESZ == 2, "16-bit operands");
    constexpr int CK = 64;
    using PL = S2Plan<PAIR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [plane 00 | 10 | 01 | 11 | W0 | W1 | W2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;  // wn: output rows 2 wn, 2 wn + 1 of the tile

    const int nN = (p.Cout + 127) / 128;
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * 128;
    const int H = p.Hout, W = p.Wout, Hi = p.Hin, Wi = p.Win;
    const int tw = PAIR ? 1 : W >> 4, tpi = (H >> 3) * tw;
    const int b = PAIR ? 2 * (tm / tpi) : tm / tpi, tt = tm - (tm / tpi) * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;

    const size_t img_bytes = (size_t)Hi * Wi * p.Cin * ESZ;
    const int nimg = PAIR ? (b + 1 < p.B ? 2 : 1) : 1;  // images under the descriptor: a missing partner reads as zeros (out of range)
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)(img_bytes * nimg));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));
    const int nchunk = p.Cin / CK;
    const int NS = nchunk * 9;

    // patch pieces, five per wave and row parity rp (rp = 0: the 9 even patch rows = taps kh 0 / 2; rp = 1: the 8 odd ones = kh 1) (pieces past the end repeat the last one: the same bytes to the same place).
    // Row parity rp: planes (rp, 0) then (rp, 1); piece pc of the set -> plane, local piece k; lane -> plane pixel i = 8 k + lane / 8 = (a, b_)
    uint32_t pvo[2][5];
    int pdst[2][5];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {
        const int n0 = (rp ? PL::NP10 : PL::NP00) >> 3, n1 = (rp ? PL::NP11 : PL::NP01) >> 3;  // pieces of the two planes: 20 + 18 / 17 + 16 (PAIR: 21 + 18 / 18 + 16)
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            int pc = r * 8 + wid;
            pc = pc < n0 + n1 ? pc : n0 + n1 - 1;
            const int cq = pc >= n0 ? 1 : 0;  // column parity of the piece's plane
            const int k = pc - cq * n0;
            const int pitch = cq ? PL::PITCH1 : PL::PITCH0, rows = rp ? 8 : 9;
            const int i = k * 8 + (lane >> 3);
            const int a = i / pitch, bc = i - a * pitch;                     // plane row, plane column
            const int wq = cq ? PL::W1 : PL::W0;
            const int pimg = PAIR && bc >= wq ? 1 : 0;                        // PAIR: the second image's columns follow the first's
            const int b_ = bc - pimg * wq;
            const int ih = 2 * oh0 - 1 + 2 * a + rp, iw = 2 * ow0 - 1 + 2 * b_ + cq;
            const bool ok = a < rows && (unsigned)ih < (unsigned)Hi && (unsigned)iw < (unsigned)Wi;
            const uint32_t lc = (uint32_t)((lane & 7) ^ (bc & 7));
            pvo[rp][r] = ok ? (uint32_t)((ih * Wi + iw) * p.Cin) * ESZ + (uint32_t)pimg * (uint32_t)img_bytes + (lc << 4) : C2W_OOB;
            pdst[rp][r] = (rp ? (cq ? PL::B11 : PL::B10) : (cq ? PL::B01 : PL::B00)) + k * 1024;
        }
    }
    uint32_t wvo[2];  // weight tile: 128 rows x 8 chunks = 2 rounds of 512 threads
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 64 * i;
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (uint32_t)(((tid & 7) ^ (row & 7)) << 4);
    }
    auto issue_w = [&](int chunk, int tap, int wslot) {
        const uint32_t so = (uint32_t)(tap * p.Cin + chunk * CK) * ESZ;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(rw, smem + PL::PBYTES + wslot * WBYTES + wid * 1024 + i * 8192, wvo[i], so);
    };
    auto issue_rows = [&](auto RPc, int chunk) {  // the two planes of one row parity
        constexpr int RP = decltype(RPc)::value;
#pragma unroll
        for (int r = 0; r < 5; ++r) glds16(rx, smem + pdst[RP][r], pvo[RP][r], (uint32_t)chunk * 128u);
    };

    // fragment addresses: A row m = + m * 2048; B: per kw the column-parity region, the lane's plane column li + (kw >> 1) and the wave's
    // first output row; row parity, row and the second output row are immediates; second K half = ^ 64
    const uint32_t offA0 = (uint32_t)(PL::PBYTES + (wm * 64 + li) * 128 + ((lg ^ (li & 7)) << 4));
    uint32_t preB0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int cq = kw & 1, pitch = cq ? PL::PITCH1 : PL::PITCH0;
        const int b_ = PAIR ? (li >> 3) * (cq ? PL::W1 : PL::W0) + (li & 7) + (kw >> 1) : li + (kw >> 1);  // PAIR: tile column >= 8 = the second image's columns
        preB0[kw] = (uint32_t)((cq ? PL::B01 : PL::B00) + (wn * 2 * pitch + b_) * 128 + ((lg ^ (b_ & 7)) << 4));
    }

    f32x4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // stage k of a chunk = tap (kh, kw) with kh = 1, 0, 2 for k / 3 = 0, 1, 2 and kw = k % 3; its weights live in ring slot k % 3
    issue_rows(IC<1>{}, 0);  // odd patch rows: the kh = 1 taps
    issue_w(0, 3, 0);  // stage 0 = tap (1, 0)
    issue_w(0, 4, 1);  // stage 1 = tap (1, 1)
    int np = 2;        // weight pieces of the next stage that may still be in flight
    u32x4_t da[4] = {}, dq[2] = {};

    auto stage = [&](auto Kc, int c) {
        constexpr int K = decltype(Kc)::value;
        constexpr int KH = K < 3 ? 1 : (K < 6 ? 0 : 2), KW = K % 3, WS = K % 3;
        constexpr int K2 = (K + 2) % 9, KH2 = K2 < 3 ? 1 : (K2 < 6 ? 0 : 2), T2 = KH2 * 3 + K2 % 3;  // the tap two stages on
        constexpr int RP = KH & 1, CQ = KW & 1, PITCH = CQ ? PL::PITCH1 : PL::PITCH0;
        constexpr int IMM = (RP ? (CQ ? PL::NP01 : PL::NP00) * 128 : 0) + (KH >> 1) * PITCH * 128;  // plane within the region + the tap's row shift
        const int s = c * 9 + K;
        const bool pnext = c + 1 < nchunk;
        // younger than this stage's weights: the previous stage's weight pieces (np) and, behind stage 0's / stage 3's, five patch pieces
        wait_vm_s2(np + ((K == 1 || K == 2 || ((K == 4 || K == 5) && pnext)) ? 5 : 0));
        __builtin_amdgcn_s_barrier();
        np = 0;
        u32x4_t a0[4], b0[2];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA0 + WS * WBYTES + m * 2048);
#pragma unroll
        for (int n = 0; n < 2; ++n) b0[n] = *(const u32x4_t*)(smem + preB0[KW] + IMM + n * (PITCH * 128));
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], dq[n], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (s + 2 < NS) {
            issue_w(K + 2 >= 9 ? c + 1 : c, T2, (K + 2) % 3);
            np = 2;
        }
        if (K == 0) issue_rows(IC<0>{}, c);               // even patch rows of THIS chunk (kh = 0, 2): last read in stage 8 of the previous one, first in stage 3
        if (K == 3 && pnext) issue_rows(IC<1>{}, c + 1);  // odd patch rows of the next chunk (kh = 1): last read in stage 2, first in its stage 0
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + WS * WBYTES + m * 2048);
#pragma unroll
        for (int n = 0; n < 2; ++n) dq[n] = *(const u32x4_t*)(smem + (preB0[KW] ^ 64u) + IMM + n * (PITCH * 128));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // deferred fragments are in registers before their slot / plane may be refilled
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        stage(IC<0>{}, c); stage(IC<1>{}, c); stage(IC<2>{}, c); stage(IC<3>{}, c); stage(IC<4>{}, c);
        stage(IC<5>{}, c); stage(IC<6>{}, c); stage(IC<7>{}, c); stage(IC<8>{}, c);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) Mma<T>::run(da[m], dq[n], acc[m][n]);

    float bv[4][4];
    epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);
    constexpr int OS = 128 * ESZ + 16;
    EpiStore<T, 128, H8_NTHR> est;
    if constexpr (PAIR) est.prefetch_pair8(p, tid, co0, ((long long)b * H + oh0) * W, H * W, nimg);
    else est.prefetch_tile16(p, tid, co0, ((long long)b * H + oh0) * W + ow0, W);
    __syncthreads();
    char* const O = smem;
    epi_acc_to_lds_n<T, 2>(O, OS, acc, bv, p.act, wm * 64, wn * 32, li, lg);
    __syncthreads();
    est.finish(p, O, OS, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// Input gradient of the stride-2 convolutions (C2W_CONV_TS2) on the halo patch.  Output pixel (2i + py, 2j + px) of parity class
// (py, px) receives   sum over kh in K(py), kw in K(px) of  w[.][kh*3 + kw][.] . dy[i + a(kh)][j + a(kw)],   K(0) = {1}, K(1) = {0, 2},
// a(0) = 1, a(1) = a(2) = 0  (conv_geom.h::src_pixel, TS2) -- a stride-1 convolution over dy with 1 / 2 / 2 / 4 taps whose result
// is stored with stride 2.  One workgroup = one class of one 8x16 tile of the dy grid: the same patch (origin (oh0 - 1, ow0 - 1),
// as for the forward conv: offset a reads patch tap a + 1), the same 3-slot weight ring and MFMA tiling as
// conv_patch_half_kernel; one launch per class.
// Replaces the gather kernel's per-tap pixel gathers (295-440 TFLOP/s at B = 128).
struct Ts2Tap { int t9, khp, kwp; };
template <int CLS> struct Ts2Class;
template <> struct Ts2Class<0> { static constexpr int NT = 1; static constexpr Ts2Tap taps[1] = {{4, 1, 1}}; };
template <> struct Ts2Class<1> { static constexpr int NT = 2; static constexpr Ts2Tap taps[2] = {{3, 1, 2}, {5, 1, 1}}; };
template <> struct Ts2Class<2> { static constexpr int NT = 2; static constexpr Ts2Tap taps[2] = {{1, 2, 1}, {7, 1, 1}}; };
template <> struct Ts2Class<3> { static constexpr int NT = 4; static constexpr Ts2Tap taps[4] = {{0, 2, 2}, {2, 2, 1}, {6, 1, 2}, {8, 1, 1}}; };

template <typename T, int CLS>
__device__ __forceinline__ void conv_patch_ts2_class(const C2wConvArgs& p, char* smem, int L) {
    typedef Ts2Class<CLS> TC;
    constexpr int NT = TC::NT;
    constexpr int PY = CLS >> 1, PX = CLS & 1;
    constexpr int ESZ = sizeof(T);
    constexpr int CK = 128 / ESZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;

    const int nN = (p.Cout + 127) / 128;
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * 128;
    const int H = p.Hin, W = p.Win;  // the dy grid; the output grid is 2H x 2W
    const int tw = W >> 4, tpi = (H >> 3) * tw;
    const int b = tm / tpi, tt = tm - b * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;

    const size_t img_bytes = (size_t)H * W * p.Cin * ESZ;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)img_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    uint32_t pvo[8];
    int pdst[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int pc = r * 4 + wid;
        pc = pc < H_NPIECE ? pc : H_NPIECE - 1;
        const int pr = pc / 3, pg = pc - pr * 3;
        const int px = pg * 8 + (lane >> 3);
        const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
        const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < 18;
        const uint32_t lc = (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));
        pvo[r] = ok ? (uint32_t)((ih * W + iw) * p.Cin) * ESZ + (lc << 4) : C2W_OOB;
        pdst[r] = pc * 1024;
    }
    uint32_t wvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (uint32_t)(((tid & 7) ^ (row & 7)) << 4);
    }
    const int nchunk = p.Cin / CK;
    const int NS = nchunk * NT;
    auto issue_stage_w = [&](int st) {  // weights of global stage st = chunk * NT + idx into ring slot st % 3
        const int c2 = st / NT, i2 = st - c2 * NT;
        int t9 = TC::taps[0].t9;
#pragma unroll
        for (int k = 1; k < NT; ++k) t9 = i2 == k ? TC::taps[k].t9 : t9;
        const uint32_t so = (uint32_t)(t9 * p.Cin + c2 * CK) * ESZ;
        const int wslot = st % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16(rw, smem + H_PBYTES + wslot * WBYTES + wid * 1024 + i * 4096, wvo[i], so);
    };
    auto issue_patch = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < 8; ++r) glds16(rx, smem + pdst[r], pvo[r], (uint32_t)chunk * 128u);
    };

    uint32_t offA[2][4], preB[2][3][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int row = wm * 64 + m * 16 + li;
            offA[ks][m] = (uint32_t)(H_PBYTES + row * 128 + (((ks * 4 + lg) ^ (row & 7)) << 4));
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int px = li + kw;
                preB[ks][kw][n] = (uint32_t)(((wn * 4 + n) * PW + px) * 128 + (((ks * 4 + lg) ^ (px & 7)) << 4));
            }
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_patch(0);
    issue_stage_w(0);
    int np = 4;  // LDS-DMA pieces of the NEXT stage that may still be in flight when a stage starts
    if (1 < NS) issue_stage_w(1);
    else np = 0;
    u32x4_t da[4] = {}, db[4] = {};

    auto stage = [&](auto IDXc, int c) {
        constexpr int IDX = decltype(IDXc)::value;
        constexpr int KH = TC::taps[IDX].khp, KW = TC::taps[IDX].kwp;
        const int s = c * NT + IDX;
        const uint32_t wso = (uint32_t)(s % 3) * WBYTES;
        wait_vm4(np);
        __builtin_amdgcn_s_barrier();
        if (IDX == 0 && c > 0) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        np = 0;
        if (s + 2 < NS) {
            issue_stage_w(s + 2);
            np = 4;
        }
        u32x4_t a0[4], b0[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA[0][m] + wso);
#pragma unroll
        for (int n = 0; n < 4; ++n) b0[n] = *(const u32x4_t*)(smem + preB[0][KW][n] + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
        if (s > 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + offA[1][m] + wso);
#pragma unroll
        for (int n = 0; n < 4; ++n) db[n] = *(const u32x4_t*)(smem + preB[1][KW][n] + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<T>::run(a0[m], b0[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // deferred fragments are in registers before their slot may be refilled
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        stage(IC<0>{}, c);
        if constexpr (NT > 1) stage(IC<1>{}, c);
        if constexpr (NT > 2) {
            stage(IC<2>{}, c);
            stage(IC<3>{}, c);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[m][n]);

    float bv[4][4];  // after the loop: the 4-tap class is within a few registers of the 256 budget
    epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);
    constexpr int OS = 128 * ESZ + 16;
    EpiStore<T, 128, H_NTHR> est;
    est.prefetch_tile16_s2(p, tid, co0, ((long long)b * (2 * H) + 2 * oh0 + PY) * (2 * W) + 2 * ow0 + PX, 2 * W);
    __syncthreads();
    char* const O = smem;
    epi_acc_to_lds<T>(O, OS, acc, bv, p.act, wm * 64, wn * 64, li, lg);
    __syncthreads();
    est.finish(p, O, OS, tid);
}

// Round 6: TWO classes per workgroup.  A class's workgroup above is mostly overhead: 1-4 taps x Cout / 64 chunks = 2-8 MFMA stages between
// a prologue (patch + two weight stages in flight), a patch reload per chunk and an epilogue with strided stores -- about 10 us per
// workgroup of which 1-4 are matrix work (128 -> 128 from 64^2: 16,384 workgroups, 0.405 ms, 0.15 of peak).  Here a workgroup stages the
// patch ONCE per chunk for two classes -- {3, 0}: 4 + 1 taps, {1, 2}: 2 + 2 taps -- keeps two accumulator sets (128 of the 256 registers a
// wave has at two workgroups per CU) and runs two epilogues: half the workgroups, half the patch loads and prologues.
struct Ts2PairTap { int t9, khp, kwp, set; };
template <int SEL> struct Ts2Pair;
template <> struct Ts2Pair<0> {  // classes 3 (set 0) and 0 (set 1)
    static constexpr int NT = 5, CLS0 = 3, CLS1 = 0;
    static constexpr Ts2PairTap taps[5] = {{0, 2, 2, 0}, {2, 2, 1, 0}, {6, 1, 2, 0}, {8, 1, 1, 0}, {4, 1, 1, 1}};
};
template <> struct Ts2Pair<1> {  // classes 1 (set 0) and 2 (set 1)
    static constexpr int NT = 4, CLS0 = 1, CLS1 = 2;
    static constexpr Ts2PairTap taps[4] = {{3, 1, 2, 0}, {5, 1, 1, 0}, {1, 2, 1, 1}, {7, 1, 1, 1}};
};

// PIPE: the second K half of a stage is deferred across the next stage's barrier (conv_patch_half_kernel's schedule: 32 more registers).  The fp16
// build does not fit with it (9 registers over; anything spilled breaks the counted vmcnt waits) and runs the two halves back to back.
template <typename T, int SEL, bool PIPE>
__device__ __forceinline__ void conv_patch_ts2_pair(const C2wConvArgs& p, char* smem, int L) {
    typedef Ts2Pair<SEL> TC;
    constexpr int NT = TC::NT;
    constexpr int ESZ = sizeof(T);
    constexpr int CK = 128 / ESZ;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;

    const int nN = (p.Cout + 127) / 128;
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * 128;
    const int H = p.Hin, W = p.Win;  // the dy grid; the output grid is 2H x 2W
    const int tw = W >> 4, tpi = (H >> 3) * tw;
    const int b = tm / tpi, tt = tm - b * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;

    const size_t img_bytes = (size_t)H * W * p.Cin * ESZ;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)img_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    const int nchunk = p.Cin / CK;
    const int NS = nchunk * NT;
    // (source offsets are recomputed where they are used: two accumulator sets leave no registers to hold them across the loop)
    auto issue_stage_w = [&](int st) {  // weights of global stage st = chunk * NT + idx into ring slot st % 3
        const int c2 = st / NT, i2 = st - c2 * NT;
        int t9 = TC::taps[0].t9;
#pragma unroll
        for (int k = 1; k < NT; ++k) t9 = i2 == k ? TC::taps[k].t9 : t9;
        const uint32_t so = (uint32_t)(t9 * p.Cin + c2 * CK) * ESZ;
        const int wslot = st % 3;
        int t_ = tid;
        asm volatile("" : "+v"(t_));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (t_ >> 3) + 32 * i;
            const uint32_t wvo = (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (uint32_t)(((t_ & 7) ^ (row & 7)) << 4);
            glds16(rw, smem + H_PBYTES + wslot * WBYTES + wid * 1024 + i * 4096, wvo, so);
        }
    };
    auto issue_patch = [&](int chunk) {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int pc = r * 4 + wid;
            pc = pc < H_NPIECE ? pc : H_NPIECE - 1;
            const int pr = pc / 3, pg = pc - pr * 3;
            const int px = pg * 8 + (l_ >> 3);
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < 18;
            const uint32_t lc = (uint32_t)((l_ & 7) ^ ((l_ >> 3) & 7));
            const uint32_t pvo = ok ? (uint32_t)((ih * W + iw) * p.Cin) * ESZ + (lc << 4) : C2W_OOB;
            glds16(rx, smem + pc * 1024, pvo, (uint32_t)chunk * 128u);
        }
    };

    // Fragment addresses: FOUR registers instead of the 32 of conv_patch_half_kernel (two accumulator sets need the rest).  Row m of the
    // A tile is 2048 bytes further (its swizzle depends on li only), pixel row n of the patch PW * 128 bytes further: immediates of the
    // ds_read; the second K half flips chunk bit 2 of the swizzled 16-byte slot = bit 6 of the address (zero in everything else).
    const uint32_t offA0 = (uint32_t)(H_PBYTES + (wm * 64 + li) * 128 + ((lg ^ (li & 7)) << 4));
    uint32_t preB0[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int px = li + kw;
        preB0[kw] = (uint32_t)((wn * 4 * PW + px) * 128 + ((lg ^ (px & 7)) << 4));
    }

    f32x4_t acc[2][4][4];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[q][m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    issue_patch(0);
    issue_stage_w(0);
    issue_stage_w(1);
    int np = 4;  // LDS-DMA pieces of the NEXT stage that may still be in flight when a stage starts
    u32x4_t da[4] = {}, db[4] = {};

    auto stage = [&](auto IDXc, int c) {
        constexpr int IDX = decltype(IDXc)::value;
        constexpr int KH = TC::taps[IDX].khp, KW = TC::taps[IDX].kwp, SET = TC::taps[IDX].set;
        constexpr int PSET = TC::taps[(IDX + NT - 1) % NT].set;  // the set the deferred second half (da, db) of the PREVIOUS stage belongs to
        const int s = c * NT + IDX;
        const uint32_t wso = (uint32_t)(s % 3) * WBYTES;
        wait_vm4(np);
        __builtin_amdgcn_s_barrier();
        if (IDX == 0 && c > 0) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        np = 0;
        if (s + 2 < NS) {
            issue_stage_w(s + 2);
            np = 4;
        }
        u32x4_t a0[4], b0[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + offA0 + wso + m * 2048);
#pragma unroll
        for (int n = 0; n < 4; ++n) b0[n] = *(const u32x4_t*)(smem + preB0[KW] + n * (PW * 128) + KH * PROW);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (PIPE) {
            if (s > 0) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[PSET][m][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) da[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + wso + m * 2048);
#pragma unroll
            for (int n = 0; n < 4; ++n) db[n] = *(const u32x4_t*)(smem + (preB0[KW] ^ 64u) + n * (PW * 128) + KH * PROW);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(a0[m], b0[n], acc[SET][m][n]);
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(a0[m], b0[n], acc[SET][m][n]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m) a0[m] = *(const u32x4_t*)(smem + (offA0 ^ 64u) + wso + m * 2048);
#pragma unroll
            for (int n = 0; n < 4; ++n) b0[n] = *(const u32x4_t*)(smem + (preB0[KW] ^ 64u) + n * (PW * 128) + KH * PROW);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(a0[m], b0[n], acc[SET][m][n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // every fragment is in registers before its slot may be refilled
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        stage(IC<0>{}, c); stage(IC<1>{}, c); stage(IC<2>{}, c); stage(IC<3>{}, c);
        if constexpr (NT > 4) stage(IC<4>{}, c);
    }
    if constexpr (PIPE) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) Mma<T>::run(da[m], db[n], acc[TC::taps[NT - 1].set][m][n]);
    }

    // the epilogues' lane coordinates are derived afresh (kept across the loop they cost registers the two accumulator sets do not leave)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, li_e = lane_e & 15, lg_e = lane_e >> 4;
    float bv[4][4];
    epi_load_bias(p, co0 + wm * 64 + lg_e * 4, bv);
    constexpr int OS = 128 * ESZ + 16;
    char* const O = smem;
    // both accumulator sets leave the registers FIRST (two output tiles of 34,816 bytes fit the loop's LDS): with one set still live the
    // epilogue's prefetch arrays do not fit next to it
    static_assert(2 * 128 * OS <= H_LDS, "two output tiles");
    __syncthreads();
    epi_acc_to_lds<T>(O, OS, acc[0], bv, p.act, wm * 64, wn * 64, li_e, lg_e);
    __builtin_amdgcn_sched_barrier(0);  // one set after the other (interleaved, the fp16 build's conversions need 9 registers more than there are)
    epi_acc_to_lds<T>(O + 128 * OS, OS, acc[1], bv, p.act, wm * 64, wn * 64, li_e, lg_e);
    __syncthreads();
    {
        constexpr int PY = TC::CLS0 >> 1, PX = TC::CLS0 & 1;
        EpiStore<T, 128, H_NTHR> est;
        est.prefetch_tile16_s2(p, tid_e, co0, ((long long)b * (2 * H) + 2 * oh0 + PY) * (2 * W) + 2 * ow0 + PX, 2 * W);
        est.finish(p, O, OS, tid_e);
    }
    {
        constexpr int PY = TC::CLS1 >> 1, PX = TC::CLS1 & 1;
        EpiStore<T, 128, H_NTHR> est;
        est.prefetch_tile16_s2(p, tid_e, co0, ((long long)b * (2 * H) + 2 * oh0 + PY) * (2 * W) + 2 * ow0 + PX, 2 * W);
        est.finish(p, O + 128 * OS, OS, tid_e);
    }
}

// blockIdx = 8 * (2 * tile-in-XCD + pair slot) + XCD: the two workgroups of a tile are neighbours on one XCD (their patch and residual
// lines meet in its L2); slot 0 = classes {3, 0} (five taps), slot 1 = {1, 2} (four).
template <typename T>
__global__ __launch_bounds__(H_NTHR, 2) void conv_patch_ts2_pairs_kernel(const C2wConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3, slot = rest & 1, j = rest >> 1;
    const int ntile = ((p.Cout + 127) / 128) * p.B * (p.Hin >> 3) * (p.Win >> 4);
    const int q = ntile >> 3, r = ntile & 7;
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (j >= (xcd < r ? q + 1 : q)) return;  // (grid padded to a multiple of 8 tiles per slot)
    constexpr bool PIPE = std::is_same<T, bf16_t>::value;
    if (slot == 0) conv_patch_ts2_pair<T, 0, PIPE>(p, smem, L);
    else conv_patch_ts2_pair<T, 1, PIPE>(p, smem, L);
}

// One kernel per class (four launches): with the four bodies in one kernel the register allocation of the 4-tap class governs
// all of them and the merged code spilled 144 VGPRs.
template <typename T, int CLS>
__global__ __launch_bounds__(H_NTHR, 2) void conv_patch_ts2_kernel(const C2wConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    conv_patch_ts2_class<T, CLS>(p, smem, L);
}

// All four classes in ONE launch (round 4): a workgroup still computes one class of one tile -- the class is picked per workgroup, so the
// register allocation is the maximum over the four bodies, not their sum -- and the four classes of a tile sit next to each other
// on the same XCD (blockIdx = 8 * (4 * tile-in-XCD + class slot) + XCD), so that they find the tile's dy patch and their common
// residual lines in that XCD's L2.  Class slots in the order 3, 1, 2, 0 (most taps first).
template <typename T>
__global__ __launch_bounds__(H_NTHR, 2) void conv_patch_ts2_all_kernel(const C2wConvArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x, xcd = bid & 7, rest = bid >> 3, slot = rest & 3, j = rest >> 2;
    const int ntile = ((p.Cout + 127) / 128) * p.B * (p.Hin >> 3) * (p.Win >> 4);
    const int q = ntile >> 3, r = ntile & 7;  // tiles are dealt to the XCDs like the single-class launch deals its workgroups
    const int L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    if (j >= (xcd < r ? q + 1 : q)) return;  // (grid padded to a multiple of 8 tiles per class slot)
    switch (slot) {
        case 0: conv_patch_ts2_class<T, 3>(p, smem, L); break;
        case 1: conv_patch_ts2_class<T, 1>(p, smem, L); break;
        case 2: conv_patch_ts2_class<T, 2>(p, smem, L); break;
        default: conv_patch_ts2_class<T, 0>(p, smem, L); break;
    }
}

template <typename T>
int launch_ts2_all(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_ts2_all_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hin >> 3) * (a.Win >> 4);
    const int ntile = nM * nN, per_xcd = (ntile + 7) / 8;
    conv_patch_ts2_all_kernel<T><<<8 * 4 * per_xcd, H_NTHR, H_LDS, st>>>(a);
    return (int)hipGetLastError();
}

template <typename T, int CLS>
int launch_ts2_class(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_ts2_kernel<T, CLS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hin >> 3) * (a.Win >> 4);
    conv_patch_ts2_kernel<T, CLS><<<nM * nN, H_NTHR, H_LDS, st>>>(a);
    return (int)hipGetLastError();
}

template <typename T>
int launch_ts2_pairs(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_ts2_pairs_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hin >> 3) * (a.Win >> 4);
    const int ntile = nM * nN, per_xcd = (ntile + 7) / 8;
    conv_patch_ts2_pairs_kernel<T><<<8 * 2 * per_xcd, H_NTHR, H_LDS, st>>>(a);
    return (int)hipGetLastError();
}

template <typename T>
int launch_ts2(const C2wConvArgs& a, hipStream_t st) {  // largest class first
    if constexpr (sizeof(T) == 2) {  // bf16: 249 registers with the deferred K half; fp16: without it (PIPE); two fp32 accumulator sets are out of the question
        if (c2w_knobs().ts2_pairs && c2w_knobs().ts2_one_launch) return launch_ts2_pairs<T>(a, st);
    }
    if (c2w_knobs().ts2_one_launch) return launch_ts2_all<T>(a, st);
    int rc = launch_ts2_class<T, 3>(a, st);
    if (rc == 0) rc = launch_ts2_class<T, 1>(a, st);
    if (rc == 0) rc = launch_ts2_class<T, 2>(a, st);
    if (rc == 0) rc = launch_ts2_class<T, 0>(a, st);
    return rc;
}

// Second launch of a split-K convolution: adds the `splitk` partial tiles in a fixed order and applies the epilogue of EpiStore::finish --
// bias, activation (none / SiLU / ReLU), multiplier (plain or silu'), residual -- to 16-byte NHWC stores.  One workgroup per 16 tile rows,
// one (row, 8-channel segment) per thread, every split's load in flight at once (a first version -- one workgroup per tile, eight rows
// per thread, one split after the other -- took 18-21 us on 76-222 tiles: a chain of dependent loads on a quarter of the chip).
template <typename T, bool PAIR>
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const C2wConvArgs p, int ntiles) {
    constexpr int ESZ = sizeof(T), PER16 = 16 / ESZ, SEGS = 128 / PER16, RPB = 256 / SEGS;  // rows per block: 16 (16-bit) / 8 (fp32)
    constexpr int BPT = 128 / RPB;
    const int Lt = blockIdx.x / BPT, rblk = blockIdx.x - Lt * BPT;
    const int nN = (p.Cout + 127) / 128;
    const int tn = Lt % nN, tm = Lt / nN, co0 = tn * 128;
    const int H = p.Hout, W = p.Wout;
    const int tw = PAIR ? 1 : W >> 4, tpi = (H >> 3) * tw;
    const int b = PAIR ? 2 * (tm / tpi) : tm / tpi, tt = tm - (tm / tpi) * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty << 3, ow0 = tx << 4;
    const int nimg = PAIR ? (b + 1 < p.B ? 2 : 1) : 1;
    const int R = rblk * RPB + (int)threadIdx.x / SEGS, cs = (int)threadIdx.x % SEGS;
    const int c = co0 + cs * PER16;
    const int trow = R >> 4, col = R & 15;
    long long pix;
    bool ok = c < p.Cout;
    if constexpr (PAIR) {
        const int img = col >> 3;
        pix = ((long long)(b + img) * H + oh0 + trow) * W + (col & 7);
        ok = ok && img < nimg;
    } else {
        pix = ((long long)b * H + oh0 + trow) * W + ow0 + col;
    }
    if (!ok) return;
    const float* const src = p.splitk_ws + (size_t)Lt * (128 * 128) + R * 128 + cs * PER16;
    const size_t sstride = (size_t)ntiles * (128 * 128);
    f32x4_t part[8][PER16 / 4];
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_)
#pragma unroll
        for (int e = 0; e < PER16 / 4; ++e)
            part[s_][e] = s_ < p.splitk ? *(const f32x4_t*)(src + s_ * sstride + 4 * e) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const size_t off = ((size_t)pix * p.ldy + c) * ESZ;
    u32x4_t mv = {}, rv = {};
    if (p.mul != nullptr) mv = *(const u32x4_t*)((const char*)p.mul + off);
    if (p.res != nullptr) rv = *(const u32x4_t*)((const char*)p.res + off);
    float f[PER16];
#pragma unroll
    for (int e = 0; e < PER16; ++e) {
        float v = 0.f;
#pragma unroll
        for (int s_ = 0; s_ < 8; ++s_) v += part[s_][e >> 2][e & 3];  // split 0 first: a fixed order
        v += (p.bias != nullptr && c + e < p.wrows) ? p.bias[c + e] : 0.f;
        if (p.act == C2W_ACT_SILU) v = silu_f(v);
        if (p.act == C2W_ACT_RELU) v = fmaxf(v, 0.f);
        f[e] = v;
    }
    if (p.mul != nullptr || p.res != nullptr) {  // on the value as the unsplit kernel stages it: rounded to the storage type first
        u32x4_t st = pack16<T>(f);
        unpack16<T>(st, f);
        if (p.mul != nullptr) {
            float gm[PER16];
            unpack16<T>(mv, gm);
#pragma unroll
            for (int e = 0; e < PER16; ++e) f[e] *= p.mulmode == C2W_MUL_DSILU ? dsilu_f(gm[e]) : gm[e];
        }
        if (p.res != nullptr) {
            float gr[PER16];
            unpack16<T>(rv, gr);
#pragma unroll
            for (int e = 0; e < PER16; ++e) f[e] += gr[e];
        }
    }
    *(u32x4_t*)((char*)p.y + off) = pack16<T>(f);
}

// at most one workgroup per CU: the eight-wave form (two waves per SIMD instead of one)
// Which 8x16-tile launches take the eight-wave form.  At most one workgroup per CU: always worth it (two waves per SIMD instead of one).  Above
// that the two forms are within 0.1 ms per step of each other (profiles/r06t_ab_half8_max_wgs_*.txt: B = 128 46.40-46.46 ms with every launch
// on eight waves against 46.47-46.58 with the 256-workgroup limit; B = 64 and the sampler: equal) -- the 16-bit builds fit two
// workgroups per CU (114-125 registers), so they take it everywhere; the fp32 builds (131-132 registers: one workgroup per CU) only
// where there is one per CU anyway.  C2W_HALF8_MAX_WGS=N overrides.
template <typename T>
static inline bool half8_wanted(long long wgs) {
    const int lim = c2w_knobs().half8_max_wgs;
    return c2w_knobs().half8 && (lim > 0 ? wgs <= lim : (sizeof(T) == 2 || wgs <= 256));
}

template <typename T, bool PAIR, bool SPLITK, bool DB>
int launch_half8_db(const C2wConvArgs& a, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_half8_kernel<T, PAIR, SPLITK, DB>, hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
        attr = true;
    }
    conv_patch_half8_kernel<T, PAIR, SPLITK, DB><<<nwg, H8_NTHR, DB ? H8_LDS_DB : H_LDS, st>>>(a);
    return (int)hipGetLastError();
}

// one workgroup per CU at most: the second patch buffer costs nothing (C2W_HALF8_DB=0: never)
template <typename T, bool PAIR, bool SPLITK>
int launch_half8(const C2wConvArgs& a, int nwg, hipStream_t st) {
    if (nwg <= 256 && c2w_knobs().half8_db) return launch_half8_db<T, PAIR, SPLITK, true>(a, nwg, st);
    return launch_half8_db<T, PAIR, SPLITK, false>(a, nwg, st);
}

template <typename T, bool PAIR>
int launch_splitk(const C2wConvArgs& a, int ntiles, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_half_kernel<T, PAIR, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr = true;
    }
    int rc;
    if (half8_wanted<T>((long long)ntiles * a.splitk)) {
        rc = launch_half8<T, PAIR, true>(a, ntiles * a.splitk, st);
    } else {
        conv_patch_half_kernel<T, PAIR, true><<<ntiles * a.splitk, H_NTHR, H_LDS, st>>>(a);
        rc = (int)hipGetLastError();
    }
    if (rc != 0) return rc;
    conv_splitk_epilogue_kernel<T, PAIR><<<ntiles * (128 / (256 / (128 / (16 / (int)sizeof(T))))), 256, 0, st>>>(a, ntiles);
    return (int)hipGetLastError();
}

template <typename T>
int launch(const C2wConvArgs& a, hipStream_t st) {  // two 8x16-tile workgroups per CU
    constexpr int ESZ = sizeof(T);
    static_assert(128 * (128 * ESZ + 16) <= H_LDS, "half-tile output rows fit");
    static bool attr_h = false;
    if (!attr_h) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_half_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr_h = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nMh = a.B * (a.Hout >> 3) * (a.Wout >> 4);
    if (a.splitk > 1) return launch_splitk<T, false>(a, nMh * nN, st);
    if (half8_wanted<T>((long long)nMh * nN)) return launch_half8<T, false, false>(a, nMh * nN, st);
    conv_patch_half_kernel<T><<<nMh * nN, H_NTHR, H_LDS, st>>>(a);
    return (int)hipGetLastError();
}

}  // namespace

namespace {
template <typename T>
int launch_pair(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_half_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = ((a.B + 1) >> 1) * (a.Hin >> 3);
    if (a.splitk > 1) return launch_splitk<T, true>(a, nM * nN, st);
    if (half8_wanted<T>((long long)nM * nN)) return launch_half8<T, true, false>(a, nM * nN, st);
    conv_patch_half_kernel<T, true><<<nM * nN, H_NTHR, H_LDS, st>>>(a);
    return (int)hipGetLastError();
}
}  // namespace

// Split-K plan of a convolution on the 8x16-tile kernels (see conv_patch_half_kernel<T, PAIR, SPLITK>): the number of workgroups each
// output tile's K chunks are dealt to (1: no split) and the scratch the partial tiles need.
int c2w_conv_splitk_plan_impl(const C2wConvArgs& a, int dtype, unsigned long long* ws_bytes) {
    if (ws_bytes != nullptr) *ws_bytes = 0;
    if (!c2w_knobs().splitk || c2w_knobs().force_gather) return 1;
    if (a.y2 != nullptr || a.ln_x != nullptr || a.lnf_y != nullptr || a.loss_sum != nullptr || (a.flags & (C2W_CONV_POOL2 | C2W_CONV_NO_Y | C2W_CONV_WPACKED)) != 0) return 1;
    if (a.act != C2W_ACT_NONE && a.act != C2W_ACT_SILU && a.act != C2W_ACT_RELU) return 1;
    const bool pair = c2w_conv_pair_eligible(a);
    if (!pair && !(c2w_conv_patch_eligible(a) && !c2w_conv_patch3_wanted(a, dtype))) return 1;
    const int nN = (a.Cout + 127) / 128;
    const long long tiles = (pair ? (long long)((a.B + 1) >> 1) * (a.Hin >> 3) : (long long)a.B * (a.Hout >> 3) * (a.Wout >> 4)) * nN;
    const int nchunk = a.Cin / (dtype == C2W_DTYPE_F32 ? 32 : 64);
    // As many workgroups per tile as keep the launch at ONE workgroup per CU (two per CU share the matrix pipe: the chain gets
    // shorter and each stage slower -- 444 workgroups of 27 stages took 31.6 us where 222 of 54 take 35.9, before the 18-us second
    // launch), at most one per chunk and 8 (the reduction's unroll).  76 tiles x 8 chunks -> 3 workgroups of 2 / 3 / 3 chunks.
    int best = (int)(256 / (tiles > 0 ? tiles : 1));
    if (best > nchunk) best = nchunk;
    if (best > 8) best = 8;
    if (best < 2) return 1;
    if (ws_bytes != nullptr) *ws_bytes = (unsigned long long)best * tiles * 128 * 128 * sizeof(float);
    return best;
}

// forward of the stride-2 convs on the parity planes of the halo patch (conv_patch_s2_kernel; 16-bit; output tiled by 8x16 pixels, or 8 pixels
// wide with two images per tile)
// Where it is taken: one workgroup per CU (121 KB of LDS) means nothing covers a workgroup's prologue and epilogue, and at Cin = 128 a
// workgroup is only 18 stages long -- measured per launch against the gather kernel (profiles/r06x_ab_s2_forward.txt): 256 -> 384 @32^2 -> 16^2
// 88 -> 64 us at B = 128 and 34 -> 22 us at B = 37; 128 -> 256 @64^2 -> 32^2 113 -> 112 / 50 -> 40 us; 128 -> 128 @128^2 -> 64^2 244 -> 251 us at
// B = 128 (4096 workgroups: slower) but 79 -> 73 us at B = 37 (1184).  So: from four K chunks on, or up to 2048 workgroups
// (C2W_CONV_S2_PATCH=0: never; =2: wherever the geometry allows).
bool c2w_conv_s2_patch_eligible(const C2wConvArgs& a, int dtype) {
    const int knob = c2w_knobs().conv_s2_patch;
    const bool pair = a.Wout == 8;  // 8-pixel-wide output: two images per tile
    const long long nwg = (pair ? (long long)((a.B + 1) >> 1) * (a.Hout >> 3) : (long long)a.B * (a.Hout >> 3) * (a.Wout >> 4)) * ((a.Cout + 127) / 128);
    const bool pays = knob == 2 || a.Cin >= 256 || nwg <= 2048;
    return knob != 0 && pays && dtype != C2W_DTYPE_F32 && a.mode == C2W_CONV_S2 && a.Hin == 2 * a.Hout && a.Win == 2 * a.Wout && (a.Hout & 7) == 0 &&
           (pair || (a.Wout & 15) == 0) && a.Cin % 64 == 0 && a.ln_x == nullptr && a.lnf_y == nullptr && a.y2 == nullptr && a.loss_sum == nullptr && a.splitk <= 1 &&
           (a.act == C2W_ACT_NONE || a.act == C2W_ACT_SILU || a.act == C2W_ACT_RELU) && (a.flags & (C2W_CONV_POOL2 | C2W_CONV_NO_Y | C2W_CONV_WPACKED)) == 0 &&
           (long long)a.Hin * a.Win * a.Cin * 4 < (1ll << 32) && nwg < (1ll << 31);
}

namespace {
template <typename T, bool PAIR>
int launch_s2(const C2wConvArgs& a, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_s2_kernel<T, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nwg = (PAIR ? ((a.B + 1) >> 1) * (a.Hout >> 3) : a.B * (a.Hout >> 3) * (a.Wout >> 4)) * nN;
    conv_patch_s2_kernel<T, PAIR><<<nwg, H8_NTHR, S2Plan<PAIR>::LDS, st>>>(a);
    return (int)hipGetLastError();
}
}  // namespace

int c2w_conv_patch_s2(const C2wConvArgs& a, int dtype, hipStream_t st) {
    if (dtype == C2W_DTYPE_BF16) return a.Wout == 8 ? launch_s2<bf16_t, true>(a, st) : launch_s2<bf16_t, false>(a, st);
    if (dtype == C2W_DTYPE_F16) return a.Wout == 8 ? launch_s2<f16_t, true>(a, st) : launch_s2<f16_t, false>(a, st);
    return C2W_ERR_BAD_ARG;
}

// input gradient of the stride-2 convs per output-parity class on the halo patch (conv_patch_ts2_kernel)
bool c2w_conv_ts2_patch_eligible(const C2wConvArgs& a) {
    return c2w_knobs().conv_ts2_patch && a.mode == C2W_CONV_TS2 && a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && (a.Hin & 7) == 0 && (a.Win & 15) == 0 &&
           a.ln_x == nullptr && a.lnf_y == nullptr && a.y2 == nullptr && a.act == C2W_ACT_NONE &&
           (long long)a.B * (a.Hin >> 3) * (a.Win >> 4) * ((a.Cout + 127) / 128) * 4 < (1ll << 31);
}

int c2w_conv_patch_ts2(const C2wConvArgs& a, int dtype, hipStream_t st) {
    if (dtype == C2W_DTYPE_F32) return launch_ts2<float>(a, st);
    if (dtype == C2W_DTYPE_BF16) return launch_ts2<bf16_t>(a, st);
    if (dtype == C2W_DTYPE_F16) return launch_ts2<f16_t>(a, st);
    return C2W_ERR_BAD_ARG;
}

// 8-pixel-wide images: two of them per 8x16 tile (conv_patch_half_kernel<T, PAIR>); no fused LayerNorm epilogues in that mode
bool c2w_conv_pair_eligible(const C2wConvArgs& a) {
    return c2w_knobs().conv_pair && a.mode == C2W_CONV_S1 && a.Hin == a.Hout && a.Win == a.Wout && a.Win == 8 && (a.Hin & 7) == 0 && a.ln_x == nullptr &&
           a.lnf_y == nullptr && (long long)((a.B + 1) >> 1) * (a.Hin >> 3) * ((a.Cout + 127) / 128) < (1ll << 31);
}

int c2w_conv_patch_pair(const C2wConvArgs& a, int dtype, hipStream_t st) {
    if (dtype == C2W_DTYPE_F32) return launch_pair<float>(a, st);
    if (dtype == C2W_DTYPE_BF16) return launch_pair<bf16_t>(a, st);
    if (dtype == C2W_DTYPE_F16) return launch_pair<f16_t>(a, st);
    return C2W_ERR_BAD_ARG;
}

bool c2w_conv_patch_eligible(const C2wConvArgs& a) {  // OUTPUT grids that 8 x 16-pixel tiles cover exactly; stride 1, or x2 upsampling folded in
    const bool geom = (a.mode == C2W_CONV_S1 && a.Hin == a.Hout && a.Win == a.Wout) ||
                      (a.mode == C2W_CONV_UP && a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && c2w_knobs().up_patch);
    return geom && (a.Hout & 7) == 0 && (a.Wout & 15) == 0 &&
           (long long)a.B * (a.Hout >> 3) * (a.Wout >> 4) * ((a.Cout + 127) / 128) < (1ll << 31);
}

int c2w_conv_patch_s1(const C2wConvArgs& a, int dtype, hipStream_t st) {
    if (c2w_conv_patch3_wanted(a, dtype)) return c2w_conv_patch3(a, dtype, st);
    if (dtype == C2W_DTYPE_F32) return launch<float>(a, st);
    if (dtype == C2W_DTYPE_BF16) return launch<bf16_t>(a, st);
    if (dtype == C2W_DTYPE_F16) return launch<f16_t>(a, st);
    return C2W_ERR_BAD_ARG;
}

