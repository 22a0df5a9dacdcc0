// Third-generation halo-patch kernel for the bf16 3x3 stride-1 convolutions: 32-channel stages, 16x16 tiles (or 8x16, three per CU).
//
// Ablation builds of this kernel (C2W_EXP, 128->128 @128^2, B = 128; 0.73 ms complete): 0.51 ms with the MFMAs removed,
// 0.42 ms with the fragment reads removed as well, 0.26 ms with the weight LDS-DMA removed too -- the matrix pipes are not
// the critical path; streaming all 295 KB of weights through LDS for every 128-pixel tile is (4.8 GB of L2 -> LDS per
// launch, 20 B/cycle/CU).  Three workgroups per CU on 8x16 tiles (TR = 8: 50 KB LDS, 168 VGPRs) therefore measured no
// faster than conv_patch_half_kernel; doubling the pixels per weight byte does:
//   * TR = 16: a workgroup owns 16x16 pixels x 128 output channels, 4 waves x (64 co x 128 px) = 128 accumulator VGPRs,
//     0.375 fragment reads per MFMA instead of 0.5, halo overhead 1.27 instead of 1.41, half the weight bytes per pixel;
//   * the patch row pitch is 20 pixels instead of 24 (18 x 20 x 128 B = 46,080 B; pieces run through the flattened pixel
//     index, 45 LDS-DMA pieces of 1 KiB);
//   * a stage is one tap x HALF a K chunk (32 channels): the weight ring is 3 x 8 KiB ([128 co][64 B] rows, XOR-swizzled
//     on the source address so that every ds_read_b128 lane group covers all 64 banks); 70.7 KB of LDS and
//     253 VGPRs => two workgroups per CU, one barrier per 32 MFMAs per wave;
//   * stages run kernel-column-major (half, kw, kh): the three taps of a column read the same pixel columns one row apart, so
//     the pixel fragments stay in registers across kh -- 22 fragment reads per three stages instead of 36 (A/B on one box,
//     128->128 @128^2: 0.606 vs 0.635 ms; the bias is loaded after the loop to make room: 16 VGPRs, no measurable cost).
// MFMA shape and the epilogue (conv_epilogue.h: bias / SiLU / pair / multiplier / residual / fused LayerNorm forward and
// backward, in two passes of 8 tile rows) are those of conv_patch_half_kernel; A = weights, B = pixels.
// Tried and abandoned: a persistent variant (workgroup walks tiles, next tile's first patch chunk requested during the epilogue,
// output staged behind the patch region): 141 spilled SGPRs -> 400-700 spilled VGPRs; the scalar state of two tiles plus the
// argument block does not fit.
// Measured and rejected on this kernel (C2W_T3V bits 1, 4): the LDS-DMA of stage s + 2 issued behind the fragment reads, or behind
// half of the stage's MFMAs, instead of right after the barrier: no change (0.636 / 0.639 vs 0.635 ms).  Also: a 4-slot weight ring (three stages of prefetch, 78.8 KB LDS) -1 %; pixel-fragment reads
// hoisted above the barrier -0.7 %; residual rows prefetched for both 8-row blocks right after staging -0.5 %; prefetching them
// next to live accumulators spills.
// Lesson kept in the code below: nothing may spill -- scratch loads return out of order with the LDS-DMA loads and break the
// counted vmcnt waits (seen as wrong weight rows at chunk boundaries with 40 spilled registers).
#include <cstdlib>

#include "conv_epilogue.h"

#ifndef C2W_T3V
#define C2W_T3V 10  // stage order / LDS-DMA placement / bias placement of conv_patch_t3_kernel (bits: see `stage` below); 10 = measured best
#endif
#ifndef C2W_T3_RING
#define C2W_T3_RING 3  // weight ring slots = stages of LDS-DMA prefetch + 1 (3: 70.7 KB of LDS, 4: 78.8 KB; both two workgroups per CU).
                       // 4 measured 7-15 % SLOWER with eight waves (profiles/r02_experiments.md): the L2 -> LDS latency (~0.76 us) is covered by two stages
#endif
#ifndef C2W_T3_DIRECT
#define C2W_T3_DIRECT 0  // elementwise epilogues of the 16x16 tile stored straight from the accumulators (t3_epi_direct)
#endif
#ifndef C2W_T3_SPLIT_EPI
#define C2W_T3_SPLIT_EPI 1  // one kernel instantiation per epilogue family (LayerNorm emission / LayerNorm backward / elementwise): by
                           // itself +-0; it is what lets the two-block operand prefetch below fit its registers
#endif
#ifndef C2W_T3_EPI2
#define C2W_T3_EPI2 14  // epilogue of the 16x16 tile: operand rows of both 8-row blocks requested up front.  Bit 0: in the all-in-one
                        // kernel (54 spilled registers, 11-17 % slower); bits 1-3: in the per-family instantiations (no spills; LayerNorm
                        // flavours 1-2.5 % faster, the step 49.09 -> 48.95 ms; profiles/r02_ab_conv_epilogues.txt)
#endif
#ifndef C2W_T3_PP
#define C2W_T3_PP 0  // two-group schedule of the 8-wave kernel (see `stage`): 1 groups = waves 0-3 / 4-7, 2 = even / odd waves
#endif
#ifndef C2W_T3_PRIO
#define C2W_T3_PRIO 0  // s_setprio 1 around a stage's MFMAs
#endif
#ifndef C2W_T3_STAGGER
#define C2W_T3_STAGGER 0  // x 8128 cycles: start delay of the second workgroup of every CU (see the kernel entry)
#endif
#ifndef C2W_T3_EPIPRIO
#define C2W_T3_EPIPRIO 0  // s_setprio N from the end of the MFMA loop on: the epilogue's VALU work competes with the CU's other workgroup's MFMAs for
                          // the SIMD's issue slots (stamps: 3 us to stage 64 accumulators); a higher priority shortens the epilogue
#endif
#ifndef C2W_T3_BIASLDS
#define C2W_T3_BIASLDS 0  // 1: the tile's 128 bias values are fetched by one LDS-DMA piece at kernel start (wave 0) into 1 KiB behind the loop's LDS
                          // and read from there after the loop -- no dependent global load at the head of the epilogue
#endif
#ifndef C2W_EXP
#define C2W_EXP 0  // diagnostic timing builds only (results are wrong): 1 no MFMA, 2 no LDS fragment reads, 4 no weight LDS-DMA, 8 no stage barrier
#endif             // in the loop, 32 no epilogue (accumulators reduced to one store per lane)

#if (C2W_T3_BIASLDS) == 2
__device__ unsigned int c2w_bias_dbg[16];  // [0] mismatching lanes, [1] lanes that read exactly 0 where the bias is not 0, [2..] a sample
extern "C" int c2w_bias_dbg_read(unsigned int* host16) { return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(c2w_bias_dbg), 64); }
extern "C" int c2w_bias_dbg_clear() { unsigned int z[16] = {}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(c2w_bias_dbg), z, 64); }
#endif
namespace {

#ifndef C2W_T3_NW
#define C2W_T3_NW 8  // waves per workgroup of the 16x16-tile kernel: 4 (wave tile 64 co x 128 px, 2 waves per SIMD) or 8 (64 co x 64 px, 4 per SIMD)
#endif
constexpr int T3_PW = 20;                     // patch row pitch in pixels (18 used)
constexpr int T3_WBYTES = 128 * 64;           // one stage of weights: 128 co x 32 ci
constexpr int T3_OS = 128 * 2 + 16;           // epilogue row stride

// TR = tile rows: 8 -> 8x16 pixels, three workgroups per CU; 16 -> 16x16 pixels (wave tile 64 co x 128 px), two per CU and
// half the weight bytes streamed per output pixel.
// NW = waves per workgroup: 2 (output-channel halves) x NW / 2 (pixel-row groups).  NW = 8 on the 16x16 tile: sixteen waves per CU,
// four per SIMD, 128 registers each -- while one of the CU's two workgroups is in its HBM-bound epilogue / next prologue the other
// still has TWO waves on every SIMD to keep the matrix pipe fed (one wave alone issues its LDS-DMA, its fragment reads and its
// barrier waits into the pipe's idle time: ~56 % busy; ablations in profiles/r02_experiments.md).
template <int TR, int NW = 4> struct T3Cfg {
    static constexpr int NTHR = 64 * NW;
    static constexpr int NPIECE = ((TR + 2) * T3_PW + 7) / 8;  // 1 KiB LDS-DMA pieces of 8 pixels: 25 / 45
    static constexpr int PBYTES = NPIECE * 1024;               // 25,600 / 46,080
    static constexpr int ROUNDS = (NPIECE + NW - 1) / NW;      // patch pieces per wave
    static constexpr int WPIECES = 8 / NW;                     // weight pieces per wave per stage (8 KiB per stage)
    static constexpr int NB = TR / (2 * NW);                   // 64-pixel blocks (4 tile rows) per wave
    static constexpr int NPASS = TR / 8;                       // epilogue passes of 128 tile pixels
    static constexpr int LDS_LOOP = PBYTES + C2W_T3_RING * T3_WBYTES;  // 50,176 / 70,656 with three slots
    static constexpr int LDS_EPI = TR * 16 * T3_OS + 512;      // output tile + LayerNorm column sums
    static constexpr int LDS_BIAS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;  // offset of the bias kilobyte (C2W_T3_BIASLDS)
    static constexpr int LDS = LDS_BIAS + ((C2W_T3_BIASLDS) ? 1024 : 0);
    static constexpr int WAVES_PER_SIMD = NW == 8 ? 4 : (TR == 8 ? 3 : 2);
    static_assert(NB >= 1 && 8 % NW == 0, "wave tiling");
};

template <int N> struct IC3 { static constexpr int value = N; };

// C2W_EXP & 16: constant-rate (100 MHz) timestamps of a workgroup's phases + where it ran, written through the otherwise unused
// second-output pointer (tools/stamp_conv3.py): [start, first patch landed, loop end, end, HW_ID, XCC_ID] per workgroup.
#if C2W_EXP & 16
__device__ unsigned long long* c2w_dbg3 = nullptr;
#define T3_STAMP(i) do { if (tid == 0) t3_stamp[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define T3_STAMP(i) (void)0
#endif

// XOR swizzles, derived for the lane groups ds_read_b128 is actually serviced in (MI355X_MICROARCH.md, LDS: {0-3,12-15,20-27},
// {4-11,16-19,28-31}, +32 -- NOT 16 consecutive lanes).  A first version assumed consecutive lanes and measured
// SQ_LDS_BANK_CONFLICT = 58 % of the LDS cycles; with these two functions the model gives zero conflicts for every tap.
//   patch pixel (128 B = 8 slots of 16 B): slot ^= col & 7
//   weight row (64 B = 4 slots):           slot ^= (-(row >> 2)) & 3
__device__ __forceinline__ uint32_t t3_pswz(int col) { return (uint32_t)(col & 7); }
__device__ __forceinline__ uint32_t t3_wswz(int row) { return (uint32_t)((4 - ((row >> 2) & 3)) & 3); }

// all but the wave's `WPIECES` youngest loads (the next stage's weight pieces) have landed; `more` false: everything
template <int WPIECES> __device__ __forceinline__ void t3_wait(bool more) {
    if (more) {
        if constexpr (WPIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
// the same with `ahead` = 0, 1 or 2 stages of weight pieces still allowed in flight (four-slot ring)
template <int WPIECES> __device__ __forceinline__ void t3_wait_n(int ahead) {
    if (ahead >= 2) {
        if constexpr (WPIECES == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        t3_wait<WPIECES>(ahead == 1);
    }
}


// ---- elementwise epilogues straight from the accumulators (C2W_T3_DIRECT) ------------------------------------------------------------
// In the MFMA layout a lane holds 4 consecutive output channels of one pixel (8 B in a 16-bit type); the four co-tiles of a wave are
// the four 32-B quarters of a pixel's 128-B half row.  Bias / activation / multiplier / residual / second output need nothing from
// another lane, so a wave can stream its 64 x 64 tile out as 8-B buffer stores the moment ITS MFMAs are done: no LDS staging, no
// workgroup barrier, early waves store while late ones still multiply.  L2 merges the four quarters (written back to back by one
// wave) into whole lines before they reach HBM.  Same arithmetic in the same order as epi_acc_to_lds + EpiStore::finish (including
// the rounding to the storage type between the two), so both paths give the same bits.
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2e_t;

template <typename T, int ACTK, int NB>
__device__ __forceinline__ void t3_epi_direct(const C2wConvArgs& p, f32x4_t (&acc)[NB][4][4], long long tile_off, int row0, int W, int co0, int co_l,
                                              int li) {  // co_l: the lane's first channel relative to the tile's first channel co0
    constexpr int ESZ = 2;
    const uint32_t span = 0x7ffffff0u;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc((char*)p.y + tile_off, span);
    const __amdgpu_buffer_rsrc_t ry2 = make_rsrc(p.y2 != nullptr ? (char*)p.y2 + tile_off : nullptr, p.y2 != nullptr ? span : 0u);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(p.res != nullptr ? (const char*)p.res + tile_off : nullptr, p.res != nullptr ? span : 0u);
    const __amdgpu_buffer_rsrc_t rmul = make_rsrc(p.mul != nullptr ? (const char*)p.mul + tile_off : nullptr, p.mul != nullptr ? span : 0u);
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.bias != nullptr ? p.bias + co0 : nullptr, p.bias != nullptr ? (uint32_t)(p.wrows - co0) * 4u : 0u);  // no bias / rows past wrows read 0
    const bool has_res = p.res != nullptr, has_mul = p.mul != nullptr;
    const uint32_t pitch = (uint32_t)p.ldy * ESZ;
    const uint32_t vlane = (uint32_t)li * pitch + (uint32_t)co_l * ESZ;
    uint32_t vo[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) vo[m] = co0 + co_l + m * 16 < p.Cout ? vlane + m * 32 : C2W_OOB;
    constexpr int NR = 4 * NB;
    u32x2e_t R[2][4], M[2][4];
    auto load_row = [&](int n, int buf) {
        const int soff = (row0 + n) * W * (int)pitch;
        if (has_res) {
#pragma unroll
            for (int m = 0; m < 4; ++m) R[buf][m] = __builtin_amdgcn_raw_buffer_load_b64(rres, vo[m], soff, 0);
        }
        if (has_mul) {
#pragma unroll
            for (int m = 0; m < 4; ++m) M[buf][m] = __builtin_amdgcn_raw_buffer_load_b64(rmul, vo[m], soff, 0);
        }
    };
    {   // the bias goes into the accumulators once (16 registers that would otherwise live through all rows)
        f32x4_t bv[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) bv[m] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rb, (co_l + m * 16) * 4, 0, 0));
        load_row(0, 0);
        load_row(1, 1);
#pragma unroll
        for (int n = 0; n < NR; ++n)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[n >> 2][m][n & 3] += bv[m];
    }
#pragma unroll
    for (int n = 0; n < NR; ++n) {
        const int soff = (row0 + n) * W * (int)pitch;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[n >> 2][m][n & 3][r];
                if constexpr (ACTK == 1) v[r] = silu_f(v[r]);
                if constexpr (ACTK == 2) v[r] = fmaxf(v[r], 0.f);
            }
            u32x2e_t pk = (u32x2e_t){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
            if (has_mul || has_res) {
                float f[4];
                unpack2<T>(pk[0], f[0], f[1]);
                unpack2<T>(pk[1], f[2], f[3]);
                if (has_mul) {
                    float gm[4];
                    unpack2<T>(M[n & 1][m][0], gm[0], gm[1]);
                    unpack2<T>(M[n & 1][m][1], gm[2], gm[3]);
                    if (p.mulmode == C2W_MUL_DSILU) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] *= dsilu_f(gm[r]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) f[r] *= gm[r];
                    }
                }
                if (has_res) {
                    float gr[4];
                    unpack2<T>(R[n & 1][m][0], gr[0], gr[1]);
                    unpack2<T>(R[n & 1][m][1], gr[2], gr[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) f[r] += gr[r];
                }
                pk = (u32x2e_t){pack2<T>(f[0], f[1]), pack2<T>(f[2], f[3])};
            }
            if ((p.act == C2W_ACT_SILU_PAIR || p.act == C2W_ACT_RELU_PAIR) && p.y2 != nullptr) {
                float a_[4], h_[4], d_[4];
                unpack2<T>(pk[0], a_[0], a_[1]);
                unpack2<T>(pk[1], a_[2], a_[3]);
                if (p.act == C2W_ACT_SILU_PAIR) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float sg = sigmoid_f(a_[r]);
                        h_[r] = a_[r] * sg;
                        d_[r] = sg + h_[r] * (1.0f - sg);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        h_[r] = fmaxf(a_[r], 0.f);
                        d_[r] = a_[r] > 0.f ? 1.f : 0.f;
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b64((u32x2e_t){pack2<T>(h_[0], h_[1]), pack2<T>(h_[2], h_[3])}, ry, vo[m], soff, 0);
                __builtin_amdgcn_raw_buffer_store_b64((u32x2e_t){pack2<T>(d_[0], d_[1]), pack2<T>(d_[2], d_[3])}, ry2, vo[m], soff, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b64(pk, ry, vo[m], soff, 0);
                if (p.y2 != nullptr) {
                    float f2[4];
                    unpack2<T>(pk[0], f2[0], f2[1]);
                    unpack2<T>(pk[1], f2[2], f2[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) f2[r] = silu_f(f2[r]);
                    __builtin_amdgcn_raw_buffer_store_b64((u32x2e_t){pack2<T>(f2[0], f2[1]), pack2<T>(f2[2], f2[3])}, ry2, vo[m], soff, 0);
                }
            }
        }
        if (n + 2 < NR) load_row(n + 2, n & 1);
    }
}

// DIRECT: the instantiation whose epilogue is t3_epi_direct (picked by the launcher for the flavours it covers; a kernel of its own
// because the register allocator, given both epilogues behind one loop, spills accumulators INSIDE the loop)
template <int TR, typename T = bf16_t, int NW = 4, int EPI = 0>  // T: bf16_t or f16_t (same bytes, other MFMA opcode and conversions)
__global__ __launch_bounds__(64 * NW, (T3Cfg<TR, NW>::WAVES_PER_SIMD)) void conv_patch_t3_kernel(const C2wConvArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    typedef T3Cfg<TR, NW> CF;
    constexpr int T3_NTHR = CF::NTHR;
    constexpr int ESZ = 2;
    constexpr int NB = CF::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [patch | W0 | W1 | W2]
#if C2W_T3_STAGGER
    // Two workgroups share a CU.  Dispatched together they run in phase -- both in their MFMA loops, then both in their HBM-bound
    // epilogue / next prologue -- and neither phase covers the other.  The second workgroup of every CU (the second batch of 256 in
    // dispatch order) starts a fraction of a tile period late; every later workgroup inherits the phase of the slot it takes over.
    if (TR == 16 && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < C2W_T3_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif

    const int tid = threadIdx.x;
#if C2W_EXP & 16
    unsigned long long t3_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    T3_STAMP(0);
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
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * 128;
    // H x W = the grid the tiles and the taps live on (= the output); with C2W_CONV_UP it is the nearest-neighbour x2 upsampling of the
    // Hin x Win source map, never materialised: patch pixel (ih, iw) is fetched from source pixel (ih >> 1, iw >> 1) (model/nn.py:184-189)
    const bool up = p.mode == C2W_CONV_UP;
    const int H = p.Hout, W = p.Wout, Ws = p.Win;
    const int tw = W >> 4, tpi = (H / TR) * tw;
    const int b = tm / tpi, tt = tm - b * tpi;
    const int ty = tt / tw, tx = tt - ty * tw;
    const int oh0 = ty * TR, ow0 = tx << 4;

    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESZ;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b * img_bytes, (uint32_t)img_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * 9 * p.Cin * ESZ));

    // patch pieces (rounds past the end repeat the last piece).  The source offsets are recomputed at every chunk start instead
    // of living in VGPRs through the loop: anything spilled would come back through scratch loads, which return out of order
    // with the LDS-DMA loads and break the counted vmcnt waits below (seen: wrong weight rows with 40 spilled registers).
    auto issue_patch = [&](int chunk) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));  // keeps the chunk-invariant part of the offsets from being hoisted out of the chunk loop (and spilled)
#pragma unroll
        for (int r = 0; r < CF::ROUNDS; ++r) {
            int pc = r * NW + wid;
            pc = pc < CF::NPIECE ? pc : CF::NPIECE - 1;
            const int f = pc * 8 + (lane_ >> 3);  // flattened patch pixel
            const int pr = f / T3_PW, px = f - pr * T3_PW;
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < 18 && pr < TR + 2;
            const uint32_t cg = (uint32_t)(lane_ & 7) ^ t3_pswz(px);
            const int spix = up ? (ih >> 1) * Ws + (iw >> 1) : ih * Ws + iw;
            const uint32_t voff = ok ? (uint32_t)(spix * p.Cin) * ESZ + (cg << 4) : C2W_OOB;
            glds16(rx, smem + pc * 1024, voff, (uint32_t)chunk * 128u);
        }
    };
    // weight stage: 128 rows x 4 slots of 16 B = 2 rounds; lane -> row = (round * 4 + wave) * 16 + lane / 4, slot = lane & 3
    constexpr bool PP = NW == 8 && (C2W_T3_PP) != 0;
    auto wvo_at = [&](int i) {  // two-group schedule: recomputed per use for the same reason as the pixel-fragment bases below
        int l = lane;
        if constexpr (PP || C2W_T3_RING == 4) asm volatile("" : "+v"(l));
        const int row = (i * NW + wid) * 16 + (l >> 2);
        const uint32_t cg = (uint32_t)(l & 3) ^ t3_wswz(row);
        return (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (cg << 4);
    };
    auto issue_w = [&](int chunk, int tap, int half, int wslot) {
        if constexpr ((C2W_EXP & 4) != 0) {
            if (chunk + tap + half > 0) return;
        }
        const uint32_t so = (uint32_t)(tap * p.Cin + chunk * 64 + half * 32) * ESZ;
#pragma unroll
        for (int i = 0; i < CF::WPIECES; ++i) glds16(rw, smem + CF::PBYTES + wslot * T3_WBYTES + (i * NW + wid) * 1024, wvo_at(i), so);
    };

    // fragment read offsets.  A: row = wm*64 + m*16 + li, its swizzle depends on (row >> 2) & 3 = (li >> 2) & 3 only, not on m, so
    // A[m] = offA + m * 1024; B: pixel (wn*4*NB + n + kh, li + kw): offB[kw] + (n + kh) * pitch; k-half 1 flips slot bit 2.
    const uint32_t offA_held = (uint32_t)(CF::PBYTES + (wm * 64 + li) * 64 + (((uint32_t)lg ^ t3_wswz(li)) << 4));
    auto offA_at = [&]() {
        if constexpr (C2W_T3_RING == 3) return offA_held;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int li_ = l & 15;
        return (uint32_t)(CF::PBYTES + (wm * 64 + li_) * 64 + (((uint32_t)(l >> 4) ^ t3_wswz(li_)) << 4));
    };
    // (experiment builds recompute the three pixel-fragment bases from the lane id where a stage needs one -- six VALU operations per
    // stage -- rather than hold them: the two-group schedule has no register left, and a spilled one comes back through scratch behind a
    // vmcnt(0))
    auto offB_at = [&](int kw) {
        int l = lane;
        if constexpr ((NW == 8 && (C2W_T3_PP) != 0) || C2W_T3_RING == 4) asm volatile("" : "+v"(l));  // default: hoisted and held
        const int px = (l & 15) + kw;
        return (uint32_t)((wn * 4 * NB * T3_PW + px) * 128 + (((uint32_t)(l >> 4) ^ t3_pswz(px)) << 4));
    };

    f32x4_t acc[NB][4][4];  // [pixel block of 4 rows][co tile][pixel row]
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[j][m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int nchunk = p.Cin / 64;
    const int NS = nchunk * 18;

    float bv[4][4];
    if constexpr (TR == 16 && (C2W_T3V & 8) == 0) epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);  // latency hidden by the loop; 16 VGPRs
    if constexpr ((C2W_T3_BIASLDS) != 0) {
        if (wid == 0) {  // oldest load of this wave: every counted wait below covers it
            const bool hb = p.bias != nullptr && co0 < p.wrows;
            const __amdgpu_buffer_rsrc_t rb = make_rsrc(hb ? (const void*)(p.bias + co0) : (const void*)p.w, hb ? (uint32_t)(p.wrows - co0) * 4u : 0u);
            glds16(rb, smem + CF::LDS_BIAS, lane < 32 ? (uint32_t)lane * 16u : C2W_OOB, 0u);
        }
    }
    issue_patch(0);
    issue_w(0, 0, 0, 0);  // stage 0 = (tap 0, half 0) in both stage orders
    if constexpr ((C2W_T3V & 2) != 0) issue_w(0, 3, 0, 1);  // kw-major: stage 1 = (kh 1, kw 0) = tap 3, half 0
    else issue_w(0, 0, 1, 1);
    if constexpr (C2W_T3_RING == 4) {
        if constexpr ((C2W_T3V & 2) != 0) issue_w(0, 6, 0, 2);  // stage 2 = (kh 2, kw 0)
        else issue_w(0, 1, 0, 2);
    }
    T3_STAMP(1);

    // stage s = chunk c x 18 + IDX; IDX -> (tap, half): weights of stage s live in ring slot s % 3 (18 % 3 == 0).
    //   C2W_T3V & 2 == 0: tap-major, IDX = 2 tap + half.
    //   C2W_T3V & 2     : IDX = half * 9 + kw * 3 + kh.  The three taps of one kernel column read the SAME pixel columns (li + kw)
    //     at rows n + kh, so the pixel fragments stay in registers across kh: 8 rows at kh = 0, one new row each at kh = 1, 2 --
    //     22 ds_read_b128 per three stages instead of 36 (LDS bytes read per MFMA 0.23 KB instead of 0.375 KB).
    //   C2W_T3V & 1: the LDS-DMA of stage s + 2 is issued behind the stage's fragment reads instead of in front of them;
    //   C2W_T3V & 4: behind the first half of the stage's MFMAs (the ring slot it fills was released by the stage's barrier).
#if C2W_EXP & 64
    uint32_t t3_dummy0 = lane;
#endif
    u32x4_t bq[4 * NB + 2];
    u32x4_t a_held[4];  // group 1 only: the weight fragments live across the barrier
    // C2W_T3_PP (NW = 8 only): the waves of a workgroup run in two groups, one per SIMD each.  Group 0 reads the fragments of stage
    // s and then runs its MFMAs; group 1 runs the MFMAs of stage s - 1 FIRST (from the registers it filled one barrier interval
    // earlier) and then reads the fragments of stage s (except around a chunk boundary).  Same barrier count, same ring discipline (everybody reads slot s % 3 inside
    // interval s), but inside an interval one wave of every SIMD is on the matrix pipe while the other waits for LDS -- a workgroup
    // that is alone in its loop (its CU neighbour is in the HBM-bound epilogue 2/3 of the time; tools/stamp_conv3.py) no longer
    // alternates "all eight waves read" / "all eight waves multiply".
    auto stage = [&](auto IDXc, int c, auto GRPc) {
        constexpr int IDX = decltype(IDXc)::value;
        constexpr int GRP = decltype(GRPc)::value;
        constexpr bool KWM = (C2W_T3V & 2) != 0;
        constexpr int HALF = KWM ? IDX / 9 : IDX % 2;
        constexpr int KW = KWM ? (IDX % 9) / 3 : (IDX / 2) % 3;
        constexpr int KH = KWM ? IDX % 3 : (IDX / 2) / 3;
        constexpr int PIDX = (IDX + 17) % 18;  // the stage before
        constexpr int PKH = KWM ? PIDX % 3 : (PIDX / 2) / 3;
        constexpr int RING = C2W_T3_RING, DIST = RING - 1;
        static_assert(RING == 3 || RING == 4, "weight ring depth");
        // ring slot of stage s = s % RING.  18 % 3 == 0: a compile-time slot; 18 % 4 == 2: odd chunks are two slots further
        const int WS = RING == 3 ? IDX % 3 : ((IDX + 2 * (c & 1)) & 3);
        const int s = c * 18 + IDX;
        auto issue_ahead = [&]() {  // weights of stage s + DIST
            constexpr int I2 = (IDX + DIST) % 18;
            constexpr int H2 = KWM ? I2 / 9 : I2 % 2;
            constexpr int T2 = KWM ? (I2 % 3) * 3 + (I2 % 9) / 3 : I2 / 2;
            issue_w(IDX + DIST < 18 ? c : c + 1, T2, H2, RING == 3 ? I2 % 3 : ((IDX + DIST + 2 * (c & 1)) & 3));
        };
        u32x4_t a_here[4];
        u32x4_t(&a)[4] = GRP == 1 ? a_held : a_here;
        auto mfmas = [&](auto KHc, auto hook) {
            constexpr int KH_ = decltype(KHc)::value;
            if constexpr ((C2W_T3_PRIO) != 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int n = 0; n < 4 * NB; ++n) {
                hook(n);
#if C2W_EXP & 64  // 16 independent VALU operations per stage next to the MFMAs: do they take matrix-pipe issue slots?
                asm volatile("v_add_u32 %0, %0, 1\n\tv_xor_b32 %0, 5, %0\n\tv_add_u32 %0, %0, 3\n\tv_xor_b32 %0, 9, %0" : "+v"(t3_dummy0));
#endif
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    if constexpr ((C2W_EXP & 1) == 0) {
                        acc[n >> 2][m][n & 3] = mfma16<T>(a[m], bq[n + KH_], acc[n >> 2][m][n & 3]);
                    } else {
                        asm volatile("" ::"v"(a[m]), "v"(bq[n + KH_]));
                    }
                }
            }
            if constexpr ((C2W_T3_PRIO) != 0) __builtin_amdgcn_s_setprio(0);
        };
        // everything but the next stage's (stages') weight pieces has landed
        if constexpr (RING == 3) t3_wait<CF::WPIECES>(s + 1 < NS);
        else t3_wait_n<CF::WPIECES>(NS - 1 - s);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // no fragment read crosses the barrier the ring is refilled behind (see the product kernel)
        if constexpr ((C2W_EXP & 8) == 0) __builtin_amdgcn_s_barrier();
        bool ahead = s + DIST < NS;
        if (IDX == 0 && c > 0) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            if (ahead) issue_ahead();
            ahead = false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // patch landed (once per chunk: no counting games here)
            __builtin_amdgcn_s_barrier();
        } else {
            if constexpr (GRP == 1 && IDX != 0) {
                mfmas(IC3<PKH>{}, [](int) {});
                __builtin_amdgcn_sched_barrier(0);  // the next fragments are read into the registers these MFMAs free, not next to them
            }
            if ((C2W_T3V & 5) == 0 && ahead) {
                issue_ahead();
                ahead = false;
            }
        }
        // C2W_T3V & 16 (with the column-major order): the pixel fragments do not depend on the stage's barrier (the patch is
        // static for the whole chunk), so the rows the NEXT stage needs are read during THIS stage's MFMAs, into the registers
        // of rows that have just died: only the four weight fragments are read between a barrier and its MFMAs.
        constexpr bool ROLL = KWM && (C2W_T3V & 16) != 0;
        static_assert(!(ROLL && GRP == 1), "the rolling pixel-fragment prefetch is not combined with the two-group schedule");
        constexpr int NXT = (IDX + 1) % 18, HALF_N = NXT / 9, KW_N = (NXT % 9) / 3;  // next stage (column-major order)
        const uint32_t offA = offA_at() + (uint32_t)(WS * T3_WBYTES);
        const uint32_t offB_kw = offB_at(KW) ^ (uint32_t)(HALF * 64);
        auto rowp = [&](int kw, int half, int row) {
            const uint32_t o = kw == KW && half == HALF ? offB_kw : (offB_at(kw) ^ (uint32_t)(half * 64));
            return (const u32x4_t*)(smem + o + row * T3_PW * 128);
        };
        if constexpr ((C2W_EXP & 2) == 0) {
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const u32x4_t*)(smem + offA + m * 1024);
            if constexpr (!KWM) {
#pragma unroll
                for (int n = 0; n < 4 * NB; ++n) bq[n + KH] = *rowp(KW, HALF, n + KH);
            } else if constexpr (!ROLL) {
                if constexpr (KH == 0) {
#pragma unroll
                    for (int n = 0; n < 4 * NB; ++n) bq[n] = *rowp(KW, HALF, n);
                } else {
                    bq[4 * NB - 1 + KH] = *rowp(KW, HALF, 4 * NB - 1 + KH);
                }
            } else {
                if constexpr (IDX == 0) {  // first stage of a chunk: nothing was prefetched (the patch has only just landed)
#pragma unroll
                    for (int n = 0; n < 4 * NB; ++n) bq[n] = *rowp(KW, HALF, n);
                }
                if constexpr (KH < 2) bq[4 * NB + KH] = *rowp(KW, HALF, 4 * NB + KH);  // the one new row of stage kh + 1
            }
        } else {
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = (u32x4_t){offA, (uint32_t)s, 3u, 4u};
#pragma unroll
            for (int n = 0; n < 4 * NB; ++n) bq[n + KH] = (u32x4_t){offB_kw, (uint32_t)s, 5u, 6u};
        }
        if ((C2W_T3V & 1) != 0 && ahead) {
            issue_ahead();
            ahead = false;
        }
        // group 1 holds nothing across a chunk boundary (the patch refill there needs the registers): its last stage runs in place
        if constexpr (GRP == 1 && IDX != 17) return;
        u32x4_t bn[4 * NB];
        constexpr bool PREF = ROLL && KH == 2 && IDX != 17 && (C2W_EXP & 2) == 0;  // next stage = first of the next kernel column
        if constexpr (PREF) {  // rows 0 and 1 died with stage kh = 1
            bn[0] = *rowp(KW_N, HALF_N, 0);
            bn[1] = *rowp(KW_N, HALF_N, 1);
        }
        mfmas(IC3<KH>{}, [&](int n) {
            if ((C2W_T3V & 4) != 0 && n == 2 * NB && ahead) issue_ahead();
            if constexpr (PREF) {
                if (n >= 1 && n + 1 < 4 * NB) bn[n + 1] = *rowp(KW_N, HALF_N, n + 1);  // row n + 1 of this column has just been used last
            }
        });
        if constexpr (PREF) {
#pragma unroll
            for (int n = 0; n < 4 * NB; ++n) bq[n] = bn[n];
        }
    };
    // (the chunk loops are spelled out: wrapped in a generic lambda the same code allocates 8 registers more and spills)
#define T3_CHUNK_LOOP(G)                                                                                                                     \
    _Pragma("unroll 1") for (int c = 0; c < nchunk; ++c) {                                                                                   \
        stage(IC3<0>{}, c, G); stage(IC3<1>{}, c, G); stage(IC3<2>{}, c, G); stage(IC3<3>{}, c, G); stage(IC3<4>{}, c, G); stage(IC3<5>{}, c, G); \
        stage(IC3<6>{}, c, G); stage(IC3<7>{}, c, G); stage(IC3<8>{}, c, G); stage(IC3<9>{}, c, G); stage(IC3<10>{}, c, G); stage(IC3<11>{}, c, G); \
        stage(IC3<12>{}, c, G); stage(IC3<13>{}, c, G); stage(IC3<14>{}, c, G); stage(IC3<15>{}, c, G); stage(IC3<16>{}, c, G); stage(IC3<17>{}, c, G); \
    }
    if constexpr (PP) {
        const int grp = (C2W_T3_PP) == 2 ? (wid & 1) : (wid >> 2);
        if (grp != 0) {
            T3_CHUNK_LOOP(IC3<1>{})
        } else {
            T3_CHUNK_LOOP(IC3<0>{})
        }
    } else {
        T3_CHUNK_LOOP(IC3<0>{})
    }
#undef T3_CHUNK_LOOP
    T3_STAMP(2);
#if C2W_EXP & 64
    if (t3_dummy0 == 0x12345u) ((uint32_t*)p.y)[tid] = t3_dummy0;
#endif

    if constexpr ((C2W_EXP & 32) != 0) {
        f32x4_t t = acc[0][0][0];
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) t += acc[j][m][n];
        if (t[0] + t[1] + t[2] + t[3] == 12345.678f) ((float*)p.y)[tid] = t[0];
        return;
    }
    constexpr bool DIRECT = EPI == 1;
    if constexpr (DIRECT) {
        static_assert(TR == 16, "16x16 tiles only");
        int lane_d = tid;
        asm volatile("" : "+v"(lane_d));
        lane_d &= 63;
        const long long tile_off = ((((long long)b * H + oh0) * W + ow0) * p.ldy + co0) * ESZ;
        const int co_l = wm * 64 + (lane_d >> 4) * 4;  // relative to the tile's first channel
        t3_epi_direct<T, 0, NB>(p, acc, tile_off, wn * 4 * NB, W, co0, co_l, lane_d & 15);
        return;
    }
    if constexpr ((C2W_T3_EPIPRIO) != 0) __builtin_amdgcn_s_setprio(C2W_T3_EPIPRIO);
    // epilogue: the residual / multiplier rows are fetched AFTER the accumulators have left the registers (the half-tile
    // kernel prefetches them next to live accumulators; that does not fit here) -- the co-resident workgroups cover the
    // exposed latency.  Output rows go through LDS in blocks of 128 (= 8 tile rows), one EpiStore pass each.
    // the epilogue's lane coordinates are derived afresh: kept across the loop they cost a register the 128-register variant does not have
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, li_e = lane_e & 15, lg_e = lane_e >> 4;
    if constexpr ((C2W_T3_BIASLDS) != 0) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x4_t t = *(const f32x4_t*)(smem + CF::LDS_BIAS + (wm * 64 + m * 16 + lg_e * 4) * 4);
            bv[m][0] = t[0]; bv[m][1] = t[1]; bv[m][2] = t[2]; bv[m][3] = t[3];
        }
#if (C2W_T3_BIASLDS) == 2
        {   // detector: the same values by the global path; count what the LDS copy got wrong, then use the global ones
            float bg[4][4];
            epi_load_bias(p, co0 + wm * 64 + lg_e * 4, bg);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (__float_as_uint(bg[m][r]) != __float_as_uint(bv[m][r])) {
                        atomicAdd(&c2w_bias_dbg[0], 1u);
                        if (bv[m][r] == 0.f) atomicAdd(&c2w_bias_dbg[1], 1u);
                        c2w_bias_dbg[2] = __float_as_uint(bv[m][r]);
                        c2w_bias_dbg[3] = __float_as_uint(bg[m][r]);
                        c2w_bias_dbg[4] = blockIdx.x;
                        c2w_bias_dbg[5] = (unsigned)tid_e;
                        c2w_bias_dbg[6] = gridDim.x;
                    }
                    bv[m][r] = bg[m][r];
                }
        }
#endif
    } else if constexpr (TR == 8 || (C2W_T3V & 8) != 0) epi_load_bias(p, co0 + wm * 64 + lg_e * 4, bv);
    __syncthreads();
    T3_STAMP(4);
    char* const O = smem;
    float* const red = (float*)(smem + TR * 16 * T3_OS);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        epi_acc_to_lds<T>(O, T3_OS, acc[j], bv, p.act, wm * 64, (wn * NB + j) * 64, li_e, lg_e);
#if C2W_EXP & 16
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    T3_STAMP(7);
#endif
#if C2W_T3_EPI2
    // both 8-row blocks' residual / multiplier rows are requested before the first block is finished: the second block's HBM latency
    // runs behind the first block's arithmetic and stores (the accumulators have left the registers, so both sets fit)
    if constexpr (CF::NPASS == 2 && (((C2W_T3_EPI2) >> (EPI == 0 ? 0 : EPI - 1)) & 1) != 0) {  // C2W_T3_EPI2: bit 0 the all-in-one kernel, bits 1 / 2 / 3 the EPI = 2 / 3 / 4 instantiations
        const bool pool2 = EPI == 0 && (p.flags & C2W_CONV_POOL2) != 0;
        EpiStore<T, 128, T3_NTHR> est0, est1;
        if (p.ln_x != nullptr && tid_e < 128) red[tid_e] = 0.f;
        if (!pool2) {
            est0.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0) * W + ow0, W);
            est1.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0 + 8) * W + ow0, W);
        }
        __syncthreads();
        T3_STAMP(5);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto& est = h == 0 ? est0 : est1;
            const char* const Oh = O + h * 128 * T3_OS;
            if constexpr (EPI == 2) est.finish_lnf(p, Oh, T3_OS, tid_e, b);
            else if constexpr (EPI == 3) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
            else if constexpr (EPI == 4) est.finish(p, Oh, T3_OS, tid_e);
            else if (pool2) est.finish_pool2(p, Oh, T3_OS, tid_e, co0, ((long long)b * (H >> 1) + ((oh0 + 8 * h) >> 1)) * (W >> 1) + (ow0 >> 1), W >> 1);
            else if (p.ln_x != nullptr) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
            else if (p.lnf_y != nullptr) est.finish_lnf(p, Oh, T3_OS, tid_e, b);
            else est.finish(p, Oh, T3_OS, tid_e);
            if (h == 0 && (EPI == 3 || (EPI == 0 && p.ln_x != nullptr))) {  // the LayerNorm column sums are re-zeroed for the second block only after everyone read them
                __syncthreads();
                if (tid_e < 128) red[tid_e] = 0.f;
                __syncthreads();
            }
            if (h == 0) T3_STAMP(6);
        }
    } else
#endif
    {
#pragma unroll
        for (int h = 0; h < CF::NPASS; ++h) {
        if (p.ln_x != nullptr && tid_e < 128) red[tid_e] = 0.f;
        EpiStore<T, 128, T3_NTHR> est;
        const bool pool2 = EPI == 0 && (p.flags & C2W_CONV_POOL2) != 0;
        if (!pool2) est.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0 + 8 * h) * W + ow0, W);
        __syncthreads();
        T3_STAMP(5 + h);
        const char* const Oh = O + h * 128 * T3_OS;
        // EPI 2 / 3 / 4: instantiations that carry one epilogue only (picked by the launcher, C2W_T3_SPLIT_EPI)
        if constexpr (EPI == 2) est.finish_lnf(p, Oh, T3_OS, tid_e, b);
        else if constexpr (EPI == 3) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
        else if constexpr (EPI == 4) est.finish(p, Oh, T3_OS, tid_e);
        else if (pool2) est.finish_pool2(p, Oh, T3_OS, tid_e, co0, ((long long)b * (H >> 1) + ((oh0 + 8 * h) >> 1)) * (W >> 1) + (ow0 >> 1), W >> 1);
        else if (p.ln_x != nullptr) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
        else if (p.lnf_y != nullptr) est.finish_lnf(p, Oh, T3_OS, tid_e, b);
        else est.finish(p, Oh, T3_OS, tid_e);
        if (h + 1 < CF::NPASS) __syncthreads();  // the LayerNorm column sums are re-zeroed for the next block only after everyone read them
    }
    }
#if C2W_EXP & 16
    T3_STAMP(3);
    if (tid == 0 && c2w_dbg3 != nullptr) {
        unsigned long long* d = c2w_dbg3 + (size_t)blockIdx.x * 16;
        d[0] = t3_stamp[0]; d[1] = t3_stamp[1]; d[2] = t3_stamp[2]; d[3] = t3_stamp[3];
        d[4] = __builtin_amdgcn_s_getreg(63492);  // HW_ID
        d[5] = __builtin_amdgcn_s_getreg(63508);  // XCC_ID
        d[6] = t3_stamp[4]; d[7] = t3_stamp[5];   // epilogue: all waves out of the loop / accumulators staged, block 0 about to be stored
        d[8] = t3_stamp[6]; d[9] = t3_stamp[7];                      // block 1 about to be stored
    }
#endif
}

// the flavours t3_epi_direct covers: everything elementwise whose activation is none or one of the pair forms
inline bool t3_direct_flavour(const C2wConvArgs& a) {
    return a.ln_x == nullptr && a.lnf_y == nullptr && (a.flags & C2W_CONV_POOL2) == 0 && a.act != C2W_ACT_SILU && a.act != C2W_ACT_RELU;
}

template <int TR, typename T, int NW, int EPI>
int t3_launch_as(const C2wConvArgs& a, hipStream_t st) {
    typedef T3Cfg<TR, NW> CF;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_t3_kernel<TR, T, NW, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hout / TR) * (a.Wout >> 4);
    conv_patch_t3_kernel<TR, T, NW, EPI><<<nM * nN, CF::NTHR, CF::LDS, st>>>(a);
    return (int)hipGetLastError();
}

template <int TR, typename T, int NW>
int t3_launch(const C2wConvArgs& a, hipStream_t st) {
#if C2W_T3_DIRECT
    static const bool off = getenv("C2W_T3_NO_DIRECT") != nullptr;
    if (TR == 16 && !off && t3_direct_flavour(a)) return t3_launch_as<16, T, NW, 1>(a, st);
#endif
#if C2W_T3_SPLIT_EPI
    if (TR == 16 && (a.flags & C2W_CONV_POOL2) == 0) {
        if (a.lnf_y != nullptr) return t3_launch_as<16, T, NW, 2>(a, st);
        if (a.ln_x != nullptr) return t3_launch_as<16, T, NW, 3>(a, st);
        return t3_launch_as<16, T, NW, 4>(a, st);
    }
#endif
    return t3_launch_as<TR, T, NW, 0>(a, st);
}

}  // namespace

#if C2W_EXP & 16
extern "C" int c2w_debug_set3(void* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(c2w_dbg3), &ptr, sizeof(void*)); }
#endif

// The 16x16-tile variant pays off where the launch still fills the chip several times over (measured on MI355X, B = 128:
// 128->128 @128^2 0.596 vs 0.621 ms, @64^2 0.156 vs 0.166 ms; 384->384 @16^2 with 384 workgroups 0.093 vs 0.079 ms).
// C2W_CONV_T3 = 0 disables it, = 16 forces it wherever the image is tiled by 16x16.  The 8-row instantiation (three
// workgroups per CU) measured no faster than conv_patch_half_kernel and is not dispatched.
bool c2w_conv_patch3_wanted(const C2wConvArgs& a, int dtype) {
    static const int mode = getenv("C2W_CONV_T3") ? atoi(getenv("C2W_CONV_T3")) : -1;
    if ((dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16) || mode == 0 || (a.Hout & 15) != 0 || (a.Wout & 15) != 0) return false;
    const long long wgs = (long long)a.B * (a.Hout >> 4) * (a.Wout >> 4) * ((a.Cout + 127) / 128);
    return mode == 16 || wgs >= 1024;
}

int c2w_conv_patch3(const C2wConvArgs& a, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? t3_launch<16, f16_t, C2W_T3_NW>(a, st) : t3_launch<16, bf16_t, C2W_T3_NW>(a, st);
}
