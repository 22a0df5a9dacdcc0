#ifndef C2W_T3_PLACE
#define C2W_T3_PLACE 0
#endif
// Halo-patch kernel for the 16-bit 3x3 stride-1 convolutions on 16x16-pixel tiles: 32-channel stages, 8 waves per workgroup.
// This is the dominant kernel of the training step (forward and input gradient of every residual-block conv and of the up-convs at
// the 128^2 / 64^2 / 32^2 levels: model/nn.py:146-159,183-189).
//
// Why this shape (ablation builds of the kernel at 128->128 @128^2, B = 128; profiles/r01_experiments.md, r02_experiments.md): with the
// MFMAs removed an 8x16-tile kernel still took 0.51 of 0.73 ms, of which 0.16 ms was streaming all 295 KB of weights through LDS for
// every 128-pixel tile (4.8 GB of L2 -> LDS per launch) -- so:
//   * a workgroup owns 16x16 pixels x 128 output channels, 8 waves x (64 co x 64 px) = 64 accumulator VGPRs, 128 registers per wave,
//     four waves per SIMD; half the weight bytes streamed per output pixel, halo overhead 1.27 instead of 1.41;
//   * the patch row pitch is 20 pixels (18 x 20 x 128 B = 46,080 B; pieces run through the flattened pixel index, 45 LDS-DMA pieces
//     of 1 KiB), staged ONCE per 64-channel K chunk, all nine taps read it at immediate LDS offsets;
//   * a stage is one tap x HALF a K chunk (32 channels); the loop body is the nine taps of one half, so K is walked in 32-channel
//     steps and a caller's promise that trailing channels are zero (kvalid: the padded edge convs) ends it early; the weight ring is 3 x 8 KiB ([128 co][64 B] rows, XOR-swizzled on the
//     source address so that every ds_read_b128 lane group covers all 64 banks), filled two stages ahead by LDS-DMA with counted
//     vmcnt waits; 71.7 KB of LDS => two workgroups per CU, one in its epilogue while the other multiplies; one barrier per stage;
//   * stages run kernel-column-major (half, kw, kh): the three taps of a column read the same pixel columns one row apart, so the
//     pixel fragments stay in registers across kh -- 22 fragment reads per three stages instead of 36;
//   * the bias reaches LDS by one LDS-DMA piece at kernel start and is read after the loop (16 VGPRs the loop does not have, and no
//     dependent global load at the head of the epilogue);
//   * one instantiation per epilogue family (LayerNorm emission / LayerNorm backward / elementwise / 2x2-pooled): in the smaller
//     kernels the residual / multiplier rows of BOTH 8-row blocks are requested before the first block is finished.
// MFMA shape and the epilogue (conv_epilogue.h) are those of conv_patch_half_kernel; A = weights, B = pixels.
// Nothing may spill: scratch loads return out of order with the LDS-DMA loads and break the counted vmcnt waits (seen once as wrong
// weight rows at chunk boundaries with 40 spilled registers; build.py refuses to link such an object: isa_checks.py).
//
// The schedule / epilogue experiments that were measured and rejected (two-group wave schedule, four-slot ring, direct stores from
// the accumulators, staggered workgroup start, LDS-DMA placement variants, the 4-wave and 8-row-tile forms), the ablation and
// timestamp builds live in lab/csrc/conv_patch3_lab.hip (repo root, not shipped), which lab/build_variant.sh compiles INSTEAD of this file;
// their results: profiles/r02_experiments.md section 2, profiles/r03_experiments.md.
#include <cstdlib>

#include "conv_epilogue.h"
#include "knobs.h"

namespace {

constexpr int T3_WAVES = 8;  // waves per workgroup: wave tile 64 co x 64 px, 4 waves per SIMD (the 4-wave form -- 64 co x 128 px, 2 per SIMD --
                             // is equal on isolated launches and 0.7 % behind inside the step)
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
    static constexpr int LDS_LOOP = PBYTES + 3 * T3_WBYTES;  // 50,176 / 70,656 with three slots
    static constexpr int LDS_EPI = TR * 16 * T3_OS + 512;      // output tile + LayerNorm column sums
    static constexpr int LDS_BIAS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;  // the tile's 128 bias values: one LDS-DMA piece (1 KiB) behind everything else
    static constexpr int LDS = LDS_BIAS + 1024;
    static constexpr int WAVES_PER_SIMD = NW == 8 ? 4 : (TR == 8 ? 3 : 2);
    static_assert(NB >= 1 && 8 % NW == 0, "wave tiling");
};

template <int N> struct IC3 { static constexpr int value = N; };


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

// WPK: the weights are the stage-major packed copy (C2W_CONV_WPACKED) -- its own instantiation: with both addressing forms in one
// kernel (a select, or a uniform branch) either the fp16 or the bf16 build spilled 2-3 registers
template <int TR, typename T = bf16_t, int NW = 4, int EPI = 0, bool WPK = false>  // T: bf16_t or f16_t (same bytes, other MFMA opcode and conversions)
__global__ __launch_bounds__(64 * NW, (T3Cfg<TR, NW>::WAVES_PER_SIMD)) void conv_patch_t3_kernel(const C2wConvArgs p) {
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    typedef T3Cfg<TR, NW> CF;
    constexpr int T3_NTHR = CF::NTHR;
    constexpr int ESZ = 2;
    constexpr int NB = CF::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [patch | W0 | W1 | W2]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    // EPI = 5: the elementwise family for at most 80 weight rows (the network's output conv: 65 rows, stored in rows of 128 channels).  The two 64-channel halves go to
    // waves 0-3 / 4-7 instead of even / odd waves -- every SIMD hosts one of each -- and waves 4-7 (`light`) compute only their first
    // 16-channel block; the weight pieces of channels 80-127 (waves 5-7) are not fetched: 20 instead of 32 MFMAs per SIMD and stage,
    // 5 of 8 KiB of weights per stage.
    // EPI = 7 (round 6): EPI 5 with the training loss fused -- the tile a = conv + bias (rounded to the storage type, in LDS) becomes
    // dY = (a - eps) * gscale in place, eps read from the half-precision noise rows the input conversion kept, sum (a - eps)^2 added to
    // loss_sum (C2wConvArgs.loss_*): the prediction is never written and the loss tail's own pass (0.36 ms per step) is gone.  (A first
    // version regenerated eps from the Philox stream here: 1.06 ms for the launch against 0.42 + 0.36 unfused -- ten dependent
    // multiply rounds on four waves per SIMD are latency-bound; profiles/r06_experiments.md.)
    constexpr bool NARROW = EPI == 5 || EPI == 7;
    const int wm = NARROW ? wid >> 2 : wid & 1, wn = NARROW ? wid & 3 : wid >> 1;
    const bool light = NARROW && wid >= 4;

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
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)(WPK ? nN * 128 : p.wrows) * 9 * p.Cin * ESZ));

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
    // C2W_CONV_WPACKED: the stage-major copy (c2w_pack_conv_weights_batched): the 128 rows x 64 B of a (tap, 32-channel half) lie in one
    // 8 KiB block, swizzle baked in -- a piece is 1 KiB of consecutive bytes (8 cache lines) instead of 16 half-used lines
    constexpr bool wpk = WPK;
    auto wvo_at = [&](int i) {
        int l = lane;
        const int row = (i * NW + wid) * 16 + (l >> 2);
        const uint32_t cg = (uint32_t)(l & 3) ^ t3_wswz(row);
        return (uint32_t)(co0 + row) * (uint32_t)(9 * p.Cin * ESZ) + (cg << 4);
    };
    auto issue_w = [&](int chunk, int tap, int half, int wslot) {
        if (NARROW && wid >= 5) return;  // pieces 5-7 = output channels 80-127: nobody reads them
        if constexpr (wpk) {
            const uint32_t so = (uint32_t)(((tap * (p.Cin >> 5) + chunk * 2 + half) * (nN * 128) + co0) * 64);
#pragma unroll
            for (int i = 0; i < CF::WPIECES; ++i)
                glds16(rw, smem + CF::PBYTES + wslot * T3_WBYTES + (i * NW + wid) * 1024, (uint32_t)(i * NW * 1024 + tid * 16), so);
            return;
        }
        const uint32_t so = (uint32_t)(tap * p.Cin + chunk * 64 + half * 32) * ESZ;
#pragma unroll
        for (int i = 0; i < CF::WPIECES; ++i) glds16(rw, smem + CF::PBYTES + wslot * T3_WBYTES + (i * NW + wid) * 1024, wvo_at(i), so);
    };

    // fragment read offsets.  A: row = wm*64 + m*16 + li, its swizzle depends on (row >> 2) & 3 = (li >> 2) & 3 only, not on m, so
    // A[m] = offA + m * 1024; B: pixel (wn*4*NB + n + kh, li + kw): offB[kw] + (n + kh) * pitch; k-half 1 flips slot bit 2.
    const uint32_t offA_held = (uint32_t)(CF::PBYTES + (wm * 64 + li) * 64 + (((uint32_t)lg ^ t3_wswz(li)) << 4));
    // Register budget (128 VGPRs = four waves per SIMD = two workgroups per CU): the A-fragment base is HELD in a register except in the
    // packed-weights LayerNorm-backward instantiation, which recomputes it per stage from the lane id (held, that instantiation needs
    // 130).  Neither a spill nor a 129th register can ship: climate2weather_amd/build.py refuses to link this translation unit if any
    // kernel with hand-counted vmcnt waits uses scratch or if a conv_patch_t3 instantiation exceeds 128 VGPRs (isa_checks.py).
    auto offA_at = [&]() {
        if constexpr (!(WPK && (EPI == 3 || EPI == 6))) return offA_held;  // packed LayerNorm backward: recomputed per stage (held, it spills 2 registers)
        int l = lane;
        asm volatile("" : "+v"(l));
        const int li_ = l & 15;
        return (uint32_t)(CF::PBYTES + (wm * 64 + li_) * 64 + (((uint32_t)(l >> 4) ^ t3_wswz(li_)) << 4));
    };
    auto offB_at = [&](int kw) {
        int l = lane;
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
    // K extent in 32-channel halves of the 64-channel chunks: all of Cin, or the caller's promise that channels >= kvalid are zero in
    // x or in w (C2wConvArgs.kvalid: the padded edge convs at C = 65 visit 3 halves instead of 4)
    const int kv = p.kvalid > 0 && p.kvalid < p.Cin ? p.kvalid : p.Cin;
    const int nhalf = (kv + 31) >> 5;
    const int NS = nhalf * 9;

    float bv[4][4];
    if (wid == 0) {  // the bias of the tile's 128 output channels -> LDS, by the wave's OLDEST load (every counted wait below covers it): the
                     // epilogue reads it from there instead of starting with a dependent global load (isolated launches 0-2.8 % faster)
        const bool hb = p.bias != nullptr && co0 < p.wrows;  // no bias / rows past wrows read as zero (out of the descriptor's range)
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(hb ? (const void*)(p.bias + co0) : (const void*)p.w, hb ? (uint32_t)(p.wrows - co0) * 4u : 0u);
        glds16(rb, smem + CF::LDS_BIAS, lane < 32 ? (uint32_t)lane * 16u : C2W_OOB, 0u);
    }
    issue_patch(0);
    issue_w(0, 0, 0, 0);  // stage 0 = (tap 0, half 0)
    issue_w(0, 3, 0, 1);  // stage 1 = (kh 1, kw 0) = tap 3, half 0

    // stage s = hc x 9 + IDX: hc = 2 x chunk + half (the 32-channel half of a 64-channel K chunk), IDX = kw * 3 + kh (kernel-column-
    // major); its weights live in ring slot s % 3 = IDX % 3 (9 % 3 == 0).  The three taps of one kernel column read the SAME pixel
    // columns (li + kw) at rows n + kh, so the pixel fragments stay in registers across kh: 4 rows at kh = 0, one new row each at
    // kh = 1, 2 -- 22 ds_read_b128 per three stages instead of 36 (LDS bytes read per MFMA 0.23 KB instead of 0.375 KB).
    u32x4_t bq[4 * NB + 2];
    auto stage = [&](auto IDXc, int hc) {
        constexpr int IDX = decltype(IDXc)::value;
        constexpr int KW = IDX / 3;
        constexpr int KH = IDX % 3;
        constexpr int WS = IDX % 3;  // ring slot of stage s = s % 3: a compile-time slot
        const int s = hc * 9 + IDX;
        const int half = hc & 1, c = hc >> 1;
        auto issue_ahead = [&]() {  // weights of stage s + 2
            constexpr int I2 = (IDX + 2) % 9;
            constexpr int T2 = (I2 % 3) * 3 + I2 / 3;
            const int h2 = IDX + 2 < 9 ? hc : hc + 1;
            issue_w(h2 >> 1, T2, h2 & 1, I2 % 3);
        };
        u32x4_t a[4];
        auto mfmas = [&](auto KHc, auto hook) {
            constexpr int KH_ = decltype(KHc)::value;
            if constexpr (NARROW) {
#pragma unroll
                for (int n = 0; n < 4 * NB; ++n) acc[n >> 2][0][n & 3] = mfma16<T>(a[0], bq[n + KH_], acc[n >> 2][0][n & 3]);
                if (!light) {  // wave-uniform
#pragma unroll
                    for (int n = 0; n < 4 * NB; ++n)
#pragma unroll
                        for (int m = 1; m < 4; ++m) acc[n >> 2][m][n & 3] = mfma16<T>(a[m], bq[n + KH_], acc[n >> 2][m][n & 3]);
                }
                return;
            }
#pragma unroll
            for (int n = 0; n < 4 * NB; ++n) {
                hook(n);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    acc[n >> 2][m][n & 3] = mfma16<T>(a[m], bq[n + KH_], acc[n >> 2][m][n & 3]);
                }
            }
        };
        // everything but the next stage's weight piece has landed
        t3_wait<CF::WPIECES>(s + 1 < NS);
        // ... and every LDS read this wave has issued is in its registers.  The slot read in stage s - 1 is refilled right behind this
        // barrier (issue_ahead below, by whichever wave gets there first), the patch buffer behind the barrier of a chunk start: a
        // fragment read still queued in the LDS when its wave arrives here can be overtaken by that refill.  hipcc places the reads'
        // own lgkmcnt waits at their first use, and it sinks MFMAs below the barrier -- with the packed 16-bit conversions of round 3
        // it left stage 5's last weight fragment (a[3], first used after the barrier) outstanding across the barrier of stage 6, and
        // about one forward in 200 under four concurrent streams computed (tap 7, m = 3) with tap 8's weights for one wave
        // (profiles/r03_experiments.md "weight ring race"; build.py / isa_checks.py checks every barrier of every LDS-DMA
        // kernel in the ISA).  Measured cost of the explicit wait: none (step 48.45 ms either way).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        bool ahead = s + 2 < NS;
        if (IDX == 0 && half == 0 && c > 0) {  // single patch buffer: every wave is past the previous chunk only now
            issue_patch(c);
            if (ahead) issue_ahead();
            ahead = false;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // patch landed (once per chunk: no counting games here)
            __builtin_amdgcn_s_barrier();
        } else {
            if (ahead && (NARROW || C2W_T3_PLACE == 0)) {
                issue_ahead();
                ahead = false;
            }
        }
        const uint32_t offA = offA_at() + (uint32_t)(WS * T3_WBYTES);
        const uint32_t offB_kw = offB_at(KW) ^ (uint32_t)(half * 64);
        auto rowp = [&](int row) { return (const u32x4_t*)(smem + offB_kw + row * T3_PW * 128); };
        a[0] = *(const u32x4_t*)(smem + offA);
        if (!light) {
#pragma unroll
            for (int m = 1; m < 4; ++m) a[m] = *(const u32x4_t*)(smem + offA + m * 1024);
        }
        {
            if constexpr (KH == 0) {
#pragma unroll
                for (int n = 0; n < 4 * NB; ++n) bq[n] = *rowp(n);
            } else {
                bq[4 * NB - 1 + KH] = *rowp(4 * NB - 1 + KH);
            }
        }
        mfmas(IC3<KH>{}, [&](int n) {
            if (C2W_T3_PLACE > 0 && n == C2W_T3_PLACE && ahead) {
                __builtin_amdgcn_sched_barrier(0);
                issue_ahead();
                ahead = false;
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };
    // (the loop body is spelled out: wrapped in a generic lambda the same code allocates 8 registers more and spills)
#pragma unroll 1
    for (int hc = 0; hc < nhalf; ++hc) {
        stage(IC3<0>{}, hc); stage(IC3<1>{}, hc); stage(IC3<2>{}, hc); stage(IC3<3>{}, hc); stage(IC3<4>{}, hc);
        stage(IC3<5>{}, hc); stage(IC3<6>{}, hc); stage(IC3<7>{}, hc); stage(IC3<8>{}, hc);
    }

    // epilogue: the residual / multiplier rows are fetched AFTER the accumulators have left the registers (the half-tile
    // kernel prefetches them next to live accumulators; that does not fit here) -- the co-resident workgroups cover the
    // exposed latency.  Output rows go through LDS in blocks of 128 (= 8 tile rows), one EpiStore pass each.
    // the epilogue's lane coordinates are derived afresh: kept across the loop they cost a register the 128-register variant does not have
    int tid_e = tid;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, li_e = lane_e & 15, lg_e = lane_e >> 4;
#pragma unroll
    for (int m = 0; m < 4; ++m) {  // after the loop: 16 VGPRs the loop does not have
        const f32x4_t t = *(const f32x4_t*)(smem + CF::LDS_BIAS + (wm * 64 + m * 16 + lg_e * 4) * 4);
        bv[m][0] = t[0]; bv[m][1] = t[1]; bv[m][2] = t[2]; bv[m][3] = t[3];
    }
    __syncthreads();
    char* const O = smem;
    float* const red = (float*)(smem + TR * 16 * T3_OS);
#pragma unroll
    for (int j = 0; j < NB; ++j)
        epi_acc_to_lds<T>(O, T3_OS, acc[j], bv, p.act, wm * 64, (wn * NB + j) * 64, li_e, lg_e);
    if constexpr (EPI == 7) {
        // items = (tile row R of 256) x (16-byte segment of the noise row: loss_lde / 8 of them): y and eps are both NHWC rows, so a thread
        // turns 8 channels of one pixel into their gradient in place; channels loss_C .. loss_lde-1 are zero on both sides
        __syncthreads();  // the whole 256-row tile is in LDS
        float gs = p.loss_gscale;
        if (p.loss_scaler != nullptr) gs *= p.loss_scaler[0];
        const int lde = p.loss_lde, nseg = lde >> 3, nitem = 256 * nseg;
        const char* const erow0 = (const char*)p.loss_eps + (((long long)b * H + oh0) * W + ow0) * (long long)lde * 2;
        float local = 0.f;
        for (int it = tid_e; it < nitem; it += T3_NTHR) {
            const int R = it / nseg, cs = it - R * nseg;
            const u32x4_t ev = *(const u32x4_t*)(erow0 + ((long long)(R >> 4) * W + (R & 15)) * lde * 2 + cs * 16);
            char* const cell = O + R * T3_OS + cs * 16;
            float yf[8], ef[8];
            unpack16<T>(*(const u32x4_t*)cell, yf);
            unpack16<f16_t>(ev, ef);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float d = yf[e] - ef[e];
                local += d * d;
                yf[e] = d * gs;
            }
            *(u32x4_t*)cell = pack16<T>(yf);
        }
        // one atomic per WORKGROUP: with one per wave (65 k atomics on a single address per launch at B = 128) the launch took 0.90 ms
        // instead of 0.42 -- the L2 serialises them; the waves' sums meet in LDS behind the tile, thread 0 adds them after the barrier below
        local = wave_sum(local);
        if (lane_e == 0) red[tid_e >> 6] = local;
    }
    // both 8-row blocks' residual / multiplier rows are requested before the first block is finished: the second block's HBM latency
    // runs behind the first block's arithmetic and stores (the accumulators have left the registers, so both sets fit)
    if constexpr (CF::NPASS == 2 && EPI != 0) {  // the per-family instantiations; in the all-in-one kernel (EPI = 0) this spills 54 registers
        EpiStore<T, 128, T3_NTHR> est0, est1;
        if ((EPI == 3 || EPI == 6) && tid_e < 128) red[tid_e] = 0.f;
        est0.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0) * W + ow0, W);
        est1.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0 + 8) * W + ow0, W);
        __syncthreads();
        if constexpr (EPI == 7) {
            if (tid_e == 0) {
                float tot = 0.f;
#pragma unroll
                for (int i = 0; i < NW; ++i) tot += red[i];
                atomicAdd(p.loss_sum, tot);
            }
        }
        typename EpiStore<T, 128, T3_NTHR>::LnColSums dmsum;  // LayerNorm backward: modulation-gradient column sums, carried over both blocks
        if constexpr (EPI == 3 || EPI == 6) dmsum.clear();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            auto& est = h == 0 ? est0 : est1;
            const char* const Oh = O + h * 128 * T3_OS;
            if constexpr (EPI == 2 || EPI == 8) est.template finish_lnf<EPI == 8>(p, Oh, T3_OS, tid_e, b);
            else if constexpr (EPI == 3) est.template finish_ln_rows<false>(p, Oh, T3_OS, tid_e, b, dmsum);
            else if constexpr (EPI == 6) est.template finish_ln_rows<true>(p, Oh, T3_OS, tid_e, b, dmsum);
            else est.finish(p, Oh, T3_OS, tid_e);
        }
        if constexpr (EPI == 3 || EPI == 6) est0.finish_ln_dm(p, tid_e, b, red, dmsum);  // one reduction per tile (was: per block, with two more barriers between)
    } else {
#pragma unroll
        for (int h = 0; h < CF::NPASS; ++h) {
        if (p.ln_x != nullptr && tid_e < 128) red[tid_e] = 0.f;
        EpiStore<T, 128, T3_NTHR> est;
        const bool pool2 = EPI == 0 && (p.flags & C2W_CONV_POOL2) != 0;
        if (!pool2) est.prefetch_tile16(p, tid_e, co0, ((long long)b * H + oh0 + 8 * h) * W + ow0, W);
        __syncthreads();
        const char* const Oh = O + h * 128 * T3_OS;
        // EPI 2 / 3 / 4: instantiations that carry one epilogue only (picked by the launcher)
        if constexpr (EPI == 2 || EPI == 8) est.template finish_lnf<EPI == 8>(p, Oh, T3_OS, tid_e, b);
        else if constexpr (EPI == 3 || EPI == 6) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
        else if constexpr (EPI == 4) est.finish(p, Oh, T3_OS, tid_e);
        else if (pool2) est.finish_pool2(p, Oh, T3_OS, tid_e, co0, ((long long)b * (H >> 1) + ((oh0 + 8 * h) >> 1)) * (W >> 1) + (ow0 >> 1), W >> 1);
        else if (p.ln_x != nullptr) est.finish_ln(p, Oh, T3_OS, tid_e, b, red);
        else if (p.lnf_y != nullptr) est.finish_lnf(p, Oh, T3_OS, tid_e, b);
        else est.finish(p, Oh, T3_OS, tid_e);
        if (h + 1 < CF::NPASS) __syncthreads();  // the LayerNorm column sums are re-zeroed for the next block only after everyone read them
    }
    }
}

template <int TR, typename T, int NW, int EPI, bool WPK>
int t3_launch_wpk(const C2wConvArgs& a, hipStream_t st) {
    typedef T3Cfg<TR, NW> CF;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_patch_t3_kernel<TR, T, NW, EPI, WPK>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS));
        attr = true;
    }
    const int nN = (a.Cout + 127) / 128;
    const int nM = a.B * (a.Hout / TR) * (a.Wout >> 4);
    conv_patch_t3_kernel<TR, T, NW, EPI, WPK><<<nM * nN, CF::NTHR, CF::LDS, st>>>(a);
    return (int)hipGetLastError();
}
template <int TR, typename T, int NW, int EPI>
int t3_launch_as(const C2wConvArgs& a, hipStream_t st) {
    return (a.flags & C2W_CONV_WPACKED) != 0 ? t3_launch_wpk<TR, T, NW, EPI, true>(a, st) : t3_launch_wpk<TR, T, NW, EPI, false>(a, st);
}

template <int TR, typename T, int NW>
int t3_launch(const C2wConvArgs& a, hipStream_t st) {
    if (TR == 16 && (a.flags & C2W_CONV_POOL2) == 0) {
        if (a.lnf_y != nullptr) return a.res_rstd != nullptr ? t3_launch_as<16, T, NW, 8>(a, st) : t3_launch_as<16, T, NW, 2>(a, st);  // 8: the residual rebuilt from normalised rows
        if (a.ln_x != nullptr) return a.ln_rstd != nullptr ? t3_launch_as<16, T, NW, 6>(a, st) : t3_launch_as<16, T, NW, 3>(a, st);
        if (a.loss_sum != nullptr) return t3_launch_as<16, T, NW, 7>(a, st);  // (c2w_conv_loss_supported: the narrow form's conditions)
        if (a.wrows <= 80 && a.Cout <= 128 && c2w_knobs().wgrad_narrow) return t3_launch_as<16, T, NW, 5>(a, st);  // the output conv: 65 weight rows
        return t3_launch_as<16, T, NW, 4>(a, st);
    }
    return t3_launch_as<TR, T, NW, 0>(a, st);
}

}  // namespace

// The 16x16-tile variant pays off where the launch still fills the chip several times over (measured on MI355X, B = 128:
// 128->128 @128^2 0.596 vs 0.621 ms, @64^2 0.156 vs 0.166 ms; 384->384 @16^2 with 384 workgroups 0.093 vs 0.079 ms).
// Knob C2W_CONV_T3 = 0 disables it, = 16 forces it wherever the image is tiled by 16x16 (knobs.h).
// Round 6: from 512 workgroups (one full round of two per CU) instead of 1024 -- what the 32^2 level has at the 8-GPU strong-scaling
// batch of 64 windows per GPU (step 26.16 -> 26.00 ms) and the 64^2 level of a one-member sampler step at L = 49 (37 windows: 6.36 k ->
// 6.64 k window-forwards/s); 256 is behind again (26.97 against 26.90 ms at B = 64).  C2W_CONV_T3_MIN_WGS overrides.
bool c2w_conv_patch3_wanted(const C2wConvArgs& a, int dtype) {
    const int mode = c2w_knobs().conv_t3;
    if ((dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16) || mode == 0 || (a.Hout & 15) != 0 || (a.Wout & 15) != 0) return false;
    const long long wgs = (long long)a.B * (a.Hout >> 4) * (a.Wout >> 4) * ((a.Cout + 127) / 128);
    return mode == 16 || wgs >= c2w_knobs().conv_t3_min_wgs;
}

int c2w_conv_patch3(const C2wConvArgs& a, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? t3_launch<16, f16_t, T3_WAVES>(a, st) : t3_launch<16, bf16_t, T3_WAVES>(a, st);
}
