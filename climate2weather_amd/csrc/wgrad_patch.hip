// Halo-patch weight-gradient GEMM for the 3x3 stride-1 convolutions on gfx950:
//     dW[co][tap][ci] += sum_q dY[q][co] * X[q shifted by tap][ci]         (+ dbias[co] += sum_q dY[q][co])
//
// The general wgrad kernel (wgrad.hip) gathers X once per tap and spends most of its issue slots on per-stage
// gather arithmetic.  Here one workgroup owns (256 B of output channels) x (128 B of input channels) x ALL 9 taps:
//   * K tile = an 8x16 block of output pixels of one image.  Per K tile it stages dY[128 px][256 B] and ONE
//     (8+2)x(16+2) input halo patch [10][24 px pitch][128 B]; the 9 taps are 9 shifted views of that patch, so each
//     input pixel is loaded once (not 9 times) and dY once -- 3.7x fewer bytes per FLOP than the gather kernel;
//   * both operands are K(pixel)-strided in memory: they are staged as they lie in HBM (direct-to-LDS 16-B loads,
//     zero padding via out-of-range buffer offsets) and transposed on the way into the matrix core
//     (bf16: ds_read_b64_tr_b16; fp32: one dword per lane).  Source-side XOR swizzles keep the transposed reads
//     conflict free (dY rows: chunk ^= (row&3)<<2 | (row>>2)&3; patch rows: chunk ^= ((pix>>1)&1)<<1 | ((pix>>3)&1)<<2);
//   * taps unrolled, every LDS address = register + immediate; one barrier per K tile (144 MFMAs per wave in bf16);
//   * 36 accumulator tiles per wave (bf16: 4 co-tiles x 1 ci-tile x 9 taps); split-K over pixel ranges across
//     workgroups, partial sums combined with fp32 atomics issued as full 256-B rows (staged through LDS per tap);
//   * the bias gradient rides along as one extra MFMA per co-tile against a vector of ones.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <utility>
#include "conv_geom.h"

#include "knobs.h"
// (schedule variants measured and rejected -- burst LDS-DMA issue, fragments one tap ahead, carried kh = 2 fragments -- and the ablation
// builds live in lab/csrc/wgrad_patch_lab.hip; results in profiles/r02_experiments.md section 1)


namespace {

// Measured alternatives (128->128 @128^2, B = 128, this layout 0.62-0.69 ms depending on the box): ablation builds run 0.49 ms
// with the LDS-DMA after the first tile removed and 0.42 ms with the MFMAs removed; 4-row K tiles in a 4-slot ring (three
// tiles of prefetch instead of one, same LDS) were SLOWER (0.75 ms; 0.22 vs 0.17 ms at 64^2): twice the barriers and 10 % more
// halo bytes cost more than the deeper prefetch gave back.
constexpr int NTHREADS = 512;
constexpr int KPX = 128;                 // pixels per K tile (8 rows x 16 cols)
constexpr int ABYTES = KPX * 256;        // dY tile
constexpr int PPITCH = 24;               // patch row pitch (pixels); 18 used
constexpr int NPIECE = 10 * 3;           // 1 KiB pieces per patch
constexpr int PBYTES = NPIECE * 1024;    // 30,720
// LDS plan: THREE dY slots + TWO patch slots (159,744 of the CU's 163,840 bytes).  With one slot pair per operand the whole next K
// tile (62 KB) is issued at the top of a tile's compute phase and must land within it -- measured, it does not: a tile's 144 MFMAs
// per wave take ~2.5 us, 62 KB through one CU's LDS-DMA path ~3.3 us behind an HBM/L2 latency, and 40 % of the wave time sat in
// the s_waitcnt + barrier of the next tile (profiles/r01q_pmc_wgrad_patch_b128.md).  The dY tile, the larger half, is therefore
// fetched TWO tiles ahead (it has a whole extra compute phase to land) and only the 30 KB patch one tile ahead: the bytes a tile
// boundary waits for are halved, and 62 KB stay in flight per CU at all times instead of draining to zero at every boundary.
constexpr int NSLOT_A = 3, NSLOT_P = 2;
constexpr int PBASE = NSLOT_A * ABYTES;                  // patch slots sit behind the dY slots
constexpr int LDS_BYTES = PBASE + NSLOT_P * PBYTES;      // 159,744

struct WpArgs {
    const void* dy;
    const void* x;
    float* dw;
    float* db;
    int B, H, W, Cin, Cout, ldy;  // H x W: the grid of dY = the grid the taps live on
    int up;                       // C2W_CONV_UP: X is the (H/2) x (W/2) source map of a nearest-neighbour x2 upsampling (patch pixel >> 1)
    int ktiles, ktiles_per_split;
    float* ws;  // optional workspace [split][tile][tap][COT][CIB] fp32: partial sums by plain stores, reduced by a second launch
    int direct;  // no split (one workgroup per output tile): the tile is added onto dw by plain read-modify-write stores, no atomics
};

// Grouped launch: the weight gradients of up to WP_MAX_ITEMS layers of ONE geometry (the residual-block convs of a level: their output
// gradients exist one after the other during the backward pass, their weight gradients are independent) as one grid.  A launch per layer
// splits K over 256 / tilesMN workgroups to fill the chip (8 K tiles per workgroup at 8x8, each followed by 295 KB of partial sums);
// together the layers fill it with a fraction of the splits.  The per-layer pointers ride in the kernel arguments and are read through the
// kernarg segment with a workgroup-uniform index (scalar loads; a by-value array indexed dynamically would be copied to scratch).
// Layer = position in the grid-wide XCD-contiguous order / workgroups per layer.
struct WpItem {
    const void* dy;
    const void* x;
    float* dw;
    float* db;
};
constexpr int WP_MAX_ITEMS = 16;
struct WpGroupArgs {
    WpArgs c;             // geometry, split plan, ws = base of the group's workspace; dy / x / dw / db unused
    int n;                // layers
    int live_per_item;    // workgroups per layer = tilesMN * nsplit; the grid is n * live_per_item
    unsigned long long ws_item_floats;
    WpItem item[WP_MAX_ITEMS];
};



// ok ? v : (an offset that is always out of the descriptor's range) as a SELECT: written as a plain ternary over the address arithmetic
// hipcc turns it into a branch around that arithmetic (s_and_saveexec + s_cbranch_execz), which cuts the MFMA stream of the main loop
// into basic blocks; the empty asm makes v a value that exists on both paths.
__device__ __forceinline__ uint32_t sel_oob(bool ok, uint32_t v) {
    asm volatile("" : "+v"(v));
    return ok ? v : C2W_OOB;
}

__device__ __forceinline__ uint32_t swzA(int row) { return (uint32_t)(((row & 3) << 2) | ((row >> 2) & 3)); }
__device__ __forceinline__ uint32_t swzP(int pix) { return (uint32_t)((((pix >> 1) & 1) << 1) | (((pix >> 3) & 1) << 2)); }

// PAIR: 8-pixel-wide images (the 8x8 level): a K tile is the same 8 rows of TWO images side by side (tile columns 0..7 = image b,
// 8..15 = image b + 1), each with its own zero halo -- patch columns 0..9 / 10..19, as in conv_patch_half_kernel<T, PAIR>.
// NARROW (16-bit only): at most 80 output channels (the network's output conv, 65).  The co-tile halves then go to waves 0-3 / 4-7
// instead of even / odd waves -- every SIMD hosts one wave of each -- and waves 4-7 compute only their first co tile (channels 64-79):
// 45 instead of 72 MFMAs per SIMD and K step; the dY rows past Cout were never fetched anyway (out-of-range lanes of the LDS-DMA).
__device__ __forceinline__ int wp_xcd_order(int bid, int nblk) {  // block -> position in an order that is contiguous per XCD (blocks b, b + 8, ... share one)
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// L: the workgroup's position in the (split, output tile) order of ONE layer
template <typename T, bool PAIR = false, bool NARROW = false>
__device__ __forceinline__ void wgrad_patch_body(const WpArgs& p, const int L) {
    constexpr int ESZ = sizeof(T);
    constexpr bool BF = ESZ == 2;
    constexpr int COT = 256 / ESZ;       // output channels per workgroup tile
    constexpr int CIB = 128 / ESZ;       // input channels per workgroup tile
    constexpr int MTW = BF ? 4 : 1;      // 16-wide co tiles per wave (x 1 ci tile x 9 taps)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    static_assert(!NARROW || (BF && !PAIR), "narrow-M form: 16-bit, 16-pixel-wide tiles");
    const int mt0 = BF ? (NARROW ? (wid >> 2) * 4 : (wid & 1) * 4) : (wid >> 1);  // first co tile of this wave
    const int nt = BF ? (NARROW ? (wid & 3) : (wid >> 1)) : (wid & 1);             // its ci tile
    const bool light = NARROW && wid >= 4;                                         // only its first co tile is live

    const int ncib = p.Cin / CIB;
    const int tilesM = (p.Cout + COT - 1) / COT;
    const int tilesMN = tilesM * ncib;
    const int split = L / tilesMN, mn = L - split * tilesMN;
    const int tm = mn / ncib, cb = mn - tm * ncib;
    const int co0 = tm * COT, ci0 = cb * CIB;
    const int H = p.H, W = p.W;
    const int tw = PAIR ? 1 : W >> 4, tpi = (H >> 3) * tw;
    const int t0 = split * p.ktiles_per_split;
    const int t1 = (t0 + p.ktiles_per_split < p.ktiles) ? t0 + p.ktiles_per_split : p.ktiles;

    // ---- staging addresses.  Each wave moves 4 dY pieces and 4 patch pieces (1 KiB = 64 lanes x 16 B) per K tile.  The per-lane source
    // offsets are recomputed from the lane number where a piece is issued (a dozen integer instructions between MFMAs) instead of
    // being kept in 16 registers across the whole tile loop: the accumulators leave no room for them (the kernel spilled with them).
    //   dY piece i: slot s = tid + 512 i -> row = s >> 4 (pixel of the tile), chunk = s & 15 (16-B chunk of the 256-B channel row);
    //     the source-side swizzle of a row does not depend on i (rows 32 apart), so offset(i) = offset(0) + i * (two tile rows).
    //   patch piece r: piece number pc = 8 r + wave (pieces past the end repeat the last one) -> patch row pc / 3, 8-pixel group pc % 3.
    const int Ws = p.up ? W >> 1 : W;
    const size_t ximg = (size_t)(p.up ? (H >> 1) * Ws : H * W) * p.Cin * ESZ;
    const size_t yimg = (size_t)H * W * p.ldy * ESZ;
    const uint32_t a_step = PAIR ? (uint32_t)(16 * p.ldy * ESZ) : (uint32_t)(2 * W * p.ldy * ESZ);  // bytes between dY pieces i and i + 1
    auto dy_voff0 = [&](int tid_) -> uint32_t {
        const int row = tid_ >> 4, pc = tid_ & 15;
        const uint32_t lc = (uint32_t)pc ^ swzA(row);
        const int c = co0 + (int)lc * (16 / ESZ);
        if constexpr (PAIR) {  // tile column >= 8: the same row of the next image
            return sel_oob(c < p.Cout, (uint32_t)((((row >> 4) * 8 + (row & 7)) * p.ldy + c) * ESZ) + (uint32_t)((row >> 3) & 1) * (uint32_t)yimg);
        } else {
            return sel_oob(c < p.Cout, (uint32_t)((((row >> 4) * W + (row & 15)) * p.ldy + c) * ESZ));
        }
    };
    auto patch_piece = [&](int r) -> int {  // wave-uniform
        const int pc = r * 8 + wid;
        return pc < NPIECE ? pc : NPIECE - 1;
    };
    auto patch_voff = [&](int r, int lane_, int oh0, int ow0) -> uint32_t {
        const int pc = patch_piece(r);
        const int pr = pc / 3, pg = pc - pr * 3;
        const int px = pg * 8 + (lane_ >> 3);
        const uint32_t lanepart = (uint32_t)(((lane_ & 7) ^ swzP(pr * PPITCH + px)) << 4) + (uint32_t)ci0 * ESZ;
        if constexpr (PAIR) {  // 8x8 images, two per K tile, each with its own zero halo: patch columns 0..9 / 10..19
            const int pimg = px >= 10 ? 1 : 0;
            const int ih = pr - 1, iw = px - 1 - 10 * pimg;
            const bool ok = (unsigned)ih < 8u && (unsigned)iw < 8u && px < 20;
            return sel_oob(ok, (uint32_t)((ih * 8 + iw) * p.Cin) * ESZ + (uint32_t)pimg * (uint32_t)(64 * p.Cin * ESZ) + lanepart);
        } else {
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < 18;
            const int spix = p.up ? (ih >> 1) * Ws + (iw >> 1) : ih * W + iw;
            return sel_oob(ok, (uint32_t)(spix * p.Cin) * ESZ + lanepart);
        }
    };

    // K tile t -> image b, tile origin (oh0, ow0)
    auto tile_origin = [&](int t, int& b, int& oh0, int& ow0) {
        b = PAIR ? 2 * t : t / tpi;  // PAIR: 8x8 images, one K tile per pair
        const int tt = PAIR ? 0 : t - (t / tpi) * tpi;
        const int ty = tt / tw, tx = tt - ty * tw;
        // integer division runs on the vector ALU: without the readfirstlane the quotients -- and every descriptor / offset derived from
        // them, live across the whole tile -- stay in vector registers
        b = __builtin_amdgcn_readfirstlane(b);
        oh0 = __builtin_amdgcn_readfirstlane(ty << 3);
        ow0 = __builtin_amdgcn_readfirstlane(tx << 4);
    };
    // LDS-DMA of one K tile = 4 patch pieces + 4 dY pieces (1 KiB each) per wave.  The descriptors and the tile origin are set up
    // once per tile (TileSrc), the pieces are issued one at a time: inside the MFMA stream of the tile being computed (see the main
    // loop), not as a burst behind the barrier -- a piece holds the issuing wave's instruction stream for 60-185 cycles, and eight of
    // them back to back on all eight waves at once left the CU's matrix pipes empty for >1000 cycles per K tile.
    struct TileSrc {
        __amdgpu_buffer_rsrc_t rsrc;
        int oh0, ow0;
        uint32_t aso;
        char* base;
    };
    auto srcA = [&](int t, int sa) {  // dY tile of K tile t -> dY slot sa
        int b, oh0, ow0;
        tile_origin(t, b, oh0, ow0);
        const uint32_t nimg = PAIR && b + 1 < p.B ? 2u : 1u;  // a missing partner image reads as zeros (out of the descriptor's range)
        TileSrc s;
        s.rsrc = make_rsrc((const char*)p.dy + (size_t)b * yimg, (uint32_t)yimg * nimg);
        s.oh0 = oh0;
        s.ow0 = ow0;
        s.aso = (uint32_t)((oh0 * W + ow0) * p.ldy) * ESZ;
        s.base = smem + sa * ABYTES + wid * 1024;
        return s;
    };
    auto srcP = [&](int t, int sp) {  // input halo patch of K tile t -> patch slot sp
        int b, oh0, ow0;
        tile_origin(t, b, oh0, ow0);
        const uint32_t nimg = PAIR && b + 1 < p.B ? 2u : 1u;
        TileSrc s;
        s.rsrc = make_rsrc((const char*)p.x + (size_t)b * ximg, (uint32_t)ximg * nimg);
        s.oh0 = oh0;
        s.ow0 = ow0;
        s.aso = 0;
        s.base = smem + PBASE + sp * PBYTES;
        return s;
    };
    // ``live`` false (no such tile: the tail of the split): the piece is still issued -- branch-free loop body, constant vmcnt
    // bookkeeping -- but reads out of range, i.e. writes zeros into a slot nobody reads again.
    auto pieceA = [&](const TileSrc& s, int i, bool live = true) {
        int tid_ = tid;
        asm volatile("" : "+v"(tid_));  // recompute here (see "staging addresses")
        const uint32_t v = dy_voff0(tid_);
        glds16(s.rsrc, s.base + i * 8192, sel_oob(live, v), s.aso + (uint32_t)i * a_step);
    };
    auto pieceP = [&](const TileSrc& s, int r, bool live = true) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const uint32_t v = patch_voff(r, lane_, s.oh0, s.ow0);
        glds16(s.rsrc, s.base + patch_piece(r) * 1024, sel_oob(live, v), 0);
    };
    auto issueA = [&](int t, int sa) {
        const TileSrc s = srcA(t, sa);
#pragma unroll
        for (int i = 0; i < 4; ++i) pieceA(s, i);
    };
    auto issueP = [&](int t, int sp) {
        const TileSrc s = srcP(t, sp);
#pragma unroll
        for (int r = 0; r < 4; ++r) pieceP(s, r);
    };

    f32x4_t acc[9][MTW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int m = 0; m < MTW; ++m) acc[t][m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.db != nullptr && cb == 0 && nt == 0;
    f32x4_t accb[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) accb[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- fragment read offsets (register part; K step and slot are added as immediates / one add per tile)
    uint32_t offA[MTW][2], offB[9][2];
    if constexpr (BF) {
        const int qq = li >> 2, pp = li & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * lg + qq + 4 * h;  // + 32*ks
            const uint32_t f = swzA(row);
#pragma unroll
            for (int m = 0; m < MTW; ++m)
                offA[m][h] = (uint32_t)(row * 256 + (((((mt0 + m) * 2 + (pp >> 1)) ^ f) & 15) << 4) + 8 * (pp & 1));
            const int r = lg >> 1, c = 8 * (lg & 1) + qq + 4 * h;  // pixel (r + 2*ks, c) of the K tile
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int pix = (r + t / 3) * PPITCH + c + t % 3 + (PAIR ? 2 * (lg & 1) : 0);
                offB[t][h] = (uint32_t)(pix * 128 + (((uint32_t)(nt * 2 + (pp >> 1)) ^ swzP(pix)) << 4) + 8 * (pp & 1));
            }
        }
    }

    // Issue order per wave (vmcnt retires in it):  P(t0) A(t0) A(t0+1) | P(t0+1) A(t0+2) | P(t0+2) A(t0+3) | ...
    // At the top of tile t the 4 youngest operations are the pieces of A(t+1) (if it exists): vmcnt(4) lets them fly on.
    if (t0 < t1) {
        issueP(t0, 0);
        issueA(t0, 0);
        const TileSrc s1 = srcA(t0 + 1 < t1 ? t0 + 1 : t0, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) pieceA(s1, i, t0 + 1 < t1);
    }
    int sa = 0, sp = 0;
    for (int t = t0; t < t1; ++t) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // all but the 4 youngest = the dY pieces of tile t + 1 (real or out of range)
        __builtin_amdgcn_s_barrier();  // tile t landed for every wave; every wave is done reading tile t - 1's slots
        // next patch -> the patch slot of tile t - 1; the dY tile after next -> the dY slot of tile t - 1 = (sa + 2) mod 3
        const bool hasP = t + 1 < t1, hasA = t + 2 < t1;
        const TileSrc nP = srcP(hasP ? t + 1 : t, sp ^ 1), nA = srcA(hasA ? t + 2 : t, sa == 0 ? 2 : sa - 1);
        if constexpr (!BF) {  // fp32 path: burst behind the barrier
#pragma unroll
            for (int r = 0; r < 4; ++r) pieceP(nP, r, hasP);
#pragma unroll
            for (int i = 0; i < 4; ++i) pieceA(nA, i, hasA);
        }
        const char* const SA = smem + sa * ABYTES;
        const char* const SP = smem + PBASE + sp * PBYTES;
        if constexpr (BF) {
            // Fragment reads are INLINE ASM.  Through the ds_read_tr16_b64 builtin hipcc (ROCm 7.2) places "s_waitcnt vmcnt(0)" in
            // front of the first LDS read that follows an LDS-DMA issue (the read might alias the DMA's destination), i.e. the next
            // tile's loads were waited for before the current tile's first MFMA: the round-1 kernel never overlapped its loads with
            // its arithmetic (its 40 % "parked" wave time).  The asm reads are invisible to that pass; what orders them is what the
            // hardware needs and nothing more: the counted vmcnt + barrier at the tile boundary (the data landed), and the counted
            // lgkmcnt below (LDS returns in order) with a sched_barrier behind it (hipcc moves MFMAs across an asm wait otherwise).
            const uint32_t sA = (uint32_t)(uintptr_t)SA, sP = (uint32_t)(uintptr_t)SP;
            auto rd = [&](tr_frag& f, uint32_t base, uint32_t o0, uint32_t o1, auto IMMc) {
                constexpr int IMM = decltype(IMMc)::value;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(base + o0), "n"(IMM));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(base + o1), "n"(IMM));
            };
            static_for<4>([&](auto KSc) {
                constexpr int ks = decltype(KSc)::value;
                {
                    // The next tiles' LDS-DMA, four pieces at a time at K-step boundaries (fragment registers are dead there), and
                    // never on both waves of a SIMD at once (waves w and w + 4 share one): a piece holds the issuing wave's
                    // instruction stream for 60-185 cycles; while one wave of the pair is held its partner has the matrix pipe to
                    // itself.  Patch pieces (waited for at the next tile boundary) first, dY pieces (a tile of slack more) later.
                    const bool first = wid < 4;
                    if ((ks == 0 && first) || (ks == 1 && !first)) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pieceP(nP, r, hasP);
                    }
                    if ((ks == 2 && first) || (ks == 3 && !first)) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) pieceA(nA, i, hasA);
                    }
                }
                constexpr int PF = 2;  // patch fragments are read PF taps ahead of their MFMAs
                tr_frag a[MTW], bq[PF + 1];
                rd(a[0], sA, offA[0][0], offA[0][1], IC<ks * 32 * 256>{});
                if (!light) {  // wave-uniform (false only in the NARROW form)
#pragma unroll
                    for (int m = 1; m < MTW; ++m) rd(a[m], sA, offA[m][0], offA[m][1], IC<ks * 32 * 256>{});
                }
                static_for<PF>([&](auto Jc) {
                    constexpr int j = decltype(Jc)::value;
                    rd(bq[j], sP, offB[j][0], offB[j][1], IC<ks * 2 * PPITCH * 128>{});
                });
                static_for<9>([&](auto TPc) {
                    constexpr int tp = decltype(TPc)::value;
                    if constexpr (tp + PF < 9) rd(bq[(tp + PF) % (PF + 1)], sP, offB[tp + PF][0], offB[tp + PF][1], IC<ks * 2 * PPITCH * 128>{});
                    constexpr int AHEAD = (tp + PF < 9 ? PF : 8 - tp) * 2;  // reads younger than tap tp's: they may stay in flight
                    if constexpr (AHEAD == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else if constexpr (AHEAD == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8_t bfr = bq[tp % (PF + 1)].vec();
                    acc[tp][0] = mfma16s<T>(a[0].vec(), bfr, acc[tp][0]);
                    if (!light) {
#pragma unroll
                        for (int m = 1; m < MTW; ++m) acc[tp][m] = mfma16s<T>(a[m].vec(), bfr, acc[tp][m]);
                    }
                    if (tp == 8 && do_bias) {
                        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones16<T>());
                        accb[0] = mfma16s<T>(a[0].vec(), ones, accb[0]);
                        if (!light) {
#pragma unroll
                            for (int m = 1; m < MTW; ++m) accb[m] = mfma16s<T>(a[m].vec(), ones, accb[m]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);  // the next reads overwrite fragment registers: not before these MFMAs are issued
                });
            });
        } else {
#pragma unroll 4
            for (int kk = 0; kk < 32; ++kk) {  // 4 pixels per MFMA: lane (i, g) feeds pixel 4*kk + g
                const int row = kk * 4 + lg;
                const float a = *(const float*)(SA + row * 256 + ((((uint32_t)(mt0 * 4 + (li >> 2))) ^ swzA(row)) << 4) + (li & 3) * 4);
                const int r = kk >> 2, c = 4 * (kk & 3) + lg;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int pix = (r + tp / 3) * PPITCH + c + tp % 3 + (PAIR && (kk & 2) ? 2 : 0);
                    const float bv = *(const float*)(SP + pix * 128 + ((((uint32_t)(nt * 4 + (li >> 2))) ^ swzP(pix)) << 4) + (li & 3) * 4);
                    acc[tp][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bv, acc[tp][0], 0, 0, 0);
                }
                if (do_bias) accb[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, 1.0f, accb[0], 0, 0, 0);
            }
        }
        sa = sa == 2 ? 0 : sa + 1;
        sp ^= 1;
    }

    // ---- epilogue: per tap, tile -> LDS [co][ci] fp32 -> atomics as whole (co, tap) rows of CIB floats
    if (t0 >= t1) return;
    if (do_bias && li == 0) {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (mt0 + m) * 16 + lg * 4 + r;
                if (co < p.Cout) atomicAdd(p.db + co, accb[m][r]);
            }
    }
    constexpr int OS = CIB + 4;
    float* const O = (float*)smem;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) O[((mt0 + m) * 16 + lg * 4 + r) * OS + nt * 16 + li] = acc[tp][m][r];
        __syncthreads();
        if (p.ws != nullptr) {  // partial sums: coalesced stores, no atomics (75 MB of atomics per launch otherwise)
            float* const dst = p.ws + (((size_t)split * tilesMN + mn) * 9 + tp) * (COT * CIB);
            for (int idx = tid; idx < COT * CIB; idx += NTHREADS) dst[idx] = O[(idx / CIB) * OS + (idx % CIB)];
        } else if (p.direct) {  // this workgroup alone owns the tile (no split): launches on a stream are ordered, nobody else adds here
            for (int idx = tid; idx < COT * CIB; idx += NTHREADS) {
                const int row = idx / CIB, col = idx - row * CIB;
                const int co = co0 + row;
                if (co < p.Cout) {
                    float* const d = p.dw + ((size_t)co * 9 + tp) * p.Cin + ci0 + col;
                    *d = *d + O[row * OS + col];
                }
            }
        } else {
            for (int idx = tid; idx < COT * CIB; idx += NTHREADS) {
                const int row = idx / CIB, col = idx - row * CIB;
                const int co = co0 + row;
                if (co < p.Cout) atomicAdd(p.dw + ((size_t)co * 9 + tp) * p.Cin + ci0 + col, O[row * OS + col]);
            }
        }
    }
}

template <typename T, bool PAIR = false, bool NARROW = false>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_patch_kernel(const WpArgs p) {
    wgrad_patch_body<T, PAIR, NARROW>(p, wp_xcd_order((int)blockIdx.x, (int)gridDim.x));
}

typedef __attribute__((address_space(4))) const char wp_kernarg_t;
__device__ __forceinline__ WpItem wp_item(int it) {  // it: workgroup-uniform
    const WpItem __attribute__((address_space(4)))* tab =
        (const WpItem __attribute__((address_space(4)))*)((wp_kernarg_t*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WpGroupArgs, item));
    WpItem r;
    r.dy = tab[it].dy;
    r.x = tab[it].x;
    r.dw = tab[it].dw;
    r.db = tab[it].db;
    return r;
}

template <typename T, bool PAIR = false>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_patch_group_kernel(const WpGroupArgs g) {
    // the XCD-contiguous order runs over the WHOLE grid: per-layer orders (first version) left the remainder blocks of every layer on
    // the same one or two XCDs -- 36 workgroups on 32 CUs there, 30 elsewhere: two rounds for the 64x64 group (1350 us instead of ~750)
    const int Lg = wp_xcd_order((int)blockIdx.x, (int)gridDim.x);
    const int it = __builtin_amdgcn_readfirstlane(Lg / g.live_per_item);
    const int L = Lg - it * g.live_per_item;
    const WpItem e = wp_item(it);
    WpArgs p = g.c;
    p.dy = e.dy;
    p.x = e.x;
    p.dw = e.dw;
    p.db = e.db;
    p.ws = g.c.ws != nullptr ? g.c.ws + (size_t)it * g.ws_item_floats : nullptr;
    wgrad_patch_body<T, PAIR, false>(p, L);
}

// dw[co][tap][ci] += sum over splits of ws[split][tile][tap][row][col].  RC float4 columns x RG split groups per block (RC * RG =
// 256 threads): every thread streams 1/RG of the splits with 16-B loads (8 in flight), the groups meet in LDS, group 0 updates dw
// (no atomics: one thread per output vector, launches on a stream are ordered).  Block shapes 16 x 16, 32 x 8 and 64 x 4 (576 to
// 2304 blocks for a 128 -> 128 layer) measured the same within noise (wgrad + reduction 618-623 us at 128^2, 143-146 us at 32^2).
#ifndef C2W_RED_COLS
#define C2W_RED_COLS 64
#endif
template <int COT, int CIB>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int tilesMN, int ncib,
                                                  int Cin, int Cout) {
    constexpr int RC = C2W_RED_COLS, RG = 256 / RC;
    __shared__ f32x4_t red[RG][RC];
    const size_t per4 = (size_t)tilesMN * 9 * COT * CIB / 4;  // float4 vectors per split
    const f32x4_t* ws4 = (const f32x4_t*)ws;
    const int q = threadIdx.x % RC, grp = threadIdx.x / RC;
    for (size_t base = (size_t)blockIdx.x * RC; base < per4; base += (size_t)gridDim.x * RC) {
        const size_t i4 = base + q;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (i4 < per4) {
#pragma unroll 8
            for (int sidx = grp; sidx < nsplit; sidx += RG) acc += ws4[(size_t)sidx * per4 + i4];
        }
        red[grp][q] = acc;
        __syncthreads();
        if (grp == 0 && i4 < per4) {
#pragma unroll
            for (int g2 = 1; g2 < RG; ++g2) acc += red[g2][q];
            const size_t i = i4 * 4;
            const int col = (int)(i % CIB);
            size_t r = i / CIB;
            const int row = (int)(r % COT);
            r /= COT;
            const int tp = (int)(r % 9);
            const int mn = (int)(r / 9);
            const int tm = mn / ncib, cb = mn - tm * ncib;
            const int co = tm * COT + row;
            if (co < Cout) {
                f32x4_t* d = (f32x4_t*)(dw + ((size_t)co * 9 + tp) * Cin + cb * CIB + col);
                *d = *d + acc;
            }
        }
        __syncthreads();
    }
}

template <int COT, int CIB>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int tilesMN, int ncib,
                                                           int Cin, int Cout) {
    wgrad_reduce_body<COT, CIB>(ws, dw, nsplit, tilesMN, ncib, Cin, Cout);
}
// the same for every layer of a grouped launch: blockIdx.y = layer
template <int COT, int CIB>
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const WpGroupArgs g, int nsplit, int tilesMN, int ncib) {
    const int it = (int)blockIdx.y;
    const WpItem e = wp_item(it);
    wgrad_reduce_body<COT, CIB>(g.c.ws + (size_t)it * g.ws_item_floats, e.dw, nsplit, tilesMN, ncib, g.c.Cin, g.c.Cout);
}

// split of the K (pixel-tile) range over workgroups: one resident workgroup per CU and ONE round: tilesMN * nsplit <= 256 (rounding
// up gave 270 workgroups for 384 -> 384 -- a second round for 14 of them: 155 us instead of one round's ~90 at 16x16); every workgroup
// costs 295 KB of partial sums
template <int ESZ, bool PAIR>
static void split_plan(const C2wConvArgs& a, int& ktiles, int& tilesMN, int& nsplit, int& ktiles_per_split) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    ktiles = PAIR ? ((a.B + 1) >> 1) * (a.Hout >> 3) : a.B * (a.Hout >> 3) * (a.Wout >> 4);
    tilesMN = ((a.Cout + COT - 1) / COT) * (a.Cin / CIB);
    nsplit = c2w_knobs().wgrad_wgs / tilesMN;
    if (nsplit > ktiles) nsplit = ktiles;
    if (nsplit < 1) nsplit = 1;
    ktiles_per_split = (ktiles + nsplit - 1) / nsplit;
    nsplit = (ktiles + ktiles_per_split - 1) / ktiles_per_split;
}

template <int ESZ, bool PAIR>
static size_t ws_need(const C2wConvArgs& a) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    int ktiles, tilesMN, nsplit, per;
    split_plan<ESZ, PAIR>(a, ktiles, tilesMN, nsplit, per);
    return nsplit > 1 ? (size_t)nsplit * tilesMN * 9 * COT * CIB * sizeof(float) : 0;
}

template <typename T, bool PAIR, bool NARROW = false>
int launch(const C2wConvArgs& a, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    WpArgs p;
    p.dy = a.y; p.x = a.x; p.dw = dw; p.db = db;
    p.B = a.B; p.H = a.Hout; p.W = a.Wout; p.Cin = a.Cin; p.Cout = a.Cout; p.ldy = a.ldy;
    p.up = a.mode == C2W_CONV_UP ? 1 : 0;
    int tilesMN, nsplit;
    split_plan<ESZ, PAIR>(a, p.ktiles, tilesMN, nsplit, p.ktiles_per_split);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)wgrad_patch_kernel<T, PAIR, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const size_t need = (size_t)nsplit * tilesMN * 9 * COT * CIB * sizeof(float);
    p.ws = (ws != nullptr && need <= ws_bytes && nsplit > 1 && !c2w_knobs().wgrad_atomics) ? ws : nullptr;
    p.direct = nsplit == 1 && !c2w_knobs().wgrad_atomics ? 1 : 0;
    wgrad_patch_kernel<T, PAIR, NARROW><<<tilesMN * nsplit, NTHREADS, LDS_BYTES, st>>>(p);
    if (p.ws != nullptr) {
        const size_t per_split = (size_t)tilesMN * 9 * COT * CIB;
        const int grid = (int)std::min<size_t>((per_split / 4 + C2W_RED_COLS - 1) / C2W_RED_COLS, 8192);
        wgrad_reduce_kernel<COT, CIB><<<grid, 256, 0, st>>>(p.ws, dw, nsplit, tilesMN, a.Cin / CIB, a.Cin, a.Cout);
    }
    return (int)hipGetLastError();
}

// ---- grouped launches.  Split plan for n layers of one geometry (T = n * tilesMN output tiles, each `ktiles` K tiles deep): the number
// of splits that minimises  rounds x (K tiles per workgroup x t_k + t_fixed) + reduction launch  over 1 ... 4 rounds of 256 workgroups,
// with t_k = 4.1 us per K tile at the clock the chip holds under this kernel and t_fixed = 16 us (prologue + a 295-KB tile added onto dw)
// or 28 us (partial sums stored and read again) -- the two constants reproduce the per-layer launches of the default network within 10 %
// (71 / 116 / 169 / 169 / 563 us predicted at the 8x8 ... 128x128 levels against 68 / 105 / 153 / 162 / 568 measured, round 4).
static void group_plan(int n, int tilesMN, int ktiles, int& nsplit, int& ktiles_per_split) {
    const int T = n * tilesMN;
    double best = 1e30;
    nsplit = 1;
    for (int r = 0; r <= 4; ++r) {  // r = 0: no split at all, whatever the number of rounds
        int ns = r == 0 ? 1 : (256 * r) / T;
        if (ns < 1) ns = 1;
        if (ns > ktiles) ns = ktiles;
        const int per = (ktiles + ns - 1) / ns;
        ns = (ktiles + per - 1) / per;
        const int rounds = (T * ns + 255) / 256;
        const double cost = rounds * (per * 4.1 + (ns > 1 ? 28.0 : 16.0)) + (ns > 1 ? 10.0 : 0.0);
        if (cost < best - 1e-9) {
            best = cost;
            nsplit = ns;
        }
    }
    ktiles_per_split = (ktiles + nsplit - 1) / nsplit;
}

template <int ESZ, bool PAIR>
static size_t group_ws_need(const C2wConvArgs& a, int n) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    int ktiles, tilesMN, ns1, per1, nsplit, per;
    split_plan<ESZ, PAIR>(a, ktiles, tilesMN, ns1, per1);
    group_plan(n, tilesMN, ktiles, nsplit, per);
    return nsplit > 1 ? (size_t)n * nsplit * tilesMN * 9 * COT * CIB * sizeof(float) : 0;
}

template <typename T, bool PAIR>
int launch_group(const C2wConvArgs& a, const C2wWgradItem* items, int n, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    WpGroupArgs g;
    WpArgs& p = g.c;
    p.dy = nullptr; p.x = nullptr; p.dw = nullptr; p.db = nullptr;
    p.B = a.B; p.H = a.Hout; p.W = a.Wout; p.Cin = a.Cin; p.Cout = a.Cout; p.ldy = a.ldy;
    p.up = a.mode == C2W_CONV_UP ? 1 : 0;
    int tilesMN, ns1, per1, nsplit;
    split_plan<ESZ, PAIR>(a, p.ktiles, tilesMN, ns1, per1);
    group_plan(n, tilesMN, p.ktiles, nsplit, p.ktiles_per_split);
    const size_t item_floats = (size_t)nsplit * tilesMN * 9 * COT * CIB;
    if (nsplit > 1 && (ws == nullptr || (size_t)n * item_floats * sizeof(float) > ws_bytes)) return C2W_ERR_BAD_ARG;  // the caller sized it with c2w_conv_wgrad_grouped_workspace_bytes
    p.ws = nsplit > 1 ? ws : nullptr;
    p.direct = nsplit == 1 ? 1 : 0;
    g.n = n;
    g.live_per_item = tilesMN * nsplit;
    g.ws_item_floats = item_floats;
    for (int i = 0; i < WP_MAX_ITEMS; ++i) {
        const C2wWgradItem& e = items[i < n ? i : n - 1];
        g.item[i].dy = e.dy; g.item[i].x = e.x; g.item[i].dw = e.dw; g.item[i].db = e.dbias;
    }
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)wgrad_patch_group_kernel<T, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    wgrad_patch_group_kernel<T, PAIR><<<g.live_per_item * n, NTHREADS, LDS_BYTES, st>>>(g);
    if (p.ws != nullptr) {
        const size_t per_split = (size_t)tilesMN * 9 * COT * CIB;
        const int grid = (int)std::min<size_t>((per_split / 4 + C2W_RED_COLS - 1) / C2W_RED_COLS, 8192);
        wgrad_reduce_group_kernel<COT, CIB><<<dim3(grid, n), 256, 0, st>>>(g, nsplit, tilesMN, a.Cin / CIB);
    }
    return (int)hipGetLastError();
}

}  // namespace

// One grouped launch serves these layers: halo-patch geometry, 2 ... WP_MAX_ITEMS of them, not the narrow-M form of the output conv
bool c2w_wgrad_patch_group_eligible(const C2wConvArgs& a, int n, int dtype) {
    const bool narrow = dtype != C2W_DTYPE_F32 && !c2w_wgrad_patch_pair(a) && a.Cout <= 80 && c2w_knobs().wgrad_narrow;  // c2w_wgrad_patch's rule
    return c2w_wgrad_patch_eligible(a) && n >= 2 && n <= WP_MAX_ITEMS && !narrow && !c2w_knobs().wgrad_atomics;
}

int c2w_wgrad_patch_group(const C2wConvArgs& a, const C2wWgradItem* items, int n, float* ws, size_t ws_bytes, int dtype, hipStream_t st) {
    if (c2w_wgrad_patch_pair(a)) {
        if (dtype == C2W_DTYPE_F32) return launch_group<float, true>(a, items, n, ws, ws_bytes, st);
        if (dtype == C2W_DTYPE_BF16) return launch_group<bf16_t, true>(a, items, n, ws, ws_bytes, st);
        if (dtype == C2W_DTYPE_F16) return launch_group<f16_t, true>(a, items, n, ws, ws_bytes, st);
        return C2W_ERR_BAD_ARG;
    }
    if (dtype == C2W_DTYPE_F32) return launch_group<float, false>(a, items, n, ws, ws_bytes, st);
    if (dtype == C2W_DTYPE_BF16) return launch_group<bf16_t, false>(a, items, n, ws, ws_bytes, st);
    if (dtype == C2W_DTYPE_F16) return launch_group<f16_t, false>(a, items, n, ws, ws_bytes, st);
    return C2W_ERR_BAD_ARG;
}

size_t c2w_wgrad_patch_group_ws_bytes(const C2wConvArgs& a, int n, int dtype) {
    const bool pair = c2w_wgrad_patch_pair(a);
    if (dtype == C2W_DTYPE_F32) return pair ? group_ws_need<4, true>(a, n) : group_ws_need<4, false>(a, n);
    return pair ? group_ws_need<2, true>(a, n) : group_ws_need<2, false>(a, n);
}

bool c2w_wgrad_patch_pair(const C2wConvArgs& a) {  // 8-pixel-wide images: two per K tile
    return c2w_knobs().conv_pair && a.mode == C2W_CONV_S1 && a.Win == 8 && a.Hin == 8;
}

bool c2w_wgrad_patch_eligible(const C2wConvArgs& a) {
    if (a.mode == C2W_CONV_UP)  // nearest-neighbour x2 upsampling folded into the patch load
        return a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && (a.Hout & 7) == 0 && (a.Wout & 15) == 0 && c2w_knobs().up_patch;
    return a.mode == C2W_CONV_S1 && a.Hin == a.Hout && a.Win == a.Wout && (a.Hin & 7) == 0 && ((a.Win & 15) == 0 || c2w_wgrad_patch_pair(a));
}

int c2w_wgrad_patch(const C2wConvArgs& a, float* dw, float* db, float* ws, size_t ws_bytes, int dtype, hipStream_t st) {
    if (c2w_wgrad_patch_pair(a)) {
        if (dtype == C2W_DTYPE_F32) return launch<float, true>(a, dw, db, ws, ws_bytes, st);
        if (dtype == C2W_DTYPE_BF16) return launch<bf16_t, true>(a, dw, db, ws, ws_bytes, st);
        if (dtype == C2W_DTYPE_F16) return launch<f16_t, true>(a, dw, db, ws, ws_bytes, st);
        return C2W_ERR_BAD_ARG;
    }
    if (dtype == C2W_DTYPE_F32) return launch<float, false>(a, dw, db, ws, ws_bytes, st);
    const bool narrow = a.Cout <= 80 && c2w_knobs().wgrad_narrow;  // the output conv (65 channels)
    if (dtype == C2W_DTYPE_BF16) return narrow ? launch<bf16_t, false, true>(a, dw, db, ws, ws_bytes, st) : launch<bf16_t, false>(a, dw, db, ws, ws_bytes, st);
    if (dtype == C2W_DTYPE_F16) return narrow ? launch<f16_t, false, true>(a, dw, db, ws, ws_bytes, st) : launch<f16_t, false>(a, dw, db, ws, ws_bytes, st);
    return C2W_ERR_BAD_ARG;
}

size_t c2w_wgrad_patch_ws_bytes(const C2wConvArgs& a, int dtype) {
    const bool pair = c2w_wgrad_patch_pair(a);
    if (dtype == C2W_DTYPE_F32) return pair ? ws_need<4, true>(a) : ws_need<4, false>(a);
    return pair ? ws_need<2, true>(a) : ws_need<2, false>(a);
}
