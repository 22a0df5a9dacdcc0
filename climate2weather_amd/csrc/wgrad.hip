// Weight-gradient GEMM on MFMA for gfx950:   dW[co][tap][ci] += sum_q dY[q][co] * X[src(q,tap)][ci]
// (the wgrad third of the 346 GFLOP/window training step; replaces autograd's conv/linear weight backward
// for model/nn.py:45,47,149,155,157,169-174,185-194 and model/score.py:56-57).
//
// GEMM view: M = output channel, N = (tap, input channel), K = output pixel -- the reduction runs over the
// NHWC *row* index, so both MFMA operands are K-strided in memory.  Tiles are staged exactly as they lie in HBM
// ([pixel][channel], direct-to-LDS 16-B loads, zero padding by out-of-range buffer offsets) and transposed on the
// way to the matrix core:  bf16 -> ds_read_b64_tr_b16 (hardware transpose read),  fp32 -> one dword per lane.
// LDS rows are XOR-swizzled on 16-B chunks (chunk ^= ((row&3)<<2 | (row>>2)&3)), applied on the source address,
// which makes both kinds of read bank-conflict free.
// Block tile (same bytes for both dtypes): 256 B of output channels x 4 "column blocks" of 128 B of (tap, ci)
// x 64 pixels per stage; 8 waves = 2 (M) x 4 (N: one column block each).  Split-K over pixel ranges across
// workgroups; partial tiles are combined with fp32 global atomics issued as 256-B contiguous wave instructions
// (tile staged through LDS first).  dW must be zero-initialised (or hold the value to accumulate onto).
#include <cstdint>
#include <cstdlib>

#include "conv_geom.h"
#include "knobs.h"

namespace {

constexpr int KT = 64;  // pixels per stage
constexpr int NTHREADS = 512;
constexpr int ABYTES = KT * 256;  // dY stage  (16 KiB)
constexpr int BBYTES = KT * 512;  // X stage   (32 KiB)
constexpr int NSLOT = 3;          // stage ring depth: loads run two stages ahead of the MFMAs

struct WgradArgs {
    const void* dy;  // [npix][ldy]
    const void* x;   // [B][Hin][Win][Cin]
    float* dw;       // [Cout][NT][Cin] fp32, accumulated
    float* db;       // [Cout] fp32 bias gradient (column sums of dY), accumulated; may be null
    int B, Hin, Win, Cin, Hout, Wout, Cout, ldy;
    int nsplit, ktiles_per_split;
    FastDiv div_hw, div_w;
    float* ws;  // optional scratch [split][tile][COT][4*CIB] fp32 for the partial sums (else fp32 atomics)
};

__device__ __forceinline__ uint32_t swz(int row) { return (uint32_t)(((row & 3) << 2) | ((row >> 2) & 3)); }

// Grouped launch (round 5): the weight gradients of up to WG_MAX_ITEMS layers of ONE geometry as one grid -- the six qkv / six proj_out
// 1x1 convs of the attention level (model/nn.py:45,47), whose per-layer launches split 128 K stages over 10-20 workgroups each and run at
// 0.05-0.11 of peak.  Same scheme as wgrad_patch_group_kernel: per-layer pointers in the kernel arguments, read through the kernarg
// segment with a workgroup-uniform index; layer = position in the grid-wide XCD-contiguous order / workgroups per layer.
struct WgItem {
    const void* dy;
    const void* x;
    float* dw;
    float* db;
};
constexpr int WG_MAX_ITEMS = 16;
struct WgradGroupArgs {
    WgradArgs c;  // geometry, split plan, ws = base of the group's workspace
    int n, live_per_item;
    unsigned long long ws_item_floats;
    WgItem item[WG_MAX_ITEMS];
};

__device__ __forceinline__ int wg_xcd_order(int bid, int nblk) {
    const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// L: the workgroup's position in the (split, output tile) order of one layer
template <typename T, int MODE>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int L) {
    constexpr int ESZ = sizeof(T);
    constexpr int NT = (MODE == C2W_CONV_1X1) ? 1 : 9;
    constexpr int COT = 256 / ESZ;  // output channels per block tile
    constexpr int CIB = 128 / ESZ;  // input channels per column block
    constexpr int MT = (ESZ == 2) ? 4 : 2;  // 16x16 MFMA tiles per wave along M
    constexpr int NTL = MT;                 // and along N (one column block per wave)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Abuf = smem;
    char* const Bbuf = smem + NSLOT * ABYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;

    const int cib_per_tap = p.Cin / CIB;
    const int nb_total = NT * cib_per_tap;       // column blocks in the whole (tap, ci) axis
    const int tilesN = (nb_total + 3) / 4;
    const int tilesM = (p.Cout + COT - 1) / COT;
    const int tilesMN = tilesM * tilesN;
    const int split = L / tilesMN, mn = L - split * tilesMN;
    const int tm = mn / tilesN, tn = mn - tm * tilesN;
    const int co0 = tm * COT;
    const int HWo = p.Hout * p.Wout;
    const int npix = p.B * HWo;
    const int nkt = (npix + KT - 1) / KT;
    const int kt0 = split * p.ktiles_per_split;
    const int kt1 = (kt0 + p.ktiles_per_split < nkt) ? kt0 + p.ktiles_per_split : nkt;

    // ---- staging slots.  dY: 64 rows x 16 chunks = 2 rounds; X: 64 rows x 32 chunks = 4 rounds.
    uint32_t avo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + NTHREADS * i, row = s >> 4, pc = s & 15;
        const uint32_t lc = (uint32_t)pc ^ swz(row);
        const int c = co0 + (int)lc * (16 / ESZ);
        avo[i] = (c < p.Cout) ? (uint32_t)row * (uint32_t)(p.ldy * ESZ) + (uint32_t)c * ESZ : C2W_OOB;
    }
    int brow[4], btap[4];
    uint32_t bco[4];  // byte offset inside the source pixel row, or OOB when the column block does not exist
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + NTHREADS * i, row = s >> 5, pc = s & 31;
        const uint32_t lc = (uint32_t)pc ^ swz(row);
        const int nb = tn * 4 + (int)(lc >> 3);
        const int tap = nb / cib_per_tap;
        brow[i] = row;
        btap[i] = (nb < nb_total) ? tap : -1;
        bco[i] = (uint32_t)((nb - tap * cib_per_tap) * 128 + (lc & 7) * 16);
    }
    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESZ;
    const size_t dy_total = (size_t)npix * p.ldy * ESZ;

    auto issue = [&](int kt, int buf) {
        const int q0 = kt * KT;
        // dY rows q0 .. q0+63 (rows past the end fall outside the descriptor -> zeros)
        const size_t aoff = (size_t)q0 * p.ldy * ESZ;
        size_t arem = dy_total - aoff;
        if (arem > 0x7fffffffu) arem = 0x7fffffffu;
        const __amdgpu_buffer_rsrc_t ra = make_rsrc((const char*)p.dy + aoff, (uint32_t)arem);
        char* const adst = Abuf + buf * ABYTES + wid * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16(ra, adst + i * 8192, avo[i], 0);
        // X rows gathered through the tap geometry, rebased at the first image of this K tile
        const int b0 = fast_div(q0, p.div_hw);
        size_t xrem = (size_t)(p.B - b0) * img_bytes;
        if (xrem > 0x7fffffffu) xrem = 0x7fffffffu;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b0 * img_bytes, (uint32_t)xrem);
        char* const bdst = Bbuf + buf * BBYTES + wid * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Q = q0 + brow[i];
            const int b = fast_div(Q, p.div_hw);
            const int rem = Q - b * HWo;
            const int oh = fast_div(rem, p.div_w);
            const int ow = rem - oh * p.Wout;
            const int tap = btap[i];
            const int kh = tap / 3, kw = tap - kh * 3;
            int ih, iw;
            const bool ok = src_pixel<MODE>(p, oh, ow, kh, kw, ih, iw) && tap >= 0 && Q < npix;
            const uint32_t voff = ok ? (uint32_t)((((b - b0) * p.Hin + ih) * p.Win + iw) * p.Cin) * ESZ + bco[i] : C2W_OOB;
            glds16(rx, bdst + i * 8192, voff, 0);
        }
    };

    f32x4_t acc[MT][NTL];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // bias gradient = dY^T . 1: the wave that owns column block 0 of column tile 0 multiplies its dY fragments by ones
    const bool do_bias = p.db != nullptr && tn == 0 && wn == 0;
    f32x4_t accb[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) accb[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // ---- fragment read offsets
    uint32_t offA[MT][2], offB[NTL][2];  // bf16: [tile][h]; fp32: [tile][0] holds the (row-independent) part
    if constexpr (ESZ == 2) {
        const int q = li >> 2, pp = li & 3;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = 8 * lg + q + 4 * h;  // + 32*ks
            const uint32_t f = swz(row);
#pragma unroll
            for (int m = 0; m < MT; ++m)
                offA[m][h] = (uint32_t)(row * 256 + ((((wm * 8 + m * 2 + (pp >> 1)) ^ f) & 15) << 4) + 8 * (pp & 1));
#pragma unroll
            for (int n = 0; n < NTL; ++n) {
                const uint32_t lcg = (uint32_t)(wn * 8 + n * 2 + (pp >> 1));
                offB[n][h] = (uint32_t)(row * 512 + ((lcg ^ f) << 4) + 8 * (pp & 1));
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            offA[m][0] = (uint32_t)((wm * 8 + m * 4 + (li >> 2)) ^ (lg << 2));  // chunk ^ (row&3)<<2 ; (kk&3) xored per step
            offA[m][1] = (uint32_t)((li & 3) * 4);
        }
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            offB[n][0] = (uint32_t)((wn * 8 + n * 4 + (li >> 2)) ^ (lg << 2));
            offB[n][1] = (uint32_t)((li & 3) * 4);
        }
    }

    if (kt0 < kt1) issue(kt0, 0);
    if (kt0 + 1 < kt1) issue(kt0 + 1, 1);
    int buf = 0;
    for (int kt = kt0; kt < kt1; ++kt) {
        // 6 LDS-DMA loads per wave per stage: all but the youngest 6 retired <=> this stage has landed
        if (kt + 1 < kt1) {
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < kt1) issue(kt + 2, buf >= 1 ? buf - 1 : NSLOT - 1);
        const char* const Ab = Abuf + buf * ABYTES;
        const char* const Bb = Bbuf + buf * BBYTES;
        if constexpr (ESZ == 2) {
            // Fragment reads in inline asm with counted lgkmcnt waits (see wgrad_patch.hip: through the ds_read_tr16_b64 builtin hipcc
            // waits vmcnt(0) -- for the stages just issued -- in front of the first read of every stage, and the loads never overlap
            // the MFMAs).
            const uint32_t sA = (uint32_t)(uintptr_t)Ab, sB = (uint32_t)(uintptr_t)Bb;
            auto rd = [&](tr_frag& f, uint32_t base, uint32_t o0, uint32_t o1, auto IMMc) {
                constexpr int IMM = decltype(IMMc)::value;
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(base + o0), "n"(IMM));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(base + o1), "n"(IMM));
            };
            static_for<2>([&](auto KSc) {
                constexpr int ks = decltype(KSc)::value;
                tr_frag a[MT], b[NTL];
#pragma unroll
                for (int m = 0; m < MT; ++m) rd(a[m], sA, offA[m][0], offA[m][1], IC<ks * 32 * 256>{});
                rd(b[0], sB, offB[0][0], offB[0][1], IC<ks * 32 * 512>{});
                rd(b[1], sB, offB[1][0], offB[1][1], IC<ks * 32 * 512>{});
                static_for<NTL>([&](auto Nc) {
                    constexpr int n = decltype(Nc)::value;
                    if constexpr (n + 2 < NTL) rd(b[n + 2], sB, offB[n + 2][0], offB[n + 2][1], IC<ks * 32 * 512>{});
                    constexpr int AHEAD = 2 * ((n + 2 < NTL ? n + 2 : NTL - 1) - n);  // reads younger than b[n]'s
                    if constexpr (AHEAD == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                    else if constexpr (AHEAD == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8_t bv = b[n].vec();
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m][n] = mfma16s<T>(a[m].vec(), bv, acc[m][n]);
                    if (n == NTL - 1 && do_bias) {
                        const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones16<T>());
#pragma unroll
                        for (int m = 0; m < MT; ++m) accb[m] = mfma16s<T>(a[m].vec(), ones, accb[m]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {  // 4 pixels per MFMA: lane (i,g) feeds k = 4*kk + g
                const int row = kk * 4 + lg;
                float a[MT], b[NTL];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = *(const float*)(Ab + row * 256 + ((offA[m][0] ^ (kk & 3)) << 4) + offA[m][1]);
#pragma unroll
                for (int n = 0; n < NTL; ++n) b[n] = *(const float*)(Bb + row * 512 + ((offB[n][0] ^ (kk & 3)) << 4) + offB[n][1]);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTL; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[m][n], 0, 0, 0);
                if (do_bias) {
#pragma unroll
                    for (int m = 0; m < MT; ++m) accb[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], 1.0f, accb[m], 0, 0, 0);
                }
            }
        }
        buf = buf + 1 == NSLOT ? 0 : buf + 1;
    }

    // ---- epilogue: tile -> LDS [co][column] fp32 -> atomics, one (co, column block) row per wave instruction
    constexpr int NCOL = 4 * CIB;
    constexpr int OS = NCOL + 4;  // padded row stride (floats)
    __syncthreads();
    float* const O = (float*)smem;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTL; ++n) {
            const int col = wn * CIB + n * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) O[(wm * (COT / 2) + m * 16 + lg * 4 + r) * OS + col] = acc[m][n][r];
        }
    __syncthreads();
    if (kt0 >= kt1) return;
    if (do_bias && li == 0) {  // every column of accb holds the same sums; lane (li=0, lg) owns rows 4*lg .. 4*lg+3
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * (COT / 2) + m * 16 + lg * 4 + r;
                if (co < p.Cout) atomicAdd(p.db + co, accb[m][r]);
            }
    }
    if (p.ws != nullptr) {  // partial tile by coalesced stores; wgrad_gather_reduce_kernel adds the splits into dw
        float* const dst = p.ws + ((size_t)split * tilesMN + mn) * (COT * NCOL);
        for (int idx = tid; idx < COT * NCOL; idx += NTHREADS) dst[idx] = O[(idx / NCOL) * OS + (idx % NCOL)];
        return;
    }
    for (int idx = tid; idx < COT * NCOL; idx += NTHREADS) {
        const int row = idx / NCOL, col = idx - row * NCOL;
        const int cb = col / CIB, cil = col - cb * CIB;
        const int nb = tn * 4 + cb;
        const int co = co0 + row;
        if (nb < nb_total && co < p.Cout) {
            const int tap = nb / cib_per_tap;
            const int ci = (nb - tap * cib_per_tap) * CIB + cil;
            atomicAdd(p.dw + ((size_t)co * NT + tap) * p.Cin + ci, O[row * OS + col]);
        }
    }
}

template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_kernel(const WgradArgs p) {
    wgrad_body<T, MODE>(p, wg_xcd_order((int)blockIdx.x, (int)gridDim.x));
}

__device__ __forceinline__ WgItem wg_item(int it) {  // it: workgroup-uniform
    typedef __attribute__((address_space(4))) const char karg_t;
    const WgItem __attribute__((address_space(4)))* tab =
        (const WgItem __attribute__((address_space(4)))*)((karg_t*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(WgradGroupArgs, item));
    WgItem r;
    r.dy = tab[it].dy;
    r.x = tab[it].x;
    r.dw = tab[it].dw;
    r.db = tab[it].db;
    return r;
}

template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void wgrad_group_kernel(const WgradGroupArgs g) {
    const int Lg = wg_xcd_order((int)blockIdx.x, (int)gridDim.x);
    const int it = __builtin_amdgcn_readfirstlane(Lg / g.live_per_item);
    const WgItem e = wg_item(it);
    WgradArgs p = g.c;
    p.dy = e.dy;
    p.x = e.x;
    p.dw = e.dw;
    p.db = e.db;
    p.ws = g.c.ws != nullptr ? g.c.ws + (size_t)it * g.ws_item_floats : nullptr;
    wgrad_body<T, MODE>(p, Lg - it * g.live_per_item);
}

// dw += sum over splits of the partial tiles (same 64 x 4 layout of a block as wgrad_reduce_kernel in wgrad_patch.hip)
template <int COT, int CIB, int NT>
__device__ __forceinline__ void wgrad_gather_reduce_body(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int tilesM,
                                                         int tilesN, int Cin, int Cout) {
    constexpr int NCOL = 4 * CIB;
    __shared__ f32x4_t red[4][64];
    const size_t per4 = (size_t)tilesM * tilesN * COT * NCOL / 4;
    const f32x4_t* ws4 = (const f32x4_t*)ws;
    const int q = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int cib_per_tap = Cin / CIB, nb_total = NT * cib_per_tap;
    for (size_t base = (size_t)blockIdx.x * 64; base < per4; base += (size_t)gridDim.x * 64) {
        const size_t i4 = base + q;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        if (i4 < per4) {
#pragma unroll 8
            for (int sidx = grp; sidx < nsplit; sidx += 4) acc += ws4[(size_t)sidx * per4 + i4];
        }
        red[grp][q] = acc;
        __syncthreads();
        if (grp == 0 && i4 < per4) {
            acc = red[0][q] + red[1][q] + red[2][q] + red[3][q];
            const size_t i = i4 * 4;
            const int col = (int)(i % NCOL);
            size_t r = i / NCOL;
            const int row = (int)(r % COT);
            const int mn = (int)(r / COT);
            const int tm = mn / tilesN, tn = mn - tm * tilesN;
            const int cb = col / CIB, cil = col - cb * CIB;
            const int nb = tn * 4 + cb;
            const int co = tm * COT + row;
            if (nb < nb_total && co < Cout) {
                const int tap = nb / cib_per_tap;
                const int ci = (nb - tap * cib_per_tap) * CIB + cil;
                f32x4_t* d = (f32x4_t*)(dw + ((size_t)co * NT + tap) * Cin + ci);
                *d = *d + acc;
            }
        }
        __syncthreads();
    }
}

template <int COT, int CIB, int NT>
__global__ __launch_bounds__(256) void wgrad_gather_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, int tilesM,
                                                                  int tilesN, int Cin, int Cout) {
    wgrad_gather_reduce_body<COT, CIB, NT>(ws, dw, nsplit, tilesM, tilesN, Cin, Cout);
}
template <int COT, int CIB, int NT>
__global__ __launch_bounds__(256) void wgrad_gather_reduce_group_kernel(const WgradGroupArgs g, int nsplit, int tilesM, int tilesN) {
    const int it = (int)blockIdx.y;
    const WgItem e = wg_item(it);
    wgrad_gather_reduce_body<COT, CIB, NT>(g.c.ws + (size_t)it * g.ws_item_floats, e.dw, nsplit, tilesM, tilesN, g.c.Cin, g.c.Cout);
}

FastDiv make_div(uint32_t d) {
    FastDiv r{0, 0};
    if (d <= 1) return r;
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;  // s = ceil(log2 d) >= 1
    r.magic = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
    r.shift = s - 1;
    return r;
}

// split of the K (pixel) range: one resident wave of workgroups (2 per CU): measured best on the 8x8 / stride-2 / upsampling shapes
// (atomics grow with the split)
template <int ESZ, int NT>
static void split_plan(const C2wConvArgs& a, int& tilesM, int& tilesN, int& nsplit, int& ktiles_per_split) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    const long long npix = (long long)a.B * a.Hout * a.Wout;
    const int nkt = (int)((npix + KT - 1) / KT);
    tilesN = (NT * (a.Cin / CIB) + 3) / 4;
    tilesM = (a.Cout + COT - 1) / COT;
    const int tilesMN = tilesM * tilesN;
    constexpr int target = 432;  // workgroups a launch aims for (measured best of 256 / 432 / 512 / 768 at the bench sizes)
    nsplit = (target + tilesMN - 1) / tilesMN;
    if (nsplit > nkt) nsplit = nkt;
    if (nsplit < 1) nsplit = 1;
    ktiles_per_split = (nkt + nsplit - 1) / nsplit;
    nsplit = (nkt + ktiles_per_split - 1) / ktiles_per_split;
}

template <int ESZ, int NT>
static size_t ws_need(const C2wConvArgs& a) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    int tilesM, tilesN, nsplit, per;
    split_plan<ESZ, NT>(a, tilesM, tilesN, nsplit, per);
    return nsplit > 1 ? (size_t)nsplit * tilesM * tilesN * COT * 4 * CIB * sizeof(float) : 0;
}

template <typename T, int MODE>
int launch(const C2wConvArgs& a, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int NT = (MODE == C2W_CONV_1X1) ? 1 : 9;
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    WgradArgs p;
    p.dy = a.y; p.x = a.x; p.dw = dw; p.db = db;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Cin = a.Cin; p.Hout = a.Hout; p.Wout = a.Wout; p.Cout = a.Cout; p.ldy = a.ldy;
    int tilesM, tilesN;
    split_plan<ESZ, NT>(a, tilesM, tilesN, p.nsplit, p.ktiles_per_split);
    const int tilesMN = tilesM * tilesN;
    p.div_hw = make_div((uint32_t)(a.Hout * a.Wout));
    p.div_w = make_div((uint32_t)a.Wout);
    constexpr int lds_main = NSLOT * (ABYTES + BBYTES);
    constexpr int lds_epi = COT * (4 * CIB + 4) * 4;
    constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)wgrad_kernel<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const size_t need = (size_t)p.nsplit * tilesMN * COT * 4 * CIB * sizeof(float);
    p.ws = (ws != nullptr && need <= ws_bytes && p.nsplit > 1 && !c2w_knobs().wgrad_atomics) ? ws : nullptr;
    wgrad_kernel<T, MODE><<<tilesMN * p.nsplit, NTHREADS, lds, st>>>(p);
    if (p.ws != nullptr) {
        const size_t per4 = (size_t)tilesMN * COT * 4 * CIB / 4;
        const int grid = (int)((per4 + 63) / 64 < 4096 ? (per4 + 63) / 64 : 4096);
        wgrad_gather_reduce_kernel<COT, CIB, NT><<<grid, 256, 0, st>>>(p.ws, dw, p.nsplit, tilesM, tilesN, a.Cin, a.Cout);
    }
    return (int)hipGetLastError();
}

template <typename T>
int launch_dtype(const C2wConvArgs& a, float* dw, float* db, float* ws, size_t ws_bytes, hipStream_t st) {
    switch (a.mode) {
        case C2W_CONV_1X1: return launch<T, C2W_CONV_1X1>(a, dw, db, ws, ws_bytes, st);
        case C2W_CONV_S1: return launch<T, C2W_CONV_S1>(a, dw, db, ws, ws_bytes, st);
        case C2W_CONV_S2: return launch<T, C2W_CONV_S2>(a, dw, db, ws, ws_bytes, st);
        case C2W_CONV_UP: return launch<T, C2W_CONV_UP>(a, dw, db, ws, ws_bytes, st);
    }
    return C2W_ERR_BAD_ARG;
}

// Split plan of a group: as many splits as bring the whole grid to ~one resident round (2 workgroups per CU), never more than the K stages
static void gather_group_plan(int n, int tilesMN, int nkt, int& nsplit, int& per) {
    nsplit = 480 / (n * tilesMN);
    if (nsplit > nkt) nsplit = nkt;
    if (nsplit < 1) nsplit = 1;
    per = (nkt + nsplit - 1) / nsplit;
    nsplit = (nkt + per - 1) / per;
}

template <int ESZ, int NT>
static size_t gather_group_ws_need(const C2wConvArgs& a, int n) {
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    int tilesM, tilesN, ns1, per1, nsplit, per;
    split_plan<ESZ, NT>(a, tilesM, tilesN, ns1, per1);
    const long long npix = (long long)a.B * a.Hout * a.Wout;
    gather_group_plan(n, tilesM * tilesN, (int)((npix + KT - 1) / KT), nsplit, per);
    return nsplit > 1 ? (size_t)n * nsplit * tilesM * tilesN * COT * 4 * CIB * sizeof(float) : 0;
}

template <typename T, int MODE>
int launch_group(const C2wConvArgs& a, const C2wWgradItem* items, int n, float* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int ESZ = sizeof(T);
    constexpr int NT = (MODE == C2W_CONV_1X1) ? 1 : 9;
    constexpr int COT = 256 / ESZ, CIB = 128 / ESZ;
    WgradGroupArgs g;
    WgradArgs& p = g.c;
    p.dy = nullptr; p.x = nullptr; p.dw = nullptr; p.db = nullptr;
    p.B = a.B; p.Hin = a.Hin; p.Win = a.Win; p.Cin = a.Cin; p.Hout = a.Hout; p.Wout = a.Wout; p.Cout = a.Cout; p.ldy = a.ldy;
    int tilesM, tilesN, ns1, per1;
    split_plan<ESZ, NT>(a, tilesM, tilesN, ns1, per1);
    const int tilesMN = tilesM * tilesN;
    const long long npix = (long long)a.B * a.Hout * a.Wout;
    gather_group_plan(n, tilesMN, (int)((npix + KT - 1) / KT), p.nsplit, p.ktiles_per_split);
    p.div_hw = make_div((uint32_t)(a.Hout * a.Wout));
    p.div_w = make_div((uint32_t)a.Wout);
    const size_t item_floats = (size_t)p.nsplit * tilesMN * COT * 4 * CIB;
    if (p.nsplit > 1 && (ws == nullptr || (size_t)n * item_floats * sizeof(float) > ws_bytes)) return C2W_ERR_BAD_ARG;
    p.ws = p.nsplit > 1 ? ws : nullptr;  // no split: one workgroup per output tile, its atomics onto dw have no partner (deterministic)
    g.n = n;
    g.live_per_item = tilesMN * p.nsplit;
    g.ws_item_floats = item_floats;
    for (int i = 0; i < WG_MAX_ITEMS; ++i) {
        const C2wWgradItem& e = items[i < n ? i : n - 1];
        g.item[i].dy = e.dy; g.item[i].x = e.x; g.item[i].dw = e.dw; g.item[i].db = e.dbias;
    }
    constexpr int lds_main = NSLOT * (ABYTES + BBYTES);
    constexpr int lds_epi = COT * (4 * CIB + 4) * 4;
    constexpr int lds = lds_main > lds_epi ? lds_main : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)wgrad_group_kernel<T, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    wgrad_group_kernel<T, MODE><<<g.live_per_item * n, NTHREADS, lds, st>>>(g);
    if (p.ws != nullptr) {
        const size_t per4 = (size_t)tilesMN * COT * 4 * CIB / 4;
        const int grid = (int)((per4 + 63) / 64 < 4096 ? (per4 + 63) / 64 : 4096);
        wgrad_gather_reduce_group_kernel<COT, CIB, NT><<<dim3(grid, n), 256, 0, st>>>(g, p.nsplit, tilesM, tilesN);
    }
    return (int)hipGetLastError();
}

}  // namespace

// Geometry is passed with the forward call's argument block: x = the forward input, y = dY (gradient w.r.t. the forward
// output, [B*Hout*Wout][ldy]); w/bias/res/mul/act are ignored.  dw is [Cout][taps][Cin] fp32 and is accumulated into;
// dbias (optional) receives the bias gradient sum_q dY[q][co] from the same pass over dY.
static int wgrad_check(const C2wConvArgs* a, int dtype) {
    if (a == nullptr) return C2W_ERR_BAD_ARG;
    if (dtype != C2W_DTYPE_F32 && dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16) return C2W_ERR_BAD_ARG;
    const int esz = dtype == C2W_DTYPE_F32 ? 4 : 2;
    if (a->Cin <= 0 || a->Cin % (128 / esz) != 0) return C2W_ERR_BAD_SHAPE;
    if (a->Cout <= 0 || a->ldy % (16 / esz) != 0 || a->Cout > a->ldy) return C2W_ERR_BAD_SHAPE;
    if (a->B <= 0 || a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0) return C2W_ERR_BAD_SHAPE;
    if ((long long)a->B * a->Hout * a->Wout >= (1ll << 31)) return C2W_ERR_BAD_SHAPE;
    if (a->mode != C2W_CONV_1X1 && a->mode != C2W_CONV_S1 && a->mode != C2W_CONV_S2 && a->mode != C2W_CONV_UP) return C2W_ERR_BAD_ARG;
    return 0;
}

extern "C" int c2w_conv_wgrad(const C2wConvArgs* a, float* dw, float* dbias, void* workspace, unsigned long long workspace_bytes, int dtype,
                              void* stream) {
    const int rc = wgrad_check(a, dtype);
    if (rc != 0) return rc;
    if (a->x == nullptr || a->y == nullptr || dw == nullptr) return C2W_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    const size_t wsb = workspace == nullptr ? 0 : (size_t)workspace_bytes;
    if (c2w_wgrad_patch_eligible(*a) && !c2w_knobs().force_gather) return c2w_wgrad_patch(*a, dw, dbias, ws, wsb, dtype, st);
    if (dtype == C2W_DTYPE_F32) return launch_dtype<float>(*a, dw, dbias, ws, wsb, st);
    if (dtype == C2W_DTYPE_BF16) return launch_dtype<bf16_t>(*a, dw, dbias, ws, wsb, st);
    return launch_dtype<f16_t>(*a, dw, dbias, ws, wsb, st);
}

// the 1x1 layers (Conv1d(k = 1) / Linear: model/nn.py:45,47) group on the gather kernel
static bool gather_group_eligible(const C2wConvArgs& a, int n) {
    return a.mode == C2W_CONV_1X1 && n >= 2 && n <= WG_MAX_ITEMS && !c2w_knobs().wgrad_atomics;
}

extern "C" int c2w_conv_wgrad_grouped_supported(const C2wConvArgs* a, int n, int dtype) {
    if (wgrad_check(a, dtype) != 0) return 0;
    if (c2w_wgrad_patch_group_eligible(*a, n, dtype) && !c2w_knobs().force_gather) return 1;
    return gather_group_eligible(*a, n) ? 1 : 0;
}

extern "C" long long c2w_conv_wgrad_grouped_workspace_bytes(const C2wConvArgs* a, int n, int dtype) {
    const int rc = wgrad_check(a, dtype);
    if (rc != 0) return rc;
    if (!c2w_conv_wgrad_grouped_supported(a, n, dtype)) return C2W_ERR_UNSUPPORTED;
    if (c2w_wgrad_patch_group_eligible(*a, n, dtype) && !c2w_knobs().force_gather) return (long long)c2w_wgrad_patch_group_ws_bytes(*a, n, dtype);
    return (long long)(dtype == C2W_DTYPE_F32 ? gather_group_ws_need<4, 1>(*a, n) : gather_group_ws_need<2, 1>(*a, n));
}

extern "C" int c2w_conv_wgrad_grouped(const C2wConvArgs* a, const C2wWgradItem* items, int n, void* workspace, unsigned long long workspace_bytes,
                                      int dtype, void* stream) {
    const int rc = wgrad_check(a, dtype);
    if (rc != 0) return rc;
    if (items == nullptr || n < 1) return C2W_ERR_BAD_ARG;
    for (int i = 0; i < n; ++i)
        if (items[i].x == nullptr || items[i].dy == nullptr || items[i].dw == nullptr) return C2W_ERR_BAD_ARG;
    if (!c2w_conv_wgrad_grouped_supported(a, n, dtype)) return C2W_ERR_UNSUPPORTED;
    float* const ws = (float*)workspace;
    const size_t wsb = workspace == nullptr ? 0 : (size_t)workspace_bytes;
    hipStream_t st = (hipStream_t)stream;
    if (c2w_wgrad_patch_group_eligible(*a, n, dtype) && !c2w_knobs().force_gather) return c2w_wgrad_patch_group(*a, items, n, ws, wsb, dtype, st);
    if (dtype == C2W_DTYPE_F32) return launch_group<float, C2W_CONV_1X1>(*a, items, n, ws, wsb, st);
    if (dtype == C2W_DTYPE_BF16) return launch_group<bf16_t, C2W_CONV_1X1>(*a, items, n, ws, wsb, st);
    return launch_group<f16_t, C2W_CONV_1X1>(*a, items, n, ws, wsb, st);
}

extern "C" int c2w_conv_wgrad_dispatch(const C2wConvArgs* a, int dtype) {
    const int rc = wgrad_check(a, dtype);
    if (rc != 0) return rc;
    if (c2w_wgrad_patch_eligible(*a) && !c2w_knobs().force_gather) return c2w_wgrad_patch_pair(*a) ? C2W_KERNEL_PATCH_PAIR : C2W_KERNEL_PATCH_8X16;
    return C2W_KERNEL_GATHER;
}

// Bytes of scratch c2w_conv_wgrad would use for this geometry (0: the launch does not split its reduction), or a negative status.
extern "C" long long c2w_conv_wgrad_workspace_bytes(const C2wConvArgs* a, int dtype) {
    const int rc = wgrad_check(a, dtype);
    if (rc != 0) return rc;
    if (c2w_wgrad_patch_eligible(*a) && !c2w_knobs().force_gather) return (long long)c2w_wgrad_patch_ws_bytes(*a, dtype);
    const int esz = dtype == C2W_DTYPE_F32 ? 4 : 2;
    if (a->mode == C2W_CONV_1X1) return (long long)(esz == 4 ? ws_need<4, 1>(*a) : ws_need<2, 1>(*a));
    return (long long)(esz == 4 ? ws_need<4, 9>(*a) : ws_need<2, 9>(*a));
}
