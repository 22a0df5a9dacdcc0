// Implicit-GEMM convolution on MFMA for gfx950 -- the forward/dgrad work-horse.
//
// Replaces every torch Conv2d / Conv1d(k=1) / Linear on the reference hot path
// (model/nn.py:45,47,149,155,157,169-174,185-194; model/score.py:56-57) and, with
// re-arranged weights, their input gradients.
//
// GEMM view (per tap t and K-chunk c):   D[co][pixel] += W[co][t][c*CK..] . X[src(pixel,t)][c*CK..]
//   * MFMA "A" operand = weights  (rows = output channel)
//   * MFMA "B" operand = pixels   (cols = output pixel)  -> each lane ends up with 4 consecutive
//     output channels of one pixel = one contiguous NHWC store.
// Block tile: 256 output pixels x 128 output channels, 8 waves (2 over channels x 4 over pixels),
// each wave a 64x64 sub-tile = 4x4 MFMA 16x16 accumulators.
// K loop: stage = (K-chunk of 128 bytes per pixel row, tap).  Both operand tiles are staged with
// direct-to-LDS buffer loads (16 B/lane) into a 2-deep ring; out-of-image taps and out-of-range
// rows are fetched through a buffer descriptor with an out-of-range offset, which returns zeros --
// zero padding costs no instructions.  LDS rows are 128 B and XOR-swizzled on the SOURCE address
// (chunk ^= row&7), so ds_read_b128 fragment reads are bank-conflict free.
// The element type only changes how a 16-byte fragment is fed to the matrix core:
//   bf16 : 1 x v_mfma_f32_16x16x32_bf16     (8 k-values per lane)
//   fp32 : 4 x v_mfma_f32_16x16x4_f32       (4 k-values per lane, exact fp32 fma chain)
#include <cstdlib>

#include "conv_epilogue.h"
#include "knobs.h"

namespace {

constexpr int BN = 128;           // output channels per block tile
constexpr int WBYTES = BN * 128;  // one weight-tile stage (16 KiB)
constexpr int NSLOT = 3;          // ring depth
// NW = pixel-side waves of the block: 4 -> 256 pixels x 128 co, 8 waves (3 x 48 KiB of LDS); 2 -> 128 pixels, 4 waves
// (3 x 32 KiB), used when the larger tile would leave the chip under-filled (8x8 level at B = 128: 128 -> 256 workgroups)
template <int NW> struct GCfg {
    static constexpr int BM = 64 * NW;         // pixels per block tile
    static constexpr int NTHREADS = 128 * NW;
    static constexpr int PBYTES = BM * 128;    // one pixel-tile stage
    static constexpr int ROWS_PER_ROUND = NTHREADS / 8;   // tile rows one round of 16-B pieces covers
    static constexpr int WROUNDS = BN / ROWS_PER_ROUND;   // 2 / 4
    static constexpr int ROUND_BYTES = NTHREADS * 16;     // 8192 / 4096
};

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

template <typename T, int MODE, int NW>
__global__ __launch_bounds__(GCfg<NW>::NTHREADS, 2) void conv_igemm_kernel(const C2wConvArgs p) {
    constexpr int BM = GCfg<NW>::BM, NTHREADS = GCfg<NW>::NTHREADS, PBYTES = GCfg<NW>::PBYTES;
    constexpr int RPR = GCfg<NW>::ROWS_PER_ROUND, WROUNDS = GCfg<NW>::WROUNDS, RB = GCfg<NW>::ROUND_BYTES;
    constexpr int ESZ = sizeof(T);
    constexpr int NT = (MODE == C2W_CONV_1X1) ? 1 : 9;
    constexpr int CK = 128 / ESZ;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 3-slot ring of (pixel tile, weight tile) stages: loads run two stages ahead of the MFMAs
    char* const Pbuf = smem;
    char* const Wbuf = smem + NSLOT * PBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int wm = wid & 1, wn = wid >> 1;

    // ---- block -> (pixel tile, channel tile); blocks b, b+8, ... share an XCD (own L2): give each XCD a
    //      contiguous run of tiles so neighbouring pixel tiles (shared halo rows, same weights) hit its L2.
    const int nN = (p.Cout + BN - 1) / BN;
    const int nblk = gridDim.x;
    int L;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tn = L % nN, tm = L / nN;
    const int co0 = tn * BN;
    const int HWo = p.Hout * p.Wout;
    const int npix = p.B * HWo;

    // ---- TS2 (input gradient of the stride-2 conv): an output pixel (oh, ow) only receives taps with kh = oh + 1 (mod 2),
    //      kw = ow + 1 (mod 2) -- 4, 2, 2 or 1 of the 9.  When the image is even-sized and a quarter of the pixels fills whole
    //      tiles, tiles are formed per parity class (heaviest class first) and their K loop walks only the class's taps.
    bool par = false;
    int pa = 0, pw = 0, psh = 0;  // row / column parity of the tile's class, log2(number of taps)
    const int Wh = p.Wout >> 1, HWq = HWo >> 2, nq = npix >> 2;
    if constexpr (MODE == C2W_CONV_TS2) {
        par = ((p.Hout | p.Wout) & 1) == 0 && nq % BM == 0;
        if (par) {
            const int cls = (tm * BM) / nq;
            pa = cls < 2;
            pw = (cls & 1) == 0;
            psh = pa + pw;
        }
    }
    // ---- buffer descriptors (wave-uniform).  x: rebased at the tile's first image so offsets stay < 2^31.
    const int b0 = (MODE == C2W_CONV_TS2 && par) ? ((tm * BM) % nq) / HWq : (tm * BM) / HWo;
    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESZ;
    size_t xrem = (size_t)(p.B - b0) * img_bytes;
    if (xrem > 0x7fffffffu) xrem = 0x7fffffffu;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)p.x + (size_t)b0 * img_bytes, (uint32_t)xrem);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (uint32_t)((size_t)p.wrows * NT * p.Cin * ESZ));

    // tile row -> output pixel (b, oh, ow); identity order unless the tile belongs to a parity class
    auto row_pixel = [&](int Q, int& b, int& oh, int& ow) {
        if (MODE == C2W_CONV_TS2 && par) {
            const int qq = Q % nq;
            b = qq / HWq;
            const int rem = qq - b * HWq;
            const int yh = rem / Wh;
            oh = 2 * yh + pa;
            ow = 2 * (rem - yh * Wh) + pw;
        } else {
            b = Q / HWo;
            const int rem = Q - b * HWo;
            oh = rem / p.Wout;
            ow = rem - oh * p.Wout;
        }
    };

    // ---- per-thread staging slots: pixel tile = 256 rows x 8 slots(16 B) = 4 rounds; weight tile = 2 rounds
    int pb[4], pyx[4];
    uint32_t plc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = (tid >> 3) + RPR * i;
        const int Q = tm * BM + q;
        int b, oh, ow;
        row_pixel(Q, b, oh, ow);
        pb[i] = (Q < npix) ? (b - b0) * p.Hin : -1;
        pyx[i] = (oh << 16) | ow;
        plc[i] = (uint32_t)(((tid & 7) ^ (q & 7)) << 4);
    }
    uint32_t wvo[WROUNDS];
#pragma unroll
    for (int i = 0; i < WROUNDS; ++i) {
        const int row = (tid >> 3) + RPR * i;
        wvo[i] = (uint32_t)(co0 + row) * (uint32_t)(NT * p.Cin * ESZ) + (uint32_t)(((tid & 7) ^ (row & 7)) << 4);
    }

    const int nchunk = p.Cin / CK;
    const int NS = (MODE == C2W_CONV_TS2 && par) ? (nchunk << psh) : nchunk * NT;

    auto issue = [&](int s, int buf) {
        int chunk, kh, kw;
        if (MODE == C2W_CONV_TS2 && par) {
            chunk = s >> psh;
            const int ti = s - (chunk << psh);  // (kh index, kw index) inside the class: kh in {0,2} if pa else {1}
            kh = pa ? 2 * (ti >> pw) : 1;
            kw = pw ? 2 * (ti & 1) : 1;
        } else {
            chunk = s / NT;
            const int tap0 = s - chunk * NT;
            kh = tap0 / 3;
            kw = tap0 - kh * 3;
        }
        const int tap = kh * 3 + kw;
        char* const pdst = Pbuf + buf * PBYTES + wid * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int ih, iw;
            const bool ok = src_pixel<MODE>(p, pyx[i] >> 16, pyx[i] & 0xffff, kh, kw, ih, iw) && pb[i] >= 0;
            const uint32_t voff = ok ? (uint32_t)(((pb[i] + ih) * p.Win + iw) * p.Cin) * ESZ + plc[i] : C2W_OOB;
            glds16(rx, pdst + i * RB, voff, (uint32_t)chunk * 128u);
        }
        char* const wdst = Wbuf + buf * WBYTES + wid * 1024;
        const uint32_t wso = (uint32_t)(tap * p.Cin + chunk * CK) * ESZ;
#pragma unroll
        for (int i = 0; i < WROUNDS; ++i) glds16(rw, wdst + i * RB, wvo[i], wso);
    };

    // ---- fragment read offsets (bytes inside a stage buffer); k-half ks toggles bit 6
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int row = wm * 64 + m * 16 + li;
        offA[m] = (uint32_t)(row * 128 + ((lg ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int row = wn * 64 + n * 16 + li;
        offB[n] = (uint32_t)(row * 128 + ((lg ^ (row & 7)) << 4));
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    float bv[4][4];  // bias: fetched now, used in the epilogue
    epi_load_bias(p, co0 + wm * 64 + lg * 4, bv);
    issue(0, 0);
    if (NS > 1) issue(1, 1);
    int slot = 0;
    for (int s = 0; s < NS; ++s) {
        // each wave issues 6 LDS-DMA loads per stage: all but the youngest 6 retired <=> stage s has landed
        if (s + 1 < NS) {
            if constexpr (WROUNDS == 2) {
                asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // no fragment read may still be queued in the LDS when its wave arrives here: the slot it reads is refilled right behind the
        // barrier (conv_patch3.hip tells how that went wrong once; today's schedule has nothing outstanding here, the wait pins it)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // stage s landed for every wave; every wave is done reading slot (s+2)%3 (= stage s-1)
        if (s + 2 < NS) issue(s + 2, slot >= 1 ? slot - 1 : NSLOT - 1);
        const char* const Pb = Pbuf + slot * PBYTES;
        const char* const Wb = Wbuf + slot * WBYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4_t a[4], b[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const u32x4_t*)(Wb + (offA[m] ^ (ks * 64)));
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *(const u32x4_t*)(Pb + (offB[n] ^ (ks * 64)));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) Mma<T>::run(a[m], b[n], acc[m][n]);
        }
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
    }

    // ---- epilogue (conv_epilogue.h): bias/activation in registers -> LDS tile [pixel][channel] -> 16-B NHWC stores
    constexpr int OS = BN * ESZ + 16;  // padded row stride
    EpiStore<T, BM, NTHREADS> est;
    est.prefetch(p, tid, co0, [&](int row) -> long long {
        const int Q = tm * BM + row;
        if (Q >= npix) return -1;
        if (MODE == C2W_CONV_TS2 && par) {
            int b, oh, ow;
            row_pixel(Q, b, oh, ow);
            return ((long long)b * p.Hout + oh) * p.Wout + ow;
        }
        return (long long)Q;
    });
    __syncthreads();
    char* const O = smem;
    epi_acc_to_lds<T>(O, OS, acc, bv, p.act, wm * 64, wn * 64, li, lg);
    __syncthreads();
    est.finish(p, O, OS, tid);
}

// ---- slow, obviously-correct direct convolution with the same argument block (debug / cross-check only)
template <typename T, int MODE>
__global__ void conv_naive_kernel(const C2wConvArgs p) {
    constexpr int NT = (MODE == C2W_CONV_1X1) ? 1 : 9;
    const int HWo = p.Hout * p.Wout;
    const long long total = (long long)p.B * HWo * p.Cout;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % p.Cout);
        const long long Q = idx / p.Cout;
        const int b = (int)(Q / HWo);
        const int rem = (int)(Q - (long long)b * HWo);
        const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
        float acc = 0.f;
        for (int tap = 0; tap < NT; ++tap) {
            int ih, iw;
            if (!src_pixel<MODE>(p, oh, ow, tap / 3, tap % 3, ih, iw)) continue;
            const T* xp = (const T*)p.x + (((size_t)b * p.Hin + ih) * p.Win + iw) * p.Cin;
            const T* wp = (const T*)p.w + ((size_t)co * NT + tap) * p.Cin;
            if (co >= p.wrows) continue;
            for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(Elem<T>::ld(wp + ci), Elem<T>::ld(xp + ci), acc);
        }
        if (p.bias && co < p.wrows) acc += p.bias[co];
        if (p.act == C2W_ACT_SILU) acc = silu_f(acc);
        if (p.act == C2W_ACT_RELU) acc = fmaxf(acc, 0.f);
        const size_t off = (size_t)Q * p.ldy + co;
        if (p.mul) {
            float g = Elem<T>::ld((const T*)p.mul + off);
            acc *= (p.mulmode == C2W_MUL_DSILU) ? dsilu_f(g) : g;
        }
        if (p.res) acc += Elem<T>::ld((const T*)p.res + off);
        if (p.act == C2W_ACT_SILU_PAIR && p.y2) {
            Elem<T>::st((T*)p.y + off, acc);
            const float a_ = Elem<T>::ld((const T*)p.y + off);  // as stored
            Elem<T>::st((T*)p.y + off, silu_f(a_));
            Elem<T>::st((T*)p.y2 + off, dsilu_f(a_));
            continue;
        }
        if (p.act == C2W_ACT_RELU_PAIR && p.y2) {
            Elem<T>::st((T*)p.y + off, acc);
            const float a_ = Elem<T>::ld((const T*)p.y + off);  // as stored
            Elem<T>::st((T*)p.y + off, fmaxf(a_, 0.f));
            Elem<T>::st((T*)p.y2 + off, a_ > 0.f ? 1.f : 0.f);
            continue;
        }
        Elem<T>::st((T*)p.y + off, acc);
        if (p.y2) Elem<T>::st((T*)p.y2 + off, silu_f(Elem<T>::ld((const T*)p.y + off)));
    }
}

template <typename T, int MODE, int NW>
int launch_tile(const C2wConvArgs& a, long long npix, int nN, hipStream_t st) {
    typedef GCfg<NW> CF;
    constexpr int ESZ = sizeof(T);
    constexpr int lds_loop = NSLOT * (CF::PBYTES + WBYTES), lds_epi = CF::BM * (BN * ESZ + 16);
    constexpr int lds = lds_loop > lds_epi ? lds_loop : lds_epi;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_igemm_kernel<T, MODE, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    const int nM = (int)((npix + CF::BM - 1) / CF::BM);
    conv_igemm_kernel<T, MODE, NW><<<nM * nN, CF::NTHREADS, lds, st>>>(a);
    return (int)hipGetLastError();
}

template <typename T, int MODE>
int launch_mode(const C2wConvArgs& a, int naive, hipStream_t st) {
    const long long npix = (long long)a.B * a.Hout * a.Wout;
    if (naive) {
        long long total = npix * a.Cout;
        int grid = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
        conv_naive_kernel<T, MODE><<<grid, 256, 0, st>>>(a);
        return (int)hipGetLastError();
    }
    const int nN = (a.Cout + BN - 1) / BN;
    // 256-pixel tiles unless they would leave CUs idle (fewer workgroups than CUs): then 128-pixel tiles, twice the workgroups
    const bool small = ((npix + 255) / 256) * nN < 256;
    return small ? launch_tile<T, MODE, 2>(a, npix, nN, st) : launch_tile<T, MODE, 4>(a, npix, nN, st);
}

template <typename T>
int launch_dtype(const C2wConvArgs& a, int naive, hipStream_t st) {
    switch (a.mode) {
        case C2W_CONV_1X1: return launch_mode<T, C2W_CONV_1X1>(a, naive, st);
        case C2W_CONV_S1: return launch_mode<T, C2W_CONV_S1>(a, naive, st);
        case C2W_CONV_S2: return launch_mode<T, C2W_CONV_S2>(a, naive, st);
        case C2W_CONV_UP: return launch_mode<T, C2W_CONV_UP>(a, naive, st);
        case C2W_CONV_TS2: return launch_mode<T, C2W_CONV_TS2>(a, naive, st);
    }
    return C2W_ERR_BAD_ARG;
}

}  // namespace

// ---- run-time knobs (knobs.h): read once, re-read on request
namespace {
C2wKnobs read_knobs() {
    auto off0 = [](const char* n) { const char* v = getenv(n); return v != nullptr && atoi(v) == 0; };  // "NAME=0" switches a default-on path off
    C2wKnobs k;
    k.force_gather = getenv("C2W_FORCE_GATHER") != nullptr;
    k.conv_t3 = getenv("C2W_CONV_T3") ? atoi(getenv("C2W_CONV_T3")) : -1;
    k.conv_pair = !off0("C2W_CONV_PAIR");
    k.conv_ts2_patch = !off0("C2W_CONV_TS2_PATCH");
    k.conv_s2_patch = getenv("C2W_CONV_S2_PATCH") ? atoi(getenv("C2W_CONV_S2_PATCH")) : 1;
    k.ts2_one_launch = getenv("C2W_TS2_FOUR_LAUNCHES") == nullptr;
    k.ts2_pairs = !off0("C2W_TS2_PAIRS");
    k.up_patch = getenv("C2W_NO_UP_PATCH") == nullptr;
    k.wgrad_narrow = getenv("C2W_NO_NARROW") == nullptr;
    k.wpacked = getenv("C2W_NO_WPACKED") == nullptr;
    k.wgrad_wgs = getenv("C2W_WGRAD_WGS") && atoi(getenv("C2W_WGRAD_WGS")) > 0 ? atoi(getenv("C2W_WGRAD_WGS")) : 256;
    k.pool2 = getenv("C2W_NO_POOL2") == nullptr;
    k.ln_fusion = getenv("C2W_NO_LN_FUSION") == nullptr;
    k.lnf = getenv("C2W_NO_LNF") == nullptr;
    k.wgrad_atomics = getenv("C2W_WGRAD_ATOMICS") != nullptr;
    k.loss_fusion = getenv("C2W_NO_LOSS_FUSION") == nullptr;
    k.ln_chain = getenv("C2W_NO_LN_CHAIN") == nullptr;
    k.splitk = getenv("C2W_NO_SPLITK") == nullptr;
    k.half8 = getenv("C2W_NO_HALF8") == nullptr;
    k.half8_db = !(getenv("C2W_HALF8_DB") && atoi(getenv("C2W_HALF8_DB")) == 0);
    k.half8_max_wgs = getenv("C2W_HALF8_MAX_WGS") && atoi(getenv("C2W_HALF8_MAX_WGS")) > 0 ? atoi(getenv("C2W_HALF8_MAX_WGS")) : 0;
    k.conv_t3_min_wgs = getenv("C2W_CONV_T3_MIN_WGS") && atoi(getenv("C2W_CONV_T3_MIN_WGS")) > 0 ? atoi(getenv("C2W_CONV_T3_MIN_WGS")) : 512;
    k.attn_valu = getenv("C2W_ATTN_VALU") != nullptr;
    return k;
}
C2wKnobs g_knobs = read_knobs();
}  // namespace
const C2wKnobs& c2w_knobs() { return g_knobs; }
extern "C" void c2w_knobs_reload(void) { g_knobs = read_knobs(); }

extern "C" int c2w_conv_patch_supported(const C2wConvArgs* a, int dtype) {
    (void)dtype;
    return a != nullptr && c2w_conv_patch_eligible(*a) && !c2w_knobs().force_gather ? 1 : 0;
}

extern "C" int c2w_conv_pool2_supported(const C2wConvArgs* a, int dtype) {
    (void)dtype;
    if (a == nullptr || a->mode != C2W_CONV_S1 || a->res != nullptr || a->mul != nullptr || a->y2 != nullptr || a->act != C2W_ACT_NONE ||
        a->ln_x != nullptr || a->lnf_y != nullptr)
        return 0;
    return c2w_conv_patch_eligible(*a) && !c2w_conv_pair_eligible(*a) && !c2w_knobs().force_gather && c2w_knobs().pool2 ? 1 : 0;
}

extern "C" int c2w_conv_lnfwd_supported(const C2wConvArgs* a, int dtype) {
    if (a == nullptr || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16)) return 0;
    if (a->Cout != 128 || a->ldy != 128 || a->mul != nullptr || a->y2 != nullptr || a->act != C2W_ACT_NONE || a->ln_x != nullptr) return 0;
    return c2w_conv_patch_eligible(*a) && !c2w_knobs().force_gather && c2w_knobs().ln_fusion && c2w_knobs().lnf ? 1 : 0;
}

extern "C" int c2w_conv_lnbwd_supported(const C2wConvArgs* a, int dtype) {
    if (a == nullptr || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16)) return 0;
    if (a->Cout != 128 || a->ldy != 128 || a->mul != nullptr || a->y2 != nullptr || a->act != C2W_ACT_NONE) return 0;
    return c2w_conv_patch_eligible(*a) && !c2w_knobs().force_gather && c2w_knobs().ln_fusion ? 1 : 0;
}

int c2w_conv_splitk_plan_impl(const C2wConvArgs& a, int dtype, unsigned long long* ws_bytes);  // conv_patch.hip
extern "C" int c2w_conv_splitk_plan(const C2wConvArgs* a, int dtype, unsigned long long* ws_bytes) {
    if (a == nullptr) return C2W_ERR_BAD_ARG;
    return c2w_conv_splitk_plan_impl(*a, dtype, ws_bytes);
}

extern "C" int c2w_conv_lnfwd_chain_supported(const C2wConvArgs* a, int dtype) {
    if (!c2w_conv_lnfwd_supported(a, dtype)) return 0;
    return c2w_knobs().ln_chain && a->mode == C2W_CONV_S1 && c2w_conv_patch3_wanted(*a, dtype) ? 1 : 0;
}

extern "C" int c2w_conv_loss_supported(const C2wConvArgs* a, int dtype) {
    if (a == nullptr || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16)) return 0;
    if (a->mode != C2W_CONV_S1 || a->wrows > 80 || a->Cout > 128 || a->ldy != 128 || a->res != nullptr || a->mul != nullptr || a->y2 != nullptr ||
        a->act != C2W_ACT_NONE || a->ln_x != nullptr || a->lnf_y != nullptr || (a->flags & C2W_CONV_POOL2) != 0)
        return 0;
    if (a->loss_sum != nullptr && (a->loss_eps == nullptr || a->loss_C <= 0 || a->loss_C > a->wrows || (a->loss_lde & 7) != 0 || a->loss_lde < a->loss_C ||
                                   a->loss_lde > 128))
        return 0;
    const C2wKnobs& k = c2w_knobs();
    return !k.force_gather && k.loss_fusion && k.wgrad_narrow && c2w_conv_patch_eligible(*a) && c2w_conv_patch3_wanted(*a, dtype) ? 1 : 0;
}

extern "C" int c2w_conv_wpacked_supported(const C2wConvArgs* a, int dtype) {
    if (a == nullptr || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16)) return 0;
    return !c2w_knobs().force_gather && c2w_knobs().wpacked && c2w_conv_patch_eligible(*a) && c2w_conv_patch3_wanted(*a, dtype) ? 1 : 0;
}

extern "C" int c2w_conv_dispatch(const C2wConvArgs* a, int dtype) {
    if (a == nullptr) return C2W_ERR_BAD_ARG;
    const bool gather = c2w_knobs().force_gather;
    if (!gather && c2w_conv_patch_eligible(*a)) return c2w_conv_patch3_wanted(*a, dtype) ? C2W_KERNEL_PATCH_16X16 : C2W_KERNEL_PATCH_8X16;
    if (!gather && c2w_conv_pair_eligible(*a)) return C2W_KERNEL_PATCH_PAIR;
    if (!gather && c2w_conv_ts2_patch_eligible(*a)) return C2W_KERNEL_PATCH_TS2;
    if (!gather && c2w_conv_s2_patch_eligible(*a, dtype)) return C2W_KERNEL_PATCH_S2;
    return C2W_KERNEL_GATHER;
}

extern "C" int c2w_conv_forward(const C2wConvArgs* a, int dtype, int naive, void* stream) {
    if (a == nullptr || a->x == nullptr || a->w == nullptr || a->y == nullptr) return C2W_ERR_BAD_ARG;
    const int esz = dtype == C2W_DTYPE_F32 ? 4 : 2;
    const int ck = 128 / esz;
    if (a->Cin <= 0 || a->Cin % ck != 0) return C2W_ERR_BAD_SHAPE;            // K-chunk granularity
    if (a->Cout <= 0 || a->Cout % (16 / esz) != 0 || a->ldy % (16 / esz) != 0) return C2W_ERR_BAD_SHAPE;
    if (a->B <= 0 || a->Hin <= 0 || a->Win <= 0 || a->Hout <= 0 || a->Wout <= 0) return C2W_ERR_BAD_SHAPE;
    if (a->Hout >= 65536 || a->Wout >= 65536) return C2W_ERR_BAD_SHAPE;
    if (a->act < C2W_ACT_NONE || a->act > C2W_ACT_RELU_PAIR) return C2W_ERR_BAD_ARG;
    if ((a->act == C2W_ACT_SILU_PAIR || a->act == C2W_ACT_RELU_PAIR) && a->y2 == nullptr) return C2W_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (a->ln_x != nullptr && (naive != 0 || !c2w_conv_lnbwd_supported(a, dtype))) return C2W_ERR_BAD_SHAPE;  // no silent unfused result
    if (a->lnf_y != nullptr && (naive != 0 || !c2w_conv_lnfwd_supported(a, dtype))) return C2W_ERR_BAD_SHAPE;
    if ((a->ln_rstd != nullptr && a->ln_x == nullptr) || (a->lnf_rstd != nullptr && a->lnf_y == nullptr)) return C2W_ERR_BAD_ARG;  // statistics of a LayerNorm that is not fused
    if (a->lnf_mean != nullptr || a->res_rstd != nullptr || a->res_mean != nullptr || a->res_m != nullptr || (a->flags & C2W_CONV_NO_Y) != 0) {
        if (a->lnf_y == nullptr || (a->res_rstd == nullptr) != (a->res_mean == nullptr) || (a->res_m != nullptr && a->res_rstd == nullptr) ||
            (a->res_rstd != nullptr && a->res == nullptr))
            return C2W_ERR_BAD_ARG;
        if (naive != 0 || !c2w_conv_lnfwd_chain_supported(a, dtype)) return C2W_ERR_BAD_SHAPE;
    }
    if ((a->flags & C2W_CONV_POOL2) != 0 && (naive != 0 || !c2w_conv_pool2_supported(a, dtype))) return C2W_ERR_BAD_SHAPE;
    if (a->loss_sum != nullptr && (naive != 0 || !c2w_conv_loss_supported(a, dtype))) return C2W_ERR_BAD_SHAPE;  // no silent unfused result
    if (a->splitk > 1) {  // exactly the plan's answer, with its scratch, or nothing
        unsigned long long need = 0;
        if (naive != 0 || a->splitk_ws == nullptr || c2w_conv_splitk_plan_impl(*a, dtype, &need) != a->splitk || a->splitk_ws_bytes < need) return C2W_ERR_BAD_SHAPE;
    }
    if ((a->flags & C2W_CONV_WPACKED) != 0 && (naive != 0 || !c2w_conv_wpacked_supported(a, dtype))) return C2W_ERR_BAD_SHAPE;  // no other kernel reads that layout
    const bool patch = naive == 0 && !c2w_knobs().force_gather;
    if (patch && c2w_conv_patch_eligible(*a)) return c2w_conv_patch_s1(*a, dtype, st);
    if (patch && c2w_conv_pair_eligible(*a)) return c2w_conv_patch_pair(*a, dtype, st);
    if (patch && c2w_conv_ts2_patch_eligible(*a)) return c2w_conv_patch_ts2(*a, dtype, st);
    if (patch && c2w_conv_s2_patch_eligible(*a, dtype)) return c2w_conv_patch_s2(*a, dtype, st);
    if (naive == 2) naive = 0;  // force the general gather kernel
    if (dtype == C2W_DTYPE_F32) return launch_dtype<float>(*a, naive, st);
    if (dtype == C2W_DTYPE_BF16) return launch_dtype<bf16_t>(*a, naive, st);
    if (dtype == C2W_DTYPE_F16) return launch_dtype<f16_t>(*a, naive, st);
    return C2W_ERR_BAD_ARG;
}
