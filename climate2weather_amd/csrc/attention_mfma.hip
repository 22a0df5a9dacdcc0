// Matrix-core version of the bottleneck attention (model/nn.py:62-85) for the shape the default network runs:
// bf16, T = 64 tokens (8x8 pixels), one head of width C (a multiple of 32, <= 512).  One workgroup of EIGHT waves per image (B = 128
// images fill half the chip's CUs, so the waves of an image are what there is to spread the work over): wave w works on the 16-row
// strip w & 3 of every 64-row matrix; the two waves of a strip split the K range of S (forward), take S and dP (backward), and each
// take half of the column tiles of every output strip.  All five products are v_mfma_f32_16x16x32_bf16:
//
//   forward :  S = q k^T (K-dim C: both operands are K-contiguous in HBM -> fragments by 16-B global loads, no LDS),
//              P = softmax(S * s^2) in fp32 (s = C^-1/4 on q and on k, model/nn.py:76-83), rounded to bf16 like the
//              reference's `weight.type(x.dtype)`, O = P v (K-dim = keys: v staged in LDS and read through the
//              transposing LDS read, P through an LDS round trip from accumulator to operand layout).
//   backward:  recompute S, P = exp(S s^2 - lse);  dP = dO v^T;  delta = rowsum(P * dP);  dS = P * (dP - delta) * s^2
//              dq = dS k,  dv = P^T dO,  dk = dS^T q  -- the three K-dim-64 products read k / dO / q from LDS with
//              transposing reads; P and dS go through LDS as bf16.  No atomics, bit-reproducible.
//
// The fp32 / general-T kernels in attention.hip stay the reference implementation and the fallback.
#include <cstdlib>
#include "common.h"
#include "knobs.h"
#include "c2w_hip.h"

namespace {

constexpr int T64 = 64;
constexpr int PP = 144;  // row pitch (bytes) of the 64 x 64 bf16 P / dS tiles in LDS (128 + 16: rows land on different banks)

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// MFMA operand for a matrix stored [k][m] (k = reduction index = LDS row, m contiguous): lane (li, lg) receives
// M[k0 + 8 lg + 0..7][col0 + li] -- two transposing 8-byte reads (rows +0..3 and +4..7 of the lane's 8-row group).
__device__ __forceinline__ bf16x8_t tr_frag(const char* base, int pitch, int k0, int col0, int li, int lg) {
    const int q = li >> 2, pp = li & 3;
    const char* p0 = base + (k0 + 8 * lg + q) * pitch + (col0 + pp * 4) * 2;
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p0 + 4 * pitch));
    return (bf16x8_t){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// reductions over the 16 lanes (li) that hold one accumulator row = one DPP row
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_get<0x128>(v);
    v += dpp_get<0x124>(v);
    v += dpp_get<0x122>(v);
    return v + dpp_get<0x121>(v);
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_get<0x128>(v));
    v = fmaxf(v, dpp_get<0x124>(v));
    v = fmaxf(v, dpp_get<0x122>(v));
    return fmaxf(v, dpp_get<0x121>(v));
}

// rows [0,64) x C bf16 of a [.][ld] matrix -> LDS [64][pitch]
__device__ __forceinline__ void stage_rows(char* dst, int pitch, const bf16_t* src, int ld, int C) {
    const int nch = C >> 3;
    for (int i = threadIdx.x; i < T64 * nch; i += blockDim.x) {
        const int r = i / nch, c = i - r * nch;
        *(u32x4_t*)(dst + r * pitch + c * 16) = *(const u32x4_t*)(src + (size_t)r * ld + c * 8);
    }
}

// acc[n] (+)= A-strip(16 rows of `a`, starting at row w*16) . B^T over the K steps [ks0, ks1) of 32, both K-contiguous in HBM (row pitches
// lda / ldb): the 16 x 64 strip of a b^T
template <typename T>
__device__ __forceinline__ void strip_abt(f32x4_t (&acc)[4], const bf16_t* a, int lda, const bf16_t* b, int ldb, int ks0, int ks1, int w, int li,
                                          int lg) {
    const bf16_t* ar = a + (size_t)(16 * w + li) * lda + 8 * lg;
    const bf16_t* br = b + (size_t)li * ldb + 8 * lg;
#pragma unroll 4
    for (int ks = ks0; ks < ks1; ++ks) {
        const bf16x8_t av = *(const bf16x8_t*)(ar + 32 * ks);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const bf16x8_t bv = *(const bf16x8_t*)(br + (size_t)(16 * n) * ldb + 32 * ks);
            acc[n] = mfma16s<T>(av, bv, acc[n]);
        }
    }
}

// a wave's 16 x 64 fp32 accumulator strip <-> its 4 KiB of the exchange area X (element [4 lg + r][16 n + li])
__device__ __forceinline__ void strip_to_lds(char* X, int w, const f32x4_t (&acc)[4], int li, int lg) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) *(float*)(X + w * 4096 + ((4 * lg + r) * 64 + 16 * n + li) * 4) = acc[n][r];
}
__device__ __forceinline__ void strip_from_lds(const char* X, int w, f32x4_t (&acc)[4], int li, int lg) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = *(const float*)(X + w * 4096 + ((4 * lg + r) * 64 + 16 * n + li) * 4);
}

// out strip (16 rows starting at 16 w) = A . M, K-dim 64:  A fragments a[ks] given, M[64][C] in LDS (pitch), column tiles [ct0, ct1)
template <typename T>
__device__ __forceinline__ void strip_times_lds(bf16_t* out, int ldo, const bf16x8_t (&a)[2], const char* M, int pitch, int ct0, int ct1, int w,
                                                int li, int lg) {
    // (computed transposed -- M's fragment as the row operand, one 8-byte store per lane instead of four 2-byte ones -- it measured
    // SLOWER: backward 38.3 -> 42.2 us, forward 18.3 -> 19.3; and such stores must convert with pack_acc2, see common.h)
    for (int ct = ct0; ct < ct1; ++ct) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) acc = mfma16s<T>(a[ks], tr_frag(M, pitch, 32 * ks, 16 * ct, li, lg), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * w + 4 * lg + r) * ldo + 16 * ct + li] = f32_to_bits16<T>(acc[r]);
    }
}

constexpr int NTA = 512;  // threads of the T = 64 kernels

template <typename T>  // T = bf16_t or f16_t: the operand format tag (pointers carry raw 16-bit patterns)
__global__ __launch_bounds__(NTA) void attn_mfma_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                            int C, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VP = C * 2 + 16;
    char* const Vl = smem;
    char* const Pl = smem + T64 * VP;
    char* const X = Pl + T64 * PP;  // 4 x 4 KiB: the upper-K partial sums of S on their way to the strip's other wave
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, w = w8 & 3, half = w8 >> 2, li = lane & 15, lg = lane >> 4;
    const int ld = 3 * C;
    const bf16_t* base = qkv + (size_t)blockIdx.x * T64 * ld;
    stage_rows(Vl, VP, base + 2 * C, ld, C);

    f32x4_t s[4] = {};
    const int nks = C / 32, ksm = nks / 2;
    strip_abt<T>(s, base, ld, base + C, ld, half ? ksm : 0, half ? nks : ksm, w, li, lg);
    if (half) strip_to_lds(X, w, s, li, lg);
    __syncthreads();
    if (!half) {
        f32x4_t s1[4];
        strip_from_lds(X, w, s1, li, lg);
        // accumulator element s[n][r] = S[16 w + 4 lg + r][16 n + li]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[n][r] = (s[n][r] + s1[n][r]) * scale2;
                m = fmaxf(m, s[n][r]);
            }
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[n][r] = __expf(s[n][r] - m);
                sum += s[n][r];
            }
            sum = row16_sum(sum);
            const float inv = 1.0f / sum;
            const int row = 16 * w + 4 * lg + r;
#pragma unroll
            for (int n = 0; n < 4; ++n) *(bf16_t*)(Pl + row * PP + (16 * n + li) * 2) = f32_to_bits16<T>(s[n][r] * inv);
            if (lse != nullptr && li == 0) lse[(size_t)blockIdx.x * T64 + row] = m + __logf(sum);
        }
    }
    __syncthreads();
    bf16x8_t pa[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) pa[ks] = *(const bf16x8_t*)(Pl + (16 * w + li) * PP + (32 * ks + 8 * lg) * 2);
    const int nct = C / 16, ctm = nct / 2;
    strip_times_lds<T>(o + (size_t)blockIdx.x * T64 * C, C, pa, Vl, VP, half ? ctm : 0, half ? nct : ctm, w, li, lg);
}

template <typename T>  // T = bf16_t or f16_t: the operand format tag (pointers carry raw 16-bit patterns)
__global__ __launch_bounds__(NTA) void attn_mfma_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                            const float* __restrict__ lse, bf16_t* __restrict__ dqkv, int C, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VP = C * 2 + 16;
    char* const B0 = smem;                 // k, later q
    char* const B1 = smem + T64 * VP;      // dO
    char* const Pl = smem + 2 * T64 * VP;  // P  [q][key]
    char* const Sl = Pl + T64 * PP;        // dS [q][key] (already times s^2)
    char* const X = Pl;                    // 4 x 4 KiB over P and dS (2 x 9 KiB), before they exist: dP on its way to the strip's other wave
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, w = w8 & 3, half = w8 >> 2, li = lane & 15, lg = lane >> 4;
    const int ld = 3 * C;
    const bf16_t* base = qkv + (size_t)blockIdx.x * T64 * ld;
    const bf16_t* dob = d_o + (size_t)blockIdx.x * T64 * C;
    bf16_t* dbase = dqkv + (size_t)blockIdx.x * T64 * ld;
    stage_rows(B0, VP, base + C, ld, C);
    stage_rows(B1, VP, dob, C, C);

    // waves 0-3: S = q k^T; waves 4-7: dP = dO v^T (dO rows have pitch C, v rows pitch 3C) -- the same strip, side by side
    f32x4_t s[4] = {}, dp[4];
    if (!half) strip_abt<T>(s, base, ld, base + C, ld, 0, C / 32, w, li, lg);
    else strip_abt<T>(s, dob, C, base + 2 * C, ld, 0, C / 32, w, li, lg);
    if (half) strip_to_lds(X, w, s, li, lg);
    __syncthreads();
    if (!half) strip_from_lds(X, w, dp, li, lg);
    __syncthreads();  // X is read: P and dS may take its place
    if (!half) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + 4 * lg + r;
            const float l = lse[(size_t)blockIdx.x * T64 + row];
            float delta = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[n][r] = __expf(s[n][r] * scale2 - l);  // P
                delta += s[n][r] * dp[n][r];
            }
            delta = row16_sum(delta);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float ds = s[n][r] * (dp[n][r] - delta) * scale2;
                *(bf16_t*)(Pl + row * PP + (16 * n + li) * 2) = f32_to_bits16<T>(s[n][r]);
                *(bf16_t*)(Sl + row * PP + (16 * n + li) * 2) = f32_to_bits16<T>(ds);
            }
        }
    }
    __syncthreads();
    const int nct = C / 16, ctm = nct / 2, ct0 = half ? ctm : 0, ct1 = half ? nct : ctm;
    bf16x8_t a[2];
    // dq strip (query rows): dS . k
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[ks] = *(const bf16x8_t*)(Sl + (16 * w + li) * PP + (32 * ks + 8 * lg) * 2);
    strip_times_lds<T>(dbase, ld, a, B0, VP, ct0, ct1, w, li, lg);
    // dv strip (key rows): P^T . dO  -- A[m = key][k = query] = P[query][key]: transposing read of the P tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[ks] = tr_frag(Pl, PP, 32 * ks, 16 * w, li, lg);
    strip_times_lds<T>(dbase + 2 * C, ld, a, B1, VP, ct0, ct1, w, li, lg);
    __syncthreads();  // every wave is done with k before q replaces it
    stage_rows(B0, VP, base, ld, C);
    __syncthreads();
    // dk strip (key rows): dS^T . q
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[ks] = tr_frag(Sl, PP, 32 * ks, 16 * w, li, lg);
    strip_times_lds<T>(dbase + C, ld, a, B0, VP, ct0, ct1, w, li, lg);
}

// ---- forward for T = 64 * nb tokens (the 256^2 variant attends over 16 x 16 = 256): one workgroup per (image, block of 64
// queries), key blocks of 64 walked with the online softmax (running row maximum m and sum l, output accumulators rescaled when
// m grows), v_j staged in LDS per block, P_j through a wave-private LDS strip.  Same operand layouts as the T = 64 kernel.
template <typename T>  // T = bf16_t or f16_t: the operand format tag (pointers carry raw 16-bit patterns)
__global__ __launch_bounds__(256, 2) void attn_mfma_fwd_blocks_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o,
                                                                       float* __restrict__ lse, int Tn, int C, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VP = C * 2 + 16;
    char* const Vl = smem;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, li = lane & 15, lg = lane >> 4;
    char* const Pw = smem + T64 * VP + w * (16 * PP);  // this wave's 16 x 64 strip of P_j
    const int ld = 3 * C, nb = Tn / T64, qb = blockIdx.x;
    const bf16_t* base = qkv + (size_t)blockIdx.y * Tn * ld;
    const bf16_t* qrow = base + (size_t)(qb * T64 + 16 * w + li) * ld + 8 * lg;

    constexpr int MAXCT = 32;  // C <= 512
    f32x4_t oacc[MAXCT];
#pragma unroll
    for (int ct = 0; ct < MAXCT; ++ct) oacc[ct] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float mrun[4], lrun[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }

    for (int j = 0; j < nb; ++j) {
        __syncthreads();  // every wave is done with the previous v block
        stage_rows(Vl, VP, base + (size_t)j * T64 * ld + 2 * C, ld, C);
        f32x4_t s[4] = {};
        {
            const bf16_t* krow = base + (size_t)(j * T64 + li) * ld + C + 8 * lg;
#pragma unroll 4
            for (int ks = 0; ks < C / 32; ++ks) {
                const bf16x8_t av = *(const bf16x8_t*)(qrow + 32 * ks);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const bf16x8_t bv = *(const bf16x8_t*)(krow + (size_t)(16 * n) * ld + 32 * ks);
                    s[n] = mfma16s<T>(av, bv, s[n]);
                }
            }
        }
        float alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[n][r] *= scale2;
                mx = fmaxf(mx, s[n][r]);
            }
            mx = fmaxf(row16_max(mx), mrun[r]);
            alpha[r] = __expf(mrun[r] - mx);  // exp(-inf) = 0 on the first block
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                s[n][r] = __expf(s[n][r] - mx);
                sum += s[n][r];
            }
            lrun[r] = lrun[r] * alpha[r] + row16_sum(sum);
            mrun[r] = mx;
#pragma unroll
            for (int n = 0; n < 4; ++n) *(bf16_t*)(Pw + (4 * lg + r) * PP + (16 * n + li) * 2) = f32_to_bits16<T>(s[n][r]);
        }
        __syncthreads();  // v_j staged (and this wave's P strip written)
        bf16x8_t pa[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) pa[ks] = *(const bf16x8_t*)(Pw + li * PP + (32 * ks + 8 * lg) * 2);
#pragma unroll
        for (int ct = 0; ct < MAXCT; ++ct) {
            if (ct * 16 < C) {
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[ct][r] *= alpha[r];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    oacc[ct] = mfma16s<T>(pa[ks], tr_frag(Vl, VP, 32 * ks, 16 * ct, li, lg), oacc[ct]);
            }
        }
    }
    bf16_t* ob = o + ((size_t)blockIdx.y * Tn + qb * T64 + 16 * w) * C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float inv = 1.0f / lrun[r];
#pragma unroll
        for (int ct = 0; ct < MAXCT; ++ct)
            if (ct * 16 < C) ob[(size_t)(4 * lg + r) * C + 16 * ct + li] = f32_to_bits16<T>(oacc[ct][r] * inv);
        if (lse != nullptr && li == 0) lse[(size_t)blockIdx.y * Tn + qb * T64 + 16 * w + 4 * lg + r] = mrun[r] + __logf(lrun[r]);
    }
}


// ---- backward for T = 64 * nb tokens (round 4; the 256^2 variant's attention backward ran on the fp32 VALU kernels: 6.8 % of its
// training step).  Flash-style in 64 x 64 blocks, two kernels, no atomics, bit-reproducible:
//   attn_mfma_bwd_kv_blocks_kernel   one workgroup per (image, KEY block j): walks the query blocks i, recomputes S_ij and dP_ij,
//                                    dv_j += P_ij^T dO_i,  dk_j += dS_ij^T q_i   (accumulators in registers: 16 rows x C/2 per wave)
//   attn_mfma_bwd_q_blocks_kernel    one workgroup per (image, QUERY block i): walks the key blocks j, dq_i += dS_ij k_j
// with P_ij = exp(S_ij s^2 - lse_i), dS_ij = P_ij (dP_ij - delta_i) s^2 and delta_i = sum_c dO_i O_i over the WHOLE row (rowdot_kernel,
// attention.hip) -- the T = 64 kernel could take it from its one block.  Same wave roles as there: wave (w, half) works on the 16-row
// strip w; half 0 computes S, half 1 dP, side by side; each takes half of the column tiles of the output strips.
constexpr int MAXHCT = 16;  // column tiles of 16 per wave half: C <= 512

template <typename T>
__device__ __forceinline__ void strip_acc_lds(f32x4_t (&acc)[MAXHCT], const bf16x8_t (&a)[2], const char* M, int pitch, int ct0, int nct, int li,
                                              int lg) {
#pragma unroll
    for (int c = 0; c < MAXHCT; ++c) {
        if (c < nct) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) acc[c] = mfma16s<T>(a[ks], tr_frag(M, pitch, 32 * ks, 16 * (ct0 + c), li, lg), acc[c]);
        }
    }
}
template <typename T>
__device__ __forceinline__ void strip_store(bf16_t* out, int ldo, const f32x4_t (&acc)[MAXHCT], int ct0, int nct, int w, int li, int lg) {
#pragma unroll
    for (int c = 0; c < MAXHCT; ++c) {
        if (c < nct) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(16 * w + 4 * lg + r) * ldo + 16 * (ct0 + c) + li] = f32_to_bits16<T>(acc[c][r]);
        }
    }
}

// P and dS of one 64 x 64 block into LDS (half 0), from S (own accumulators) and dP (from the strip's other wave, through X)
template <typename T, bool WANT_P>
__device__ __forceinline__ void block_p_ds(char* Pl, char* Sl, const f32x4_t (&s)[4], const f32x4_t (&dp)[4], const float* lse_i, const float* del_i,
                                           float scale2, int w, int li, int lg) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * w + 4 * lg + r;
        const float l = lse_i[row], delta = del_i[row];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float p = __expf(s[n][r] * scale2 - l);
            if (WANT_P) *(bf16_t*)(Pl + row * PP + (16 * n + li) * 2) = f32_to_bits16<T>(p);
            *(bf16_t*)(Sl + row * PP + (16 * n + li) * 2) = f32_to_bits16<T>(p * (dp[n][r] - delta) * scale2);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(NTA) void attn_mfma_bwd_kv_blocks_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                      const float* __restrict__ lse, const float* __restrict__ delta,
                                                                      bf16_t* __restrict__ dqkv, int Tn, int C, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VP = C * 2 + 16;
    char* const B0 = smem;                 // q_i
    char* const B1 = smem + T64 * VP;      // dO_i
    char* const Pl = smem + 2 * T64 * VP;  // P_ij  [q][key]
    char* const Sl = Pl + T64 * PP;        // dS_ij [q][key]
    char* const X = Pl;                    // dP on its way to the strip's other wave (before P / dS exist)
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, w = w8 & 3, half = w8 >> 2, li = lane & 15, lg = lane >> 4;
    const int ld = 3 * C, nb = Tn / T64, j = blockIdx.x;
    const bf16_t* base = qkv + (size_t)blockIdx.y * Tn * ld;
    const bf16_t* dob = d_o + (size_t)blockIdx.y * Tn * C;
    const bf16_t* kj = base + (size_t)j * T64 * ld + C;
    const bf16_t* vj = base + (size_t)j * T64 * ld + 2 * C;
    const int nct = C / 16, ctm = nct / 2, ct0 = half ? ctm : 0, nc = half ? nct - ctm : ctm;
    f32x4_t dk[MAXHCT], dv[MAXHCT];
#pragma unroll
    for (int c = 0; c < MAXHCT; ++c) dk[c] = dv[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < nb; ++i) {
        const bf16_t* qi = base + (size_t)i * T64 * ld;
        const bf16_t* doi = dob + (size_t)i * T64 * C;
        __syncthreads();  // every wave is done with the previous block's q, dO, P, dS
        stage_rows(B0, VP, qi, ld, C);
        stage_rows(B1, VP, doi, C, C);
        f32x4_t s[4] = {}, dp[4];
        if (!half) strip_abt<T>(s, qi, ld, kj, ld, 0, C / 32, w, li, lg);
        else strip_abt<T>(s, doi, C, vj, ld, 0, C / 32, w, li, lg);
        if (half) strip_to_lds(X, w, s, li, lg);
        __syncthreads();
        if (!half) strip_from_lds(X, w, dp, li, lg);
        __syncthreads();  // X is read: P and dS may take its place
        if (!half) block_p_ds<T, true>(Pl, Sl, s, dp, lse + (size_t)blockIdx.y * Tn + i * T64, delta + (size_t)blockIdx.y * Tn + i * T64, scale2, w, li, lg);
        __syncthreads();
        bf16x8_t a[2];
        // dv strip (key rows 16 w ..): P^T . dO_i
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[ks] = tr_frag(Pl, PP, 32 * ks, 16 * w, li, lg);
        strip_acc_lds<T>(dv, a, B1, VP, ct0, nc, li, lg);
        // dk strip: dS^T . q_i
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[ks] = tr_frag(Sl, PP, 32 * ks, 16 * w, li, lg);
        strip_acc_lds<T>(dk, a, B0, VP, ct0, nc, li, lg);
    }
    bf16_t* dbase = dqkv + ((size_t)blockIdx.y * Tn + (size_t)j * T64) * ld;
    strip_store<T>(dbase + C, ld, dk, ct0, nc, w, li, lg);
    strip_store<T>(dbase + 2 * C, ld, dv, ct0, nc, w, li, lg);
}

template <typename T>
__global__ __launch_bounds__(NTA) void attn_mfma_bwd_q_blocks_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ d_o,
                                                                     const float* __restrict__ lse, const float* __restrict__ delta,
                                                                     bf16_t* __restrict__ dqkv, int Tn, int C, float scale2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VP = C * 2 + 16;
    char* const B0 = smem;             // k_j
    char* const Pl = smem + T64 * VP;  // (unused P slot: keeps the X overlay's 16 KiB inside the buffer)
    char* const Sl = Pl + T64 * PP;    // dS_ij [q][key]
    char* const X = Pl;
    const int tid = threadIdx.x, lane = tid & 63, w8 = tid >> 6, w = w8 & 3, half = w8 >> 2, li = lane & 15, lg = lane >> 4;
    const int ld = 3 * C, nb = Tn / T64, i = blockIdx.x;
    const bf16_t* base = qkv + (size_t)blockIdx.y * Tn * ld;
    const bf16_t* qi = base + (size_t)i * T64 * ld;
    const bf16_t* doi = d_o + ((size_t)blockIdx.y * Tn + (size_t)i * T64) * C;
    const float* lse_i = lse + (size_t)blockIdx.y * Tn + i * T64;
    const float* del_i = delta + (size_t)blockIdx.y * Tn + i * T64;
    const int nct = C / 16, ctm = nct / 2, ct0 = half ? ctm : 0, nc = half ? nct - ctm : ctm;
    f32x4_t dq[MAXHCT];
#pragma unroll
    for (int c = 0; c < MAXHCT; ++c) dq[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < nb; ++j) {
        const bf16_t* kj = base + (size_t)j * T64 * ld + C;
        const bf16_t* vj = base + (size_t)j * T64 * ld + 2 * C;
        __syncthreads();  // every wave is done with the previous k block and dS
        stage_rows(B0, VP, kj, ld, C);
        f32x4_t s[4] = {}, dp[4];
        if (!half) strip_abt<T>(s, qi, ld, kj, ld, 0, C / 32, w, li, lg);
        else strip_abt<T>(s, doi, C, vj, ld, 0, C / 32, w, li, lg);
        if (half) strip_to_lds(X, w, s, li, lg);
        __syncthreads();
        if (!half) strip_from_lds(X, w, dp, li, lg);
        __syncthreads();
        if (!half) block_p_ds<T, false>(Pl, Sl, s, dp, lse_i, del_i, scale2, w, li, lg);
        __syncthreads();
        bf16x8_t a[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) a[ks] = *(const bf16x8_t*)(Sl + (16 * w + li) * PP + (32 * ks + 8 * lg) * 2);
        strip_acc_lds<T>(dq, a, B0, VP, ct0, nc, li, lg);
    }
    strip_store<T>(dqkv + ((size_t)blockIdx.y * Tn + (size_t)i * T64) * ld, ld, dq, ct0, nc, w, li, lg);
}

}  // namespace

static bool is16(int dtype) { return dtype == C2W_DTYPE_BF16 || dtype == C2W_DTYPE_F16; }

bool c2w_attention_mfma_eligible(int B, int Tn, int C, int dtype) {
    return is16(dtype) && Tn == T64 && C % 32 == 0 && C <= 512 && B > 0 && !c2w_knobs().attn_valu;
}

namespace {
template <typename T>
int fwd_launch(const void* qkv, void* o, float* lse, int B, int C, hipStream_t st) {
    const int lds = T64 * (C * 2 + 16) + T64 * PP + 4 * 4096;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_mfma_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    attn_mfma_fwd_kernel<T><<<B, NTA, lds, st>>>((const bf16_t*)qkv, (bf16_t*)o, lse, C, 1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}
template <typename T>
int bwd_launch(const void* qkv, const void* d_o, const float* lse, void* dqkv, int B, int C, hipStream_t st) {
    const int lds = 2 * T64 * (C * 2 + 16) + 2 * T64 * PP;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_mfma_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    attn_mfma_bwd_kernel<T><<<B, NTA, lds, st>>>((const bf16_t*)qkv, (const bf16_t*)d_o, lse, (bf16_t*)dqkv, C, 1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}
template <typename T>
int blocks_launch(const void* qkv, void* o, float* lse, int B, int Tn, int C, hipStream_t st) {
    const int lds = T64 * (C * 2 + 16) + 4 * 16 * PP;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_mfma_fwd_blocks_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    attn_mfma_fwd_blocks_kernel<T><<<dim3(Tn / T64, B), 256, lds, st>>>((const bf16_t*)qkv, (bf16_t*)o, lse, Tn, C, 1.0f / sqrtf((float)C));
    return (int)hipGetLastError();
}
}  // namespace

int c2w_attention_mfma_forward(const void* qkv, void* o, float* lse, int B, int C, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? fwd_launch<f16_t>(qkv, o, lse, B, C, st) : fwd_launch<bf16_t>(qkv, o, lse, B, C, st);
}

int c2w_attention_mfma_backward(const void* qkv, const void* d_o, const float* lse, void* dqkv, int B, int C, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? bwd_launch<f16_t>(qkv, d_o, lse, dqkv, B, C, st) : bwd_launch<bf16_t>(qkv, d_o, lse, dqkv, B, C, st);
}

// T a multiple of 64 beyond 64: forward with the online softmax over key blocks, backward in 64 x 64 blocks (two kernels)
bool c2w_attention_mfma_blocks_eligible(int B, int Tn, int C, int dtype) {
    return is16(dtype) && Tn > T64 && Tn % T64 == 0 && Tn <= 4096 && C % 32 == 0 && C <= 512 && B > 0 && !c2w_knobs().attn_valu;
}

int c2w_attention_mfma_blocks_forward(const void* qkv, void* o, float* lse, int B, int Tn, int C, int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? blocks_launch<f16_t>(qkv, o, lse, B, Tn, C, st) : blocks_launch<bf16_t>(qkv, o, lse, B, Tn, C, st);
}

namespace {
template <typename T>
int blocks_bwd_launch(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv, int B, int Tn, int C, hipStream_t st) {
    const int VP = C * 2 + 16;
    const int lds_kv = 2 * T64 * VP + 2 * T64 * PP, lds_q = T64 * VP + 2 * T64 * PP;
    static bool attr = false;
    if (!attr) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_mfma_bwd_kv_blocks_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_mfma_bwd_q_blocks_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const float scale2 = 1.0f / sqrtf((float)C);
    const dim3 grid(Tn / T64, B);
    attn_mfma_bwd_kv_blocks_kernel<T><<<grid, NTA, lds_kv, st>>>((const bf16_t*)qkv, (const bf16_t*)d_o, lse, delta, (bf16_t*)dqkv, Tn, C, scale2);
    attn_mfma_bwd_q_blocks_kernel<T><<<grid, NTA, lds_q, st>>>((const bf16_t*)qkv, (const bf16_t*)d_o, lse, delta, (bf16_t*)dqkv, Tn, C, scale2);
    return (int)hipGetLastError();
}
}  // namespace

// backward of the T = 64 nb shapes: delta[row] = sum_c dO O must have been computed (attention.hip: rowdot_kernel) into `delta`
int c2w_attention_mfma_blocks_backward(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv, int B, int Tn, int C,
                                       int dtype, hipStream_t st) {
    return dtype == C2W_DTYPE_F16 ? blocks_bwd_launch<f16_t>(qkv, d_o, lse, delta, dqkv, B, Tn, C, st)
                                  : blocks_bwd_launch<bf16_t>(qkv, d_o, lse, delta, dqkv, B, Tn, C, st);
}
