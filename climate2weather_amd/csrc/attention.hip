// Single-head QKV attention of the bottleneck AttentionBlock (model/nn.py:62-85): tokens = pixels of one
// image (T = H*W = 64 at the default 8x8 level), head dim = C (512).  0.05 of the 116 GFLOP forward, so these
// are compact fp32 VALU kernels: one block per (image, 16-row tile), whole score rows live in LDS.
//   forward : P = softmax_s( (q.s)(k.s) ), s = C^-1/4 (model/nn.py:76-83), O = P v ; saves lse[t] = log sum exp
//   backward: flash-style recompute from (q,k,v,lse) with delta[t] = sum_c dO[t][c] O[t][c]:
//             pass A (query tiles): dq ;  pass B (key tiles): dk, dv   -- no atomics, bit-reproducible
// qkv layout: [B][T][3C] rows (the NHWC output of the 1x1 qkv conv): q | k | v channel blocks.
#include "common.h"
#include "c2w_hip.h"

// matrix-core path for bf16, T = 64 (attention_mfma.hip)
bool c2w_attention_mfma_eligible(int B, int Tn, int C, int dtype);
int c2w_attention_mfma_forward(const void* qkv, void* o, float* lse, int B, int C, int dtype, hipStream_t st);
int c2w_attention_mfma_backward(const void* qkv, const void* d_o, const float* lse, void* dqkv, int B, int C, int dtype, hipStream_t st);
bool c2w_attention_mfma_blocks_eligible(int B, int Tn, int C, int dtype);
int c2w_attention_mfma_blocks_forward(const void* qkv, void* o, float* lse, int B, int Tn, int C, int dtype, hipStream_t st);
int c2w_attention_mfma_blocks_backward(const void* qkv, const void* d_o, const float* lse, const float* delta, void* dqkv, int B, int Tn, int C,
                                       int dtype, hipStream_t st);

namespace {

constexpr int TR = 16;  // rows per tile

// A_lds[r][c] (fp32) <- rows [row0, row0+TR) of M (zeros past nrows)
template <typename T>
__device__ void load_rows(float* A_lds, const T* M, int ld, int row0, int nrows, int C) {
    constexpr int P = Elem<T>::PER16;
    const int nvec = C / P;
    for (int i = threadIdx.x; i < TR * nvec; i += blockDim.x) {
        const int r = i / nvec, v = i - r * nvec;
        float f[P];
        if (row0 + r < nrows) {
            unpack16<T>(*(const u32x4_t*)(M + (size_t)(row0 + r) * ld + v * P), f);
        } else {
#pragma unroll
            for (int e = 0; e < P; ++e) f[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < P; ++e) A_lds[r * C + v * P + e] = f[e];
    }
}

// S_lds[r][j] = scale * sum_c A_lds[r][c] * Bm[j][c],  j < Tn
template <typename T>
__device__ void rows_dot(float* S_lds, const float* A_lds, const T* Bm, int ld, int Tn, int C, float scale) {
    constexpr int P = Elem<T>::PER16;
    for (int j = threadIdx.x; j < Tn; j += blockDim.x) {
        float acc[TR];
#pragma unroll
        for (int r = 0; r < TR; ++r) acc[r] = 0.f;
        const T* brow = Bm + (size_t)j * ld;
        for (int c = 0; c < C; c += P) {
            float f[P];
            unpack16<T>(*(const u32x4_t*)(brow + c), f);
#pragma unroll
            for (int r = 0; r < TR; ++r) {
#pragma unroll
                for (int e = 0; e < P; ++e) acc[r] = fmaf(A_lds[r * C + c + e], f[e], acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < TR; ++r) S_lds[r * Tn + j] = acc[r] * scale;
    }
}

// out[row0 + r][c] = scale * sum_j W_lds[r][j] * M[j][c]   (rows past nrows skipped)
template <typename T>
__device__ void rows_mix(T* out, int ldo, const float* W_lds, const T* M, int ld, int row0, int nrows, int Tn, int C, float scale) {
    constexpr int P = Elem<T>::PER16;
    const int nvec = C / P;
    // thread -> (row group of 4, channel vector)
    for (int i = threadIdx.x; i < (TR / 4) * nvec; i += blockDim.x) {
        const int rg = i / nvec, v = i - rg * nvec;
        float acc[4][P];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < P; ++e) acc[r][e] = 0.f;
        for (int j = 0; j < Tn; ++j) {
            float f[P];
            unpack16<T>(*(const u32x4_t*)(M + (size_t)j * ld + v * P), f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float w = W_lds[(rg * 4 + r) * Tn + j];
#pragma unroll
                for (int e = 0; e < P; ++e) acc[r][e] = fmaf(w, f[e], acc[r][e]);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + rg * 4 + r;
            if (row < nrows) {
#pragma unroll
                for (int e = 0; e < P; ++e) acc[r][e] *= scale;
                *(u32x4_t*)(out + (size_t)row * ldo + v * P) = pack16<T>(acc[r]);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, T* __restrict__ o, float* __restrict__ lse, int Tn, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* A = sm;            // [TR][C]
    float* S = sm + TR * C;   // [TR][Tn]
    const int b = blockIdx.y, row0 = blockIdx.x * TR;
    const T* base = qkv + (size_t)b * Tn * 3 * C;
    const float scale2 = 1.0f / sqrtf((float)C);  // (C^-1/4)^2
    load_rows<T>(A, base, 3 * C, row0, Tn, C);
    __syncthreads();
    rows_dot<T>(S, A, base + C, 3 * C, Tn, C, scale2);
    __syncthreads();
    {  // row softmax: 16 lanes per row
        const int r = threadIdx.x >> 4, j = threadIdx.x & 15;
        float mx = -INFINITY;
        for (int s = j; s < Tn; s += 16) mx = fmaxf(mx, S[r * Tn + s]);
#pragma unroll
        for (int o2 = 8; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
        float sum = 0.f;
        for (int s = j; s < Tn; s += 16) {
            const float e = expf(S[r * Tn + s] - mx);
            S[r * Tn + s] = e;
            sum += e;
        }
#pragma unroll
        for (int o2 = 8; o2 > 0; o2 >>= 1) sum += __shfl_xor(sum, o2, 64);
        const float inv = 1.0f / sum;
        for (int s = j; s < Tn; s += 16) S[r * Tn + s] *= inv;
        if (j == 0 && lse != nullptr && row0 + r < Tn) lse[(size_t)b * Tn + row0 + r] = mx + logf(sum);
    }
    __syncthreads();
    rows_mix<T>(o + (size_t)b * Tn * C, C, S, base + 2 * C, 3 * C, row0, Tn, Tn, C, 1.0f);
}

// MODE 0: query tile -> dq ;  MODE 1: key tile -> dk, dv
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const T* __restrict__ dO, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, T* __restrict__ dqkv, int Tn, int C) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* A = sm;                 // [TR][C]
    float* S = sm + TR * C;        // [TR][Tn]  -> P or P^T, later dS
    float* D = S + TR * Tn;        // [TR][Tn]  -> dP or dP^T
    const int b = blockIdx.y, row0 = blockIdx.x * TR;
    const T* base = qkv + (size_t)b * Tn * 3 * C;
    const T* dOb = dO + (size_t)b * Tn * C;
    T* dbase = dqkv + (size_t)b * Tn * 3 * C;
    const float* lse_b = lse + (size_t)b * Tn;
    const float* del_b = delta + (size_t)b * Tn;
    const float scale2 = 1.0f / sqrtf((float)C);
    if (MODE == 0) {
        load_rows<T>(A, base, 3 * C, row0, Tn, C);  // q tile
        __syncthreads();
        rows_dot<T>(S, A, base + C, 3 * C, Tn, C, scale2);  // S[t][s]
        __syncthreads();
        load_rows<T>(A, dOb, C, row0, Tn, C);  // dO tile
        __syncthreads();
        rows_dot<T>(D, A, base + 2 * C, 3 * C, Tn, C, 1.0f);  // dP[t][s] = dO[t] . v[s]
        __syncthreads();
        for (int i = threadIdx.x; i < TR * Tn; i += blockDim.x) {
            const int r = i / Tn, t = row0 + r;
            if (t < Tn) {
                const float p = expf(S[i] - lse_b[t]);
                S[i] = p * (D[i] - del_b[t]);
            } else {
                S[i] = 0.f;
            }
        }
        __syncthreads();
        rows_mix<T>(dbase, 3 * C, S, base + C, 3 * C, row0, Tn, Tn, C, scale2);  // dq = scale2 * dS k
    } else {
        load_rows<T>(A, base + C, 3 * C, row0, Tn, C);  // k tile
        __syncthreads();
        rows_dot<T>(S, A, base, 3 * C, Tn, C, scale2);  // S^T[s][t]
        __syncthreads();
        load_rows<T>(A, base + 2 * C, 3 * C, row0, Tn, C);  // v tile
        __syncthreads();
        rows_dot<T>(D, A, dOb, C, Tn, C, 1.0f);  // dP^T[s][t] = v[s] . dO[t]
        __syncthreads();
        for (int i = threadIdx.x; i < TR * Tn; i += blockDim.x) {
            const int t = i % Tn;
            const float p = expf(S[i] - lse_b[t]);
            S[i] = p;                        // P^T
            D[i] = p * (D[i] - del_b[t]);    // dS^T
        }
        __syncthreads();
        rows_mix<T>(dbase + 2 * C, 3 * C, S, dOb, C, row0, Tn, Tn, C, 1.0f);     // dv = P^T dO
        rows_mix<T>(dbase + C, 3 * C, D, base, 3 * C, row0, Tn, Tn, C, scale2);  // dk = scale2 * dS^T q
    }
}

// delta[row] = sum_c a[row][c] * b[row][c]
template <typename T>
__global__ __launch_bounds__(256) void rowdot_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ out, long long rows, int C) {
    constexpr int P = Elem<T>::PER16;
    const int sub = threadIdx.x >> 4, j = threadIdx.x & 15;
    for (long long r = (long long)blockIdx.x * 16 + sub; r < rows; r += (long long)gridDim.x * 16) {
        float s = 0.f;
        for (int c = j * P; c < C; c += 16 * P) {
            float f[P], g[P];
            unpack16<T>(*(const u32x4_t*)(a + r * C + c), f);
            unpack16<T>(*(const u32x4_t*)(b + r * C + c), g);
#pragma unroll
            for (int e = 0; e < P; ++e) s = fmaf(f[e], g[e], s);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (j == 0) out[r] = s;
    }
}

template <typename K>
int set_lds(K kernel, int bytes) {
    return (int)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename T>
int valu_forward(const void* qkv, void* o, float* lse, int Tn, int C, dim3 grid, int lds, hipStream_t st) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_fwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attn_fwd_kernel<T><<<grid, 256, lds, st>>>((const T*)qkv, (T*)o, lse, Tn, C);
    return (int)hipGetLastError();
}

template <typename T>
int valu_backward(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dqkv, int Tn, int C, long long rows,
                  dim3 grid, int rgrid, int lds, hipStream_t st) {
    rowdot_kernel<T><<<rgrid, 256, 0, st>>>((const T*)d_o, (const T*)o, delta_ws, rows, C);
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_kernel<T, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_bwd_kernel<T, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attn_bwd_kernel<T, 0><<<grid, 256, lds, st>>>((const T*)qkv, (const T*)d_o, lse, delta_ws, (T*)dqkv, Tn, C);
    attn_bwd_kernel<T, 1><<<grid, 256, lds, st>>>((const T*)qkv, (const T*)d_o, lse, delta_ws, (T*)dqkv, Tn, C);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int c2w_attention_forward(const void* qkv, void* o, float* lse, int B, int Tn, int C, int dtype, void* stream) {
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    if (!qkv || !o || B <= 0 || Tn <= 0 || C <= 0 || C % P) return C2W_ERR_BAD_SHAPE;
    if (c2w_attention_mfma_eligible(B, Tn, C, dtype)) return c2w_attention_mfma_forward(qkv, o, lse, B, C, dtype, (hipStream_t)stream);
    if (c2w_attention_mfma_blocks_eligible(B, Tn, C, dtype))
        return c2w_attention_mfma_blocks_forward(qkv, o, lse, B, Tn, C, dtype, (hipStream_t)stream);
    const int lds = (TR * C + TR * Tn) * 4;
    if (lds > 160 * 1024) return C2W_ERR_UNSUPPORTED;
    dim3 grid((Tn + TR - 1) / TR, B);
    if (dtype == C2W_DTYPE_F32) return valu_forward<float>(qkv, o, lse, Tn, C, grid, lds, (hipStream_t)stream);
    if (dtype == C2W_DTYPE_BF16) return valu_forward<bf16_t>(qkv, o, lse, Tn, C, grid, lds, (hipStream_t)stream);
    if (dtype == C2W_DTYPE_F16) return valu_forward<f16_t>(qkv, o, lse, Tn, C, grid, lds, (hipStream_t)stream);
    return C2W_ERR_BAD_ARG;
}

extern "C" int c2w_attention_backward(const void* qkv, const void* o, const void* d_o, const float* lse, float* delta_ws, void* dqkv, int B,
                                      int Tn, int C, int dtype, void* stream) {
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    if (!qkv || !o || !d_o || !lse || !delta_ws || !dqkv || B <= 0 || Tn <= 0 || C % P) return C2W_ERR_BAD_SHAPE;
    if (c2w_attention_mfma_eligible(B, Tn, C, dtype)) return c2w_attention_mfma_backward(qkv, d_o, lse, dqkv, B, C, dtype, (hipStream_t)stream);
    if (c2w_attention_mfma_blocks_eligible(B, Tn, C, dtype)) {  // T = 64 nb: delta over the whole row first, then the two block kernels
        const long long nrows = (long long)B * Tn;
        const int rg = (int)((nrows + 15) / 16 < 4096 ? (nrows + 15) / 16 : 4096);
        if (dtype == C2W_DTYPE_F16) rowdot_kernel<f16_t><<<rg, 256, 0, (hipStream_t)stream>>>((const f16_t*)d_o, (const f16_t*)o, delta_ws, nrows, C);
        else rowdot_kernel<bf16_t><<<rg, 256, 0, (hipStream_t)stream>>>((const bf16_t*)d_o, (const bf16_t*)o, delta_ws, nrows, C);
        return c2w_attention_mfma_blocks_backward(qkv, d_o, lse, delta_ws, dqkv, B, Tn, C, dtype, (hipStream_t)stream);
    }
    const int lds = (TR * C + 2 * TR * Tn) * 4;
    if (lds > 160 * 1024) return C2W_ERR_UNSUPPORTED;
    dim3 grid((Tn + TR - 1) / TR, B);
    hipStream_t st = (hipStream_t)stream;
    const long long rows = (long long)B * Tn;
    const int rgrid = (int)((rows + 15) / 16 < 4096 ? (rows + 15) / 16 : 4096);
    if (dtype == C2W_DTYPE_F32) return valu_backward<float>(qkv, o, d_o, lse, delta_ws, dqkv, Tn, C, rows, grid, rgrid, lds, st);
    if (dtype == C2W_DTYPE_BF16) return valu_backward<bf16_t>(qkv, o, d_o, lse, delta_ws, dqkv, Tn, C, rows, grid, rgrid, lds, st);
    if (dtype == C2W_DTYPE_F16) return valu_backward<f16_t>(qkv, o, d_o, lse, delta_ws, dqkv, Tn, C, rows, grid, rgrid, lds, st);
    return C2W_ERR_BAD_ARG;
}
