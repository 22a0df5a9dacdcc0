// HBM-bound kernels around the convolutions: channel LayerNorm (+ time modulation) forward/backward,
// layout changes at the NCHW fp32 boundary, noise process / loss, reductions, optimizer.
// All are streaming kernels: 16-byte vector accesses, fp32 math, wave-level (16-lane sub-group) reductions.
#include <algorithm>
#include <cstdlib>
#include "common.h"
#include "philox.h"
#include "c2w_hip.h"

namespace {

constexpr int LN_MAXV = 8;  // 16 lanes x LN_MAXV vectors of 16 B per pixel  (C <= 1024 bf16 / 512 fp32)

__device__ __forceinline__ float sub16_sum(float v) {  // sum over the 16 lanes that share a pixel
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// y = LN_C(x + m[b])   zuko.nn.LayerNorm as used at model/nn.py:44,154,183 fused with the broadcast add of
// model/nn.py:28.  16 lanes per pixel, whole channel row in registers, two-pass mean/variance.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ m, T* __restrict__ y,
                                                     long long npix, int HW, int C, int ldm, float eps, float inv_den) {
    constexpr int P = Elem<T>::PER16;
    const int sub = threadIdx.x >> 4, j = threadIdx.x & 15;
    constexpr int nv = NV;
    for (long long pix = (long long)blockIdx.x * 16 + sub; pix < npix; pix += (long long)gridDim.x * 16) {
        const T* xr = x + pix * C;
        const float* mr = m ? m + (size_t)(ldm ? (pix / HW) : 0) * ldm : nullptr;
        float f[NV][P];
        float s = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
                unpack16<T>(*(const u32x4_t*)(xr + c), f[v]);
                if (mr) {
#pragma unroll
                    for (int e = 0; e < P; ++e) f[v][e] += mr[c + e];
                }
#pragma unroll
                for (int e = 0; e < P; ++e) s += f[v][e];
            }
        }
        const float mean = sub16_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    f[v][e] -= mean;
                    q += f[v][e] * f[v][e];
                }
            }
        }
        const float rs = 1.0f / sqrtf(sub16_sum(q) * inv_den + eps);
        T* yr = y + pix * C;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
#pragma unroll
                for (int e = 0; e < P; ++e) f[v][e] *= rs;
                *(u32x4_t*)(yr + c) = pack16<T>(f[v]);
            }
        }
    }
}

// dx = dres + d/dx [ LN_C(x + m) ] . dy ;  dm[b] += sum_pixels of the LN part (the modulation gradient).
// With s = sqrt(var + eps), xh = (x+m-mean)/s, den = C-1 (unbiased) or C:
//   dxm = ( dy - mean(dy) - xh * sum(dy*xh)/den ) / s
// grid = (chunks per image, images): every pixel of a block belongs to one image, so the modulation gradient is
// reduced in registers -> per-sub-group partial rows in LDS (plain stores) -> ONE contiguous global atomic sweep per block.
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ m,
                                                     const T* __restrict__ dres, T* __restrict__ dx, float* __restrict__ dm,
                                                     int HW, int C, int ldm, float eps, float inv_den, int pix_per_block) {
    constexpr int P = Elem<T>::PER16;
    extern __shared__ __attribute__((aligned(16))) float ln_red[];  // [16 pixel sub-groups][C], only with dm
    const int sub = threadIdx.x >> 4, j = threadIdx.x & 15;
    constexpr int nv = NV;
    const long long b = blockIdx.y;
    float am[NV][P];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int e = 0; e < P; ++e) am[v][e] = 0.f;
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = (p0 + pix_per_block < HW) ? p0 + pix_per_block : HW;
    const float* mr = m ? m + (size_t)(ldm ? b : 0) * ldm : nullptr;
    float mv[NV][P];  // the image's modulation row, loaded once (every pixel of the block belongs to image b)
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        const int c = (v * 16 + j) * P;
#pragma unroll
        for (int e = 0; e < P; ++e) mv[v][e] = (mr && v < nv && c < C) ? mr[c + e] : 0.f;
    }
    for (int pp = p0 + sub; pp < p1; pp += 16) {
        const long long pix = b * HW + pp;
        const T* xr = x + pix * C;
        const T* gr = dy + pix * C;
        float f[NV][P], g[NV][P];
        float s = 0.f, sg = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
                unpack16<T>(*(const u32x4_t*)(xr + c), f[v]);
                unpack16<T>(*(const u32x4_t*)(gr + c), g[v]);
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    f[v][e] += mv[v][e];
                    s += f[v][e];
                    sg += g[v][e];
                }
            }
        }
        const float mean = sub16_sum(s) / (float)C;
        const float gmean = sub16_sum(sg) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    f[v][e] -= mean;
                    q += f[v][e] * f[v][e];
                }
            }
        }
        const float rs = 1.0f / sqrtf(sub16_sum(q) * inv_den + eps);
        float d = 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    f[v][e] *= rs;  // xhat
                    d += g[v][e] * f[v][e];
                }
            }
        }
        const float dot = sub16_sum(d) * inv_den;
        T* dxr = dx + pix * C;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
                float o[P];
#pragma unroll
                for (int e = 0; e < P; ++e) {
                    o[e] = (g[v][e] - gmean - f[v][e] * dot) * rs;
                    am[v][e] += o[e];
                }
                if (dres) {
                    float r[P];
                    unpack16<T>(*(const u32x4_t*)(dres + pix * C + c), r);
#pragma unroll
                    for (int e = 0; e < P; ++e) o[e] += r[e];
                }
                *(u32x4_t*)(dxr + c) = pack16<T>(o);
            }
        }
    }
    if (dm != nullptr) {
        // Every sub-group stores its partial column sums as plain 16-B vectors, then thread c adds the 16 partials of
        // channel c (consecutive threads -> consecutive banks) and issues the block's one global atomic for it.  (LDS
        // atomics on this layout -- lanes 8 floats apart, four lanes per address -- serialised 16-fold and cost 2-5x the
        // whole streaming pass at the small levels: 95 vs 18 us at 16x16 x 384 channels.)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int c = (v * 16 + j) * P;
            if (v < nv && c < C) {
#pragma unroll
                for (int e = 0; e < P; e += 4)
                    *(f32x4_t*)(ln_red + (size_t)sub * C + c + e) = (f32x4_t){am[v][e], am[v][e + 1], am[v][e + 2], am[v][e + 3]};
            }
        }
        __syncthreads();
        float* dr = dm + (size_t)(ldm ? b : 0) * ldm;
        for (int c = threadIdx.x; c < C; c += 256) {
            float sum = 0.f;
#pragma unroll
            for (int sb = 0; sb < 16; ++sb) sum += ln_red[sb * C + c];
            atomicAdd(dr + c, sum);
        }
    }
}

// out[c] += sum over rows of a[row][c]   (bias gradients).  Consecutive threads read consecutive 16-B vectors of a row
// (coalesced); row groups are combined through LDS and each block issues one contiguous atomic sweep.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ a, float* __restrict__ out, long long rows, int C, int lda,
                                                     int rows_per_block) {
    constexpr int P = Elem<T>::PER16;
    __shared__ float red[256 * P];
    const int vbase = blockIdx.y * 256;              // this block's slice of the row: up to 256 vectors of 16 B
    const int nvec = (C / P - vbase) < 256 ? (C / P - vbase) : 256;
    a += (size_t)vbase * P;
    out += (size_t)vbase * P;
    C = nvec * P;
    const int ngrp = 256 / nvec;
    const int vec = threadIdx.x % nvec, grp = threadIdx.x / nvec;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
    float acc[P];
#pragma unroll
    for (int e = 0; e < P; ++e) acc[e] = 0.f;
    if (grp < ngrp) {
        for (long long r = r0 + grp; r < r1; r += ngrp) {
            float f[P];
            unpack16<T>(*(const u32x4_t*)(a + r * lda + vec * P), f);
#pragma unroll
            for (int e = 0; e < P; ++e) acc[e] += f[e];
        }
#pragma unroll
        for (int e = 0; e < P; ++e) red[(grp * nvec + vec) * P + e] = acc[e];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int g = 0; g < ngrp; ++g) s += red[g * C + c];
        atomicAdd(out + c, s);
    }
}

template <typename T, int OP>  // OP 0: y = silu(x); 1: y = a * silu'(x)
__global__ __launch_bounds__(256) void silu_kernel(const T* __restrict__ x, const T* __restrict__ a, T* __restrict__ y, long long nvec) {
    constexpr int P = Elem<T>::PER16;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
        float f[P];
        unpack16<T>(((const u32x4_t*)x)[i], f);
        if (OP == 0) {
#pragma unroll
            for (int e = 0; e < P; ++e) f[e] = silu_f(f[e]);
        } else {
            float g[P];
            unpack16<T>(((const u32x4_t*)a)[i], g);
#pragma unroll
            for (int e = 0; e < P; ++e) f[e] = g[e] * dsilu_f(f[e]);
        }
        ((u32x4_t*)y)[i] = pack16<T>(f);
    }
}

// dx[b][h][w][:] = sum_{i,j<2} g[b][2h+i][2w+j][:]      (adjoint of Upsample(nearest, x2), model/nn.py:184)
template <typename T>
__global__ __launch_bounds__(256) void sumpool2_kernel(const T* __restrict__ g, T* __restrict__ dx, int B, int H, int W, int C) {
    constexpr int P = Elem<T>::PER16;
    const int nvec = C / P;
    const long long total = (long long)B * H * W * nvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int vec = (int)(i % nvec);
        long long pix = i / nvec;
        const int w = (int)(pix % W);
        pix /= W;
        const int h = (int)(pix % H);
        const int b = (int)(pix / H);
        float acc[P];
#pragma unroll
        for (int e = 0; e < P; ++e) acc[e] = 0.f;
#pragma unroll
        for (int di = 0; di < 2; ++di)
#pragma unroll
            for (int dj = 0; dj < 2; ++dj) {
                float f[P];
                unpack16<T>(*(const u32x4_t*)(g + ((((size_t)b * 2 * H + 2 * h + di) * 2 * W) + 2 * w + dj) * C + vec * P), f);
#pragma unroll
                for (int e = 0; e < P; ++e) acc[e] += f[e];
            }
        ((u32x4_t*)dx)[i] = pack16<T>(acc);
    }
}

// y[b][2h+i][2w+j][:] = x[b][h][w][:]   (Upsample(nearest, x2), model/nn.py:184, materialised so that the 3x3 conv after it
// and its weight gradient run on the halo-patch kernels; the gather-mode `UP` convolution reads x directly instead)
template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C) {
    constexpr int P = Elem<T>::PER16;
    const int nvec = C / P;
    const long long total = (long long)B * 2 * H * 2 * W * nvec;  // one 16-B vector of the OUTPUT per thread: contiguous stores
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int vec = (int)(i % nvec);
        long long pix = i / nvec;
        const int ow = (int)(pix % (2 * W));
        pix /= 2 * W;
        const int oh = (int)(pix % (2 * H));
        const int b = (int)(pix / (2 * H));
        ((u32x4_t*)y)[i] = *(const u32x4_t*)(x + (((size_t)b * H + (oh >> 1)) * W + (ow >> 1)) * C + vec * P);
    }
}

// NCHW fp32 -> NHWC T with channel padding (zeros).  Optional fused noise process xt = mu[b] x + sigma[b] eps
// (src/thor/pipelines.py:22-25): pass eps (NCHW fp32) and musig[b] = {mu, sigma}.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                           const float* __restrict__ musig, T* __restrict__ y, int B, int C, int HW, int ldc) {
    constexpr int P = Elem<T>::PER16;
    const int nvec = ldc / P;
    const long long total = (long long)B * HW * nvec;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // consecutive threads -> consecutive pixels of one (b, channel-vector): reads coalesce per channel plane
        const int pix = (int)(i % HW);
        const long long r = i / HW;
        const int vec = (int)(r % nvec);
        const int b = (int)(r / nvec);
        float f[P];
#pragma unroll
        for (int e = 0; e < P; ++e) {
            const int c = vec * P + e;
            float v = 0.f;
            if (c < C) {
                const size_t o = ((size_t)b * C + c) * HW + pix;
                v = x[o];
                if (eps) v = musig[2 * b] * v + musig[2 * b + 1] * eps[o];
            }
            f[e] = v;
        }
        *(u32x4_t*)(y + ((size_t)b * HW + pix) * ldc + vec * P) = pack16<T>(f);
    }
}

// NHWC T -> NCHW fp32 (first C channels)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const T* __restrict__ y, float* __restrict__ out, int B, int C, int HW, int ldc) {
    const long long total = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long r = i / HW;
        const int c = (int)(r % C);
        const int b = (int)(r / C);
        out[i] = Elem<T>::ld(y + ((size_t)b * HW + pix) * ldc + c);
    }
}

// Training loss tail (src/thor/pipelines.py:35 + training_loop.py:377):  l = mean((y - eps)^2) * scale
//   dy[b][pix][c] = 2 (y - eps) * scale / N   (NHWC T, padded channels = 0);  loss_sum += sum (y-eps)^2 (fp32 atomics)
template <typename T>
__global__ __launch_bounds__(256) void mse_loss_grad_kernel(const T* __restrict__ y, const float* __restrict__ eps, T* __restrict__ dy,
                                                            float* __restrict__ loss_sum, int B, int C, int HW, int ldc, float gscale,
                                                            const float* __restrict__ dscale) {
    if (dscale != nullptr) gscale *= dscale[0];  // dynamic loss scale (c2w_grad_scaler_*), read on the device
    constexpr int P = Elem<T>::PER16;
    const int nvec = ldc / P;
    const long long total = (long long)B * HW * nvec;
    float local = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pix = (int)(i % HW);
        const long long r = i / HW;
        const int vec = (int)(r % nvec);
        const int b = (int)(r / nvec);
        const size_t o = ((size_t)b * HW + pix) * ldc + vec * P;
        float f[P];
        unpack16<T>(*(const u32x4_t*)(y + o), f);
#pragma unroll
        for (int e = 0; e < P; ++e) {
            const int c = vec * P + e;
            float d = 0.f;
            if (c < C) {
                d = f[e] - eps[((size_t)b * C + c) * HW + pix];
                local += d * d;
            }
            f[e] = d * gscale;
        }
        *(u32x4_t*)(dy + o) = pack16<T>(f);
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, local);
}

// ---- LDS-tiled layout kernels: one block = one image x 64 pixels.  Channel planes (NCHW fp32) are read as coalesced rows
//      into an LDS tile [channel][pixel], NHWC rows leave as contiguous 16-B vectors (the whole 64-pixel x ldc span of
//      the output is one contiguous byte range).  Replaces per-thread strided 16-B stores / 4-B gathers.
constexpr int LT_PT = 64;  // pixels per tile (256-B runs of every channel plane; ldc = 128 channels -> 33 KB of LDS)
constexpr int LT_LD = LT_PT + 1;

// Counter-based normal noise (philox_normal4 / philox_normal1): philox.h
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, long long n, uint32_t k0, uint32_t k1) {
    const long long nb = (n + 3) >> 2;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < nb; b += (long long)gridDim.x * blockDim.x) {
        const f32x4_t v = philox_normal4(k0, k1, (unsigned long long)b);
        if (4 * b + 3 < n) {
            *(f32x4_t*)(out + 4 * b) = v;
        } else {
            for (int i = 0; 4 * b + i < n; ++i) out[4 * b + i] = v[i];
        }
    }
}

// tile[c][px] = f(plane values) for c < ldc (zero beyond C / beyond HW)
template <typename F>
__device__ __forceinline__ void lt_load_planes(float* tile, int C, int ldc, int HW, int p0, size_t img_off, F&& f) {
    const int tid = threadIdx.x;
    if ((HW & 3) == 0) {  // 16-B loads: 16 lanes cover one channel row of the tile
        const int q = tid & (LT_PT / 4 - 1), cb = tid / (LT_PT / 4);
        const bool in = p0 + 4 * q < HW;
#pragma unroll 4
        for (int c = cb; c < ldc; c += 256 / (LT_PT / 4)) {
            f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (c < C && in) v = f(img_off + (size_t)c * HW + p0 + 4 * q);
            float* d = tile + c * LT_LD + 4 * q;
            d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
        }
    } else {
        const int px = tid & (LT_PT - 1), cb = tid / LT_PT;
        const bool in = p0 + px < HW;
#pragma unroll 4
        for (int c = cb; c < ldc; c += 256 / LT_PT) {
            float v = 0.f;
            if (c < C && in) v = f(img_off + (size_t)c * HW + p0 + px, 0);
            tile[c * LT_LD + px] = v;
        }
    }
}

template <typename T, bool PHILOX = false>
__global__ __launch_bounds__(256) void nchw_to_nhwc_tiled_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                                 const float* __restrict__ musig, T* __restrict__ y, int B, int C, int HW,
                                                                 int ldc, long long img_stride, uint32_t k0 = 0, uint32_t k1 = 0,
                                                                 const long long* __restrict__ img_off = nullptr) {
    constexpr int P = Elem<T>::PER16;
    extern __shared__ float lt_tile[];
    const int ntile = (HW + LT_PT - 1) / LT_PT, nvec = ldc / P;
    const float* const x_all = x;
    for (int blk = blockIdx.x; blk < B * ntile; blk += gridDim.x) {
        const int b = blk / ntile, p0 = (blk - b * ntile) * LT_PT;
        // img_off: image b starts img_off[b] floats into x (windows picked out of a dataset array).  The element index the noise
        // stream is addressed with stays the dense one, b * img_stride + ..., whatever the image's place in memory.
        if (img_off != nullptr) x = x_all + (img_off[b] - (long long)b * img_stride);
        float mu = 1.f, sg = 0.f;
        if (eps || PHILOX) { mu = musig[2 * b]; sg = musig[2 * b + 1]; }
        struct Ld {
            const float *x, *eps; float mu, sg; uint32_t k0, k1;
            // x_t = fma(sigma, eps, mu * x) with the product rounded on its own: spelled out so that the kernel that reads eps
            // and the one that regenerates it round identically (left to the compiler, the two contracted differently)
            static __device__ __forceinline__ float mix(float mu, float xv, float sg, float e) { return __fmaf_rn(sg, e, __fmul_rn(mu, xv)); }
            __device__ __forceinline__ f32x4_t operator()(size_t o) const {
                f32x4_t v = *(const f32x4_t*)(x + o);
                if constexpr (PHILOX) {
                    const f32x4_t e = philox_normal4(k0, k1, (unsigned long long)o >> 2);  // o is a multiple of 4 on this path
                    v = (f32x4_t){mix(mu, v[0], sg, e[0]), mix(mu, v[1], sg, e[1]), mix(mu, v[2], sg, e[2]), mix(mu, v[3], sg, e[3])};
                } else if (eps) {
                    const f32x4_t e = *(const f32x4_t*)(eps + o);
                    v = (f32x4_t){mix(mu, v[0], sg, e[0]), mix(mu, v[1], sg, e[1]), mix(mu, v[2], sg, e[2]), mix(mu, v[3], sg, e[3])};
                }
                return v;
            }
            __device__ __forceinline__ float operator()(size_t o, int) const {
                if constexpr (PHILOX) return mix(mu, x[o], sg, philox_normal1(k0, k1, (unsigned long long)o));
                return eps ? mix(mu, x[o], sg, eps[o]) : x[o];
            }
        } ld{x, eps, mu, sg, k0, k1};
        lt_load_planes(lt_tile, C, ldc, HW, p0, (size_t)b * img_stride, ld);  // img_stride < C*HW: overlapping windows of a trajectory
        __syncthreads();
        const int npx = min(LT_PT, HW - p0);
        for (int i = threadIdx.x; i < npx * nvec; i += 256) {
            const int px = i / nvec, v = i - px * nvec;
            float f[P];
#pragma unroll
            for (int e = 0; e < P; ++e) f[e] = lt_tile[(v * P + e) * LT_LD + px];
            *(u32x4_t*)(y + ((size_t)b * HW + p0) * ldc + (size_t)i * P) = pack16<T>(f);
        }
        __syncthreads();
    }
}

// nchw_to_nhwc_tiled_kernel<T, true> that KEEPS the noise (round 6): eps is rounded to half precision, mixed into x_t in that rounded
// form and written as NHWC rows erows [B*HW][lde] -- the loss tail reads it back (conv_patch_t3_kernel EPI 7) instead of running the
// generator a second time.  Second LDS tile [lde][LT_LD] for the noise.  HW % 4 == 0; 16-bit outputs.
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_eps_kernel(const float* __restrict__ x, const float* __restrict__ musig, T* __restrict__ y,
                                                               f16_t* __restrict__ erows, int B, int C, int HW, int ldc, int lde,
                                                               long long img_stride, uint32_t k0, uint32_t k1,
                                                               const long long* __restrict__ img_off) {
    constexpr int P = Elem<T>::PER16;
    static_assert(P == 8, "16-bit storage");
    extern __shared__ float lt_tile[];
    float* const et = lt_tile + (size_t)ldc * LT_LD;
    const int ntile = (HW + LT_PT - 1) / LT_PT, nvec = ldc / P, nve = lde / 8;
    const float* const x_all = x;
    const int q = threadIdx.x & (LT_PT / 4 - 1), cb = threadIdx.x / (LT_PT / 4);
    const int cmax = ldc > lde ? ldc : lde;
    for (int blk = blockIdx.x; blk < B * ntile; blk += gridDim.x) {
        const int b = blk / ntile, p0 = (blk - b * ntile) * LT_PT;
        if (img_off != nullptr) x = x_all + (img_off[b] - (long long)b * img_stride);
        const float mu = musig[2 * b], sg = musig[2 * b + 1];
        const bool in = p0 + 4 * q < HW;
#pragma unroll 2
        for (int c = cb; c < cmax; c += 256 / (LT_PT / 4)) {
            f32x4_t v = (f32x4_t){0.f, 0.f, 0.f, 0.f}, e = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (c < C && in) {
                const size_t o = (size_t)b * img_stride + (size_t)c * HW + p0 + 4 * q;
                const f32x4_t xv = *(const f32x4_t*)(x + o);
                const f32x4_t en = philox_normal4(k0, k1, (unsigned long long)o >> 2);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    e[k] = (float)(f16_t)en[k];  // the step's noise: the stream rounded to half precision (RNE), here and in the loss
                    v[k] = __fmaf_rn(sg, e[k], __fmul_rn(mu, xv[k]));
                }
            }
            if (c < ldc) {
                float* d = lt_tile + c * LT_LD + 4 * q;
                d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
            }
            if (c < lde) {
                float* d = et + c * LT_LD + 4 * q;
                d[0] = e[0]; d[1] = e[1]; d[2] = e[2]; d[3] = e[3];
            }
        }
        __syncthreads();
        const int npx = min(LT_PT, HW - p0);
        for (int i = threadIdx.x; i < npx * nvec; i += 256) {
            const int px = i / nvec, v = i - px * nvec;
            float f[P];
#pragma unroll
            for (int k = 0; k < P; ++k) f[k] = lt_tile[(v * P + k) * LT_LD + px];
            *(u32x4_t*)(y + ((size_t)b * HW + p0) * ldc + (size_t)i * P) = pack16<T>(f);
        }
        for (int i = threadIdx.x; i < npx * nve; i += 256) {
            const int px = i / nve, v = i - px * nve;
            float f[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] = et[(v * 8 + k) * LT_LD + px];
            *(u32x4_t*)(erows + ((size_t)b * HW + p0) * lde + (size_t)i * 8) = pack16<f16_t>(f);
        }
        __syncthreads();
    }
}

template <typename T, bool PHILOX = false>
__global__ __launch_bounds__(256) void mse_loss_grad_tiled_kernel(const T* __restrict__ y, const float* __restrict__ eps, T* __restrict__ dy,
                                                                  float* __restrict__ loss_sum, int B, int C, int HW, int ldc, float gscale,
                                                                  const float* __restrict__ dscale, uint32_t k0 = 0, uint32_t k1 = 0) {
    if (dscale != nullptr) gscale *= dscale[0];
    constexpr int P = Elem<T>::PER16;
    extern __shared__ float lt_tile[];
    const int ntile = (HW + LT_PT - 1) / LT_PT, nvec = ldc / P;
    float local = 0.f;
    for (int blk = blockIdx.x; blk < B * ntile; blk += gridDim.x) {
        const int b = blk / ntile, p0 = (blk - b * ntile) * LT_PT;
        struct Ld {
            const float* eps; uint32_t k0, k1;
            __device__ __forceinline__ f32x4_t operator()(size_t o) const {
                if constexpr (PHILOX) return philox_normal4(k0, k1, (unsigned long long)o >> 2);
                return *(const f32x4_t*)(eps + o);
            }
            __device__ __forceinline__ float operator()(size_t o, int) const {
                if constexpr (PHILOX) return philox_normal1(k0, k1, (unsigned long long)o);
                return eps[o];
            }
        } ld{eps, k0, k1};
        lt_load_planes(lt_tile, C, ldc, HW, p0, (size_t)b * C * HW, ld);
        __syncthreads();
        const int npx = min(LT_PT, HW - p0);
        for (int i = threadIdx.x; i < npx * nvec; i += 256) {
            const int px = i / nvec, v = i - px * nvec;
            const size_t o = ((size_t)b * HW + p0) * ldc + (size_t)i * P;
            float f[P];
            unpack16<T>(*(const u32x4_t*)(y + o), f);
#pragma unroll
            for (int e = 0; e < P; ++e) {
                float d = 0.f;
                if (v * P + e < C) {
                    d = f[e] - lt_tile[(v * P + e) * LT_LD + px];
                    local += d * d;
                }
                f[e] = d * gscale;
            }
            *(u32x4_t*)(dy + o) = pack16<T>(f);
        }
        __syncthreads();
    }
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, local);
}

// NHWC T -> NCHW fp32 through the same LDS tile: rows of 64 pixels x ldc channels are read as one contiguous run, channel planes
// leave as 256-B runs (the untiled kernel gathered 2-byte elements at a stride of ldc: 2 ms for 32 x 80 x 256^2)
template <typename T>
__global__ __launch_bounds__(256) void nhwc_to_nchw_tiled_kernel(const T* __restrict__ y, float* __restrict__ out, int B, int C, int HW, int ldc) {
    constexpr int P = Elem<T>::PER16;
    extern __shared__ float lt_tile[];
    const int ntile = (HW + LT_PT - 1) / LT_PT, nvec = ldc / P;
    for (int blk = blockIdx.x; blk < B * ntile; blk += gridDim.x) {
        const int b = blk / ntile, p0 = (blk - b * ntile) * LT_PT;
        const int npx = min(LT_PT, HW - p0);
        for (int i = threadIdx.x; i < npx * nvec; i += 256) {
            const int px = i / nvec, v = i - px * nvec;
            float f[P];
            unpack16<T>(*(const u32x4_t*)(y + ((size_t)b * HW + p0) * ldc + (size_t)i * P), f);
#pragma unroll
            for (int e = 0; e < P; ++e) lt_tile[(v * P + e) * LT_LD + px] = f[e];
        }
        __syncthreads();
        const int px = threadIdx.x & (LT_PT - 1), cb = threadIdx.x / LT_PT;
        if (px < npx) {
#pragma unroll 4
            for (int c = cb; c < C; c += 256 / LT_PT) out[((size_t)b * C + c) * HW + p0 + px] = lt_tile[c * LT_LD + px];
        }
        __syncthreads();
    }
}

// Unreduced training loss of the module path (src/thor/pipelines.py:35: ``(eps_pred - eps) ** 2``, the caller takes .mean()):
// out[b][c][px] (NCHW fp32) = (y[b][px][c] - eps[b][c][px])^2 through the same LDS tile; eps read from memory (SQ == 1) or
// regenerated from the step's Philox stream (SQ == 2: four consecutive pixels of a channel plane are one counter).  HW % 4 == 0.
template <typename T, int SQ>
__global__ __launch_bounds__(256) void sq_err_tiled_kernel(const T* __restrict__ y, const float* __restrict__ eps, float* __restrict__ out,
                                                           float* __restrict__ loss_sum, int B, int C, int HW, int ldc, uint32_t k0, uint32_t k1) {
    constexpr int P = Elem<T>::PER16;
    extern __shared__ float lt_tile[];
    const int ntile = (HW + LT_PT - 1) / LT_PT, nvec = ldc / P;
    float local = 0.f;
    for (int blk = blockIdx.x; blk < B * ntile; blk += gridDim.x) {
        const int b = blk / ntile, p0 = (blk - b * ntile) * LT_PT;
        const int npx = min(LT_PT, HW - p0);
        for (int i = threadIdx.x; i < npx * nvec; i += 256) {
            const int px = i / nvec, v = i - px * nvec;
            float f[P];
            unpack16<T>(*(const u32x4_t*)(y + ((size_t)b * HW + p0) * ldc + (size_t)i * P), f);
#pragma unroll
            for (int e = 0; e < P; ++e) lt_tile[(v * P + e) * LT_LD + px] = f[e];
        }
        __syncthreads();
        const int q = threadIdx.x & (LT_PT / 4 - 1), cb = threadIdx.x / (LT_PT / 4);
        if (4 * q < npx) {
#pragma unroll 2
            for (int c = cb; c < C; c += 256 / (LT_PT / 4)) {
                const size_t o = ((size_t)b * C + c) * HW + p0 + 4 * q;
                const f32x4_t e = SQ == 2 ? philox_normal4(k0, k1, (unsigned long long)o >> 2) : *(const f32x4_t*)(eps + o);
                const float* t = lt_tile + c * LT_LD + 4 * q;
                // eps as VALUES before the subtraction: without the (empty) asm statements hipcc contracts the Box-Muller product inside
                // philox_normal4 with the difference (fma(-r, cos, y): one rounding less than y - eps of the materialised stream)
                float e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
                asm volatile("" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3));
                const float d0 = t[0] - e0, d1 = t[1] - e1, d2 = t[2] - e2, d3 = t[3] - e3;
                const f32x4_t sq = (f32x4_t){__fmul_rn(d0, d0), __fmul_rn(d1, d1), __fmul_rn(d2, d2), __fmul_rn(d3, d3)};
                *(f32x4_t*)(out + o) = sq;
                local += (sq[0] + sq[1]) + (sq[2] + sq[3]);
            }
        }
        __syncthreads();
    }
    if (loss_sum != nullptr) {  // sum of the tensor just written: the caller's .mean() without a second pass over it
        local = wave_sum(local);
        if ((threadIdx.x & 63) == 0) atomicAdd(loss_sum, local);
    }
}

// timestep_embedding (model/score.py:14-34): out[b] = [cos(t f_i) | sin(t f_i)], f_i = exp(-ln(max_period) i/half)
__global__ void timestep_embedding_kernel(const float* __restrict__ t, float* __restrict__ out, int n, int dim, float max_period) {
    const int half = dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * dim) return;
    const int b = i / dim, k = i - b * dim;
    float v = 0.f;
    if (k < 2 * half) {
        const int kk = k < half ? k : k - half;
        const float freq = expf(-logf(max_period) * (float)kk / (float)half);
        const float a = t[b] * freq;
        v = k < half ? cosf(a) : sinf(a);
    }
    out[i] = v;
}

// VP-cosine schedule (src/thor/pipelines.py:13-20): musig[b] = {mu(t_b), sigma(t_b)}
__global__ void mu_sigma_kernel(const float* __restrict__ t, float* __restrict__ musig, int n, float eta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float c = cosf(acosf(sqrtf(eta)) * t[i]);
    const float a = c * c;
    musig[2 * i] = a;
    musig[2 * i + 1] = sqrtf(1.f - a * a + eta * eta);
}

template <typename TO>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ in, TO* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        Elem<TO>::st(out + i, in[i]);
}

// w[r][tap][k] (fp32, row stride NT*ldk) -> out[k][tap'][r] (T, row stride NT*ldr), tap' = NT-1-tap when flip:
// the operand of the input-gradient convolution.  Only r < R, k < K are written (padding stays as initialised).
template <typename TO>
__global__ __launch_bounds__(256) void weight_transpose_kernel(const float* __restrict__ w, TO* __restrict__ out, int R, int NT, int K,
                                                               int ldk, int ldr, int flip) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int otap = flip ? NT - 1 - tap : tap;
    const int r0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, k = k0 + tx;
        tile[i][tx] = (r < R && k < K) ? w[((size_t)r * NT + tap) * ldk + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int k = k0 + i, r = r0 + tx;
        if (k < K && r < R) Elem<TO>::st(out + ((size_t)k * NT + otap) * ldr + r, tile[tx][i]);
    }
}

// The same for every conv of the network in ONE launch: desc[j] = {w_off, out_off, R, NT, K, ldk, ldr, flip} (offsets in
// elements into `flat` / `out`); blockIdx.y = conv, blockIdx.x strides over its 32 x 32 x tap tiles.
template <typename TO>
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const float* __restrict__ flat, TO* __restrict__ outb,
                                                                       const long long* __restrict__ desc) {
    __shared__ float tile[32][33];
    const long long* d = desc + (size_t)blockIdx.y * 8;
    const float* w = flat + d[0];
    TO* out = outb + d[1];
    const int R = (int)d[2], NT = (int)d[3], K = (int)d[4], ldk = (int)d[5], ldr = (int)d[6], flip = (int)d[7];
    const int nk = (K + 31) / 32, nr = (R + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < nk * nr * NT; t += gridDim.x) {
        const int tap = t / (nk * nr), rem = t - tap * nk * nr;
        const int r0 = (rem / nk) * 32, k0 = (rem - (rem / nk) * nk) * 32;
        const int otap = flip ? NT - 1 - tap : tap;
        for (int i = ty; i < 32; i += 8) {
            const int r = r0 + i, k = k0 + tx;
            tile[i][tx] = (r < R && k < K) ? w[((size_t)r * NT + tap) * ldk + k] : 0.f;
        }
        __syncthreads();
        for (int i = ty; i < 32; i += 8) {
            const int k = k0 + i, r = r0 + tx;
            if (k < K && r < R) Elem<TO>::st(out + ((size_t)k * NT + otap) * ldr + r, tile[tx][i]);
        }
        __syncthreads();
    }
}

// Stage-major weight copies for conv_patch_t3_kernel (c2w_pack_conv_weights_batched in c2w_hip.h): one thread per 16-byte slot of the
// destination; blockIdx.y = matrix, blockIdx.x strides over its 9 * (cin / 32) * rows_pad * 4 slots.
__global__ __launch_bounds__(256) void pack_conv_weights_kernel(const uint16_t* __restrict__ srcb, uint16_t* __restrict__ dstb,
                                                                const long long* __restrict__ desc) {
    const long long* d = desc + (size_t)blockIdx.y * 4;
    const uint16_t* src = srcb + d[0];
    uint16_t* dst = dstb + d[1];
    const int rows = (int)d[2], cin = (int)d[3], nh = cin >> 5, rows_pad = (rows + 127) & ~127;
    const long long nslot = (long long)9 * nh * rows_pad * 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nslot; i += (long long)gridDim.x * 256) {
        const int s = (int)(i & 3);            // destination slot of the 64-byte row
        const long long rr = i >> 2;
        const int r = (int)(rr % rows_pad);
        const int th = (int)(rr / rows_pad);   // tap * nh + h
        const int tap = th / nh, h = th - tap * nh;
        const int q = s ^ ((4 - ((r >> 2) & 3)) & 3);  // source chunk that lands in slot s (t3_wswz)
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (r < rows) v = *(const u32x4_t*)(src + ((size_t)r * 9 + tap) * cin + h * 32 + q * 8);
        *(u32x4_t*)(dst + (size_t)i * 8) = v;
    }
}

// Fused AdamW (torch.optim.AdamW, train.py:176-181) + EMA (src/thor/ema.py:23-27) + low-precision shadow refresh.
// With a scaler state (c2w_grad_scaler_*): gradients are divided by the loss scale; a step whose gradients held inf/nan changes
// nothing but the EMA (GradScaler.step skips optimizer.step, the reference still calls ema.update: training_loop.py:383-389);
// the bias corrections count the steps actually taken (state[3], on the device -- the host never learns about a skip).
template <typename TS>
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, float* __restrict__ ema, TS* __restrict__ shadow,
                                                        long long n, float lr, float beta1, float beta2, float eps, float wd, float bc1,
                                                        float bc2_sqrt, float ema_rate, float grad_scale,
                                                        const float* __restrict__ scaler) {
    bool skip = false;
    if (scaler != nullptr) {
        grad_scale /= scaler[0];
        skip = scaler[2] != 0.f;
        const double step = (double)scaler[3] + 1.0;
        bc1 = (float)(1.0 - pow((double)beta1, step));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, step));
    }
    if (skip) {
        if (ema)
            for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
                ema[i] = ema_rate * ema[i] + (1.f - ema_rate) * p[i];
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        float pi = p[i] * (1.f - lr * wd);
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
        m[i] = mi;
        v[i] = vi;
        if (ema) ema[i] = ema_rate * ema[i] + (1.f - ema_rate) * pi;
        if (shadow) Elem<TS>::st(shadow + i, pi);
    }
}

// ---- dynamic loss scale, device resident.  state = {scale, growth tracker, found_inf, optimizer steps taken}
__global__ __launch_bounds__(256) void scaler_check_kernel(const float* __restrict__ g, long long n, float* __restrict__ state) {
    bool bad = false;
    const long long nv = n >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
        const f32x4_t q = *(const f32x4_t*)(g + 4 * i);
        // finite <=> exponent field not all ones
        bad |= ((__float_as_uint(q[0]) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(q[1]) & 0x7f800000u) == 0x7f800000u) |
               ((__float_as_uint(q[2]) & 0x7f800000u) == 0x7f800000u) | ((__float_as_uint(q[3]) & 0x7f800000u) == 0x7f800000u);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) bad |= (__float_as_uint(g[(nv << 2) + threadIdx.x]) & 0x7f800000u) == 0x7f800000u;
    if (__any(bad) && (threadIdx.x & 63) == 0) state[2] = 1.f;  // benign race: every writer stores the same value
}

__global__ void scaler_init_kernel(float* __restrict__ state, float init_scale) {
    if (threadIdx.x < 4) state[threadIdx.x] = threadIdx.x == 0 ? init_scale : 0.f;
}

__global__ void scaler_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[2] != 0.f) {
        state[0] *= backoff;
        state[1] = 0.f;
    } else {
        state[3] += 1.f;
        const float tr = state[1] + 1.f;
        if (tr >= (float)interval) {
            state[0] *= growth;
            state[1] = 0.f;
        } else {
            state[1] = tr;
        }
    }
    state[2] = 0.f;
}

// src/thor/ema.py:23-27 over a flat buffer
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ ema, const float* __restrict__ p, long long n, float rate) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        ema[i] = rate * ema[i] + (1.f - rate) * p[i];
}

inline int grid_for(long long n, int per_block = 256, int cap = 8192) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

#define DISPATCH_T(dtype, CALL)                                                         \
    do {                                                                                \
        if ((dtype) == C2W_DTYPE_F32) { using T = float; CALL; }                        \
        else if ((dtype) == C2W_DTYPE_BF16) { using T = bf16_t; CALL; }                 \
        else if ((dtype) == C2W_DTYPE_F16) { using T = f16_t; CALL; }                   \
        else return C2W_ERR_BAD_ARG;                                                    \
    } while (0)

static inline bool vec_ok(int dtype, int C) { return C > 0 && C % (dtype == C2W_DTYPE_F32 ? 4 : 8) == 0; }

extern "C" int c2w_ln_forward(const void* x, const float* m, void* y, long long npix, int HW, int C, int ldm, float eps, int unbiased,
                              int dtype, void* stream) {
    if (!x || !y || !vec_ok(dtype, C) || C > 16 * LN_MAXV * (dtype == C2W_DTYPE_F32 ? 4 : 8) || C < 2) return C2W_ERR_BAD_SHAPE;
    const float inv_den = 1.0f / (float)(unbiased ? C - 1 : C);
    const int nv = (C + 16 * (dtype == C2W_DTYPE_F32 ? 4 : 8) - 1) / (16 * (dtype == C2W_DTYPE_F32 ? 4 : 8));
#define LN_FWD(NVV) DISPATCH_T(dtype, (ln_fwd_kernel<T, NVV><<<grid_for(npix, 16, 16384), 256, 0, (hipStream_t)stream>>>( \
    (const T*)x, m, (T*)y, npix, HW, C, ldm, eps, inv_den)))
    if (nv <= 1) LN_FWD(1); else if (nv == 2) LN_FWD(2); else if (nv == 3) LN_FWD(3); else if (nv == 4) LN_FWD(4);
    else if (nv <= 6) LN_FWD(6); else LN_FWD(8);
#undef LN_FWD
    return (int)hipGetLastError();
}

extern "C" int c2w_ln_backward(const void* dy, const void* x, const float* m, const void* dres, void* dx, float* dm, long long npix, int HW,
                               int C, int ldm, float eps, int unbiased, int dtype, void* stream) {
    if (!dy || !x || !dx || !vec_ok(dtype, C) || C > 16 * LN_MAXV * (dtype == C2W_DTYPE_F32 ? 4 : 8) || C < 2) return C2W_ERR_BAD_SHAPE;
    const float inv_den = 1.0f / (float)(unbiased ? C - 1 : C);
    if (HW <= 0 || npix % HW != 0) return C2W_ERR_BAD_SHAPE;
    // pixels per block: 512 at the large levels; fewer where that would leave the chip with under ~2048 blocks (32x32 and
    // below at B = 128: 256 / 128 blocks of 4 waves were latency-bound at 3.5 TB/s)
    long long want = (npix / 2048 + 15) / 16 * 16;
    if (want < 16) want = 16;
    if (want > 512) want = 512;
    const int ppb = HW < want ? HW : (int)want;
    dim3 grid((HW + ppb - 1) / ppb, (unsigned)(npix / HW));
    const int nv = (C + 16 * (dtype == C2W_DTYPE_F32 ? 4 : 8) - 1) / (16 * (dtype == C2W_DTYPE_F32 ? 4 : 8));
    const size_t red_bytes = dm != nullptr ? (size_t)16 * C * sizeof(float) : 0;  // <= 64 KiB (C <= 1024)
#define LN_BWD(NVV) DISPATCH_T(dtype, (ln_bwd_kernel<T, NVV><<<grid, 256, red_bytes, (hipStream_t)stream>>>( \
    (const T*)dy, (const T*)x, m, (const T*)dres, (T*)dx, dm, HW, C, ldm, eps, inv_den, ppb)))
    if (nv <= 1) LN_BWD(1); else if (nv == 2) LN_BWD(2); else if (nv == 3) LN_BWD(3); else if (nv == 4) LN_BWD(4);
    else if (nv <= 6) LN_BWD(6); else LN_BWD(8);
#undef LN_BWD
    return (int)hipGetLastError();
}

extern "C" int c2w_colsum(const void* a, float* out, long long rows, int C, int lda, int dtype, void* stream) {
    if (!a || !out || !vec_ok(dtype, C) || !vec_ok(dtype, lda)) return C2W_ERR_BAD_SHAPE;
    const int rpb = 2048;
    const int nvec_total = C / (dtype == C2W_DTYPE_F32 ? 4 : 8);
    const dim3 grid((unsigned)((rows + rpb - 1) / rpb), (unsigned)((nvec_total + 255) / 256));
    DISPATCH_T(dtype, (colsum_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>((const T*)a, out, rows, C, lda, rpb)));
    return (int)hipGetLastError();
}

extern "C" int c2w_silu(const void* x, void* y, long long n, int dtype, void* stream) {
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    if (!x || !y || n % P) return C2W_ERR_BAD_SHAPE;
    DISPATCH_T(dtype, (silu_kernel<T, 0><<<grid_for(n / P), 256, 0, (hipStream_t)stream>>>((const T*)x, nullptr, (T*)y, n / P)));
    return (int)hipGetLastError();
}

extern "C" int c2w_silu_backward(const void* x, const void* dy, void* dx, long long n, int dtype, void* stream) {
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    if (!x || !dy || !dx || n % P) return C2W_ERR_BAD_SHAPE;
    DISPATCH_T(dtype, (silu_kernel<T, 1><<<grid_for(n / P), 256, 0, (hipStream_t)stream>>>((const T*)x, (const T*)dy, (T*)dx, n / P)));
    return (int)hipGetLastError();
}

extern "C" int c2w_sumpool2(const void* g, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
    if (!g || !dx || !vec_ok(dtype, C)) return C2W_ERR_BAD_SHAPE;
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    DISPATCH_T(dtype, (sumpool2_kernel<T><<<grid_for((long long)B * H * W * (C / P)), 256, 0, (hipStream_t)stream>>>((const T*)g, (T*)dx, B, H,
                                                                                                                  W, C)));
    return (int)hipGetLastError();
}

extern "C" int c2w_upsample2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    if (!x || !y || !vec_ok(dtype, C)) return C2W_ERR_BAD_SHAPE;
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    DISPATCH_T(dtype, (upsample2_kernel<T><<<grid_for((long long)B * 4 * H * W * (C / P), 256, 16384), 256, 0, (hipStream_t)stream>>>(
                          (const T*)x, (T*)y, B, H, W, C)));
    return (int)hipGetLastError();
}

extern "C" int c2w_nchw_to_nhwc(const float* x, const float* eps, const float* musig, void* y, int B, int C, int HW, int ldc, int dtype,
                                void* stream) {
    if (!x || !y || !vec_ok(dtype, ldc) || ldc < C || (eps && !musig)) return C2W_ERR_BAD_SHAPE;
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds <= 64 * 1024) {
        const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
        DISPATCH_T(dtype, (nchw_to_nhwc_tiled_kernel<T><<<nblk, 256, lds, (hipStream_t)stream>>>(x, eps, musig, (T*)y, B, C, HW, ldc,
                                                                                                 (long long)C * HW)));
        return (int)hipGetLastError();
    }
    DISPATCH_T(dtype, (nchw_to_nhwc_kernel<T><<<grid_for((long long)B * HW * (ldc / P)), 256, 0, (hipStream_t)stream>>>(x, eps, musig, (T*)y,
                                                                                                                     B, C, HW, ldc)));
    return (int)hipGetLastError();
}

// used by c2w_window_gather (sampler.hip): image b starts img_stride floats after image b - 1 (windows of a trajectory overlap)
int c2w_planes_to_rows_strided(const float* x, void* y, int B, int C, int HW, int ldc, long long img_stride, int dtype, hipStream_t st) {
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (!vec_ok(dtype, ldc) || lds > 64 * 1024 || (img_stride & 3) != 0) return C2W_ERR_UNSUPPORTED;
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
    DISPATCH_T(dtype, (nchw_to_nhwc_tiled_kernel<T><<<nblk, 256, lds, st>>>(x, nullptr, nullptr, (T*)y, B, C, HW, ldc, img_stride)));
    return (int)hipGetLastError();
}

extern "C" int c2w_nhwc_to_nchw(const void* y, float* out, int B, int C, int HW, int ldc, int dtype, void* stream) {
    if (!y || !out || ldc < C) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (vec_ok(dtype, ldc) && lds <= 64 * 1024) {
        const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
        DISPATCH_T(dtype, (nhwc_to_nchw_tiled_kernel<T><<<nblk, 256, lds, (hipStream_t)stream>>>((const T*)y, out, B, C, HW, ldc)));
        return (int)hipGetLastError();
    }
    DISPATCH_T(dtype, (nhwc_to_nchw_kernel<T><<<grid_for((long long)B * C * HW), 256, 0, (hipStream_t)stream>>>((const T*)y, out, B, C, HW,
                                                                                                             ldc)));
    return (int)hipGetLastError();
}

extern "C" int c2w_mse_loss_grad_scaled(const void* y, const float* eps, void* dy, float* loss_sum, int B, int C, int HW, int ldc, float gscale,
                                        const float* scaler_state, int dtype, void* stream) {
    if (!y || !eps || !dy || !loss_sum || !vec_ok(dtype, ldc) || ldc < C) return C2W_ERR_BAD_SHAPE;
    const int P = dtype == C2W_DTYPE_F32 ? 4 : 8;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds <= 64 * 1024) {
        const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 2048);
        DISPATCH_T(dtype, (mse_loss_grad_tiled_kernel<T><<<nblk, 256, lds, (hipStream_t)stream>>>((const T*)y, eps, (T*)dy, loss_sum, B, C, HW,
                                                                                                  ldc, gscale, scaler_state)));
        return (int)hipGetLastError();
    }
    DISPATCH_T(dtype, (mse_loss_grad_kernel<T><<<grid_for((long long)B * HW * (ldc / P), 256, 2048), 256, 0, (hipStream_t)stream>>>(
                          (const T*)y, eps, (T*)dy, loss_sum, B, C, HW, ldc, gscale, scaler_state)));
    return (int)hipGetLastError();
}

extern "C" int c2w_mse_loss_grad(const void* y, const float* eps, void* dy, float* loss_sum, int B, int C, int HW, int ldc, float gscale,
                                 int dtype, void* stream) {
    return c2w_mse_loss_grad_scaled(y, eps, dy, loss_sum, B, C, HW, ldc, gscale, nullptr, dtype, stream);
}

// ---- regenerated noise (philox_normal4): the noise tensor of the training step never exists in memory
extern "C" int c2w_philox_normal(float* out, long long n, unsigned long long seed, void* stream) {
    if (!out || n <= 0 || ((uintptr_t)out & 15) != 0) return C2W_ERR_BAD_ARG;
    philox_normal_kernel<<<grid_for((n + 3) / 4), 256, 0, (hipStream_t)stream>>>(out, n, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (int)hipGetLastError();
}

extern "C" int c2w_nchw_to_nhwc_noise(const float* x, unsigned long long seed, const float* musig, void* y, int B, int C, int HW, int ldc,
                                      int dtype, void* stream) {
    if (!x || !y || !musig || !vec_ok(dtype, ldc) || ldc < C) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds > 64 * 1024) return C2W_ERR_UNSUPPORTED;  // caller materialises the stream with c2w_philox_normal instead
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
    DISPATCH_T(dtype, (nchw_to_nhwc_tiled_kernel<T, true><<<nblk, 256, lds, (hipStream_t)stream>>>(
                          x, nullptr, musig, (T*)y, B, C, HW, ldc, (long long)C * HW, (uint32_t)seed, (uint32_t)(seed >> 32))));
    return (int)hipGetLastError();
}

extern "C" int c2w_nchw_to_nhwc_noise_rows(const float* x, const long long* img_off, unsigned long long seed, const float* musig, void* y, void* erows,
                                           int B, int C, int HW, int ldc, int lde, int dtype, void* stream) {
    if (!x || !y || !erows || !musig || dtype == C2W_DTYPE_F32 || !vec_ok(dtype, ldc) || ldc < C || lde < C || (lde & 7) != 0) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)(ldc + lde) * LT_LD * sizeof(float);
    if (lds > 64 * 1024 || (HW & 3) != 0) return C2W_ERR_UNSUPPORTED;  // callers fall back to the regenerating pair
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
    if (dtype == C2W_DTYPE_BF16)
        nchw_to_nhwc_eps_kernel<bf16_t><<<nblk, 256, lds, (hipStream_t)stream>>>(x, musig, (bf16_t*)y, (f16_t*)erows, B, C, HW, ldc, lde, (long long)C * HW,
                                                                               (uint32_t)seed, (uint32_t)(seed >> 32), img_off);
    else
        nchw_to_nhwc_eps_kernel<f16_t><<<nblk, 256, lds, (hipStream_t)stream>>>(x, musig, (f16_t*)y, (f16_t*)erows, B, C, HW, ldc, lde, (long long)C * HW,
                                                                              (uint32_t)seed, (uint32_t)(seed >> 32), img_off);
    return (int)hipGetLastError();
}

extern "C" int c2w_windows_to_nhwc_noise(const float* data, const long long* img_off, unsigned long long seed, const float* musig, void* y,
                                         int B, int C, int HW, int ldc, int dtype, void* stream) {
    if (!data || !img_off || !y || !musig || !vec_ok(dtype, ldc) || ldc < C) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds > 64 * 1024) return C2W_ERR_UNSUPPORTED;  // caller gathers the windows and takes the dense path
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 65536);
    DISPATCH_T(dtype, (nchw_to_nhwc_tiled_kernel<T, true><<<nblk, 256, lds, (hipStream_t)stream>>>(
                          data, nullptr, musig, (T*)y, B, C, HW, ldc, (long long)C * HW, (uint32_t)seed, (uint32_t)(seed >> 32), img_off)));
    return (int)hipGetLastError();
}

extern "C" int c2w_mse_loss_grad_noise(const void* y, unsigned long long seed, void* dy, float* loss_sum, int B, int C, int HW, int ldc,
                                       float gscale, const float* scaler_state, int dtype, void* stream) {
    if (!y || !dy || !loss_sum || !vec_ok(dtype, ldc) || ldc < C) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds > 64 * 1024) return C2W_ERR_UNSUPPORTED;
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 2048);
    DISPATCH_T(dtype, (mse_loss_grad_tiled_kernel<T, true><<<nblk, 256, lds, (hipStream_t)stream>>>(
                          (const T*)y, nullptr, (T*)dy, loss_sum, B, C, HW, ldc, gscale, scaler_state, (uint32_t)seed, (uint32_t)(seed >> 32))));
    return (int)hipGetLastError();
}

static int sq_err_launch(const void* y, const float* eps, unsigned long long seed, bool regen, float* out, float* loss_sum, int B, int C, int HW,
                         int ldc, int dtype, void* stream) {
    if (!y || !out || (!regen && !eps) || !vec_ok(dtype, ldc) || ldc < C) return C2W_ERR_BAD_SHAPE;
    const size_t lds = (size_t)ldc * LT_LD * sizeof(float);
    if (lds > 64 * 1024 || (HW & 3) != 0 || ((uintptr_t)out & 15) != 0 || (eps && ((uintptr_t)eps & 15) != 0)) return C2W_ERR_UNSUPPORTED;
    const int nblk = (int)std::min<long long>((long long)B * ((HW + LT_PT - 1) / LT_PT), 4096);  // bounded: one atomic per wave
    if (regen) {
        DISPATCH_T(dtype, (sq_err_tiled_kernel<T, 2><<<nblk, 256, lds, (hipStream_t)stream>>>((const T*)y, nullptr, out, loss_sum, B, C, HW, ldc,
                                                                                              (uint32_t)seed, (uint32_t)(seed >> 32))));
    } else {
        DISPATCH_T(dtype, (sq_err_tiled_kernel<T, 1><<<nblk, 256, lds, (hipStream_t)stream>>>((const T*)y, eps, out, loss_sum, B, C, HW, ldc, 0u,
                                                                                              0u)));
    }
    return (int)hipGetLastError();
}

extern "C" int c2w_sq_err(const void* y, const float* eps, float* out, float* loss_sum, int B, int C, int HW, int ldc, int dtype, void* stream) {
    return sq_err_launch(y, eps, 0ull, false, out, loss_sum, B, C, HW, ldc, dtype, stream);
}

extern "C" int c2w_sq_err_noise(const void* y, unsigned long long seed, float* out, float* loss_sum, int B, int C, int HW, int ldc, int dtype,
                                void* stream) {
    return sq_err_launch(y, nullptr, seed, true, out, loss_sum, B, C, HW, ldc, dtype, stream);
}

// One-row Linear (model/score.py:56-57,62-67 and the modulation projections model/nn.py:149 when t is ONE value for the whole batch --
// every network call of the sampler): y[r] = act(b[r] + W[r][:K] . x).  As a GEMM that is a 16-pixel MFMA tile with one live column on
// 4 workgroups walking K (30 us per layer, three layers in a row); as a matrix-vector product it is the weight matrix read once: one
// wave per output row, coalesced 16-byte loads, a shuffle reduction.
__global__ __launch_bounds__(256) void gemv_f32_kernel(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                                       float* __restrict__ y, int rows, int K, int ldk, int act) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* wr = W + (size_t)r * ldk;
    float s = 0.f;
    if ((K & 3) == 0 && (ldk & 3) == 0) {
        for (int k = lane * 4; k < K; k += 256) {
            const float4 a = *(const float4*)(wr + k), b = *(const float4*)(x + k);
            s = fmaf(a.x, b.x, s);
            s = fmaf(a.y, b.y, s);
            s = fmaf(a.z, b.z, s);
            s = fmaf(a.w, b.w, s);
        }
    } else {
        for (int k = lane; k < K; k += 64) s = fmaf(wr[k], x[k], s);
    }
    s = wave_sum(s);
    if (lane == 0) {
        s += bias != nullptr ? bias[r] : 0.f;
        y[r] = act == C2W_ACT_SILU ? silu_f(s) : (act == C2W_ACT_RELU ? fmaxf(s, 0.f) : s);
    }
}

extern "C" int c2w_gemv_f32(const float* x, const float* W, const float* bias, float* y, int rows, int K, int ldk, int act, void* stream) {
    if (!x || !W || !y || rows <= 0 || K <= 0 || ldk < K) return C2W_ERR_BAD_SHAPE;
    if (act != C2W_ACT_NONE && act != C2W_ACT_SILU && act != C2W_ACT_RELU) return C2W_ERR_BAD_ARG;
    gemv_f32_kernel<<<(rows + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, W, bias, y, rows, K, ldk, act);
    return (int)hipGetLastError();
}

// A device scalar made readable by the host without a stream synchronisation (training_loop.py:385 reads the loss back every step):
// one thread copies the value into a slot of host memory (pinned, device-visible) and then stores the sequence number the host polls
// for -- release at system scope, so the value is there when the number is.
__global__ void publish_scalar_kernel(const float* __restrict__ src, int* slot, int seq) {
    const float v = *src;
    __hip_atomic_store(slot, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(slot + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int c2w_publish_scalar(const float* src, int* host_slot, int seq, void* stream) {
    if (!src || !host_slot) return C2W_ERR_BAD_ARG;
    publish_scalar_kernel<<<1, 1, 0, (hipStream_t)stream>>>(src, host_slot, seq);
    return (int)hipGetLastError();
}

extern "C" int c2w_timestep_embedding(const float* t, float* out, int n, int dim, float max_period, void* stream) {
    if (!t || !out || n <= 0 || dim <= 0) return C2W_ERR_BAD_SHAPE;
    timestep_embedding_kernel<<<(n * dim + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, out, n, dim, max_period);
    return (int)hipGetLastError();
}

extern "C" int c2w_mu_sigma(const float* t, float* musig, int n, float eta, void* stream) {
    if (!t || !musig || n <= 0) return C2W_ERR_BAD_SHAPE;
    mu_sigma_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(t, musig, n, eta);
    return (int)hipGetLastError();
}

extern "C" int c2w_cast_f32(const float* in, void* out, long long n, int dtype, void* stream) {
    if (!in || !out) return C2W_ERR_BAD_ARG;
    DISPATCH_T(dtype, (cast_f32_kernel<T><<<grid_for(n), 256, 0, (hipStream_t)stream>>>(in, (T*)out, n)));
    return (int)hipGetLastError();
}

extern "C" int c2w_weight_transpose(const float* w, void* out, int R, int NT, int K, int ldk, int ldr, int flip, int dtype, void* stream) {
    if (!w || !out || R <= 0 || NT <= 0 || K <= 0 || ldk < K || ldr < R) return C2W_ERR_BAD_SHAPE;
    dim3 grid((K + 31) / 32, (R + 31) / 32, NT);
    DISPATCH_T(dtype, (weight_transpose_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(w, (T*)out, R, NT, K, ldk, ldr, flip)));
    return (int)hipGetLastError();
}

extern "C" int c2w_weight_transpose_batched(const float* flat, void* out, const long long* desc, int nconv, int dtype, void* stream) {
    if (!flat || !out || !desc || nconv <= 0) return C2W_ERR_BAD_ARG;
    dim3 grid(256, nconv);
    DISPATCH_T(dtype, (weight_transpose_batched_kernel<T><<<grid, 256, 0, (hipStream_t)stream>>>(flat, (T*)out, desc)));
    return (int)hipGetLastError();
}

extern "C" int c2w_pack_conv_weights_batched(const void* src, void* dst, const long long* desc, int n, int dtype, void* stream) {
    if (!src || !dst || !desc || n <= 0 || (dtype != C2W_DTYPE_BF16 && dtype != C2W_DTYPE_F16)) return C2W_ERR_BAD_ARG;
    dim3 grid(128, n);
    pack_conv_weights_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const uint16_t*)src, (uint16_t*)dst, desc);
    return (int)hipGetLastError();
}

extern "C" int c2w_adamw_ema_scaled(float* p, const float* g, float* m, float* v, float* ema, void* shadow, int shadow_dtype, long long n,
                                    float lr, float beta1, float beta2, float eps, float weight_decay, int step, float ema_rate,
                                    float grad_scale, const float* scaler_state, void* stream) {
    if (!p || !g || !m || !v || step < 1) return C2W_ERR_BAD_ARG;
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    if (shadow != nullptr && shadow_dtype == C2W_DTYPE_F16)
        adamw_ema_kernel<f16_t><<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, ema, (f16_t*)shadow, n, lr, beta1, beta2, eps,
                                                                              weight_decay, bc1, bc2_sqrt, ema_rate, grad_scale, scaler_state);
    else if (shadow == nullptr || shadow_dtype == C2W_DTYPE_BF16)
        adamw_ema_kernel<bf16_t><<<grid_for(n), 256, 0, (hipStream_t)stream>>>(p, g, m, v, ema, (bf16_t*)shadow, n, lr, beta1, beta2, eps,
                                                                               weight_decay, bc1, bc2_sqrt, ema_rate, grad_scale, scaler_state);
    else
        return C2W_ERR_BAD_ARG;
    return (int)hipGetLastError();
}

extern "C" int c2w_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow_bf16, long long n, float lr, float beta1,
                             float beta2, float eps, float weight_decay, int step, float ema_rate, float grad_scale, void* stream) {
    return c2w_adamw_ema_scaled(p, g, m, v, ema, shadow_bf16, C2W_DTYPE_BF16, n, lr, beta1, beta2, eps, weight_decay, step, ema_rate,
                                grad_scale, nullptr, stream);
}

extern "C" int c2w_grad_scaler_init(float* state, float init_scale, void* stream) {
    if (!state || !(init_scale > 0.f)) return C2W_ERR_BAD_ARG;
    scaler_init_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, init_scale);
    return (int)hipGetLastError();
}

extern "C" int c2w_grad_scaler_check(const float* g, long long n, float* state, void* stream) {
    if (!g || !state || n <= 0 || ((uintptr_t)g & 15) != 0) return C2W_ERR_BAD_ARG;
    scaler_check_kernel<<<grid_for(n / 4 + 1, 256, 4096), 256, 0, (hipStream_t)stream>>>(g, n, state);
    return (int)hipGetLastError();
}

extern "C" int c2w_grad_scaler_update(float* state, float growth, float backoff, int growth_interval, void* stream) {
    if (!state || growth_interval < 1) return C2W_ERR_BAD_ARG;
    scaler_update_kernel<<<1, 64, 0, (hipStream_t)stream>>>(state, growth, backoff, growth_interval);
    return (int)hipGetLastError();
}

extern "C" int c2w_ema_update(float* ema, const float* p, long long n, float rate, void* stream) {
    if (!ema || !p) return C2W_ERR_BAD_ARG;
    ema_kernel<<<grid_for(n), 256, 0, (hipStream_t)stream>>>(ema, p, n, rate);
    return (int)hipGetLastError();
}

extern "C" const char* c2w_target(void) { return "gfx950"; }
