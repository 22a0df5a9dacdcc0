// Shared epilogue of the forward/dgrad implicit-GEMM kernels: accumulators (+bias, activation) -> LDS rows
// [pixel][channel] -> coalesced 16-B NHWC stores with the fused multiplier / residual / second output.
//
// Measured with in-kernel stamps (tools/stamp_conv_patch.py) the first version of this code took 10.5k of a tile's 37.9k
// cycles: bias values were fetched in 4 dependent round trips and every 16-B store waited for its own residual load.
// Now the bias is loaded before the main loop, and the store pass is unrolled in groups of 8 segments per thread with
// all of a group's global loads issued before the first use.
#pragma once
#include "conv_geom.h"

// bias of the 16 output channels this lane's accumulator rows cover (wave tile 64 co: 4 m-tiles x rows 4*lg..4*lg+3)
__device__ __forceinline__ void epi_load_bias(const C2wConvArgs& p, int co_base, float (&bv)[4][4]) {
    const bool has = p.bias != nullptr;
    const float* bp = has ? p.bias : (const float*)p.w;  // never dereferenced out of range: index clamped, value masked
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co_base + m * 16 + r;
            const int idx = (has && co < p.wrows) ? co : 0;
            const float v = has ? bp[idx] : 0.f;
            bv[m][r] = (has && co < p.wrows) ? v : 0.f;
        }
}

// one wave's 64 (co) x 64 (pixel) accumulator tile -> LDS rows; row0 = first pixel row of the wave inside O
template <typename T>
__device__ __forceinline__ void epi_acc_to_lds(char* O, int OS, const f32x4_t (&acc)[4][4], const float (&bv)[4][4], int act, int col0, int row0,
                                               int li, int lg) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int col = col0 + m * 16 + lg * 4;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int row = row0 + n * 16 + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[m][n][r] + bv[m][r];
                if (act == C2W_ACT_SILU) v[r] = silu_f(v[r]);
            }
            if constexpr (sizeof(T) == 4) {
                *(f32x4_t*)(O + row * OS + col * 4) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {
                *(u32x2_t*)(O + row * OS + col * 2) = (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
}

// LDS rows [NROWS][128 channels] -> global.  pix(row) maps a tile row to the NHWC pixel index, or -1 if the row is outside.
// Split in two so that the residual / multiplier loads fly while the accumulators are staged through LDS:
//   EpiStore st; st.prefetch(...);  __syncthreads(); epi_acc_to_lds(...); __syncthreads();  st.finish(...);
template <typename T, int NROWS, int NTHR>
struct EpiStore {
    static constexpr int ESZ = sizeof(T);
    static constexpr int SEGS = 128 * ESZ / 16;
    static constexpr int PER16 = 16 / ESZ;
    static constexpr int NIT = NROWS * SEGS / NTHR;
    static constexpr bool EARLY = NIT <= 8;  // bf16: 8 segments per thread fit in registers next to the accumulators
    long long off[NIT];
    u32x4_t rr[EARLY ? NIT : 1], mm[EARLY ? NIT : 1];

    template <typename PixFn>
    __device__ __forceinline__ void prefetch(const C2wConvArgs& p, int tid, int co0, PixFn pix) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int seg = tid + i * NTHR;
            const int row = seg / SEGS, cs = seg - row * SEGS;
            const int c = co0 + cs * PER16;
            const long long Q = pix(row);
            off[i] = (Q >= 0 && c < p.Cout) ? (long long)((Q * p.ldy + c) * ESZ) : -1;
        }
        if constexpr (EARLY) {
            if (p.res != nullptr) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) rr[i] = *(const u32x4_t*)((const char*)p.res + (off[i] >= 0 ? off[i] : 0));
            }
            if (p.mul != nullptr) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) mm[i] = *(const u32x4_t*)((const char*)p.mul + (off[i] >= 0 ? off[i] : 0));
            }
        }
    }

    __device__ __forceinline__ void finish(const C2wConvArgs& p, const char* O, int OS, int tid) {
        constexpr int GRP = NIT < 8 ? NIT : 8;
#pragma unroll
        for (int g0 = 0; g0 < NIT; g0 += GRP) {
            u32x4_t v[GRP], r2[GRP], m2[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const int seg = tid + (g0 + i) * NTHR;
                const int row = seg / SEGS, cs = seg - row * SEGS;
                v[i] = *(const u32x4_t*)(O + row * OS + cs * 16);
            }
            if constexpr (!EARLY) {
                if (p.res != nullptr) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) r2[i] = *(const u32x4_t*)((const char*)p.res + (off[g0 + i] >= 0 ? off[g0 + i] : 0));
                }
                if (p.mul != nullptr) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) m2[i] = *(const u32x4_t*)((const char*)p.mul + (off[g0 + i] >= 0 ? off[g0 + i] : 0));
                }
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const long long o = off[g0 + i];
                if (p.mul != nullptr || p.res != nullptr) {
                    float f[PER16];
                    unpack16<T>(v[i], f);
                    if (p.mul != nullptr) {
                        float gm[PER16];
                        unpack16<T>(EARLY ? mm[EARLY ? g0 + i : 0] : m2[i], gm);
#pragma unroll
                        for (int e = 0; e < PER16; ++e) f[e] *= (p.mulmode == C2W_MUL_DSILU) ? dsilu_f(gm[e]) : gm[e];
                    }
                    if (p.res != nullptr) {
                        float gr[PER16];
                        unpack16<T>(EARLY ? rr[EARLY ? g0 + i : 0] : r2[i], gr);
#pragma unroll
                        for (int e = 0; e < PER16; ++e) f[e] += gr[e];
                    }
                    v[i] = pack16<T>(f);
                }
                if (o >= 0) {
                    *(u32x4_t*)((char*)p.y + o) = v[i];
                    if (p.y2 != nullptr) {  // second output: silu of the stored value (training keeps pre-activation and activation)
                        float f2[PER16];
                        unpack16<T>(v[i], f2);
#pragma unroll
                        for (int e = 0; e < PER16; ++e) f2[e] = silu_f(f2[e]);
                        *(u32x4_t*)((char*)p.y2 + o) = pack16<T>(f2);
                    }
                }
            }
        }
    }
};
