// Shared epilogue of the forward/dgrad implicit-GEMM kernels: accumulators (+bias, activation) -> LDS rows
// [pixel][channel] -> coalesced 16-B NHWC stores with the fused multiplier / residual / second output.
//
// Measured with in-kernel stamps (tools/stamp_conv_patch.py) the first version of this code took 10.5k of a tile's 37.9k
// cycles: bias values were fetched in 4 dependent round trips and every 16-B store waited for its own residual load.
// Now the bias is loaded before the main loop, and the store pass is unrolled in groups of 8 segments per thread with
// all of a group's global loads issued before the first use.
#pragma once
#include "conv_geom.h"

// the epilogues' 16-B output stores carry the non-temporal hint (written once, read by a later kernel, never by this one): 1-2 % on
// isolated launches of every flavour, 0.2 % on the step (profiles/r02_ab_conv_epilogues.txt)
__device__ __forceinline__ void epi_st(char* ptr, const u32x4_t& v) { __builtin_nontemporal_store(v, (u32x4_t*)ptr); }


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

// one wave's 64 (co) x 64 (pixel) accumulator tile -> LDS rows; row0 = first pixel row of the wave inside O.
// SILU is a template parameter: with a run-time `act` the compiler evaluated the 64 exp/rcp pairs per lane for every
// conv and selected afterwards (stamps: 3.3k of a 27k-cycle tile, whether or not the activation was requested).
template <typename T, int ACTK>  // 0 none, 1 SiLU, 2 ReLU
__device__ __forceinline__ void epi_acc_to_lds_impl(char* O, int OS, const f32x4_t (&acc)[4][4], const float (&bv)[4][4], int col0, int row0,
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
                if constexpr (ACTK == 1) v[r] = silu_f(v[r]);
                if constexpr (ACTK == 2) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (sizeof(T) == 4) {
                *(f32x4_t*)(O + row * OS + col * 4) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {
                *(u32x2_t*)(O + row * OS + col * 2) = (u32x2_t){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
            }
        }
    }
}

template <typename T>
__device__ __forceinline__ void epi_acc_to_lds(char* O, int OS, const f32x4_t (&acc)[4][4], const float (&bv)[4][4], int act, int col0, int row0,
                                               int li, int lg) {
    if (act == C2W_ACT_SILU) {  // wave-uniform branch
        epi_acc_to_lds_impl<T, 1>(O, OS, acc, bv, col0, row0, li, lg);
    } else if (act == C2W_ACT_RELU) {
        epi_acc_to_lds_impl<T, 2>(O, OS, acc, bv, col0, row0, li, lg);
    } else {
        epi_acc_to_lds_impl<T, 0>(O, OS, acc, bv, col0, row0, li, lg);
    }
}

// The same for a wave tile of 64 (co) x 16 N (pixel) accumulators (conv_patch_half8_kernel: eight waves, two pixel rows each).
template <typename T, int N>
__device__ __forceinline__ void epi_acc_to_lds_n(char* O, int OS, const f32x4_t (&acc)[4][N], const float (&bv)[4][4], int act, int col0, int row0,
                                                 int li, int lg) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int col = col0 + m * 16 + lg * 4;
#pragma unroll
        for (int n = 0; n < N; ++n) {
            const int row = row0 + n * 16 + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[m][n][r] + bv[m][r];
                if (act == C2W_ACT_SILU) v[r] = silu_f(v[r]);  // (wave-uniform; 32 values per lane here, not 64: no separate instantiations)
                if (act == C2W_ACT_RELU) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (sizeof(T) == 4) {
                *(f32x4_t*)(O + row * OS + col * 4) = (f32x4_t){v[0], v[1], v[2], v[3]};
            } else {
                *(u32x2_t*)(O + row * OS + col * 2) = (u32x2_t){pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3])};
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
    // tiles that lie inside one image (prefetch_tile16*): only the channel bound can mask a thread, so all of its segments share ONE
    // validity -- a bool tested once instead of a 64-bit compare + exec-mask branch in front of every store (`hipcc -S`, round 3)
    bool same_valid = false, valid0 = false;
    __device__ __forceinline__ bool ok(int i) const { return same_valid ? valid0 : off[i] >= 0; }
    // LDS address of the thread's i-th segment: row = tid / SEGS + i * (NTHR / SEGS) -- unsigned, so that the i-dependent part is an
    // immediate offset of the ds_read (the signed seg / SEGS cost eight VALU operations per segment)
    static __device__ __forceinline__ const char* lds_seg(const char* O, int OS, int tid, int i) {
        const unsigned r0 = (unsigned)tid / SEGS, cs = (unsigned)tid % SEGS;
        return O + (r0 + (unsigned)i * (NTHR / SEGS)) * (unsigned)OS + cs * 16u;
    }

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
        issue_prefetch(p);
    }

    // Same for a tile that is 16 pixels wide inside one image (halo-patch kernels): pix0 = NHWC index of the tile's first
    // pixel, W = image width.  Every row of the tile is inside the image, so only the channel bound can mask a thread, and
    // the 8 (16) offsets differ by wave-uniform constants: one 64-bit multiply per thread instead of one per segment.
    __device__ __forceinline__ void prefetch_tile16(const C2wConvArgs& p, int tid, int co0, long long pix0, int W) {
        constexpr int RS = NTHR / SEGS;  // tile rows between two consecutive segments of a thread
        static_assert(RS % 16 == 0 || 16 % RS == 0, "row step and tile width must nest");
        const int r0 = tid / SEGS, cs = tid - r0 * SEGS;
        const int c = co0 + cs * PER16;
        const long long pitch = (long long)p.ldy * ESZ;
        const long long off0 = c < p.Cout ? ((pix0 + (long long)(r0 >> 4) * W + (r0 & 15)) * p.ldy + c) * ESZ : -1;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const long long d = (long long)((RS * i) >> 4) * W * pitch + (long long)((RS * i) & 15) * pitch;
            off[i] = off0 >= 0 ? off0 + d : -1;
        }
        same_valid = true;
        valid0 = off0 >= 0;
        issue_prefetch(p);
    }

    // 16-pixel-wide tile whose pixels are stored with stride 2 in both directions (one output-parity class of the stride-2
    // input gradient): pix0 = NHWC index of the tile's first OUTPUT pixel, Wo = output image width.
    __device__ __forceinline__ void prefetch_tile16_s2(const C2wConvArgs& p, int tid, int co0, long long pix0, int Wo) {
        constexpr int RS = NTHR / SEGS;
        static_assert(RS % 16 == 0 || 16 % RS == 0, "row step and tile width must nest");
        const int r0 = tid / SEGS, cs = tid - r0 * SEGS;
        const int c = co0 + cs * PER16;
        const long long pitch = (long long)p.ldy * ESZ;
        const long long off0 = c < p.Cout ? ((pix0 + (long long)(r0 >> 4) * 2 * Wo + (r0 & 15) * 2) * p.ldy + c) * ESZ : -1;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const long long d = (long long)((RS * i) >> 4) * 2 * Wo * pitch + (long long)((RS * i) & 15) * 2 * pitch;
            off[i] = off0 >= 0 ? off0 + d : -1;
        }
        same_valid = true;
        valid0 = off0 >= 0;
        issue_prefetch(p);
    }

    // Tile of conv_patch_half_kernel<T, PAIR>: LDS row R = 16 * tile row + column; columns 0..7 are image `pixA`'s row, 8..15 the
    // same row of the next image (HW pixels further), which is missing when nimg == 1.
    __device__ __forceinline__ void prefetch_pair8(const C2wConvArgs& p, int tid, int co0, long long pixA, int HW, int nimg) {
        constexpr int RS = NTHR / SEGS;
        const int r0 = tid / SEGS, cs = tid - r0 * SEGS;
        const int c = co0 + cs * PER16;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int R = r0 + RS * i, trow = R >> 4, col = R & 15, img = col >> 3;
            const long long pix = pixA + (long long)img * HW + trow * 8 + (col & 7);
            off[i] = (c < p.Cout && img < nimg) ? (pix * p.ldy + c) * ESZ : -1;
        }
        issue_prefetch(p);
    }

    __device__ __forceinline__ void issue_prefetch(const C2wConvArgs& p) {
        if constexpr (EARLY) {
            if (p.res != nullptr) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) rr[i] = *(const u32x4_t*)((const char*)p.res + (ok(i) ? off[i] : 0));
            }
            const void* const mulp = p.ln_x != nullptr ? p.ln_x : p.mul;  // LN mode: the multiplier slot carries the LN input rows
            if (mulp != nullptr) {
#pragma unroll
                for (int i = 0; i < NIT; ++i) mm[i] = *(const u32x4_t*)((const char*)mulp + (ok(i) ? off[i] : 0));
            }
        }
    }

    // ---- fused LayerNorm backward (bf16, the tile holds whole 128-channel rows, all of one image `img`):
    //   O rows = g (input gradient of conv1 = gradient w.r.t. the LN output);  u = ln_x + ln_m[img];  xh = (u - mean)/s
    //   y = res + ( g - mean(g) - xh * sum(g*xh)/den ) / s ;   ln_dm[img][c] += column sums of the LN part
    // Same arithmetic, in the same order, as ln_bwd_kernel (pointwise.hip): 16 lanes share a pixel row.
    // `red`: 128 floats of LDS outside O, zeroed by the caller before the barrier that precedes this call.
    typedef __attribute__((ext_vector_type(2))) float ln_f2;  // pairs -> v_pk_{add,mul,fma}_f32: half the VALU issue slots
    struct LnColSums {  // per-thread column sums of the LayerNorm part: the thread's 8 channels over the rows it has finished
        ln_f2 am[4];
        __device__ __forceinline__ void clear() {
#pragma unroll
            for (int k = 0; k < 4; ++k) am[k] = (ln_f2){0.f, 0.f};
        }
    };
    // one call = one pass of NROWS tile rows and the modulation-gradient reduction behind it (tiles of one pass)
    __device__ __forceinline__ void finish_ln(const C2wConvArgs& p, const char* O, int OS, int tid, int img, float* red) {
        LnColSums cs;
        cs.clear();
        if (p.ln_rstd != nullptr) finish_ln_rows<true>(p, O, OS, tid, img, cs);  // kernel argument: uniform
        else finish_ln_rows<false>(p, O, OS, tid, img, cs);
        finish_ln_dm(p, tid, img, red, cs);
    }
    // the rows of one pass; the column sums are carried in `acc` (a tile of several passes reduces them once: finish_ln_dm).
    // STORED: the forward kept the normalised rows and their 1/sigma (C2wConvArgs.ln_rstd) -- a template parameter, because the 16x16-tile
    // kernel has no registers for both forms in one instantiation (with a uniform branch its LayerNorm-backward kernels spilled 24 registers)
    template <bool STORED>
    __device__ __forceinline__ void finish_ln_rows(const C2wConvArgs& p, const char* O, int OS, int tid, int img, LnColSums& acc) {
        static_assert(EARLY && SEGS == 16 && PER16 == 8, "fused LN backward: 16-bit tiles only");
        typedef ln_f2 f2;
        const int cs = tid & (SEGS - 1);
        if constexpr (STORED) {
            finish_ln_rows_stored(p, O, OS, tid, acc);
            return;
        }
        f2 m2[4];
        if (p.ln_m != nullptr) {
            const float* mr = p.ln_m + (size_t)(p.ln_ldm ? img : 0) * p.ln_ldm + cs * PER16;
            const f32x4_t ma = *(const f32x4_t*)mr, mb = *(const f32x4_t*)(mr + 4);
            m2[0] = (f2){ma[0], ma[1]}; m2[1] = (f2){ma[2], ma[3]}; m2[2] = (f2){mb[0], mb[1]}; m2[3] = (f2){mb[2], mb[3]};
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) m2[k] = (f2){0.f, 0.f};
        }
        const float inv_den = 1.0f / (float)(128 - (p.ln_unbiased ? 1 : 0));
        f2 (&am)[4] = acc.am;
        auto unpack2 = [](const u32x4_t& v, f2* f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo, hi;
                ::unpack2<T>(v[k], lo, hi);
                f[k] = (f2){lo, hi};
            }
        };
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            f2 g[4], u[4];
            unpack2(*(const u32x4_t*)lds_seg(O, OS, tid, i), g);
            unpack2(mm[i], u);
            f2 s2 = (f2){0.f, 0.f}, sg2 = (f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] += m2[k];
                s2 += u[k];
                sg2 += g[k];
            }
            const float mean = sub16(s2[0] + s2[1]) * (1.0f / 128.0f);
            const float gmean = sub16(sg2[0] + sg2[1]) * (1.0f / 128.0f);
            f2 q2 = (f2){0.f, 0.f}, d2 = (f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] -= mean;            // centred input; xhat = u * rs
                q2 += u[k] * u[k];
                d2 += g[k] * u[k];
            }
            const float rs = __builtin_amdgcn_rsqf(sub16(q2[0] + q2[1]) * inv_den + p.ln_eps);
            const float dot = sub16(d2[0] + d2[1]) * rs * inv_den;  // sum(g * xhat) / den
            // (g - gmean - xhat * dot) * rs  =  g * rs - gmean * rs - u * (rs * rs * dot)
            const float c0 = gmean * rs, c2 = rs * rs * dot;
            f2 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = g[k] * rs - c0;
                o[k] -= u[k] * c2;
                am[k] += o[k];
            }
            if (p.res != nullptr) {
                f2 r[4];
                unpack2(rr[i], r);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += r[k];
            }
            u32x4_t out;
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = pack2<T>(o[k][0], o[k][1]);
            if (ok(i)) epi_st((char*)p.y + off[i], out);
        }
    }
    // mm[] = the normalised rows the forward kept, ln_rstd their 1/sigma:  rs * (g - mean(g) - xhat * sum(g * xhat) / den)
    __device__ __forceinline__ void finish_ln_rows_stored(const C2wConvArgs& p, const char* O, int OS, int tid, LnColSums& acc) {
        typedef ln_f2 f2;
        const float inv_den = 1.0f / (float)(128 - (p.ln_unbiased ? 1 : 0));
        f2 (&am)[4] = acc.am;
        auto unpack2 = [](const u32x4_t& v, f2* f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo, hi;
                ::unpack2<T>(v[k], lo, hi);
                f[k] = (f2){lo, hi};
            }
        };
        float rsv[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) rsv[i] = p.ln_rstd[ok(i) ? off[i] >> 8 : 0];  // 16 lanes share a pixel row (256 B): one address
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            f2 g[4], xh[4];
            unpack2(*(const u32x4_t*)lds_seg(O, OS, tid, i), g);
            unpack2(mm[i], xh);
            f2 sg2 = (f2){0.f, 0.f}, d2 = (f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                sg2 += g[k];
                d2 += g[k] * xh[k];
            }
            const float rs = rsv[i];
            const float c0 = sub16(sg2[0] + sg2[1]) * (1.0f / 128.0f) * rs;  // mean(g) / sigma
            const float c2 = sub16(d2[0] + d2[1]) * inv_den * rs;             // sum(g * xhat) / den / sigma
            f2 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = g[k] * rs - c0;
                o[k] -= xh[k] * c2;
                am[k] += o[k];
            }
            if (p.res != nullptr) {
                f2 r[4];
                unpack2(rr[i], r);
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] += r[k];
            }
            u32x4_t out;
#pragma unroll
            for (int k = 0; k < 4; ++k) out[k] = pack2<T>(o[k][0], o[k][1]);
            if (ok(i)) epi_st((char*)p.y + off[i], out);
        }
    }
    // column sums -> ln_dm[img]: across the four pixel rows of a wave in registers (lanes l, l+16, l+32, l+48 hold the same channels),
    // across the waves through 128 floats of LDS (`red`, zeroed by the caller before the barrier that precedes the first pass), then
    // one global atomic per channel.  Every thread of the workgroup must call it (barrier inside).
    __device__ __forceinline__ void finish_ln_dm(const C2wConvArgs& p, int tid, int img, float* red, const LnColSums& acc) {
        if (p.ln_dm == nullptr) return;  // kernel argument: uniform
        const int cs = tid & (SEGS - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v = acc.am[k][h];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                if ((tid & 63) < 16) atomicAdd(&red[cs * PER16 + 2 * k + h], v);
            }
        __syncthreads();
        if (tid < 128) atomicAdd(p.ln_dm + (size_t)(p.ln_ldm ? img : 0) * p.ln_ldm + tid, red[tid]);
    }

    // ---- fused LayerNorm FORWARD of the consumer (bf16 tile with whole 128-channel rows of one image `img`):
    //   y = O (+ res), stored; lnf_y = LN_C(y_as_stored + lnf_m[img])  -- the arithmetic of ln_fwd_kernel (two-pass variance)
    // Round 6 (C2wConvArgs.res_rstd / res_mean / res_m, C2W_CONV_NO_Y, lnf_mean):
    //   RESN: `res` holds the NORMALISED rows h = LN(x + res_m) an earlier launch emitted, with that LayerNorm's per-pixel statistics:
    //         the residual is rebuilt as x = h / rstd + mean - res_m[img] (the block input itself was never written);
    //   C2W_CONV_NO_Y: y is not stored (its only readers are this LayerNorm and the next block's residual, which rebuilds it the same
    //         way from lnf_y, lnf_rstd and lnf_mean); the LayerNorm then sees the fp32 sum instead of its 16-bit rounding.
    template <bool RESN = false>
    __device__ __forceinline__ void finish_lnf(const C2wConvArgs& p, const char* O, int OS, int tid, int img) {
        static_assert(EARLY && SEGS == 16 && PER16 == 8, "fused LN forward: 16-bit tiles only");
        typedef __attribute__((ext_vector_type(2))) float f2;
        const int cs = tid & (SEGS - 1);
        const bool store_y = (p.flags & C2W_CONV_NO_Y) == 0;  // kernel argument: uniform
        f2 m2[4];
        if (p.lnf_m != nullptr) {
            const float* mr = p.lnf_m + (size_t)(p.ln_ldm ? img : 0) * p.ln_ldm + cs * PER16;
            const f32x4_t ma = *(const f32x4_t*)mr, mb = *(const f32x4_t*)(mr + 4);
            m2[0] = (f2){ma[0], ma[1]}; m2[1] = (f2){ma[2], ma[3]}; m2[2] = (f2){mb[0], mb[1]}; m2[3] = (f2){mb[2], mb[3]};
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) m2[k] = (f2){0.f, 0.f};
        }
        f2 rm2[4];
        float rsig[RESN ? NIT : 1], rmean[RESN ? NIT : 1];
        if constexpr (RESN) {
            if (p.res_m != nullptr) {
                const float* mr = p.res_m + (size_t)(p.ln_ldm ? img : 0) * p.ln_ldm + cs * PER16;
                const f32x4_t ma = *(const f32x4_t*)mr, mb = *(const f32x4_t*)(mr + 4);
                rm2[0] = (f2){ma[0], ma[1]}; rm2[1] = (f2){ma[2], ma[3]}; rm2[2] = (f2){mb[0], mb[1]}; rm2[3] = (f2){mb[2], mb[3]};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) rm2[k] = (f2){0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) {  // 16 lanes share a pixel row (256 B): one address each
                rsig[i] = p.res_rstd[ok(i) ? off[i] >> 8 : 0];
                rmean[i] = p.res_mean[ok(i) ? off[i] >> 8 : 0];
            }
#pragma unroll
            for (int i = 0; i < NIT; ++i) rsig[i] = __builtin_amdgcn_rcpf(rsig[i]);  // sigma
        }
        const float inv_den = 1.0f / (float)(128 - (p.ln_unbiased ? 1 : 0));
        auto unpack2 = [](const u32x4_t& v, f2* f) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo, hi;
                ::unpack2<T>(v[k], lo, hi);
                f[k] = (f2){lo, hi};
            }
        };
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            u32x4_t out = *(const u32x4_t*)lds_seg(O, OS, tid, i);
            f2 u[4];
            unpack2(out, u);
            if (RESN || p.res != nullptr) {
                f2 r[4];
                unpack2(rr[i], r);
                if constexpr (RESN) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = r[k] * rsig[i] + (rmean[i] - rm2[k]);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) u[k] += r[k];
                if (store_y) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) out[k] = pack2<T>(u[k][0], u[k][1]);
                }
            }
            if (store_y) {
                if (ok(i)) epi_st((char*)p.y + off[i], out);
                unpack2(out, u);  // the values as stored (bf16), like the separate LN pass would read them
            }
            f2 s2 = (f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] += m2[k];
                s2 += u[k];
            }
            const float mean = sub16(s2[0] + s2[1]) * (1.0f / 128.0f);
            f2 q2 = (f2){0.f, 0.f};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] -= mean;
                q2 += u[k] * u[k];
            }
            const float rs = __builtin_amdgcn_rsqf(sub16(q2[0] + q2[1]) * inv_den + p.ln_eps);
            u32x4_t ln;
#pragma unroll
            for (int k = 0; k < 4; ++k) ln[k] = pack2<T>(u[k][0] * rs, u[k][1] * rs);
            if (ok(i)) epi_st((char*)p.lnf_y + off[i], ln);
            // the row's 1/sigma (and mean) for the backward / the next block's residual: one lane per pixel row (rows are 256 B:
            // pixel = byte offset >> 8)
            if (cs == 0 && ok(i)) {
                if (p.lnf_rstd != nullptr) p.lnf_rstd[off[i] >> 8] = rs;
                if (p.lnf_mean != nullptr) p.lnf_mean[off[i] >> 8] = mean;
            }
        }
    }

    // ---- 2x2 sum pooling of a pass of 128 tile pixels (8 tile rows x 16 columns in LDS rows R = 16 * row + col) -> 4 x 8 pooled
    // pixels: the adjoint of Upsample(nearest, x2) (model/nn.py:184) on the input gradient of the up-conv.  Each thread adds the
    // four rows of a pooled pixel in the order c2w_sumpool2 does ((g00 + g01) + g10) + g11, in fp32, from the values as they
    // would have been stored.  ppix0 = NHWC index of the pass's first POOLED pixel, Wp = pooled image width.
    __device__ __forceinline__ void finish_pool2(const C2wConvArgs& p, const char* O, int OS, int tid, int co0, long long ppix0, int Wp) {
        static_assert(NROWS == 128, "one pass = 8 tile rows");
        constexpr int ITEMS = 32 * SEGS;
#pragma unroll
        for (int i0 = 0; i0 < ITEMS; i0 += NTHR) {
            const int idx = i0 + tid;
            if (ITEMS % NTHR != 0 && idx >= ITEMS) break;
            const int pp = idx / SEGS, cs = idx - pp * SEGS;
            const int pr = pp >> 3, pc = pp & 7;
            const int r00 = (2 * pr) * 16 + 2 * pc;
            float s[PER16], f[PER16];
            unpack16<T>(*(const u32x4_t*)(O + r00 * OS + cs * 16), s);
            unpack16<T>(*(const u32x4_t*)(O + (r00 + 1) * OS + cs * 16), f);
#pragma unroll
            for (int e = 0; e < PER16; ++e) s[e] += f[e];
            unpack16<T>(*(const u32x4_t*)(O + (r00 + 16) * OS + cs * 16), f);
#pragma unroll
            for (int e = 0; e < PER16; ++e) s[e] += f[e];
            unpack16<T>(*(const u32x4_t*)(O + (r00 + 17) * OS + cs * 16), f);
#pragma unroll
            for (int e = 0; e < PER16; ++e) s[e] += f[e];
            const int c = co0 + cs * PER16;
            if (c < p.Cout) *(u32x4_t*)((char*)p.y + ((ppix0 + (long long)pr * Wp + pc) * p.ldy + c) * ESZ) = pack16<T>(s);
        }
    }

    // sum over the 16 lanes that share a pixel row = one DPP row: four row rotations on the VALU (no LDS-pipe shuffles)
    template <int CTRL>
    static __device__ __forceinline__ float dpp_add(float v) {
        return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
    }
    static __device__ __forceinline__ float sub16(float v) {
        v = dpp_add<0x128>(v);  // row_ror:8
        v = dpp_add<0x124>(v);  // row_ror:4
        v = dpp_add<0x122>(v);  // row_ror:2
        return dpp_add<0x121>(v);  // row_ror:1
    }

    __device__ __forceinline__ void finish(const C2wConvArgs& p, const char* O, int OS, int tid) {
        constexpr int GRP = NIT < 8 ? NIT : 8;
        if (same_valid && !valid0) return;  // a thread past the channel bound stores nothing (no cross-lane operation in this flavour)
#pragma unroll
        for (int g0 = 0; g0 < NIT; g0 += GRP) {
            u32x4_t v[GRP], r2[GRP], m2[GRP];
#pragma unroll
            for (int i = 0; i < GRP; ++i) v[i] = *(const u32x4_t*)lds_seg(O, OS, tid, g0 + i);
            if constexpr (!EARLY) {
                if (p.res != nullptr) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) r2[i] = *(const u32x4_t*)((const char*)p.res + (ok(g0 + i) ? off[g0 + i] : 0));
                }
                if (p.mul != nullptr) {
#pragma unroll
                    for (int i = 0; i < GRP; ++i) m2[i] = *(const u32x4_t*)((const char*)p.mul + (ok(g0 + i) ? off[g0 + i] : 0));
                }
            }
#pragma unroll
            for (int i = 0; i < GRP; ++i) {
                const long long o = off[g0 + i];
                const bool st = ok(g0 + i);
                if (p.mul != nullptr || p.res != nullptr) {
                    float f[PER16];
                    unpack16<T>(v[i], f);
                    if (p.mul != nullptr) {
                        float gm[PER16];
                        unpack16<T>(EARLY ? mm[EARLY ? g0 + i : 0] : m2[i], gm);
                        if (p.mulmode == C2W_MUL_DSILU) {  // wave-uniform branch: the plain multiplier pays no exp/rcp
#pragma unroll
                            for (int e = 0; e < PER16; ++e) f[e] *= dsilu_f(gm[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < PER16; ++e) f[e] *= gm[e];
                        }
                    }
                    if (p.res != nullptr) {
                        float gr[PER16];
                        unpack16<T>(EARLY ? rr[EARLY ? g0 + i : 0] : r2[i], gr);
#pragma unroll
                        for (int e = 0; e < PER16; ++e) f[e] += gr[e];
                    }
                    v[i] = pack16<T>(f);
                }
                if (p.act == C2W_ACT_SILU_PAIR && p.y2 != nullptr) {  // y = silu(a), y2 = silu'(a): one exp/rcp pair serves both
                    float a_[PER16], h_[PER16], d_[PER16];
                    unpack16<T>(v[i], a_);
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        const float sg = sigmoid_f(a_[e]);
                        h_[e] = a_[e] * sg;
                        d_[e] = sg + h_[e] * (1.0f - sg);
                    }
                    if (st) {
                        epi_st((char*)p.y + o, pack16<T>(h_));
                        epi_st((char*)p.y2 + o, pack16<T>(d_));
                    }
                } else if (p.act == C2W_ACT_RELU_PAIR && p.y2 != nullptr) {  // y = max(a, 0), y2 = (a > 0)
                    float a_[PER16], h_[PER16], d_[PER16];
                    unpack16<T>(v[i], a_);
#pragma unroll
                    for (int e = 0; e < PER16; ++e) {
                        h_[e] = fmaxf(a_[e], 0.f);
                        d_[e] = a_[e] > 0.f ? 1.f : 0.f;
                    }
                    if (st) {
                        epi_st((char*)p.y + o, pack16<T>(h_));
                        epi_st((char*)p.y2 + o, pack16<T>(d_));
                    }
                } else if (st) {
                    epi_st((char*)p.y + o, v[i]);
                    if (p.y2 != nullptr) {  // second output: silu of the stored value (training keeps pre-activation and activation)
                        float f2[PER16];
                        unpack16<T>(v[i], f2);
#pragma unroll
                        for (int e = 0; e < PER16; ++e) f2[e] = silu_f(f2[e]);
                        epi_st((char*)p.y2 + o, pack16<T>(f2));
                    }
                }
            }
        }
    }
};
