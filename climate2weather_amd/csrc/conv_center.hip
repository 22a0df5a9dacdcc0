// The output convolution of the network as the SAMPLER needs it.
//
// src/thor/score.py:76-88 (fold) keeps, of the w * F channels the network writes for a window (model/nn.py:194: 3x3, 128 -> w * F),
// only the centre frame's F -- every channel only for the first and the last window of a trajectory.  At w = 13 the reference computes
// 13 frames per window and throws 12 away.  This kernel computes the kept rows only and writes them where fold() would put them:
//
//     out[b][c][pix] (fp32 planes, b-stride `ostride`)  =  round_T( bias[r0 + c] + sum_{tap, ci} w[r0 + c][tap][ci] * x[b][src(pix, tap)][ci] ),   c < nr <= 16
//
// i.e. rows r0 .. r0 + nr - 1 of the [wrows][9][Cin] weight matrix (r0 = k * F, nr = F for the centre frame), rounded through the
// compute type like the full convolution's output rows, stored as the fp32 planes of the eps trajectory (L, F, H, W) -- the
// separate window_scatter pass over the 128-channel output rows is gone too.  The first / last window's other frames go through the
// full convolution (engine.py).
//
// A workgroup walks 8x16-pixel tiles, 4 waves x (16 weight rows x 32 pixels).  All nine taps of the 16 rows stay in LDS for the
// whole launch (36 KiB at Cin = 128: there is no weight ring to pace), the halo patch comes in 64-channel chunks as in
// conv_patch_half_kernel (30 KiB, same piece layout and swizzle): 66 KiB, two workgroups per CU, no hand-counted waits.  The launch is
// bound by the patch stream (the input is read once, 1.1x with halos), not by its 72 MFMAs per wave and chunk.
#include <cstdint>

#include <hip/hip_runtime.h>

#include "c2w_hip.h"
#include "common.h"

namespace {

constexpr int CC_NTHR = 256;
constexpr int CC_PW = 24;                   // patch row pitch in pixels (18 used)
constexpr int CC_PROW = CC_PW * 128;        // bytes per patch row of one 64-channel chunk
constexpr int CC_NPIECE = 10 * 3;           // (8 + 2) patch rows x 3 pieces of 8 pixels
constexpr int CC_PBYTES = CC_NPIECE * 1024; // 30,720
constexpr int CC_WTILE = 16 * 128;          // one (chunk, tap) weight tile: 16 rows x 128 B

template <typename T>
__global__ __launch_bounds__(CC_NTHR, 2) void conv_center_kernel(const T* __restrict__ x, const T* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ out, int ntiles, int H, int W, int Cin, int wrows, int r0, int nr,
                                                                long long ostride) {
    constexpr int ESZ = 2, CK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [weights: nchunk x 9 tiles of 2 KiB | patch chunk]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const int nchunk = Cin / CK;
    const int WBYTES = nchunk * 9 * CC_WTILE;
    char* const P = smem + WBYTES;

    const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, (uint32_t)((size_t)wrows * 9 * Cin * ESZ));
    // weight pieces, once per workgroup: tile (chunk, tap) = 16 rows x 128 B = two pieces of 8 rows; rows at or past wrows read as zeros
    {
        const int row8 = lane >> 3;
        const int npiece = nchunk * 9 * 2;
        for (int pc = wid; pc < npiece; pc += 4) {
            const int tile = pc >> 1, row = (pc & 1) * 8 + row8;
            const int chunk = tile / 9, tap = tile - chunk * 9;
            const uint32_t vo = (uint32_t)(r0 + row) * (uint32_t)(9 * Cin * ESZ) + (uint32_t)(((lane & 7) ^ (row & 7)) << 4);
            glds16(rw, smem + pc * 1024, vo, (uint32_t)(tap * Cin + chunk * CK) * ESZ);
        }
    }
    uint32_t offA[2], offB[2][3][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        offA[ks] = (uint32_t)(li * 128 + (((ks * 4 + lg) ^ (li & 7)) << 4));
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int px = li + kw;
                offB[ks][kw][n] = (uint32_t)(((2 * wid + n) * CC_PW + px) * 128 + (((ks * 4 + lg) ^ (px & 7)) << 4));
            }
    }
    const int tw = W >> 4, tpi = (H >> 3) * tw;
    const size_t img_bytes = (size_t)H * W * Cin * ESZ;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (bias != nullptr && 4 * lg + r < nr) ? bias[r0 + 4 * lg + r] : 0.f;

    // persistent over tiles (the grid is at most a few workgroups per CU): the weights stay, adjacent workgroups take adjacent tiles
    for (int L = blockIdx.x; L < ntiles; L += gridDim.x) {
        const int b = L / tpi, tt = L - b * tpi;
        const int ty = tt / tw, tx = tt - ty * tw;
        const int oh0 = ty << 3, ow0 = tx << 4;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc((const char*)x + (size_t)b * img_bytes, (uint32_t)img_bytes);
        // patch pieces: 30 pieces of 8 pixels x 128 B over 4 waves = 8 rounds (pieces past the end repeat the last one)
        uint32_t pvo[8];
        int pdst[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int pc = r * 4 + wid;
            pc = pc < CC_NPIECE ? pc : CC_NPIECE - 1;
            const int pr = pc / 3, pg = pc - pr * 3;
            const int px = pg * 8 + (lane >> 3);
            const int ih = oh0 - 1 + pr, iw = ow0 - 1 + px;
            const bool ok = (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W && px < 18;
            const uint32_t lc = (uint32_t)((lane & 7) ^ ((lane >> 3) & 7));  // LDS position d of pixel px holds source chunk d ^ (px & 7)
            pvo[r] = ok ? (uint32_t)((ih * W + iw) * Cin) * ESZ + (lc << 4) : C2W_OOB;
            pdst[r] = pc * 1024;
        }
        f32x4_t acc[2] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}, (f32x4_t){0.f, 0.f, 0.f, 0.f}};
        for (int c = 0; c < nchunk; ++c) {
            if (c > 0 || L != (int)blockIdx.x) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __syncthreads();  // every wave has read the patch that is overwritten next
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) glds16(rx, P + pdst[r], pvo[r], (uint32_t)c * 128u);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const char* const Wc = smem + c * 9 * CC_WTILE;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const u32x4_t a = *(const u32x4_t*)(Wc + tap * CC_WTILE + offA[ks]);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const u32x4_t bq = *(const u32x4_t*)(P + offB[ks][kw][n] + kh * CC_PROW);
                        acc[n] = mfma16<T>(a, bq, acc[n]);
                    }
                }
            }
        }
        // lane (li, lg) holds rows 4 lg .. 4 lg + 3 of pixel (2 wid + n, li) of the tile
        float* const ob = out + (long long)b * ostride;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 4 * lg + r;
            if (ch < nr) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    T t;
                    Elem<T>::st(&t, acc[n][r] + bv[r]);
                    ob[(long long)ch * H * W + (long long)(oh0 + 2 * wid + n) * W + ow0 + li] = Elem<T>::ld(&t);
                }
            }
        }
    }
}

template <typename T>
int launch(const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int Cin, int wrows, int r0, int nr, long long ostride,
           hipStream_t st) {
    const int lds = (Cin / 64) * 9 * CC_WTILE + CC_PBYTES;
    static int attr_lds = 0;
    if (lds > attr_lds) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)conv_center_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    const long long tiles = (long long)B * (H >> 3) * (W >> 4);
    const int grid = (int)(tiles < 1024 ? tiles : 1024);  // two co-resident workgroups per CU and two more queued behind them
    conv_center_kernel<T><<<grid, CC_NTHR, lds, st>>>((const T*)x, (const T*)w, bias, out, (int)tiles, H, W, Cin, wrows, r0, nr, ostride);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int c2w_conv_center_supported(int H, int W, int Cin, int nr, int dtype) {
    return (dtype == C2W_DTYPE_BF16 || dtype == C2W_DTYPE_F16) && H > 0 && W > 0 && (H & 7) == 0 && (W & 15) == 0 && (Cin == 64 || Cin == 128) && nr >= 1 &&
           nr <= 16 && (long long)H * W * Cin * 2 < (1ll << 31);
}

extern "C" int c2w_conv_center(const void* x, const void* w, const float* bias, float* out, int B, int H, int W, int Cin, int wrows, int r0, int nr,
                               long long ostride, int dtype, void* stream) {
    if (!x || !w || !out || B <= 0 || r0 < 0 || r0 + nr > wrows) return C2W_ERR_BAD_ARG;
    if (!c2w_conv_center_supported(H, W, Cin, nr, dtype) || (long long)B * (H >> 3) * (W >> 4) >= (1ll << 31)) return C2W_ERR_BAD_SHAPE;
    if (dtype == C2W_DTYPE_BF16) return launch<bf16_t>(x, w, bias, out, B, H, W, Cin, wrows, r0, nr, ostride, (hipStream_t)stream);
    return launch<f16_t>(x, w, bias, out, B, H, W, Cin, wrows, r0, nr, ostride, (hipStream_t)stream);
}
