// Geometry shared by the forward/dgrad implicit GEMM (conv_igemm.hip) and the weight-gradient GEMM (wgrad.hip).
#pragma once
#include "common.h"
#include "c2w_hip.h"

// source pixel of output pixel (oh,ow) under tap (kh,kw); returns false when it is padding
template <int MODE, typename A>
__device__ __forceinline__ bool src_pixel(const A& p, int oh, int ow, int kh, int kw, int& ih, int& iw) {
    if constexpr (MODE == C2W_CONV_1X1) {
        ih = oh; iw = ow;
        return true;
    } else if constexpr (MODE == C2W_CONV_S1) {
        ih = oh + kh - 1; iw = ow + kw - 1;
        return (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
    } else if constexpr (MODE == C2W_CONV_S2) {
        ih = 2 * oh + kh - 1; iw = 2 * ow + kw - 1;
        return (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
    } else if constexpr (MODE == C2W_CONV_UP) {  // conv3x3(nearest_up2(x)): zero padding lives in the upsampled frame
        int uh = oh + kh - 1, uw = ow + kw - 1;
        ih = uh >> 1; iw = uw >> 1;
        return (unsigned)uh < (unsigned)p.Hout && (unsigned)uw < (unsigned)p.Wout;
    } else {  // C2W_CONV_TS2: input-gradient of the stride-2 conv; x := dy, y := dx, ih = (oh + 1 - kh)/2
        int th = oh + 1 - kh, tw = ow + 1 - kw;
        ih = th >> 1; iw = tw >> 1;
        return ((th | tw) & 1) == 0 && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
    }
}


// exact n / d for n < 2^31 with a host-precomputed (magic, shift): q = umulhi(n, magic) >> shift; magic == 0 means d == 1
struct FastDiv {
    uint32_t magic, shift;
};
__device__ __forceinline__ int fast_div(int n, FastDiv d) { return d.magic ? (int)(__umulhi((uint32_t)n, d.magic) >> d.shift) : n; }

// halo-patch kernel for 3x3 stride-1 convolutions on 16x16-tileable images (conv_patch.hip)
bool c2w_conv_patch_eligible(const C2wConvArgs& a);
int c2w_conv_patch_s1(const C2wConvArgs& a, int dtype, hipStream_t st);
// stride-2 forward on the parity planes of the halo patch (conv_patch.hip; 16-bit)
bool c2w_conv_s2_patch_eligible(const C2wConvArgs& a, int dtype);
int c2w_conv_patch_s2(const C2wConvArgs& a, int dtype, hipStream_t st);
// stride-2 input gradient per output-parity class on the halo patch (conv_patch.hip)
bool c2w_conv_ts2_patch_eligible(const C2wConvArgs& a);
int c2w_conv_patch_ts2(const C2wConvArgs& a, int dtype, hipStream_t st);
// 8-pixel-wide images, two per 8x16 tile (conv_patch.hip)
bool c2w_conv_pair_eligible(const C2wConvArgs& a);
int c2w_conv_patch_pair(const C2wConvArgs& a, int dtype, hipStream_t st);
// bf16 variant on 16x16-pixel tiles with a 32-channel weight ring (conv_patch3.hip)
bool c2w_conv_patch3_wanted(const C2wConvArgs& a, int dtype);
int c2w_conv_patch3(const C2wConvArgs& a, int dtype, hipStream_t st);

// halo-patch weight-gradient kernel for the same convolutions (wgrad_patch.hip)
bool c2w_wgrad_patch_eligible(const C2wConvArgs& a);
bool c2w_wgrad_patch_pair(const C2wConvArgs& a);  // 8-pixel-wide images: two per K tile
// ws / ws_bytes: the caller's scratch buffer for the split-K partial sums (NULL / too small: fp32 atomics)
int c2w_wgrad_patch(const C2wConvArgs& a, float* dw, float* db, float* ws, size_t ws_bytes, int dtype, hipStream_t st);
size_t c2w_wgrad_patch_ws_bytes(const C2wConvArgs& a, int dtype);
// the weight gradients of n layers of one geometry as one launch (wgrad_patch.hip)
bool c2w_wgrad_patch_group_eligible(const C2wConvArgs& a, int n, int dtype);
int c2w_wgrad_patch_group(const C2wConvArgs& a, const C2wWgradItem* items, int n, float* ws, size_t ws_bytes, int dtype, hipStream_t st);
size_t c2w_wgrad_patch_group_ws_bytes(const C2wConvArgs& a, int n, int dtype);
