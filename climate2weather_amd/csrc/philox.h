// Philox4x32-10 + Box-Muller: the counter-based N(0,1) stream every kernel that needs the training step's noise regenerates
// (pointwise.hip: input conversion, loss tail, c2w_philox_normal; conv_patch3.hip: the loss fused into the output conv's epilogue).
#pragma once
#include "common.h"

// Counter-based normal noise: Philox4x32-10 (key = seed, counter = index of a block of four consecutive elements) + Box-Muller.
// Element e of a stream is lane e & 3 of block e >> 2, so any kernel can regenerate eps[e] instead of reading it: the training
// step's eps = randn_like(x) (src/thor/pipelines.py:22-25) is never written to or read from HBM (3 x 545 MB per step at B = 128).
__device__ __forceinline__ f32x4_t philox_normal4(uint32_t k0, uint32_t k1, unsigned long long blk) {
    uint32_t c0 = (uint32_t)blk, c1 = (uint32_t)(blk >> 32), c2 = 0u, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u0 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c2 >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(c3 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float ra = sqrtf(-2.0f * __logf(u0)), rb = sqrtf(-2.0f * __logf(u2));
    float sa, ca, sb, cb;
    __sincosf(6.28318530717958647692f * u1, &sa, &ca);
    __sincosf(6.28318530717958647692f * u3, &sb, &cb);
    return (f32x4_t){ra * ca, ra * sa, rb * cb, rb * sb};
}
__device__ __forceinline__ float philox_normal1(uint32_t k0, uint32_t k1, unsigned long long e) { return philox_normal4(k0, k1, e >> 2)[(int)(e & 3)]; }

