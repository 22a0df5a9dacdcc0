// Shared device helpers for the gfx950 (CDNA4) kernels of climate2weather_amd.
// Wave = 64 lanes everywhere.  Activations are NHWC ("pixel rows" of C channels);
// T is the storage type: float (parity mode) or bf16 (throughput mode); all
// arithmetic outside the MFMA operands is fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;  // 8 x bf16 bit patterns (one MFMA operand)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

typedef uint16_t bf16_t;  // raw bf16 bits

#define C2W_OOB 0x80000000u  // buffer voffset that is always out of range (-> loads return 0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int SZ = 4;
    static constexpr int PER16 = 4;  // elements per 16-byte vector
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int SZ = 2;
    static constexpr int PER16 = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector <-> fp32 lanes
template <typename T> __device__ __forceinline__ void unpack16(const u32x4_t& v, float* f);
template <> __device__ __forceinline__ void unpack16<float>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
}
template <> __device__ __forceinline__ void unpack16<bf16_t>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(v[i] << 16);
        f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
}
template <typename T> __device__ __forceinline__ u32x4_t pack16(const float* f);
template <> __device__ __forceinline__ u32x4_t pack16<float>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(f[i]);
    return v;
}
template <> __device__ __forceinline__ u32x4_t pack16<bf16_t>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return v;
}

// SiLU and its derivative on the fast transcendental path (v_exp_f32 + v_rcp_f32, ~1 ulp each): the epilogues evaluate
// them 64x per thread per tile, where an IEEE division (v_div_scale/fmas/fixup) costs more than the rest of the epilogue.
__device__ __forceinline__ float sigmoid_f(float a) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * a));
}
__device__ __forceinline__ float silu_f(float a) { return a * sigmoid_f(a); }
__device__ __forceinline__ float dsilu_f(float a) {
    const float s = sigmoid_f(a);
    return s * (1.0f + a * (1.0f - s));
}

// wave-uniform buffer descriptor over [base, base+bytes)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

typedef __attribute__((address_space(3))) void lds_void_t;
// async global -> LDS, 16 B per lane: LDS dst = (wave-uniform) lds_base + lane*16, src = rsrc + voff + soff
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t rsrc, void* lds_base, uint32_t voff, uint32_t soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_void_t*)lds_base, 16, (int)voff, (int)soff, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

#define HIP_CHECK_RET(expr)                 \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
