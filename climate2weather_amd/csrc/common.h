// Shared device helpers for the gfx950 (CDNA4) kernels of climate2weather_amd.
// Wave = 64 lanes everywhere.  Activations are NHWC ("pixel rows" of C channels);
// T is the storage type: float (parity mode), bf16 (throughput mode) or fp16 (the
// reference's autocast type, train.py:98 "16-mixed"; 10 mantissa bits, narrow range:
// training in it needs the loss scale of c2w_grad_scaler_*); all arithmetic outside
// the MFMA operands is fp32.
#pragma once
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;  // 8 x bf16 bit patterns (one MFMA operand)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

typedef uint16_t bf16_t;  // raw bf16 bits
typedef _Float16 f16_t;   // IEEE half: a distinct 2-byte type, so templates tell the two 16-bit formats apart
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

#define C2W_OOB 0x80000000u  // buffer voffset that is always out of range (-> loads return 0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(uint16_t, b);
}
// Two fp32 -> one dword of two 16-bit values, ONE instruction each (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32: RNE, NaN stays NaN, fp16
// overflow -> inf).  Written as a 2-vector conversion: converted one at a time and or-ed together, hipcc emitted a v_cvt_pk per VALUE
// plus a shift and an or per pair -- 16 instructions per 8 outputs instead of 4, in every epilogue (round 3, `hipcc -S`).
typedef __attribute__((ext_vector_type(2))) float c2w_f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 c2w_bf16x2_t;
// The bf16 form is issued from a volatile asm statement: same instruction, but hipcc then keeps the conversions where the epilogues
// wrote them (between the activation arithmetic of neighbouring values) instead of clustering them in front of the LDS writes.  Same-box
// A/B on the fixed kernel, three rounds: SiLU epilogues of the dominant conv 547 -> 525 and 591 -> 572 us, step 48.48 -> 48.12 ms
// (profiles/r03_experiments.md section 5); with an s_nop behind it, or as a non-volatile asm, 0.1-0.2 ms of that are lost again.
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    uint32_t d;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(lo), "v"(hi));
    return d;
}

// NOT for values that come straight out of the matrix core: hipcc's hazard recognizer does not look into an asm statement, so the wait
// states between an MFMA and the first read of its result are missing and the conversion reads the registers too early (seen: NaNs in
// the attention output product).  Such values go through pack_acc2 (the compiler's own conversion, hazards handled);
// climate2weather_amd/build.py (isa_checks.py) looks for an MFMA result feeding an asm conversion in the ISA of every kernel it compiles and refuses to link one.
__device__ __forceinline__ uint32_t pack_bf16x2_plain(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((c2w_f32x2_t){lo, hi}, c2w_bf16x2_t));
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((c2w_f32x2_t){lo, hi}, f16x2_t));
}
// two 16-bit storage values in one dword <-> fp32
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
// two accumulator values (straight from an MFMA) -> one dword of two 16-bit values
template <typename T> __device__ __forceinline__ uint32_t pack_acc2(float lo, float hi);
template <> __device__ __forceinline__ uint32_t pack_acc2<bf16_t>(float lo, float hi) { return pack_bf16x2_plain(lo, hi); }
template <> __device__ __forceinline__ uint32_t pack_acc2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
template <typename T> __device__ __forceinline__ void unpack2(uint32_t v, float& lo, float& hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t v, float& lo, float& hi) {
    lo = __uint_as_float(v << 16);
    hi = __uint_as_float(v & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t v, float& lo, float& hi) {
    const f16x2_t h = __builtin_bit_cast(f16x2_t, v);
    lo = (float)h[0];
    hi = (float)h[1];
}
// scalar store conversion of the 16-bit types (raw bits)
template <typename T> __device__ __forceinline__ uint16_t f32_to_bits16(float f);
template <> __device__ __forceinline__ uint16_t f32_to_bits16<bf16_t>(float f) { return f32_to_bf16(f); }
template <> __device__ __forceinline__ uint16_t f32_to_bits16<f16_t>(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

// one matrix-core step on 16-bit operands: 16x16 tile, K = 32 (8 values per lane in 16 B)
template <typename T> __device__ __forceinline__ f32x4_t mfma16(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c);
template <> __device__ __forceinline__ f32x4_t mfma16<bf16_t>(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4_t mfma16<f16_t>(const u32x4_t& a, const u32x4_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
// same on operands held as eight 16-bit patterns
template <typename T> __device__ __forceinline__ f32x4_t mfma16s(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    return mfma16<T>(__builtin_bit_cast(u32x4_t, a), __builtin_bit_cast(u32x4_t, b), c);
}
// operand of eight ones (column sums through the matrix core: the bias gradient)
template <typename T> __device__ __forceinline__ u32x4_t ones16();
template <> __device__ __forceinline__ u32x4_t ones16<bf16_t>() { return (u32x4_t){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; }
template <> __device__ __forceinline__ u32x4_t ones16<f16_t>() { return (u32x4_t){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; }

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

template <> struct Elem<f16_t> {
    static constexpr int SZ = 2;
    static constexpr int PER16 = 8;
    __device__ static __forceinline__ float ld(const f16_t* p) { return (float)*p; }
    __device__ static __forceinline__ void st(f16_t* p, float v) { *p = (_Float16)v; }
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
template <> __device__ __forceinline__ void unpack16<f16_t>(const u32x4_t& v, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) unpack2<f16_t>(v[i], f[2 * i], f[2 * i + 1]);
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

template <> __device__ __forceinline__ u32x4_t pack16<f16_t>(const float* f) {
    u32x4_t v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = pack_f16x2(f[2 * i], f[2 * i + 1]);
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

template <int N> struct IC { static constexpr int value = N; };
template <typename F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(IC<I>{}), ...); }
// f(IC<0>{}), ..., f(IC<N - 1>{}): loop indices that are constant expressions (asm immediates, register-array subscripts)
template <int N, typename F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// one 16x32 (or 32x16) 16-bit MFMA operand = two transposing 8-byte LDS reads
typedef __attribute__((ext_vector_type(2))) int tr_half;
struct tr_frag {
    tr_half lo, hi;
    __device__ __forceinline__ bf16x8_t vec() const { return __builtin_bit_cast(bf16x8_t, (__attribute__((ext_vector_type(4))) int){lo[0], lo[1], hi[0], hi[1]}); }
};

#define HIP_CHECK_RET(expr)                 \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
