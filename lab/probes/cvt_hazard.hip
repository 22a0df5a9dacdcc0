// Probe: is the result of v_cvt_pk_bf16_f32 safe to read as LDS / global store data by the wave's very next instruction on gfx950?
// Each lane converts a fresh pair every iteration into a register that holds a marker, writes it to LDS (or global) with the next
// instruction, reads it back and compares with the software conversion.  A stale read shows up as the marker (or the previous
// iteration's value) in memory.  Run alone (one wave per SIMD issues back to back) and beside an MFMA-heavy kernel on a second stream.
//   hipcc --offload-arch=gfx950 -O2 -o cvt_hazard lab/probes/cvt_hazard.hip && ./cvt_hazard
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ uint32_t sw_bf16(float f) {  // RNE, finite inputs only
    uint32_t u = __float_as_uint(f);
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

template <int MODE>
__global__ void probe(uint32_t* out, uint32_t* gbuf, int iters) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t laddr = tid * 8;
    uint32_t* gp = gbuf + ((size_t)blockIdx.x * blockDim.x + tid) * 2;
    uint32_t bad = 0, stale_marker = 0, stale_prev = 0;
    float a = 1.0f + tid * 0.001f + blockIdx.x * 0.37f, b = -2.0f - tid * 0.002f;
    uint32_t prev = 0;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0001f + 0.25f;
        b = b * 0.9999f - 0.125f;
        if (a > 1e6f) a = 1.0f;
        if (b < -1e6f) b = -1.0f;
        const float a2 = a * 1.5f, b2 = b * 0.5f;
        const uint32_t want0 = sw_bf16(a) | (sw_bf16(b) << 16), want1 = sw_bf16(a2) | (sw_bf16(b2) << 16);
        uint32_t d0 = 0xdead0000u | (i & 0xffff), d1 = 0xbeef0000u | (i & 0xffff), r0, r1;
        if (MODE == 0) {  // two conversions, then a 64-bit LDS write of the pair: the second conversion is adjacent to the write
            asm volatile("v_mov_b32 %0, %3\n\tv_mov_b32 %1, %4\n\ts_nop 4\n\t"
                         "v_cvt_pk_bf16_f32 %0, %5, %6\n\tv_cvt_pk_bf16_f32 %1, %7, %8\n\t"
                         "ds_write2_b32 %2, %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1) : "v"(laddr), "v"(d0), "v"(d1), "v"(a), "v"(b), "v"(a2), "v"(b2) : "memory");
            r0 = lds[tid * 2];
            r1 = lds[tid * 2 + 1];
        }
        if (r0 != want0 || r1 != want1) {
            ++bad;
            if ((r0 >> 16) == 0xdead || (r1 >> 16) == 0xbeef) ++stale_marker;
            if (r1 == prev) ++stale_prev;
        }
        prev = want1;
        __builtin_amdgcn_s_sleep(0);
    }
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], stale_marker);
    atomicAdd(&out[2], stale_prev);
}

// the register-pair form, as the epilogue has it: v[N:N+1] written by two conversions, read by ds_write_b64 / global_store_dwordx2
template <int MODE>
__global__ void probe_pair(uint32_t* out, uint32_t* gbuf, int iters) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x;
    const uint32_t laddr = tid * 8;
    uint32_t* gp = gbuf + ((size_t)blockIdx.x * blockDim.x + tid) * 2;
    uint32_t bad = 0, stale_marker = 0;
    float a = 1.0f + tid * 0.001f + blockIdx.x * 0.37f, b = -2.0f - tid * 0.002f;
    for (int i = 0; i < iters; ++i) {
        a = a * 1.0001f + 0.25f;
        b = b * 0.9999f - 0.125f;
        if (a > 1e6f) a = 1.0f;
        if (b < -1e6f) b = -1.0f;
        const float a2 = a * 1.5f, b2 = b * 0.5f;
        const uint32_t want0 = sw_bf16(a) | (sw_bf16(b) << 16), want1 = sw_bf16(a2) | (sw_bf16(b2) << 16);
        const uint32_t m0 = 0xdead0000u | (i & 0xffff), m1 = 0xbeef0000u | (i & 0xffff);
        uint32_t r0, r1;
        if (MODE == 0) {
            asm volatile("v_mov_b32 v100, %5\n\tv_mov_b32 v101, %6\n\ts_nop 4\n\t"
                         "v_cvt_pk_bf16_f32 v100, %1, %2\n\tv_cvt_pk_bf16_f32 v101, %3, %4\n\t"
                         "ds_write_b64 %0, v[100:101]\n\ts_waitcnt lgkmcnt(0)"
                         : : "v"(laddr), "v"(a), "v"(b), "v"(a2), "v"(b2), "v"(m0), "v"(m1) : "memory", "v100", "v101");
            r0 = lds[tid * 2];
            r1 = lds[tid * 2 + 1];
        } else {
            asm volatile("v_mov_b32 v100, %5\n\tv_mov_b32 v101, %6\n\ts_nop 4\n\t"
                         "v_cvt_pk_bf16_f32 v100, %1, %2\n\tv_cvt_pk_bf16_f32 v101, %3, %4\n\t"
                         "global_store_dwordx2 %0, v[100:101], off\n\ts_waitcnt vmcnt(0)"
                         : : "v"(gp), "v"(a), "v"(b), "v"(a2), "v"(b2), "v"(m0), "v"(m1) : "memory", "v100", "v101");
            r0 = __builtin_nontemporal_load(gp);
            r1 = __builtin_nontemporal_load(gp + 1);
        }
        if (r0 != want0 || r1 != want1) {
            ++bad;
            if ((r0 >> 16) == 0xdead || (r1 >> 16) == 0xbeef) ++stale_marker;
        }
    }
    atomicAdd(&out[0], bad);
    atomicAdd(&out[1], stale_marker);
}

// something that keeps the matrix cores and the other SIMD slots busy on a second stream
__global__ void mfma_noise(float* sink, int iters) {
    typedef __attribute__((ext_vector_type(8))) short bf16x8;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x); b[i] = (short)(0x3f00 + i); }
    f32x4 c = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    if (c[0] == 123.456f) sink[0] = c[1];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    uint32_t *out, *gbuf;
    float* sink;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&gbuf, (size_t)4096 * 512 * 8));
    CK(hipMalloc(&sink, 64));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    const struct { int grid, block; bool noise; const char* what; } cfg[] = {
        {256, 64, false, "one wave per CU, alone"}, {1024, 256, false, "one wave per SIMD, 1024 workgroups"}, {4096, 512, false, "two waves per SIMD, 4096 workgroups"},
        {256, 64, true, "one wave per CU beside MFMA noise"}, {2048, 256, true, "one wave per SIMD beside MFMA noise"}, {4096, 512, true, "two waves per SIMD beside MFMA noise"}};
    for (int form = 0; form < 3; ++form) {
        for (auto& c : cfg) {
            uint32_t h[4] = {0, 0, 0, 0};
            CK(hipMemsetAsync(out, 0, 64, s0));
            CK(hipStreamSynchronize(s0));
            if (c.noise) mfma_noise<<<2048, 256, 0, s1>>>(sink, iters * 40);
            if (form == 0) probe<0><<<c.grid, c.block, c.block * 8, s0>>>(out, gbuf, iters);
            else if (form == 1) probe_pair<0><<<c.grid, c.block, c.block * 8, s0>>>(out, gbuf, iters);
            else probe_pair<1><<<c.grid, c.block, c.block * 8, s0>>>(out, gbuf, iters);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            printf("%-22s %-44s conversions %.3e  wrong %u (marker seen %u, previous value seen %u)\n",
                   form == 0 ? "ds_write2_b32" : form == 1 ? "ds_write_b64 pair" : "global_store_dwordx2", c.what, 2.0 * iters * c.grid * c.block, h[0], h[1], h[2]);
        }
    }
    return 0;
}
