// Probe: how fast can the waves of a CU stream an L2-resident table into LDS -- by LDS-DMA (buffer_load ... lds) or through registers
// (buffer_load_dwordx4 -> ds_write_b128) -- in the access pattern of conv_patch_t3's weight ring (8 waves, one 1 KiB piece per wave and
// stage, 3 slots, one barrier per stage, two workgroups per CU)?  No MFMAs, no fragment reads: the fill path alone.
//   hipcc --offload-arch=gfx950 -O3 -o fill_probe lab/probes/fill_probe.hip && ./fill_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((address_space(3))) void lds_void_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00027000); }

template <int MODE>  // 0: LDS-DMA, 1: registers, 2: half the waves each way
__global__ __launch_bounds__(512, 4) void fill(const char* __restrict__ table, uint32_t table_bytes, int tiles, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const __amdgpu_buffer_rsrc_t r = rsrc(table, table_bytes);
    const uint32_t voff = (uint32_t)(wid * 1024 + lane * 16);
    const int NS = 36;
    uint32_t acc = 0;
    for (int t = 0; t < tiles; ++t) {
        u32x4 reg0 = {0, 0, 0, 0}, reg1 = {0, 0, 0, 0};
        const bool dma = MODE == 0 || (MODE == 2 && (wid & 1) == 0);
        auto issue = [&](int s, u32x4& dst) {
            const uint32_t so = (uint32_t)(s * 8192);
            if (dma) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void_t*)(smem + (s % 3) * 8192 + wid * 1024), 16, (int)voff, (int)so, 0, 0);
            else dst = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)so, 0);
        };
        issue(0, reg0);
        issue(1, reg1);
        for (int s = 0; s < NS; ++s) {
            // stage s must be in LDS: the DMA form waits for its piece; the register form writes the piece it loaded two stages ago
            if (dma) {
                if (s + 1 < NS) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                u32x4& cur = (s & 1) ? reg1 : reg0;
                if (s + 1 < NS) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                *(u32x4*)(smem + (s % 3) * 8192 + wid * 1024 + lane * 16) = cur;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 2 < NS) issue(s + 2, (s & 1) ? reg1 : reg0);
            if (s == NS - 1) acc += *(const uint32_t*)(smem + ((tid * 16) % 24576));
        }
        __builtin_amdgcn_s_barrier();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const uint32_t table_bytes = 36 * 8192;
    char* table;
    uint32_t* sink;
    CK(hipMalloc(&table, table_bytes));
    CK(hipMemset(table, 1, table_bytes));
    CK(hipMalloc(&sink, 64));
    const int tiles = 64, grid = 512, lds = 71 * 1024;  // two workgroups per CU, like the conv kernel
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&]() {
            if (mode == 0) fill<0><<<grid, 512, lds>>>(table, table_bytes, tiles, sink);
            else if (mode == 1) fill<1><<<grid, 512, lds>>>(table, table_bytes, tiles, sink);
            else fill<2><<<grid, 512, lds>>>(table, table_bytes, tiles, sink);
        };
        if (mode == 0) CK(hipFuncSetAttribute((const void*)fill<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (mode == 1) CK(hipFuncSetAttribute((const void*)fill<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (mode == 2) CK(hipFuncSetAttribute((const void*)fill<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        ms /= 5;
        const double bytes = (double)grid * tiles * 36 * 8192;
        printf("%-34s %7.3f ms  %6.2f TB/s chip-wide  %5.1f GB/s per CU  (%.0f ns per 8 KiB stage and workgroup)\n",
               mode == 0 ? "LDS-DMA" : mode == 1 ? "registers (load -> ds_write_b128)" : "half the waves each way", ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256,
               ms * 1e6 / (tiles * 36.0 * (grid / 512.0)));
    }
    return 0;
}
