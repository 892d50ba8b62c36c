// Microbenchmark: how fast can ONE CU pull L2-resident data into LDS?
//   mode 0: global_load_lds_dwordx4 (LDS-DMA), full 128-B lines per 8 lanes
//   mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128
//   mode 2: global_load_dwordx4 -> VGPR only (no LDS write)
// 256 workgroups (one per CU) x WAVES waves, each workgroup streams its own 512 KiB window (L2 resident after the
// first pass) REPS times.  Prints bytes/clk/CU at the measured kernel time (clock 2.4 GHz assumed).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

template <int MODE, int WAVES, int DEPTH>
__global__ __launch_bounds__(WAVES * 64) void stream_kernel(const char* __restrict__ src, int reps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t window = 64 * 1024;   // 2 MiB per XCD: L2 resident
    const char* base = src + (size_t)blockIdx.x * window;
    // one "piece" = 1 KiB (64 lanes x 16 B); a wave handles pieces wave, wave+WAVES, ... of each 64 KiB chunk
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int chunk = 0; chunk < 8; ++chunk) {     // the same 64 KiB window, 8 times per rep
            const char* cb = base;
            asm volatile("" : "+v"(cb));               // opaque: the loads cannot be hoisted out of the loops
            char* lb = lds + (chunk & 1) * 65536;
            if constexpr (MODE == 0) {
#pragma unroll
                for (int p = 0; p < 64 / WAVES; ++p) {
                    const int piece = wave + p * WAVES;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(cb + piece * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(lb + piece * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH) : "memory");
            } else {
                u32x4 v[64 / WAVES];
#pragma unroll
                for (int p = 0; p < 64 / WAVES; ++p) {
                    const int piece = wave + p * WAVES;
                    v[p] = *reinterpret_cast<const u32x4*>(cb + piece * 1024 + lane * 16);
                }
#pragma unroll
                for (int p = 0; p < 64 / WAVES; ++p) {
                    const int piece = wave + p * WAVES;
                    if constexpr (MODE == 1) *reinterpret_cast<u32x4*>(lb + piece * 1024 + lane * 16) = v[p];
                    else acc ^= v[p][0] ^ v[p][3];
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE != 2) acc = *reinterpret_cast<unsigned*>(lds + (threadIdx.x * 16) % 65536);
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE, int WAVES, int DEPTH>
void run(const char* name, const char* buf, unsigned* sink) {
    const int reps = 64;
    hipFuncSetAttribute((const void*)stream_kernel<MODE, WAVES, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    stream_kernel<MODE, WAVES, DEPTH><<<256, WAVES * 64, 131072>>>(buf, 2, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    stream_kernel<MODE, WAVES, DEPTH><<<256, WAVES * 64, 131072>>>(buf, reps, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes_per_cu = (double)reps * 512 * 1024;
    printf("%-44s %8.1f us  %6.1f B/clk/CU  %6.2f TB/s aggregate\n", name, ms * 1e3, bytes_per_cu / (ms * 1e-3 * 2.4e9),
           bytes_per_cu * 256 / (ms * 1e-3) / 1e12);
}

int main() {
    char* buf; unsigned* sink;
    hipMalloc(&buf, (size_t)256 * 512 * 1024);  // windows are 64 KiB apart * 8? no: blockIdx * window hipMalloc(&sink, 64);
    hipMemset(buf, 1, (size_t)256 * 512 * 1024);
    run<0, 8, 0>("LDS-DMA, 8 waves, wait all per 64 KiB", buf, sink);
    run<0, 8, 8>("LDS-DMA, 8 waves, 1 chunk in flight", buf, sink);
    run<0, 4, 16>("LDS-DMA, 4 waves, 1 chunk in flight", buf, sink);
    run<0, 16, 4>("LDS-DMA, 16 waves, 1 chunk in flight", buf, sink);
    run<1, 8, 0>("VGPR + ds_write_b128, 8 waves", buf, sink);
    run<1, 16, 0>("VGPR + ds_write_b128, 16 waves", buf, sink);
    run<2, 8, 0>("VGPR only, 8 waves", buf, sink);
    run<2, 16, 0>("VGPR only, 16 waves", buf, sink);
    return 0;
}
