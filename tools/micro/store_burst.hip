// What does the store burst of a GEMM epilogue cost, and does its shape matter?  256 workgroups x 256 threads each write a
// 256 x 256 bf16 tile (128 KiB) of a [32768, N] matrix (N = 3072: row stride 6 KiB) the way the 4-wave GEMM's epilogue does --
//   A: per wave instruction 16 rows x 64 B (lane (c, g): row c, 16 B at column chunk g), the two 64-B halves of a 128-B line
//      written 8 instructions apart (the product epilogue),
//   B: per wave instruction 8 rows x 128 B (whole lines),
//   C: per wave instruction 2 rows x 512 B (a wave writes whole tile rows),
// plain or non-temporal; six tiles per workgroup back to back (one qkv launch's worth of output: 201 MB).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int NT>
__global__ __launch_bounds__(256) void burst(char* out, int N, int n_tiles, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn2 = wave & 1;
    const u32x4 v = u32x4{(unsigned)threadIdx.x, 2u, 3u, 4u};
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = blockIdx.x + t * gridDim.x;
        const int tm = tile / n_tiles, tn = tile - tm * n_tiles;
        char* base = out + ((size_t)tm * 256 * N + tn * 256) * 2;
        // wave (wm, wn2) owns rows [128 wm, +128) x columns [128 wn2, +128): 128 x 128 x 2 B = 32 KiB = 32 instructions of 1 KiB
        if (MODE == 0) {
            const int c = lane & 15, g = lane >> 4;
            for (int h = 0; h < 2; ++h)
                for (int p = 0; p < 2; ++p)
                    for (int mi = 0; mi < 8; ++mi) {
                        char* a = base + ((size_t)(128 * wm + 16 * mi + c) * N + 128 * wn2 + 64 * h + 32 * p + 8 * g) * 2;
                        if (NT) __builtin_nontemporal_store(v, (u32x4*)a); else *(u32x4*)a = v;
                    }
        } else if (MODE == 1) {
            const int r = lane >> 3, ch = lane & 7;
            for (int h = 0; h < 2; ++h)
                for (int i = 0; i < 16; ++i) {
                    char* a = base + ((size_t)(128 * wm + 8 * i + r) * N + 128 * wn2 + 64 * h) * 2 + ch * 16;
                    if (NT) __builtin_nontemporal_store(v, (u32x4*)a); else *(u32x4*)a = v;
                }
        } else {
            // two whole 512-B tile rows per instruction; the wave writes rows [64 wave, +64)
            const int r = lane >> 5, ch = lane & 31;
            for (int i = 0; i < 32; ++i) {
                char* a = base + ((size_t)(64 * wave + 2 * i + r) * N) * 2 + ch * 16;
                if (NT) __builtin_nontemporal_store(v, (u32x4*)a); else *(u32x4*)a = v;
            }
        }
    }
}

template <int MODE, int NT>
void run(const char* name, char* out, int N) {
    const int n_tiles = N / 256, tiles = 128 * n_tiles, per = tiles / 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) burst<MODE, NT><<<256, 256>>>(out, N, n_tiles, per);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) burst<MODE, NT><<<256, 256>>>(out, N, n_tiles, per);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 32768.0 * N * 2;
    printf("%-64s %7.1f us per %5.1f MB = %5.2f TB/s  (%.2f us per 33.5 MB round)\n", name, ms / 20 * 1e3, bytes / 1e6,
           bytes / (ms / 20 * 1e-3) / 1e12, ms / 20 * 1e3 / per);
}

int main() {
    const int N = 3072;
    char* out; hipMalloc(&out, (size_t)32768 * N * 2);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0>("A 16 rows x 64 B per instruction, plain", out, N);
        run<0, 1>("A 16 rows x 64 B per instruction, non-temporal (product)", out, N);
        run<1, 0>("B  8 rows x 128 B per instruction, plain", out, N);
        run<1, 1>("B  8 rows x 128 B per instruction, non-temporal", out, N);
        run<2, 0>("C  2 rows x 512 B per instruction, plain", out, N);
        run<2, 1>("C  2 rows x 512 B per instruction, non-temporal", out, N);
    }
    return 0;
}
