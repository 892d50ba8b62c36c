// Sustained MFMA rate of the whole chip: every SIMD runs WAVES_PER_SIMD waves of back-to-back
// v_mfma_f32_16x16x32_bf16 on register operands (no memory).  Gives the practical matrix peak (clock under load).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int NACC>
__global__ __launch_bounds__(512) void mfma_kernel(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(float)(threadIdx.x + j); b[j] = (__bf16)(float)(j + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1.2345f) out[0] = s;
}

template <int NACC>
void run(int waves_per_wg, const char* name, float* out) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mfma_kernel<NACC><<<256, waves_per_wg * 64>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mfma_kernel<NACC><<<256, waves_per_wg * 64>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * waves_per_wg * iters * NACC * 16384.0;
    const double tf = flops / (ms * 1e-3) / 1e12;
    // one 16x16x32 bf16 MFMA occupies a SIMD's matrix pipe for 16 cycles (8 passes of 2... per the guide: 1017 flop/clk/SIMD)
    const double clk = flops / (256.0 * 4 * 1024.0) / (ms * 1e-3) / 1e9;
    printf("%-40s %8.2f ms  %7.1f TFLOP/s   -> %.2f GHz if the pipe is 100%% busy\n", name, ms, tf, clk);
}

int main() {
    float* out; hipMalloc(&out, 64);
    run<8>(4, "4 waves/CU (1 per SIMD), 8 accumulators", out);
    run<8>(8, "8 waves/CU (2 per SIMD), 8 accumulators", out);
    run<16>(8, "8 waves/CU, 16 accumulators", out);
    run<4>(8, "8 waves/CU, 4 accumulators", out);
    run<8>(8, "repeat: 8 waves/CU, 8 accumulators", out);
    return 0;
}
