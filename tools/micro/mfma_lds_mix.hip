// The 256x256 GEMM's K-step without its memory side: every wave reads its A / W fragments of a K-tile from LDS (24 ds_read_b128
// per 64-K tile, the product kernel's addressing and swizzle) and multiplies them into 128 accumulator registers -- once with
// v_mfma_f32_16x16x32_bf16 (the product kernel: 64 MFMAs per K-tile per wave, 8 issue cycles each) and once with
// v_mfma_f32_32x32x16_bf16 (32 MFMAs of twice the work: half the issue slots and half the operand-register reads per flop).
// Random data in LDS (the chip's clock under load depends on what toggles).  Answers: would the 32x32 form sustain more?
// Measured (round 2): 1.44-1.69 PFLOP/s for both forms (the register-only loop: 2.06); the plain-compiled four-wave
// 128 x 128 layout below 1.20-1.23 (its 256 accumulators wander between AGPRs and VGPRs unless pinned: gemm.hip notes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
template <int SHAPE, int F16 = 0>
__global__ __launch_bounds__(512, 2) void mix_kernel(const uint4* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // two operand buffers of 64 KiB: X | W each 32 KiB
    for (int i = threadIdx.x; i < 131072 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave & 3, wm = wave >> 2;
    float s = 0.f;
    if constexpr (SHAPE == 0) {
        const int g = lane >> 4, c = lane & 15, f = (c >> 1) & 7;
        const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
        const int xrow = (128 * wm + c) * 128, wrow = 32768 + (64 * wn + c) * 128;
        f32x4 acc[4][8];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const char* b = smem + (it & 1) * 65536;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int slot = kk ? slot1 : slot0;
                bf16x8 w[4], x[8];
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) w[ni] = *reinterpret_cast<const bf16x8*>(b + wrow + slot + ni * 2048);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) x[mi] = *reinterpret_cast<const bf16x8*>(b + xrow + slot + mi * 2048);
#pragma unroll
                for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni) {
                        if constexpr (F16) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w[ni]), __builtin_bit_cast(f16x8, x[mi]), acc[ni][mi], 0, 0, 0);
                        else acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ni], x[mi], acc[ni][mi], 0, 0, 0);
                    }
            }
        }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) s += acc[ni][mi][0] + acc[ni][mi][3];
    } else {
        // 32x32x16: lane (r = lane & 31, h = lane >> 5) holds row r, k = 8 h .. 8 h + 7 of a 16-wide k-step
        const int r = lane & 31, h = lane >> 5, f = (r >> 1) & 7;
        const int xrow = (128 * wm + r) * 128, wrow = 32768 + (64 * wn + r) * 128;
        f32x16 acc[2][4];
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ni][mi][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* b = smem + (it & 1) * 65536;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int slot = ((2 * ks + h) ^ f) << 4;
                bf16x8 w[2], x[4];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) w[ni] = *reinterpret_cast<const bf16x8*>(b + wrow + slot + ni * 4096);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) x[mi] = *reinterpret_cast<const bf16x8*>(b + xrow + slot + mi * 4096);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ni], x[mi], acc[ni][mi], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) s += acc[ni][mi][0] + acc[ni][mi][15];
    }
    if (s == 1.2345f) out[0] = s;
}

// the vendor-style layout: FOUR waves, 128 x 128 per wave (256 accumulator registers: one wave per SIMD, the whole 512-register
// file), 32 fragment reads per wave per K-tile = 128 per CU instead of 192
__global__ __launch_bounds__(256) void mix4_kernel(const uint4* __restrict__ src, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 131072 / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = src[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wn = wave & 1, wm = wave >> 1;
    const int g = lane >> 4, c = lane & 15, f = (c >> 1) & 7;
    const int slot0 = ((0 + g) ^ f) << 4, slot1 = ((4 + g) ^ f) << 4;
    const int xrow = (128 * wm + c) * 128, wrow = 32768 + (128 * wn + c) * 128;
    f32x4 acc[8][8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const char* b = smem + (it & 1) * 65536;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int slot = kk ? slot1 : slot0;
            bf16x8 w[8], x[8];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) w[ni] = *reinterpret_cast<const bf16x8*>(b + wrow + slot + ni * 2048);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) x[mi] = *reinterpret_cast<const bf16x8*>(b + xrow + slot + mi * 2048);
#pragma unroll
            for (int mi = 0; mi < 8; ++mi)
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ni], x[mi], acc[ni][mi], 0, 0, 0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) s += acc[ni][mi][0] + acc[ni][mi][3];
    if (s == 1.2345f) out[0] = s;
}

void run4(const char* name, const uint4* src, float* out) {
    const int iters = 3000;
    hipFuncSetAttribute((const void*)mix4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix4_kernel<<<256, 256, 131072>>>(src, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix4_kernel<<<256, 256, 131072>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * iters * 2.0 * 256 * 256 * 64;
    printf("%-52s %8.2f ms  %7.1f TFLOP/s  (%.0f cycles per K-tile at 2.0 GHz)\n", name, ms, flops / (ms * 1e-3) / 1e12,
           ms * 1e-3 / iters * 2.0e9);
}

template <int SHAPE, int F16 = 0>
void run(const char* name, const uint4* src, float* out) {
    const int iters = 3000;
    hipFuncSetAttribute((const void*)mix_kernel<SHAPE, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix_kernel<SHAPE, F16><<<256, 512, 131072>>>(src, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix_kernel<SHAPE, F16><<<256, 512, 131072>>>(src, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 256.0 * iters * 2.0 * 256 * 256 * 64;      // one 256 x 256 x 64 K-tile per iteration per workgroup
    printf("%-52s %8.2f ms  %7.1f TFLOP/s  (%.0f cycles per K-tile at 2.0 GHz)\n", name, ms, flops / (ms * 1e-3) / 1e12,
           ms * 1e-3 / iters * 2.0e9);
}

// operand data: what toggles sets the power, and the power cap sets the clock.  mode 0: uniform [-2, 2) in both operands;
// 1: X uniform [-2, 2), W scaled by 1/32 (weights ~ K^-0.5); 2: zeros; 3: normal(0, 1) X, normal(0, 1/32) W.  f16: the same
// VALUES rounded to fp16 instead of bf16.
static float frand() { return (rand() & 0xFFFFFF) / 16777216.0f; }
static float nrand() { float u = frand() + 1e-7f, v = frand(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }
static unsigned short to_bf16(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static unsigned short to_f16(float v) { _Float16 h = (_Float16)v; unsigned short r; memcpy(&r, &h, 2); return r; }
void fill(unsigned short* h, int mode, bool f16) {
    for (int buf = 0; buf < 2; ++buf)
        for (int i = 0; i < 32768; ++i) {
            const bool is_w = i >= 16384;                 // each 64 KiB buffer: X (32 KiB) | W (32 KiB)
            float v = 0.f;
            if (mode == 0) v = (frand() - 0.5f) * 4.0f;
            if (mode == 1) v = (frand() - 0.5f) * 4.0f * (is_w ? 1.0f / 32 : 1.0f);
            if (mode == 3) v = nrand() * (is_w ? 1.0f / 32 : 1.0f);
            h[buf * 32768 + i] = f16 ? to_f16(v) : to_bf16(v);
        }
}
#include <cmath>
int main() {
    uint4* src; float* out;
    hipMalloc(&src, 131072); hipMalloc(&out, 64);
    unsigned short* h = (unsigned short*)malloc(131072);
    srand(1);
    const char* modes[4] = {"uniform [-2,2) both", "X uniform [-2,2), W / 32", "zeros", "X normal(0,1), W normal(0,1/32)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            printf("-- data: %s\n", modes[mode]);
            fill(h, mode, false); hipMemcpy(src, h, 131072, hipMemcpyHostToDevice);
            run<0, 0>("16x16x32 bf16: 24 ds_read_b128 + 64 MFMA per wave K-tile", src, out);
            run<1, 0>("32x32x16 bf16: 24 ds_read_b128 + 32 MFMA per wave K-tile", src, out);
            fill(h, mode, true); hipMemcpy(src, h, 131072, hipMemcpyHostToDevice);
            run<0, 1>("16x16x32 f16 : 24 ds_read_b128 + 64 MFMA per wave K-tile", src, out);
        }
    return 0;
}
