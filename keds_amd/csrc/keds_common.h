// Shared device/host helpers for libkeds_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/keds_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define KEDS_WAVE 64

// ---- error plumbing (host) ---------------------------------------------------------
void keds_set_error(const char* fmt, ...);
int keds_check_launch(const char* what);

#define KEDS_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            keds_set_error(__VA_ARGS__);         \
            return KEDS_E_ARG;                   \
        }                                        \
    } while (0)

// ---- profiling (host): event pair around a launch when enabled -----------------------
struct KedsProfScope {
    int klass;
    hipStream_t stream;
    void* slot;
    KedsProfScope(int klass, hipStream_t s);
    ~KedsProfScope();
};

static inline size_t keds_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers ----------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
    return __uint_as_float(((unsigned int)b) << 16);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D block id: blocks that the dispatcher deals to one XCD
// (b, b+8, b+16, ...) get a contiguous run of logical ids, so neighbouring tiles share an L2.
// Speed only; any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
