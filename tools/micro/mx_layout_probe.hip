// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3) operand / scale lane maps with exact data.
// Hypothesis: lane l holds A[row l&15][k = 32*(l>>4) + j] (byte j of its 32-byte operand), B[k = 32*(l>>4)+j][col l&15];
// its scale VGPR byte 0 (e8m0) scales exactly those 32 k of that row / column; D as the bf16 16x16 forms.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void probe(const unsigned char* A /*[16][128]*/, const unsigned char* B /*[128][16]*/, const unsigned char* sA /*[16][4]*/,
                      const unsigned char* sB /*[16][4]*/, float* D /*[16][16]*/) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    union { i32x8 v; unsigned char b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        a.b[j] = A[r * 128 + 32 * g + j];
        b.b[j] = B[(32 * g + j) * 16 + r];
    }
    const int sa = sA[r * 4 + g], sb = sB[r * 4 + g];
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];   // row = 4*(l>>4)+i, col = l&15
}

static unsigned char enc_e4m3(float v) {   // exact for the small integers / halves used here
    if (v == 0) return 0;
    unsigned char s = v < 0 ? 0x80 : 0;
    float a = fabsf(v);
    int e = (int)floorf(log2f(a));
    float m = a / ldexpf(1.f, e) - 1.f;   // [0,1)
    int mi = (int)lrintf(m * 8);
    int be = e + 7;
    if (be <= 0) { mi = (int)lrintf(a / ldexpf(1.f, -9)); return s | mi; }
    return s | (be << 3) | mi;
}
static float dec_e4m3(unsigned char b) {
    int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}

int run(int scale_mode);
int main() { int rc = 0; for (int m = 0; m < 4; ++m) rc |= run(m); return rc; }
int run(int scale_mode) {
    unsigned char hA[16 * 128], hB[128 * 16], hsA[64], hsB[64];
    srand(7);
    const float vals[] = {0, 1, -1, 2, -2, 0.5f, 3, -3, 4, 1.5f, -0.5f, 6};
    for (int i = 0; i < 16 * 128; ++i) hA[i] = enc_e4m3(vals[rand() % 12]);
    for (int i = 0; i < 128 * 16; ++i) hB[i] = enc_e4m3(vals[rand() % 12]);
    // scale_mode 0: all 1.0; 1: A scales vary per row only; 2: A scales vary per (row, k-block); 3: both operands vary
    for (int i = 0; i < 64; ++i) {
        hsA[i] = 127; hsB[i] = 127;
        if (scale_mode == 1) hsA[i] = 127 + ((i / 4) % 5) - 2;
        if (scale_mode >= 2) hsA[i] = 127 + (rand() % 5) - 2;
        if (scale_mode == 3) hsB[i] = 127 + (rand() % 5) - 2;
    }
    unsigned char *dA, *dB, *dsA, *dsB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dsA, 64); hipMalloc(&dsB, 64); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipMemcpy(dsA, hsA, 64, hipMemcpyHostToDevice); hipMemcpy(dsB, hsB, 64, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dsA, dsB, dD);
    float hD[256]; hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; double maxerr = 0;
    for (int m = 0; m < 16; ++m)
        for (int n = 0; n < 16; ++n) {
            double ref = 0;
            for (int k = 0; k < 128; ++k)
                ref += (double)dec_e4m3(hA[m * 128 + k]) * ldexp(1.0, hsA[m * 4 + k / 32] - 127) * dec_e4m3(hB[k * 16 + n]) *
                       ldexp(1.0, hsB[n * 4 + k / 32] - 127);
            double e = fabs(ref - hD[m * 16 + n]);
            if (e > 1e-3) ++bad;
            if (e > maxerr) maxerr = e;
        }
    printf("scale_mode %d: ", scale_mode);
    printf("hypothesis A[row l&15][32(l>>4)+j], B[32(l>>4)+j][col l&15], scale byte0 per (row/col, k-block l>>4): %s (bad %d of 256, max err %.4g)\n",
           bad ? "WRONG" : "CONFIRMED", bad, maxerr);
    printf("D[0][0..3] = %g %g %g %g\n", hD[0], hD[1], hD[2], hD[3]);
    return bad != 0;
}
