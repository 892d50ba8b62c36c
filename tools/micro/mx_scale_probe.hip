// Which operand bytes does lane (r, g0)'s e8m0 scale apply to in v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 x fp8)?
// A = one-hot 1.0 at (row 3, lane group g1, byte j1), B = all ones, scale of lane (3, g0) = 2.0, all others 1.0:
// D[3][0] == 2 iff that byte is scaled by lane g0's scale.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void probe(int g0, int g1, int j1, int which /*0: scale on A, 1: scale on B*/, float* out, const int* svals) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    union { i32x8 v; unsigned char b[32]; } a, b;
    for (int j = 0; j < 32; ++j) { a.b[j] = 0; b.b[j] = 0x38; }          // 0x38 = 1.0 in e4m3
    if (which == 0) { if (r == 3 && g == g1) a.b[j1] = 0x38; }
    else { for (int j = 0; j < 32; ++j) { a.b[j] = 0x38; b.b[j] = 0; } if (r == 3 && g == g1) b.b[j1] = 0x38; }
    int sa = svals[l], sb = svals[l];            // 127 from memory (a literal operand is mis-read as ~0)
    if (r == 3 && g == g0) { if (which == 0) sa = svals[64]; else sb = svals[64]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a.v, b.v, c, 0, 0, 0, sa, 0, sb);
    // which==0: D[3][0] -> lane with 4*g+i == 3 -> g = 0, i = 3, col r = 0 ; which==1: D[0][3] -> g=0,i=0, r=3
    if (which == 0 && l == 0) out[0] = c[3];
    if (which == 1 && l == 3) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 4);
    int hs[65]; for (int i = 0; i < 64; ++i) hs[i] = 127; hs[64] = 128;
    int* ds; hipMalloc(&ds, sizeof hs); hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        printf("%s operand: rows = lane group g1 of the data byte, columns = byte j1 (0..31); entry = lane group g0 whose scale applies\n", which ? "B" : "A");
        for (int g1 = 0; g1 < 4; ++g1) {
            printf("  g1=%d: ", g1);
            for (int j1 = 0; j1 < 32; ++j1) {
                int found = -1;
                for (int g0 = 0; g0 < 4; ++g0) {
                    probe<<<1, 64>>>(g0, g1, j1, which, d, ds);
                    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
                    if (h == 2.f) found = found < 0 ? g0 : 9;
                }
                printf("%d", found);
            }
            printf("\n");
        }
    }
    return 0;
}
