// Developer micro-test: does v_mfma_f32_16x16x4_f32, fed k in the order (8t+0, 8t+4, 8t+1, 8t+5 | 8t+2, 8t+6, 8t+3, 8t+7),
// reproduce bit for bit the f32 chain of v_mfma_f32_32x32x2_f32 fed (8t+j | 8t+4+j), j = 0..3 -- i.e. is its internal
// accumulation order k-slot 0, 1, 2, 3?  32 rows x 16 columns x K, random data.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma16_order.hip -o build_dbg/mfma16_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 96;

__global__ void k32(const float *A, const float *B, float *D) {  // A[32][K], B[32][K] (B rows = columns j), D[32][32]
    const int lane = threadIdx.x, li = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    for (int t = 0; t < K / 8; t++)
        for (int j = 0; j < 4; j++) {
            const int k = 8 * t + 4 * half + j;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[li * K + k], B[li * K + k], acc, 0, 0, 0);
        }
    for (int reg = 0; reg < 16; reg++) D[((reg & 3) + 8 * (reg >> 2) + 4 * half) * 32 + li] = acc[reg];
}

__global__ void k16(const float *A, const float *B, float *D) {  // rows 0..31 x columns 0..15
    const int lane = threadIdx.x, i = lane & 15, s = lane >> 4;
    f32x4 acc0, acc1;  // row blocks 0..15, 16..31
    for (int r = 0; r < 4; r++) acc0[r] = acc1[r] = 0.f;
    for (int t = 0; t < K / 8; t++)
        for (int u = 0; u < 2; u++) {
            const int k = 8 * t + 4 * (s & 1) + (s >> 1) + 2 * u;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * K + k], B[i * K + k], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(16 + i) * K + k], B[i * K + k], acc1, 0, 0, 0);
        }
    for (int r = 0; r < 4; r++) {
        D[(4 * s + r) * 32 + i] = acc0[r];
        D[(16 + 4 * s + r) * 32 + i] = acc1[r];
    }
}

int main() {
    float hA[32 * K], hB[32 * K], h32[32 * 32], h16[32 * 32];
    srand(7);
    for (int i = 0; i < 32 * K; i++) {
        hA[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
        hB[i] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
    }
    float *A, *B, *D32, *D16;
    (void)hipMalloc(&A, sizeof(hA));
    (void)hipMalloc(&B, sizeof(hB));
    (void)hipMalloc(&D32, sizeof(h32));
    (void)hipMalloc(&D16, sizeof(h16));
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    (void)hipMemset(D16, 0, sizeof(h16));
    k32<<<1, 64>>>(A, B, D32);
    k16<<<1, 64>>>(A, B, D16);
    (void)hipMemcpy(h32, D32, sizeof(h32), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h16, D16, sizeof(h16), hipMemcpyDeviceToHost);
    int diff = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 16; j++) diff += memcmp(&h32[i * 32 + j], &h16[i * 32 + j], 4) != 0;
    printf("32 x 16 outputs, K = %d: %d differ bit-wise (sample %.9g vs %.9g)\n", K, diff, h32[5 * 32 + 3], h16[5 * 32 + 3]);
    return diff != 0;
}
