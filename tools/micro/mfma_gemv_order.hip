// Developer micro-test: the GEMV summation order of kernels.hpp (lane l accumulates elements 256 c + 4 l + j with one fmaf
// each, the 64 lane partials are added in the xor butterfly's tree) reproduced bit for bit on the f32 matrix cores: one
// v_mfma_f32_32x32x2_f32 chain per LANE PARTIAL (K = 2 per instruction: 2 NCH instructions per partial; or v_mfma_f32_16x16x4_f32:
// K = 4, NCH instructions -- the production kernel's form, ivf_route_mfma16_kernel), the 64 partial
// tiles added pairwise in the butterfly's association (a binary counter of pending tiles).  32 rows x 32 queries x 768,
// random data, rows of very different scale, denormal products.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I hnsw-clj_amd/csrc tools/micro/mfma_gemv_order.hip -o /tmp/mfma_gemv_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include "kernels.hpp"
using namespace hg;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NCH = 3, D = 768;

__global__ void k_valu(const float *A, const float *B, float *out) {  // out[row][query]: wave per (row), loops queries
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x;
    float4 r[NCH];
    load_row<NCH>(r, A + row * D, D / 4, lane, true);
    for (int qi = 0; qi < 32; qi++) {
        float4 q[NCH];
        load_query<NCH>(q, B + qi * D, D, lane);
        const float s = wave_sum(lane_partial<NCH, false>(q, r));
        if (lane == 0) out[row * 32 + qi] = s;
    }
}

__global__ void k_mfma(const float *A, const float *B, float *out) {
    const int lane = threadIdx.x, i = lane & 31, kh = lane >> 5;
    f32x16 st[7];  // pending partial tiles, one per level of the tree
#pragma unroll
    for (int l = 0; l < 64; l++) {
        f32x16 P;
#pragma unroll
        for (int g = 0; g < 16; g++) P[g] = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const float4 a4 = *reinterpret_cast<const float4 *>(A + i * D + 256 * c + 4 * l);
            const float4 b4 = *reinterpret_cast<const float4 *>(B + i * D + 256 * c + 4 * l);
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? a4.y : a4.x, kh ? b4.y : b4.x, P, 0, 0, 0);
            P = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? a4.w : a4.z, kh ? b4.w : b4.z, P, 0, 0, 0);
        }
        int t = l, lvl = 0;
        while (t & 1) {  // (compile-time after unrolling)
#pragma unroll
            for (int g = 0; g < 16; g++) P[g] = st[lvl][g] + P[g];
            t >>= 1;
            lvl++;
        }
        st[lvl] = P;
    }
    for (int g = 0; g < 16; g++) out[((g & 3) + 8 * (g >> 2) + 4 * kh) * 32 + i] = st[6][g];
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_mfma16(const float *A, const float *B, float *out) {  // rows 0..15 x queries 0..15: the production form
    const int lane = threadIdx.x, i = lane & 15, ks = lane >> 4;
    f32x4 st[7];
#pragma unroll
    for (int l = 0; l < 64; l++) {
        f32x4 P = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < NCH; c++)
            P = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * D + 256 * c + 4 * l + ks], B[i * D + 256 * c + 4 * l + ks], P, 0, 0, 0);
        int t = l, lvl = 0;
        while (t & 1) {
            P = st[lvl] + P;
            t >>= 1;
            lvl++;
        }
        st[lvl] = P;
    }
    for (int g = 0; g < 4; g++) out[(4 * ks + g) * 32 + i] = st[6][g];
}

int main() {
    static float hA[32 * D], hB[32 * D], h1[32 * 32], h2[32 * 32];
    srand(11);
    for (int r = 0; r < 32; r++) {
        const float sc = std::exp((rand() / (float)RAND_MAX - 0.5f) * 24.0f);
        for (int k = 0; k < D; k++) {
            hA[r * D + k] = (rand() / (float)RAND_MAX - 0.5f) * sc;
            hB[r * D + k] = (rand() / (float)RAND_MAX - 0.5f) * 3.0f;
        }
    }
    for (int k = 0; k < D; k++) hA[5 * D + k] = (rand() / (float)RAND_MAX - 0.5f) * 1e-30f;   // denormal products with row 6 of B
    for (int k = 0; k < D; k++) hB[6 * D + k] = (rand() / (float)RAND_MAX - 0.5f) * 1e-12f;
    for (int k = 0; k < D; k++) hA[7 * D + k] = (k & 1) ? -0.0f : 0.0f;
    float *A, *B, *O1, *O2;
    (void)hipMalloc(&A, sizeof(hA));
    (void)hipMalloc(&B, sizeof(hB));
    (void)hipMalloc(&O1, sizeof(h1));
    (void)hipMalloc(&O2, sizeof(h2));
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    k_valu<<<32, 64>>>(A, B, O1);
    k_mfma<<<1, 64>>>(A, B, O2);
    (void)hipMemcpy(h1, O1, sizeof(h1), hipMemcpyDeviceToHost);
    (void)hipMemcpy(h2, O2, sizeof(h2), hipMemcpyDeviceToHost);
    int bad = 0, badden = 0;
    for (int r = 0; r < 32; r++)
        for (int q = 0; q < 32; q++)
            if (memcmp(&h1[r * 32 + q], &h2[r * 32 + q], 4)) {
                if (r == 5 || q == 6) badden++;
                else if (bad++ < 8) printf("differs at row %d query %d: valu %.9g mfma %.9g\n", r, q, h1[r * 32 + q], h2[r * 32 + q]);
            }
    static float h3[32 * 32];
    (void)hipMemset(O2, 0, sizeof(h2));
    k_mfma16<<<1, 64>>>(A, B, O2);
    (void)hipMemcpy(h3, O2, sizeof(h3), hipMemcpyDeviceToHost);
    int bad16 = 0;
    for (int r = 0; r < 16; r++)
        for (int q = 0; q < 16; q++)
            if (memcmp(&h1[r * 32 + q], &h3[r * 32 + q], 4)) bad16++;
    printf("v_mfma_f32_16x16x4_f32 form (rows / queries 0..15, denormal row 5 and query 6 included): %d of 256 differ\n", bad16);
    bad += bad16;
    printf("GEMV order on the matrix cores: %d of 1024 differ outside the denormal row / query; %d of 63 differ on them (valu[5][6] = %g, mfma = %g)\n",
           bad, badden, h1[5 * 32 + 6], h2[5 * 32 + 6]);
    return bad ? 1 : 0;
}
