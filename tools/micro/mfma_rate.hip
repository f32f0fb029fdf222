// Developer micro-benchmark: issue rate of v_mfma_f32_32x32x2_f32 chains on gfx950 as a function of waves per SIMD
// and independent accumulators per wave (the tile kernel runs 2 waves per SIMD with ONE accumulator chain each).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_rate.hip -o build_dbg/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void chain(float *out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++)
        for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; j++)
        for (int i = 0; i < 16; i++) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int waves_per_wg, int wgs) {
    float *out;
    hipMalloc(&out, sizeof(float) * wgs * waves_per_wg * 64);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    chain<NACC><<<wgs, waves_per_wg * 64>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e0);
    chain<NACC><<<wgs, waves_per_wg * 64>>>(out, iters, 1.0f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = 2.0 * 32 * 32 * 2 * 16.0 * iters * waves_per_wg * wgs;
    printf("acc/wave %d  waves/WG %d  WGs %d: %.3f ms  %.1f TFLOP/s\n", NACC, waves_per_wg, wgs, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    run<1>(4, 256);
    run<1>(8, 256);
    run<1>(16, 256);
    run<2>(4, 256);
    run<2>(8, 256);
    run<4>(4, 256);
    run<1>(8, 2560);
    return 0;
}
