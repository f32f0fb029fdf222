// Developer micro-benchmark: what do the instruction mixes of a ONE-WAVE sequential loop (solo_kernels.hpp's sequencer) cost?
// One wave per CU slot, dependent chains; prints shader cycles per iteration of each mix.
// build: hipcc -O3 --offload-arch=gfx950 -o /tmp/one_wave_rates tools/micro/one_wave_rates.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

constexpr int N = 4096;

__global__ void k(unsigned long long *out, float *sink, int waves) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float v = lane * 0.5f + 1.0f;
    uint32_t u = lane * 2654435761u;
    lds[threadIdx.x] = u;
    __syncthreads();
    if (wave >= waves) return;
    unsigned long long t0, t1;
    // A: dependent VALU chain
    t0 = clock64();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[0] = t1 - t0;
    // B: readlane -> VALU with the SGPR -> readlane ...
    t0 = clock64();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
            v = v + s;
        }
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[1] = t1 - t0;
    // C: v_cmp -> ballot -> popcount -> VALU
    t0 = clock64();
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int c = __popcll(__ballot(v > 3.0f + j));
            v = v + c;
        }
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[2] = t1 - t0;
    // D: a data-dependent uniform branch per step (taken half the time)
    t0 = clock64();
    int acc = 0;
    for (int i = 0; i < N; i++) {
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const int c = __builtin_amdgcn_readfirstlane(u >> (j + (i & 3))) & 1;
            if (c) acc += 3;
            else acc ^= 5;
            u = u * 1664525u + acc;
        }
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[3] = t1 - t0;
    // E: dependent LDS read chain
    t0 = clock64();
    uint32_t idx = lane;
    for (int i = 0; i < N; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) idx = lds[idx & 63] & 63;
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[4] = t1 - t0;
    // F: the sequencer's buffer insertion (ballot rank, two DPP shifts, selects, bit-mask update), straight-line
    t0 = clock64();
    float bd = v;
    uint32_t bi = u;
    uint64_t bun = 0x5555;
    int nb = 20;
    for (int i = 0; i < N; i++) {
        const float dj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bd), i & 15));
        const uint32_t idj = __builtin_amdgcn_readlane(bi, i & 7);
        const int r = __popcll(__ballot(lane < nb && bd <= dj));
        const float sd = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(bd), __float_as_int(bd), 0x138, 0xF, 0xF, false));
        const uint32_t si = __builtin_amdgcn_update_dpp((int)bi, (int)bi, 0x138, 0xF, 0xF, false);
        bd = lane > r ? sd : (lane == r ? dj : bd);
        bi = lane > r ? si : (lane == r ? idj : bi);
        const uint64_t lowm = (1ull << r) - 1ull;
        bun = (bun & lowm) | ((bun & ~lowm) << 1) | (1ull << r);
        nb = (nb + 1) & 31;
    }
    t1 = clock64();
    if (lane == 0 && wave == 0) out[5] = t1 - t0;
    sink[threadIdx.x] = v + u + acc + idx + bd + bi + (float)bun;
}

int main() {
    unsigned long long *d, h[8];
    float *s;
    hipMalloc(&d, 64);
    hipMalloc(&s, 4096);
    for (int waves = 1; waves <= 4; waves += 3) {
        hipMemset(d, 0, 64);
        k<<<1, 256, 1024>>>(d, s, waves);
        hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
        printf("%d active wave(s) in the workgroup: cycles per step -- A dependent fma %.1f, B readlane+add %.1f, C cmp/ballot/popc+add %.1f, "
               "D branchy step %.1f, E dependent ds_read %.1f, F buffer insertion %.1f\n", waves,
               h[0] / (8.0 * N), h[1] / (8.0 * N), h[2] / (8.0 * N), h[3] / (8.0 * N), h[4] / (8.0 * N), h[5] / (1.0 * N));
    }
    return 0;
}
