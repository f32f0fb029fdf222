// Developer micro-benchmark + check: the k smallest of <= 1024 keys by one 256-thread workgroup -- hg::topk_small_wg (now
// histograms, topk_hist_wg) against the bisection it replaces and against a CPU sort, on distributions that stress it: uniform, an outlier plus a concentrated rest (centroid
// distances of clustered data), hundreds of equal distances, fewer keys than k, all-ones holes.
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o /tmp/topk_select tools/micro/topk_select.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../hnsw-clj_amd/csrc/kernels.hpp"
using namespace hg;

template <bool OLD>
__global__ __launch_bounds__(256) void sel(const uint64_t *keys, int n, int k, uint64_t *out, unsigned long long *stamps, int reps) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    const uint64_t *in = keys + static_cast<int64_t>(blockIdx.x) * n;
    unsigned long long t0 = wall_clock64();
    const uint64_t *fin = nullptr;
    for (int r = 0; r < reps; r++) {
        fin = topk_small_wg(n, k, lists, lists + kNWave * k, lists + (kNWave + 1) * k,
                            [&](int64_t i) { return __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }, !OLD);
        __syncthreads();
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x < 64 && fin && threadIdx.x < k) out[blockIdx.x * k + threadIdx.x] = fin[threadIdx.x];
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t1 - t0;
}

static uint64_t mk(float d, uint32_t pos) {
    uint32_t u;
    memcpy(&u, &d, 4);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (static_cast<uint64_t>(u) << 32) | pos;
}

int main() {
    const int nb = 64;
    uint64_t *dk, *dout;
    unsigned long long *dst;
    hipMalloc(&dk, sizeof(uint64_t) * nb * 1024);
    hipMalloc(&dout, sizeof(uint64_t) * nb * 64);
    hipMalloc(&dst, 64);
    int bad = 0;
    const char *names[] = {"uniform", "outlier + concentrated", "300 equal distances", "fewer than k", "holes", "all equal", "two values"};
    for (int dist = 0; dist < 7; dist++)
        for (int n : {1024, 973, 320, 40, 7})
            for (int k : {10, 32, 64, 1}) {
                std::vector<uint64_t> h(static_cast<size_t>(nb) * n);
                srand(dist * 1000 + n + k);
                for (int b = 0; b < nb; b++)
                    for (int i = 0; i < n; i++) {
                        float d;
                        const float u = rand() / (float)RAND_MAX;
                        switch (dist) {
                            case 0: d = u; break;
                            case 1: d = i == 17 ? 0.04f : 0.95f + 0.1f * u; break;
                            case 2: d = i < 300 ? 0.5f : 0.4f + 0.2f * u; break;
                            case 3: d = u; break;
                            case 4: d = u; break;
                            case 5: d = 0.25f; break;
                            default: d = (rand() & 1) ? 0.5f : 0.75f; break;
                        }
                        uint64_t key = mk(d, i);
                        if (dist == 3 && i >= 5) key = ~0ull;
                        if (dist == 4 && (rand() % 3) == 0) key = ~0ull;
                        h[static_cast<size_t>(b) * n + i] = key;
                    }
                hipMemcpy(dk, h.data(), sizeof(uint64_t) * h.size(), hipMemcpyHostToDevice);
                std::vector<uint64_t> got[2];
                double us[2];
                for (int v = 0; v < 2; v++) {
                    hipMemset(dout, 0xff, sizeof(uint64_t) * nb * 64);
                    const size_t lds = sizeof(uint64_t) * (2 * kNWave + 1) * k;
                    if (v == 0) hipLaunchKernelGGL(sel<false>, dim3(nb), dim3(256), lds, 0, dk, n, k, dout, dst, 1);
                    else hipLaunchKernelGGL(sel<true>, dim3(nb), dim3(256), lds, 0, dk, n, k, dout, dst, 1);
                    hipDeviceSynchronize();
                    got[v].resize(static_cast<size_t>(nb) * k);
                    hipMemcpy(got[v].data(), dout, sizeof(uint64_t) * nb * k, hipMemcpyDeviceToHost);
                    // time: 50 selections back to back inside one workgroup
                    if (v == 0) hipLaunchKernelGGL(sel<false>, dim3(1), dim3(256), lds, 0, dk, n, k, dout, dst, 50);
                    else hipLaunchKernelGGL(sel<true>, dim3(1), dim3(256), lds, 0, dk, n, k, dout, dst, 50);
                    hipDeviceSynchronize();
                    unsigned long long t;
                    hipMemcpy(&t, dst, 8, hipMemcpyDeviceToHost);
                    us[v] = t * 0.01 / 50;
                }
                int wrong[2] = {0, 0};
                for (int b = 0; b < nb; b++) {
                    std::vector<uint64_t> ref(h.begin() + static_cast<size_t>(b) * n, h.begin() + static_cast<size_t>(b + 1) * n);
                    std::sort(ref.begin(), ref.end());
                    for (int v = 0; v < 2; v++)
                        for (int i = 0; i < k; i++) {
                            const uint64_t want = i < n ? ref[i] : ~0ull;
                            if (got[v][static_cast<size_t>(b) * k + i] != want) wrong[v]++;
                        }
                }
                bad += wrong[0] + wrong[1];
                printf("%-24s n %4d k %2d   histograms %6.2f us   bisection %6.2f us   wrong %d / %d\n", names[dist], n, k, us[0], us[1], wrong[0], wrong[1]);
            }
    printf(bad ? "FAILED\n" : "all selections equal the CPU sort\n");
    return bad ? 1 : 0;
}
