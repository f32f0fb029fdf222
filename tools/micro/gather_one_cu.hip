// Micro-benchmark: what can ONE workgroup (one CU) gather per dependent step?  The single-query HNSW traversal is a chain
// of steps "fetch ~20 random 3 KB rows of a 96 MB table, reduce, decide": this measures the memory part of a step in
// isolation.  usage: gather_one_cu [rows_per_step] [waves] [table_MB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int RB>
__global__ __launch_bounds__(1024) void gather(const float *table, const int *ids, int rows_per_step, int steps, int ld, float *out,
                                              int dependent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float acc = 0.f;
    int off = 0;
    for (int s = 0; s < steps; s++) {
        const int *row_ids = ids + (size_t)s * rows_per_step;
        for (int j0 = wave * RB; j0 < rows_per_step; j0 += nw * RB) {
            float4 r[RB][3];
#pragma unroll
            for (int x = 0; x < RB; x++) {
                if (j0 + x < rows_per_step) {
                    const int id = (row_ids[j0 + x] + off) & 0x7fff;  // 32768-row table window
                    const float4 *rp = reinterpret_cast<const float4 *>(table + (size_t)id * ld);
                    r[x][0] = rp[lane];
                    r[x][1] = rp[64 + lane];
                    r[x][2] = rp[128 + lane];
                }
            }
#pragma unroll
            for (int x = 0; x < RB; x++)
                if (j0 + x < rows_per_step) acc += r[x][0].x + r[x][1].y + r[x][2].z;
        }
        if (dependent) {  // the next step's addresses depend on this step's data (as a traversal's do)
            float v = acc;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
            off = (__float_as_int(v) >> 3) & 1;
            __syncthreads();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// The same gather with LDS-direct loads (global_load_lds_dwordx4: no VGPR destination): does a lone CU keep more bytes in
// flight that way?  Each wave lands its rows in its own LDS area (RB rows x 3 KB), waits, and reads them back.
template <int RB>
__global__ __launch_bounds__(1024) void gather_lds(const float *table, const int *ids, int rows_per_step, int steps, int ld, float *out,
                                                   int dependent) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float *area = reinterpret_cast<float *>(smem) + (size_t)wave * RB * 768;
    float acc = 0.f;
    int off = 0;
    for (int s = 0; s < steps; s++) {
        const int *row_ids = ids + (size_t)s * rows_per_step;
        for (int j0 = wave * RB; j0 < rows_per_step; j0 += nw * RB) {
#pragma unroll
            for (int x = 0; x < RB; x++) {
                if (j0 + x < rows_per_step) {
                    const int id = (row_ids[j0 + x] + off) & 0x7fff;
                    const float *rp = table + (size_t)id * ld;
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        __builtin_amdgcn_global_load_lds(rp + c * 256 + lane * 4, (__attribute__((address_space(3))) void *)(area + x * 768 + c * 256),
                                                         16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int x = 0; x < RB; x++)
                if (j0 + x < rows_per_step) {
                    const float4 a = *reinterpret_cast<const float4 *>(area + x * 768 + lane * 4);
                    const float4 b = *reinterpret_cast<const float4 *>(area + x * 768 + 256 + lane * 4);
                    const float4 c = *reinterpret_cast<const float4 *>(area + x * 768 + 512 + lane * 4);
                    acc += a.x + b.y + c.z;
                }
        }
        if (dependent) {
            float v = acc;
            for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
            off = (__float_as_int(v) >> 3) & 1;
            __syncthreads();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// optional background load: does the lone workgroup's memory latency depend on how busy the rest of the chip is?
__global__ void stream_bg(const float4 *src, size_t n4, float *out, int iters) {
    float acc = 0.f;
    for (int it = 0; it < iters; it++)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) acc += src[i].x;
    if (acc == 1.2345f) out[0] = acc;
}

int main(int argc, char **argv) {
    const int rps = argc > 1 ? atoi(argv[1]) : 20, waves = argc > 2 ? atoi(argv[2]) : 4, mb = argc > 3 ? atoi(argv[3]) : 96;
    const int ld = 768, steps = 2000;
    const size_t nrows = (size_t)mb * 1024 * 1024 / (ld * 4);
    float *table, *out;
    int *ids;
    CK(hipMalloc(&table, nrows * ld * 4));
    CK(hipMemset(table, 0, nrows * ld * 4));
    CK(hipMalloc(&out, 4));
    std::vector<int> h((size_t)steps * rps);
    unsigned s = 1;
    const int seq = argc > 5 ? atoi(argv[5]) : 0;  // 1: consecutive rows instead of random ones
    size_t cnt = 0;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = seq ? (int)(cnt++ % (nrows < 32768 ? nrows : 32768)) : (int)((s >> 8) % (nrows < 32768 ? nrows : 32768)); }
    CK(hipMalloc(&ids, h.size() * 4));
    CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int bg = argc > 4 ? atoi(argv[4]) : 0;  // background workgroups streaming a 1 GB buffer meanwhile
    float4 *big = nullptr;
    hipStream_t st2;
    CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    if (bg) {
        CK(hipMalloc(&big, 1ull << 30));
        CK(hipMemset(big, 0, 1ull << 30));
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int dep = 0; dep < 2; dep++)
        for (int rep = 0; rep < 2; rep++) {
            if (bg) hipLaunchKernelGGL(stream_bg, dim3(bg), dim3(256), 0, st2, big, (size_t)(1ull << 30) / 16, out, 40);
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(gather<8>, dim3(1), dim3(waves * 64), 0, 0, table, ids, rps, steps, ld, out, dep);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (bg) CK(hipStreamSynchronize(st2));
            if (rep && !bg && waves <= 6) {  // the LDS-direct form: 8 rows x 3 KB per wave of LDS
                hipFuncSetAttribute(reinterpret_cast<const void *>(&gather_lds<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(gather_lds<8>, dim3(1), dim3(waves * 64), (size_t)waves * 8 * 768 * 4, 0, table, ids, rps, steps, ld, out, dep);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms2;
                CK(hipEventElapsedTime(&ms2, e0, e1));
                printf("   LDS-direct loads (global_load_lds_dwordx4): %.2f us per step = %.1f GB/s\n", ms2 * 1e3 / steps, rps * 3072.0 * steps / (ms2 * 1e-3) / 1e9);
            }
            if (rep) printf("[bg %d%s] %s steps, %d rows x 3 KB per step, %d waves, %d MB table: %.2f us per step = %.1f GB/s on one CU\n",
                            bg, seq ? " seq" : "", dep ? "dependent" : "independent", rps, waves, mb, ms * 1e3 / steps, rps * 3072.0 * steps / (ms * 1e-3) / 1e9);
        }
    return 0;
}
