// Developer micro-benchmark: does LDS / global traffic overlap with an f32 MFMA chain on gfx950?
// 8 waves per workgroup (two per SIMD), one workgroup per CU, per "K-step" 16 x v_mfma_f32_32x32x2_f32 and,
// by mode bit: 1 = 8 ds_read_b128 feeding the NEXT step's operands, 2 = 4 ds_write_b128, 4 = 4 global_load_dwordx4
// from an L2-resident table (stored to LDS by the writes of a later step when bit 2 is set).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_overlap.hip -o build_dbg/mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, bool MFMA, int NACC, int PRIO>
__global__ __launch_bounds__(1024) void k(const float4 *tab, int tab_mask, float *out, int steps) {
    extern __shared__ float4 lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 *slab = lds + wave * 320;  // 32 rows x 36 floats = 288 float4 (padded to 320)
    for (int i = lane; i < 320; i += 64) slab[i] = make_float4(1.f, 0.5f, 0.25f, 0.125f);
    __syncthreads();
    f32x16 acc, acc2;
    for (int i = 0; i < 16; i++) acc[i] = acc2[i] = 0.f;
    float4 a[4], b[4], na[4], nb[4], st[4], st2[4];
    for (int t = 0; t < 4; t++) {
        a[t] = slab[(lane & 31) * 9 + (lane >> 5) + 2 * t];
        b[t] = a[t];
        st[t] = make_float4(1.f, 1.f, 1.f, 1.f);
        st2[t] = st[t];
    }
    // every wave instruction reads 1 KiB contiguous (64 lanes x 16 B) at a pseudo-random place of the table
    unsigned idx = (blockIdx.x * 16 + wave) * 977u * 64u + lane;
    for (int s = 0; s < steps; s++) {
        if (MFMA) {
#pragma unroll
            for (int t = 0; t < 1; t++) {
                if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[t].x, acc, 0, 0, 0);
                if (NACC == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[t].y, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[t].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[t].z, acc, 0, 0, 0);
                if (NACC == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[t].w, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[t].w, acc, 0, 0, 0);
                if (PRIO) __builtin_amdgcn_s_setprio(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 2) {
#pragma unroll
            for (int t = 0; t < ((MODE & 8) ? 2 : 4); t++) slab[(lane >> 3) * 9 + (lane & 7) + 72 * t] = st[t];
        }
        if (MODE & 1) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                na[t] = slab[(lane & 31) * 9 + (lane >> 5) + 2 * t];
                nb[t] = slab[((lane + 1) & 31) * 9 + (lane >> 5) + 2 * t];
            }
        }
        if (MODE & 4) {
#pragma unroll
            for (int t = 0; t < ((MODE & 8) ? 2 : 4); t++) {
                st[t] = st2[t];
                st2[t] = tab[(idx + 64 * 131 * t) & tab_mask];
            }
            idx += 7919u * 64u;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MFMA) {
            if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
#pragma unroll
            for (int t = 1; t < 4; t++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].x, b[t].x, acc, 0, 0, 0);
                if (NACC == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[t].y, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].y, b[t].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].z, b[t].z, acc, 0, 0, 0);
                if (NACC == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[t].w, acc2, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t].w, b[t].w, acc, 0, 0, 0);
            }
            if (PRIO) __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE & 1) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                a[t] = na[t];
                b[t] = nb[t];
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc[i] + acc2[i];
    for (int t = 0; t < 4; t++) s += st[t].x + st2[t].y + a[t].x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, bool MFMA, int NACC, int PRIO>
void run(const float4 *tab, int mask, float *out, int waves) {
    const int steps = 2048 * 8 / waves, wgs = 256 * 8;   // same MFMA count for every wave count
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, MFMA, NACC, PRIO>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    k<MODE, MFMA, NACC, PRIO><<<wgs, waves * 64, 140 * 1024>>>(tab, mask, out, steps);
    (void)hipEventRecord(e0);
    k<MODE, MFMA, NACC, PRIO><<<wgs, waves * 64, 140 * 1024>>>(tab, mask, out, steps);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    double flop = MFMA ? 2.0 * 32 * 32 * 2 * 16.0 * steps * waves * wgs : 0.0;
    double gb = (MODE & 4) ? ((MODE & 8) ? 2.0 : 4.0) * 1024 * steps * 1.0 * waves * wgs / 1e9 : 0.0;
    printf("waves/WG %2d acc %d prio %d | mfma %d lds-read %d lds-write %d global %d : %8.3f ms  %6.1f TFLOP/s  %6.2f TB/s loaded\n", waves, NACC,
           PRIO, MFMA, MODE & 1, (MODE >> 1) & 1, (MODE >> 2) & 1, ms, flop / ms / 1e9, gb / ms);
}

int main() {
    const int n4 = 3 << 16;  // 3 MiB table (fits every XCD's 4 MiB L2, like the centroid table of an assignment pass)
    float4 *tab;
    float *out;
    (void)hipMalloc(&tab, sizeof(float4) * n4);
    (void)hipMemset(tab, 0, sizeof(float4) * n4);
    (void)hipMalloc(&out, sizeof(float) * 256 * 8 * 1024);
    const int mask = (1 << 17) - 1;  // the low 2 MiB of it (a power of two for the index mask)
    run<0, true, 1, 0>(tab, mask, out, 8);
    run<7, true, 1, 0>(tab, mask, out, 8);
    run<15, true, 1, 0>(tab, mask, out, 8);   // half the global loads and LDS stores per MFMA
    run<15, true, 2, 0>(tab, mask, out, 8);
    run<15, true, 2, 2>(tab, mask, out, 8);
    run<14, true, 2, 0>(tab, mask, out, 8);   // ... and no LDS reads
    run<7, true, 1, 2>(tab, mask, out, 8);
    run<7, false, 1, 0>(tab, mask, out, 8);
    return 0;
}
