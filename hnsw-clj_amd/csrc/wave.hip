// wave.hip -- launcher of the one-wave-per-query HNSW traversal of large launches (wave_kernels.hpp).
// Reference: search-layer-ultra / search-knn, src/hnsw/ultra_fast.clj:151-212, 346-374 (the same traversal, same results).
#include "engine.hpp"
#include "wave_kernels.hpp"

namespace hg {

int launch_hnsw_wave(hnswgpu_index *idx, const HnswArgs &a, int grid, size_t lds, bool vg, hipStream_t st) {
    const int nch = idx->nch;
    const bool l2 = a.metric == METRIC_L2;
#define CALL_W(N, R, L, V)                                                                                   \
    do {                                                                                                     \
        if (lds > 48 * 1024)                                                                                 \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_wave_kernel<N, R, L, V>),        \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
        hipLaunchKernelGGL((hnsw_wave_kernel<N, R, L, V>), dim3(grid), dim3(kWave), lds, st, a);             \
    } while (0)
#define CALL_WV(N, R, L)               \
    do {                               \
        if (vg) CALL_W(N, R, L, true); \
        else CALL_W(N, R, L, false);   \
    } while (0)
    // f32 rows in flight per wave: as hnsw.hip's CALLR -- with the rejection test a hop fetches f32 rows for a handful of
    // neighbours only
#define CALL_WR(N, R, RF)                   \
    do {                                    \
        if (a.qrows) {                      \
            if (l2) CALL_WV(N, RF, true);   \
            else CALL_WV(N, RF, false);     \
        } else {                            \
            if (l2) CALL_WV(N, R, true);    \
            else CALL_WV(N, R, false);      \
        }                                   \
    } while (0)
    switch (nch) {
        case 1: CALL_WR(1, 8, 4); break;
        case 2: CALL_WR(2, 8, 4); break;
        case 3: CALL_WR(3, 8, 2); break;
        case 4: CALL_WR(4, 4, 4); break;
        case 6: CALL_WR(6, 4, 4); break;
        case 8: CALL_WR(8, 2, 2); break;
        case 12: CALL_WR(12, 2, 2); break;
        default: set_error("unsupported row length"); return HNSWGPU_ELIMIT;
    }
#undef CALL_WR
#undef CALL_WV
#undef CALL_W
    HG_HIP(hipGetLastError());
    return 0;
}

}  // namespace hg
