// tile_kernels.hpp -- the batched form of the distance scan for gfx950: when MANY queries share the
// same rows (a large query batch against IVF lists, every base row against the centroid table in
// k-means assignment, ground-truth kNN), the per-(query,row) GEMV of scan_kernel re-reads each row once
// per query.  Here a workgroup keeps a group of up to 32 queries resident in LDS, streams the rows
// through LDS once, and lets the matrix cores do the reduction over D:
//     v_mfma_f32_32x32x2_f32  (f32 in, f32 accumulate: exact f32, bit-for-bit a k-ordered fmaf chain --
//     cdna_hip_programming.md "FP32-input MFMA"), one 32-row x 32-query tile per wave.
// This is a real reuse win, not a reshaping for its own sake: the row bytes fetched from HBM drop by
// the group size, and the dot products need no cross-lane shuffles at all.
//
// Numeric contract (oracle/oracle.c "MFMA order" mimics it): for a (row, query) pair the accumulator
// is one f32 fmaf chain over k in the order 8t+j, 8t+4+j (j = 0..3, t ascending) -- lane half 0 of an
// MFMA carries k = 8t+j, half 1 carries k = 8t+4+j, and the instruction adds half 0's product first.
// Norms are the same sqrt(reduce(v.v)) values as everywhere else (row_norms_kernel).
// Cosine and dot only: rooted L2 from |q|^2+|v|^2-2q.v would lose the exact zeros the reference's
// tests pin, so L2 batches run l2_group_kernel (l2_kernels.hpp: same groups and outputs, GEMV arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hg {

constexpr int kTileWaves = 8;    // waves per workgroup (two per SIMD), each owns 32 rows of a tile
constexpr int kTileThreads = kTileWaves * 64;
constexpr int kTileRows = 32 * kTileWaves;  // rows per tile
constexpr int kTileQ = 32;      // MFMA N: queries per group
constexpr int kTileK = 32;      // K per staging step
constexpr int kTileLdA = kTileK + 4;  // padded LDS row of the A (rows) tile: conflict-free ds_read_b128
constexpr int kL2MaxDim = 1024;      // register-row group kernel (l2_kernels.hpp): 32 resident queries x 4 KiB
constexpr int kTileMaxDim = 3072;     // rows up to this length are supported by the callers' row loaders

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The query group is resident in LDS one K-PHASE at a time: all of K when a row has at most 28 K-steps (dim <= 896),
// otherwise phases of 24 K-steps (768 columns, the 96 KiB a 32-query group may take).  A tile's accumulators
// live across its phases, so the f32 chain over k is the same one whatever the phase length.  (Before, dims
// above 896 / 1792 kept only 16 / 8 queries resident and left half / three quarters of every 32-column MFMA empty.)
__host__ __device__ inline int tile_phase_steps(int dim) {
    const int nk = (dim + kTileK - 1) / kTileK;
    return nk <= 28 ? nk : 24;
}
__host__ __device__ inline int tile_ldq(int dim) {  // LDS row of one query: the phase's columns (+4: bank spread)
    const int cols = tile_phase_steps(dim) * kTileK;
    return ((cols + 63) / 64) * 64 + 4;
}
// queries resident per workgroup = the MFMA tile's 32 columns, for every supported dim
__host__ __device__ inline int tile_tq(int dim) { return kTileQ; }
// chunks a segment of `rows` rows is cut into: round(tiles / tiles_per_chunk), at least 1, at most max_chunks
__host__ __device__ inline int64_t tile_nchunks(int64_t rows, int64_t chunk_rows, int64_t max_chunks) {
    const int64_t tiles = (rows + kTileRows - 1) / kTileRows, tpc = chunk_rows / kTileRows;
    int64_t nch = (tiles + tpc / 2) / tpc;
    return nch < 1 ? 1 : (nch > max_chunks ? max_chunks : nch);
}
__host__ inline size_t tile_lds_bytes(int dim) {
    return sizeof(float) * (static_cast<size_t>(tile_tq(dim)) * tile_ldq(dim) + kTileRows * kTileLdA) +
           sizeof(float) * kTileQ + sizeof(int64_t) * kTileQ + sizeof(float) * kTileRows + sizeof(int32_t) * kTileQ + 16;
}

struct GroupMember {
    int32_t q;         // query index
    int32_t pad;
    int64_t out_base;  // distances of this (query, segment) go to out[out_base + (row - row_begin)]
};

struct TileArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld;
    int32_t dim;
    int32_t metric;
    const float *Qp;  // queries padded to stride ld (zero filled), 16-B aligned rows
    const float *q_norms;
    // explicit groups (IVF): group g -> segment seg_of[g], members [mem_begin[g], mem_begin[g] + mem_cnt[g])
    const int32_t *grp_seg;
    const int32_t *grp_mem_begin;
    const int32_t *grp_mem_cnt;
    const int32_t *ngroups;  // device scalar (grid is an upper bound)
    // explicit mode: dense work list built on the device -- workgroup b serves chunk wi_chunk[b] of group
    // wi_group[b] for b < *nitems and exits otherwise, so every idle workgroup sits at the END of the grid.
    // (Interleaved empty workgroups halve the CU occupancy: each needs the whole 136 KiB LDS slot of a CU.)
    const int32_t *wi_group;
    const int32_t *wi_chunk;
    const int32_t *nitems;
    int32_t *work_ctr;  // persistent mode: 8 per-XCD item counters (zeroed by the plan kernel); nullptr = one item per workgroup
    const GroupMember *members;
    const int64_t *seg_off;  // row range of segment s = [seg_off[s], seg_off[s+1])
    // implicit groups (assignment / exact kNN): group g = queries [tq*g, tq*g + tq), every group scans rows
    // [0, nrows_all) and writes out[q * out_stride + row]
    int64_t nrows_all;
    int32_t nq;
    int64_t out_stride;
    int32_t chunk_rows;  // multiple of kTileRows
    int32_t nchunks;
    float *out;
    // k = 1 (k-means assignment, nearest-row queries): when set, nothing dense is written; every query's
    // best make_key(distance, row) is folded in registers and merged with one 64-bit atomicMin per wave
    // (implicit groups only; the caller presets out_key[q] = ~0)
    unsigned long long *out_key;
    int32_t gemv_order;  // cosine / dot through the register-row group kernel (GEMV summation order) instead of MFMA tiles
    int32_t dbg;  // developer ablation switches (-DHG_DIAG builds: hnswgpu_debug_set_ablation); 0 in the product
    unsigned long long *dbg_buf;  // diagnostic builds only: per-workgroup {start, end, hw id, tiles} stamps
};

// Per-query top-k over a dense distance array (written by tile_scan_kernel): one wave per query.
// ord = position in the array = position in the query's concatenated candidate stream.
struct SelectArgs {
    const float *dist;     // query q's candidates are dist[q * stride .. + (q_cnt ? q_cnt[q] : cnt_all))
    const int32_t *q_cnt;
    int64_t stride;
    int64_t cnt_all;
    int32_t nq, k;
    int32_t wpq;  // waves per query (set by launch_select)
    int32_t vec4;  // candidates read as float4 (set by launch_select when every query's array is 16-B aligned)
    uint32_t *out_ord;
    float *out_dist;
};

}  // namespace hg
