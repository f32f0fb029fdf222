// engine.hip -- index lifetime, distance seams, exact kNN and the shared scan/merge launchers of
// libhnswgpu.so (C ABI: include/hnswgpu.h).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>

#include <thread>

#include "engine.hpp"
#include "tile_kernels.hpp"
#include "l2_kernels.hpp"
#include "stream_kernels.hpp"

namespace hg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int pick_nch(int64_t ld) {
    int64_t nvec = ld / 4;
    int need = static_cast<int>((nvec + kWave - 1) / kWave);
    const int opts[] = {1, 2, 3, 4, 6, 8, 12};
    for (int o : opts)
        if (o >= need) return o;
    return 0;
}

static int rb_of(int nch) { return nch <= 3 ? 8 : (nch <= 6 ? 4 : 2); }
int scan_rows_per_iter(int nch) { return kNWave * rb_of(nch); }


int launch_norms(int nch, const float *rows, int64_t ld, int64_t n, float *out, hipStream_t st) {
    if (n <= 0) return 0;
    unsigned grid = static_cast<unsigned>((n + kNWave - 1) / kNWave);
#define CALL(N, R, L) hipLaunchKernelGGL((row_norms_kernel<N>), dim3(grid), dim3(kWG), 0, st, rows, ld, n, out)
    HG_DISPATCH(nch, false, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// int8 codes + per-row bound terms of `n` rows (kernels.hpp: quantize_rows_kernel)
int quantize_rows(hnswgpu_index *idx, const float *rows, int64_t n, uint32_t **crows, float4 **cmeta, hipStream_t st) {
    HG_HIP(hipMalloc(reinterpret_cast<void **>(crows), sizeof(uint32_t) * kWave * idx->nch * n));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(cmeta), sizeof(float4) * n));
    unsigned grid = static_cast<unsigned>((n + kNWave - 1) / kNWave);
#define CALL(N, R, L) \
    hipLaunchKernelGGL((quantize_rows_kernel<N>), dim3(grid), dim3(kWG), 0, st, rows, idx->ld, n, idx->metric, *crows, *cmeta)
    HG_DISPATCH(idx->nch, false, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// An int8 row is 64 x NCH dwords (256 B at dim <= 256): below dim 128 it saves no cache line against the f32 row and the
// bounds pass would only add a memory round trip -- mode 1 leaves such handles without codes (mode 2 forces them).
static bool codes_wanted(const hnswgpu_index *idx) {
    return idx->rejection_mode == 2 || (idx->rejection_mode == 1 && idx->dim >= 128);
}

// The int8 rows of the traversal's rejection test and of the k-means++ bounds pass (base order, lane layout): once per
// handle (the base rows never change), on the first graph or build.
int ensure_qrows(hnswgpu_index *idx, hipStream_t st) {
    if (!codes_wanted(idx) || idx->d_qrows || idx->n <= 0) return 0;
    return quantize_rows(idx, idx->d_base, idx->n, &idx->d_qrows, &idx->d_qmeta, st);
}

// The int8 rows of the IVF list scan's bounds pass (stream_kernels.hpp): list order, MFMA tile layout -- blocks of 32
// rows x 32-byte steps -- plus the per-row bound terms; ONE copy, once per set of lists.
int ensure_list_codes(hnswgpu_index *idx, hipStream_t st) {
    if (!codes_wanted(idx) || idx->d_lctile || idx->n <= 0 || idx->nlist <= 0) return 0;
    const int64_t n = idx->n, blocks = (n + 31) / 32;
    const size_t bytes = static_cast<size_t>(blocks) * 32 * idx->nch * 256;
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_lctile), bytes));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_lcmeta), sizeof(float4) * n));
    HG_HIP(hipMemsetAsync(idx->d_lctile, 0, bytes, st));
    unsigned grid = static_cast<unsigned>((n + kNWave - 1) / kNWave);
#define CALL(N, R, L)                                                                                                   \
    hipLaunchKernelGGL((quantize_rows_tile_kernel<N>), dim3(grid), dim3(kWG), 0, st, idx->d_lrows, idx->ld, n, idx->metric, \
                       idx->d_lctile, idx->d_lcmeta)
    HG_DISPATCH(idx->nch, false, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// The half-precision copy of the list rows (stream_kernels.hpp, step 1b): list order, row-major, + (scale, E, 0, 1/|v|)
// per row.  Half the size of the f32 rows again; HNSWGPU_IVF_HALF=0 leaves it out (the searches then go from the int8
// bounds straight to the f32 rows at every batch size).
int ensure_list_half(hnswgpu_index *idx, hipStream_t st) {
    const bool wanted = tune(HNSWGPU_TUNE_IVF_HALF, 1) != 0;
    if (!wanted || !idx->d_lctile || idx->d_lhalf || idx->n <= 0 || idx->nlist <= 0) return 0;
    const int64_t n = idx->n;
    // (an optional accelerator: a device too full for it -- +50 % of the base -- searches without it)
    if (hipMalloc(reinterpret_cast<void **>(&idx->d_lhalf), sizeof(uint16_t) * static_cast<size_t>(n) * idx->ld) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&idx->d_lhmeta), sizeof(float4) * n) != hipSuccess) {
        (void)hipGetLastError();
        if (idx->d_lhalf) (void)hipFree(idx->d_lhalf);
        idx->d_lhalf = nullptr;
        idx->d_lhmeta = nullptr;
        return 0;
    }
    unsigned grid = static_cast<unsigned>((n + kNWave - 1) / kNWave);
#define CALL(N, R, L)                                                                                                   \
    hipLaunchKernelGGL((quantize_rows_half_kernel<N>), dim3(grid), dim3(kWG), 0, st, idx->d_lrows, idx->ld, n, idx->metric, \
                       idx->d_lhalf, idx->d_lhmeta)
    HG_DISPATCH(idx->nch, false, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// LDS of finish_wg: W lists + final list + (ord, dist) of the result + the query's probe table + the bisection merge's scratch
static size_t finish_lds_bytes(int k, int nprobe) {
    return sizeof(uint64_t) * (kNWave + 2) * k + (sizeof(int64_t) + 2 * sizeof(uint32_t)) * ((nprobe + 1) & ~1) +
           (k <= kWave ? sizeof(uint64_t) * kNWave * k : 0);
}

int launch_heavy(const HeavyArgs &a, hipStream_t st) {
    hipLaunchKernelGGL(ivf_heavy_kernel, dim3(1), dim3(1024), 0, st, a);
    HG_HIP(hipGetLastError());
    return 0;
}

// the home lists through the matrix cores in half precision (stream_kernels.hpp, step 1a): `blocks` bounds the work list
int launch_home(const HomeArgs &a, int64_t blocks, int nch, hipStream_t st) {
    if (blocks <= 0) return 0;
    HG_REQUIRE(a.ld % 128 == 0, HNSWGPU_EINVAL, "home-list pass: rows of whole 128-element steps only");
    // operand loads in flight per wave: the kernel lives on them (one query group per 16-row block is 24 KB of rows and
    // ~1.5 us of a CU's share of HBM); a depth has to divide the steps of a row
    const int S = static_cast<int>(a.ld / 32), want = static_cast<int>(tune(HNSWGPU_TUNE_HOME_DEPTH, 12));
    int pf = 4;
    for (int c : {8, 12, 16, 24})
        if (c <= want && S % c == 0) pf = c;
#define CALLP(N, P) hipLaunchKernelGGL((ivf_home_kernel<N, P>), dim3(static_cast<unsigned>(blocks)), dim3(kWG), 0, st, a)
#define CALL(N, R, L)                  \
    do {                               \
        switch (pf) {                  \
            case 24: CALLP(N, 24); break; \
            case 16: CALLP(N, 16); break; \
            case 12: CALLP(N, 12); break; \
            case 8: CALLP(N, 8); break;   \
            default: CALLP(N, 4); break;  \
        }                              \
    } while (0)
    HG_DISPATCH(nch, false, CALL);
#undef CALL
#undef CALLP
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_mid(const MidArgs &a0, int nch, hipStream_t st) {
    MidArgs a = a0;
    int64_t blocks = static_cast<int64_t>(a.nq) * a.slices;
    if (a.qorder) blocks = (static_cast<int64_t>(a.nq) + 7) / 8 * 8;  // ordered queries: whole rounds over the eight XCDs
    if (a.todo) blocks = static_cast<int64_t>(std::min(a.nq, 1024)) * a.todo_slices;  // (slots that walk the list of queries with work)
    if (blocks <= 0) return 0;
    a.main_blocks = static_cast<int32_t>(blocks);
    if (a.heavy_cnt)  // + the workgroups that walk the list of heavy queries
        blocks += static_cast<int64_t>(kHeavySlots) * a.heavy_slices;
    HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "half-precision pass grid too large");
    const bool l2 = a.metric == METRIC_L2;
    const size_t lds = sizeof(float) * static_cast<size_t>(std::max(a.compact, 0));
    HG_REQUIRE(lds <= 48 * 1024, HNSWGPU_ELIMIT, "half-precision pass: compaction buffer too large");
    const bool wide_env = tune(HNSWGPU_TUNE_MID_WIDE, 1) != 0;  // 0 = never (A/B)
    // eight waves per query where the grid alone does not fill the chip (cosine batch 1024: 235 -> 216 us; the Euclidean kernel,
    // heavier per element, measured 2 - 3 % slower with them and keeps four)
    const bool wide = wide_env && a.slices == 1 && !a.todo && a.nq < 2048 && a.k <= 8 * kWave && a.metric != METRIC_L2;
#define CALL(N, R, L)                                                                                                             \
    do {                                                                                                                          \
        if (wide) hipLaunchKernelGGL((ivf_mid_kernel<N, R, L, 8>), dim3(static_cast<unsigned>(blocks)), dim3(8 * kWave), lds, st, a); \
        else hipLaunchKernelGGL((ivf_mid_kernel<N, R, L, 4>), dim3(static_cast<unsigned>(blocks)), dim3(kWG), lds, st, a);        \
    } while (0)
    HG_DISPATCH(nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_stream_bounds(const StreamArgs &a, int64_t blocks, int nch, bool narrow, hipStream_t st, int qblocks) {
    if (blocks <= 0) return 0;
    HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "bounds pass grid too large");
    const bool wide = (qblocks == 2 || qblocks == 4) && !narrow && a.defer;  // several 32-query column blocks per group (the work list was built for them)
    HG_REQUIRE(qblocks == 1 || wide, HNSWGPU_EINVAL, "bounds pass: several column blocks need the wide deferring epilogue");
    const bool two = wide && qblocks == 2, four = wide && qblocks == 4;
    if (wide) count_launch(HNSWGPU_COUNT_BOUNDS_TWO_BLOCKS);
    const size_t lds = stream_lds_bytes(nch, narrow, qblocks, stream_waves(qblocks));
    HG_REQUIRE(lds <= 160 * 1024, HNSWGPU_ELIMIT, "bounds pass: the group's codes do not fit the LDS");
#define CALLV(N, NARROW, DEFER, QB)                                                                                    \
    do {                                                                                                               \
        static bool attr_done[64] = {};                                                                                \
        if (lds > 48 * 1024 && attr_needed(attr_done))                                                                 \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&stream_bounds_kernel<N, NARROW, DEFER, QB>),    \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                       \
        hipLaunchKernelGGL((stream_bounds_kernel<N, NARROW, DEFER, QB>), dim3(static_cast<unsigned>(blocks)),          \
                           dim3(stream_waves(QB) * kWave), lds, st, a);                                                \
    } while (0)
#define CALL(N, R, L)                         \
    do {                                      \
        if (narrow) CALLV(N, true, false, 1); \
        else if (four) CALLV(N, false, true, 4); \
        else if (two) CALLV(N, false, true, 2); \
        else if (a.defer) CALLV(N, false, true, 1); \
        else CALLV(N, false, false, 1);       \
    } while (0)
    HG_DISPATCH(nch, false, CALL);
#undef CALL
#undef CALLV
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_finish(const FinishArgs &a, int nch, hipStream_t st) {
    int64_t blocks = static_cast<int64_t>(a.nq) * a.slices;
    if (a.qorder) blocks = (static_cast<int64_t>(a.nq) + 7) / 8 * 8;  // ordered queries: whole rounds over the eight XCDs
    if (blocks <= 0) return 0;
    HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "finish grid too large");
    const size_t lds = finish_lds_bytes(a.k, a.nprobe);
    HG_REQUIRE(lds <= 64 * 1024, HNSWGPU_ELIMIT, "k too large for the finish kernel (k=%d)", a.k);
    const bool l2 = a.metric == METRIC_L2;
#define CALL(N, R, L)                                                                                                   \
    do {                                                                                                                \
        static bool attr_done[64] = {};                                                                                 \
        if (lds > 48 * 1024 && attr_needed(attr_done))                                                                  \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ivf_finish_kernel<N, (R > 2 ? R / 2 : R), L>),   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));                         \
        /* half the rows in flight of the streaming kernels: a gathered row costs this kernel 24 registers */          \
        hipLaunchKernelGGL((ivf_finish_kernel<N, (R > 2 ? R / 2 : R), L>), dim3(static_cast<unsigned>(blocks)), dim3(kWG), lds, st, a); \
        if (a.heavy_cnt) /* the heavy queries, heavy_slices workgroups each */                                          \
            hipLaunchKernelGGL((ivf_finish_heavy_kernel<N, (R > 2 ? R / 2 : R), L>), dim3(kHeavySlots * a.heavy_slices), dim3(kWG), lds, st, a); \
    } while (0)
    HG_DISPATCH(nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_query_prep(const PrepArgs &a, int nch, hipStream_t st) {
    if (a.nq <= 0) return 0;
    const bool l2 = a.metric == METRIC_L2;
#define CALL(N, R, L) hipLaunchKernelGGL((ivf_query_prep_kernel<N, R, L>), dim3(a.nq), dim3(kWG), 0, st, a)
    HG_DISPATCH(nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_scan(int nch, const ScanArgs &a0, hipStream_t st) {
    ScanArgs a = a0;
    int64_t blocks = static_cast<int64_t>(a.npairs) * a.nchunks;
    if (blocks <= 0) return 0;
    if (a.order) {
        if (a.run < 8) a.run = 8;
        blocks = (blocks + 8 * a.run - 1) / (8 * a.run) * (8 * a.run);  // whole runs on every XCD
    }
    HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "scan grid too large (%lld blocks)", (long long)blocks);
    size_t lds = a.mode == MODE_TOPK ? sizeof(uint64_t) * kNWave * a.k : 0;
    if (a.done) lds = sizeof(uint64_t) * (kNWave + 2) * a.k;  // fused tail: W lists + final list + (ord, dist) of the result
    HG_REQUIRE(lds <= 64 * 1024, HNSWGPU_ELIMIT, "k too large for the scan kernel (k=%d)", a.k);
    bool l2 = a.metric == METRIC_L2;
#define CALL_ROLE(N, R, L, ROLE) \
    hipLaunchKernelGGL((scan_kernel<N, R, L, ROLE>), dim3(static_cast<unsigned>(blocks)), dim3(kWG), lds, st, a)
#define CALL(N, R, L)                                          \
    switch (a.role) {                                          \
        case ROLE_ROUTE: CALL_ROLE(N, R, L, ROLE_ROUTE); break;   \
        case ROLE_ASSIGN: CALL_ROLE(N, R, L, ROLE_ASSIGN); break; \
        case ROLE_SEED: CALL_ROLE(N, R, L, ROLE_SEED); break;     \
        case ROLE_EXACT: CALL_ROLE(N, R, L, ROLE_EXACT); break;   \
        default: CALL_ROLE(N, R, L, ROLE_LIST_SCAN); break;       \
    }
    HG_DISPATCH(nch, l2, CALL);
#undef CALL
#undef CALL_ROLE
    HG_HIP(hipGetLastError());
    return 0;
}

// hipFuncSetAttribute is per device: remember it per (call site, device), not once per process (one process may
// hold handles on several GPUs -- the partitioned and IVF-HNSW mirrors do)
bool attr_needed(bool (&done)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (done[dev]) return false;
    done[dev] = true;
    return true;
}

__global__ __launch_bounds__(1024) void merge_topk_kernel(MergeArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int64_t q = blockIdx.x;
    merge_topk_wg(a, blockIdx.x, blockDim.x >> 6, smem, a.out_ord + q * a.k, a.out_dist + q * a.k);
}

int launch_merge(const MergeArgs &a, hipStream_t st) {
    if (a.nq <= 0) return 0;
    // waves per query: ~2048 keys per wave, but never more workgroup-waves than the batch needs to fill the chip
    int64_t w = std::min<int64_t>(16, std::max<int64_t>(1, a.keys_per_query / 2048));
    while (w > 1 && static_cast<int64_t>(a.nq) * w > 8192) w /= 2;
    if (const int64_t e = tune(HNSWGPU_TUNE_MERGE_W, 0)) w = std::max<int64_t>(1, e);
    size_t lds = sizeof(uint64_t) * (w + 1) * a.k;
    HG_REQUIRE(lds <= 150 * 1024, HNSWGPU_ELIMIT, "k too large for the merge kernel");
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        if (attr_needed(attr_done)) {
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&merge_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        }
    }
    hipLaunchKernelGGL(merge_topk_kernel, dim3(a.nq), dim3(static_cast<unsigned>(w * kWave)), lds, st, a);
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_gather(int nch, GatherArgs a, int32_t nq, hipStream_t st) {
    if (a.m <= 0 || nq <= 0) return 0;
    bool l2 = a.metric == METRIC_L2;
#define CALL(N, R, L)                                                                                         \
    do {                                                                                                      \
        a.blocks_per_query = (a.m + kNWave * R - 1) / (kNWave * R);                                           \
        int64_t blocks = static_cast<int64_t>(a.blocks_per_query) * nq;                                       \
        HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "gather grid too large");                           \
        hipLaunchKernelGGL((gather_dist_kernel<N, R, L>), dim3(static_cast<unsigned>(blocks)), dim3(kWG), 0, st, a); \
    } while (0)
    HG_DISPATCH(nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

void prof_begin(hnswgpu_index *idx, int slot, hipStream_t st, hipEvent_t *e0) {
    *e0 = nullptr;
    if (!idx->prof || slot < 0) return;
    if (hipEventCreate(e0) != hipSuccess) {
        *e0 = nullptr;
        return;
    }
    (void)hipEventRecord(*e0, st);
}
void prof_end(hnswgpu_index *idx, int slot, hipStream_t st, hipEvent_t e0) {
    if (!e0) return;
    hipEvent_t e1;
    if (hipEventCreate(&e1) != hipSuccess) {
        (void)hipEventDestroy(e0);
        return;
    }
    (void)hipEventRecord(e1, st);
    idx->prof_ev[slot].push_back({e0, e1});
}

int plan_chunks(int nch, int64_t max_rows, int64_t mean_rows, int64_t npairs, int32_t *chunk_rows, bool list_pairs) {
    // Enough NON-EMPTY workgroups to fill 256 CUs several times over, but no chunk below one loop trip
    // of every wave.  The chunk size comes from the MEAN segment length (short lists then simply leave
    // their trailing chunks empty -- those workgroups exit at once); the chunk count from the longest.
    const int64_t per_iter = scan_rows_per_iter(nch);
    if (max_rows < 1) max_rows = 1;
    if (mean_rows < 1) mean_rows = 1;
    if (mean_rows > max_rows) mean_rows = max_rows;
    const int64_t env_blocks = tune(HNSWGPU_TUNE_SCAN_BLOCKS, 0);  // tuning override
    // measured on 1M x 768 / 1024 k-means lists (tools/sweep_scan_blocks.py).  (query, list) pairs run in list
    // order: a target of 4096-6144 workgroups is best for every small batch (batch 32: 0.348 ms against 0.43 at 32768
    // and 0.45 at 1024; batch 16: 0.215 against 0.237; batch 8: 0.135 against 0.144) -- ~250 rows per workgroup, a
    // query loaded once per 8 loop trips, 8x fewer partial lists for the merge.  One query streaming a whole table
    // (k-means++ rounds, single-query exact kNN) wants the finest chunks instead: 473 us against 511 per 1M x 768 pass.
    // A handful of queries: fewer, longer chunks still -- the merge of the per-wave partial lists is then the larger
    // part of the search (one query, 32 lists: 69 us end to end at 512 workgroups, 78 at 4096).
    const int64_t small_target = std::min<int64_t>(4096, std::max<int64_t>(512, 8 * npairs));
    int64_t target_blocks = env_blocks > 0 ? env_blocks
                                           : (list_pairs ? (npairs <= 4096 ? small_target : 16384) : (npairs <= 1536 ? 32768 : 16384));
    int64_t want = (target_blocks + npairs - 1) / (npairs > 0 ? npairs : 1);
    if (want < 1) want = 1;
    if (want == 1) mean_rows = max_rows;  // already enough pairs: one workgroup per pair, no empty chunks
    int64_t max_chunks = (mean_rows + per_iter - 1) / per_iter;
    if (want > max_chunks) want = max_chunks;
    int64_t cr = (mean_rows + want - 1) / want;
    cr = ((cr + per_iter - 1) / per_iter) * per_iter;
    *chunk_rows = static_cast<int32_t>(cr);
    return static_cast<int>((max_rows + cr - 1) / cr);
}

int scan_topk(hnswgpu_index *idx, ScanArgs a, int32_t nq, int32_t pairs_per_query, int64_t max_rows,
              hipStream_t st, int prof_slot, int64_t mean_rows) {
    a.mode = MODE_TOPK;
    a.npairs = nq * pairs_per_query;
    a.nchunks = plan_chunks(idx->nch, max_rows, mean_rows > 0 ? mean_rows : max_rows, a.npairs, &a.chunk_rows,
                            a.pairs != nullptr);
    int64_t keys_per_query = static_cast<int64_t>(pairs_per_query) * a.nchunks * kNWave * a.k;
    HG_TRY(idx->s_partial.ensure(sizeof(uint64_t) * keys_per_query * nq));
    HG_TRY(idx->s_ord.ensure(sizeof(uint32_t) * static_cast<size_t>(nq) * a.k));
    HG_TRY(idx->s_dist.ensure(sizeof(float) * static_cast<size_t>(nq) * a.k));
    a.partial = idx->s_partial.as<uint64_t>();
    hipEvent_t e0;
    prof_begin(idx, prof_slot, st, &e0);
    HG_TRY(launch_scan(idx->nch, a, st));
    prof_end(idx, prof_slot, st, e0);
    MergeArgs m;
    m.partial = a.partial;
    m.keys_per_query = keys_per_query;
    m.nq = nq;
    m.k = a.k;
    m.out_ord = idx->s_ord.as<uint32_t>();
    m.out_dist = idx->s_dist.as<float>();
    HG_TRY(launch_merge(m, st));
    return 0;
}

// zero-initialised per-query counters of the fused tails (each tail resets its own counter: zero between calls)
int ensure_counters(hnswgpu_index *idx, size_t n, hipStream_t st) {
    const size_t bytes = sizeof(uint32_t) * (2 * n + 4);  // [n] scan tails | [n] route tails | 2 words of the folded work list | the queries finished (flagged calls)
    if (bytes <= idx->s_done.cap) return 0;
    HG_TRY(idx->s_done.ensure(bytes));
    HG_HIP(hipMemsetAsync(idx->s_done.p, 0, idx->s_done.cap, st));
    idx->s_done_n = (idx->s_done.cap - 4 * sizeof(uint32_t)) / (sizeof(uint32_t) * 2);
    return 0;
}

// IVF list scan whose last workgroup per query merges, decodes and writes the final results (ScanArgs::done).
int scan_fused(hnswgpu_index *idx, ScanArgs a, int32_t nq, int32_t pairs_per_query, int64_t max_rows, int64_t mean_rows,
               hipStream_t st, int prof_slot) {
    a.mode = MODE_TOPK;
    a.npairs = nq * pairs_per_query;
    a.nchunks = plan_chunks(idx->nch, max_rows, mean_rows > 0 ? mean_rows : max_rows, a.npairs, &a.chunk_rows, true);
    a.wg_merge = a.k <= kWave ? 1 : 0;
    const int64_t keys_per_query = static_cast<int64_t>(pairs_per_query) * a.nchunks * (a.wg_merge ? 1 : kNWave) * a.k;
    HG_TRY(idx->s_partial.ensure(sizeof(uint64_t) * keys_per_query * nq));
    HG_TRY(ensure_counters(idx, nq, st));
    a.partial = idx->s_partial.as<uint64_t>();
    a.done = idx->s_done.as<uint32_t>();
    a.pairs_per_query = pairs_per_query;
    hipEvent_t e0;
    prof_begin(idx, prof_slot, st, &e0);
    HG_TRY(launch_scan(idx->nch, a, st));
    prof_end(idx, prof_slot, st, e0);
    return 0;
}

// ---- centroid routing of a small batch in ONE launch (search-ivf-flat's centroid ranking, ivf_flat.clj:261-269) ----
// Every workgroup computes the distances of a GROUP of up to `qgroup` queries to a slice of the centroid table (the GEMV
// order of scan_kernel: same bits; a wave fetches its eight centroid rows once and walks the group's queries over them),
// the last workgroup of a query picks the nprobe nearest (select_topk_wg: keys (distance, centroid), the stable sort of
// :266-268) and writes the query's probe table -- what used to be three launches (scan, select, probe_pairs).  For the
// survivor stream of the list scan (stream_kernels.hpp) the same tail also files the query's (query, list) pairs into
// the per-list buckets the bounds pass reads, and seeds the query's threshold.
struct RouteArgs {
    const float *cent;
    const float *cnorms;
    int64_t ld;
    int32_t nlist;
    const float *Q;
    int64_t qld;
    int32_t dim, metric, nq, nprobe;
    int32_t rows_per_block, blocks_per_query;
    int32_t qgroup;   // ivf_route_dist_kernel: queries per workgroup
    float *dense;     // [nq][nlist]
    uint32_t *done;   // [nq], zero between calls
    const int64_t *listoff;
    const int64_t *glistoff;
    Pair *pairs;
    int32_t *probes;  // optional
    int32_t *qcnt;    // optional
    // optional, for the survivor stream of the list scan (stream_kernels.hpp): the query's int8 codes + bound scalars,
    // tau seeded from the head of the candidate stream, an empty survivor list, the pairs filed by list
    uint32_t *qcodes;
    QueryScal *qscal;
    uint32_t *tau, *surv_cnt;
    int32_t k;               // of the search: the first threshold is the k-th smallest distance of the stream's head
    int32_t seed_rows;       // ... of its first max(k, seed_rows) rows (stream_seed_rows)
    const float *rows;       // list rows (f32) + norms
    const float *row_norms;
    const uint2 *half;       // optional: their half-precision copy + meta words -- the threshold seed reads those (seed_tau_wg)
    const float4 *hmeta;
    uint32_t *bk_cnt;        // [nlist] members filed per list (zeroed before the launch), or null
    uint2 *bk_mem;           // [nlist][bk_cap] (query, offset of the list in the query's candidate stream)
    int32_t bk_cap;
    int32_t home;            // the home-list pass follows: no seed for a query whose nearest list holds k rows (its threshold comes from there)
    int32_t defer_file;      // the pairs are filed under their lists by ivf_bucket_fill_kernel behind the tail (large batches), not here
    int32_t wl_on;           // kWorklistParts extra workgroups of the tail's launch build the bounds pass's work list (worklist_part_wg)
    WorklistArgs wl;
    unsigned long long *dbg;  // -DHG_IVF_STAMPS diagnostic builds only
};

// The nprobe <= 64 nearest of a query's centroid distances (keys (distance, position): the stable order of
// ivf_flat.clj:266-268), ascending: topk_small_wg (kernels.hpp).  Returns the keys in LDS to wave 0, null to the others.
template <bool COH>
__device__ __forceinline__ const uint64_t *select_small_wg(const float *in, int64_t n, int k, unsigned char *smem) {
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);  // [kNWave][k] | fin [k] | scratch [kNWave][k]
    return topk_small_wg(n, k, lists, lists + kNWave * k, lists + (kNWave + 1) * k, [&](int64_t i) {
        const float v = COH ? coherent_load(in + i) : in[i];
        return make_key(v, static_cast<uint32_t>(i));
    });
}

// The probe table of query qi from its nprobe nearest lists (keys: (distance, list), ascending, in LDS): offsets of the probed
// lists in the query's candidate stream, the pairs filed under their lists, the empty survivor list.  One wave.
constexpr int kRouteHead = 16;
__device__ __forceinline__ void route_probe_table(const RouteArgs &a, int qi, const uint64_t *keys, int lane, Pair *head_s,
                                                  uint32_t *tail_cover, int64_t *tail_qcnt) {
    // the query's probe table: offsets of the probed lists in its candidate stream (probe_pairs_kernel, one wave)
    uint32_t carry = 0, gcarry = 0;
    bool over = false;
    for (int p0 = 0; p0 < a.nprobe; p0 += kWave) {
        const int p = p0 + lane;
        uint32_t l = 0xffffffffu;
        if (p < a.nprobe) {
            const uint64_t key = keys[p];
            if (key != ~0ull) l = static_cast<uint32_t>(key);
        }
        Pair pr;
        pr.q = qi;
        pr.pad = 0;
        pr.row_begin = pr.row_end = 0;
        uint32_t glen = 0;
        if (l != 0xffffffffu) {
            pr.row_begin = a.listoff[l];
            pr.row_end = a.listoff[l + 1];
            glen = static_cast<uint32_t>(a.glistoff[l + 1] - a.glistoff[l]);
        }
        const uint32_t len = static_cast<uint32_t>(pr.row_end - pr.row_begin);
        uint32_t incl = len, gincl = glen;
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off, kWave), go = __shfl_up(gincl, off, kWave);
            if (lane >= off) {
                incl += o;
                gincl += go;
            }
        }
        pr.ord_base = carry + incl - len;
        pr.gord_base = gcarry + gincl - glen;
        if (p < a.nprobe) {
            a.pairs[static_cast<int64_t>(qi) * a.nprobe + p] = pr;
            if (a.probes) a.probes[static_cast<int64_t>(qi) * a.nprobe + p] = l == 0xffffffffu ? -1 : static_cast<int32_t>(l);
            if (head_s && p < kRouteHead) head_s[p] = pr;
            if (tail_cover && (p == kRouteHead - 1 || (p < kRouteHead && p == a.nprobe - 1))) *tail_cover = carry + incl;  // candidates the head covers
            if (a.bk_cnt && len > 0 && !a.defer_file) {  // file the pair under its list: the bounds pass serves a list's members together
                const uint32_t slot = __hip_atomic_fetch_add(a.bk_cnt + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < static_cast<uint32_t>(a.bk_cap))
                    a.bk_mem[static_cast<int64_t>(l) * a.bk_cap + slot] = make_uint2(static_cast<uint32_t>(qi), pr.ord_base);
                else
                    over = true;
            }
        }
        carry += __shfl(incl, kWave - 1, kWave);
        gcarry += __shfl(gincl, kWave - 1, kWave);
    }
    if (a.qcnt && lane == 0) a.qcnt[qi] = static_cast<int32_t>(carry);
    const bool any_over = __ballot(over) != 0;
    if (lane == 0) {
        if (tail_qcnt) *tail_qcnt = carry;
        // an empty survivor list -- or, when a bucket was full, the mark that sends the query through the finish
        // kernel's fallback (the plain f32 scan of all its candidates)
        if (a.surv_cnt) a.surv_cnt[qi] = any_over ? 0x80000000u : 0u;
    }
    wait_stores_acked();  // the probe table may be read back by the other waves below
    // (this query's pairs are filed: the work list's workgroups count the queries in)
    if (a.wl_on && lane == 0) (void)__hip_atomic_fetch_add(a.wl.filed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The tail of the routing of query qi, run by one whole workgroup once all of the query's centroid distances are in
// a.dense: pick the nprobe nearest, write the probe table, file the pairs by list, seed the threshold.  COH: the
// distances were written by other workgroups of THIS launch (agent-scope loads), not by an earlier one.
template <int NCH>
__device__ __forceinline__ void route_encode(const RouteArgs &a, int qi, const float4 (&q)[NCH], int lane);

template <int NCH, int RB, bool L2, bool COH>
__device__ __forceinline__ void route_tail_wg(const RouteArgs &a, int qi, unsigned char *smem, bool encode_here = false) {
    __shared__ int64_t tail_qcnt;
    __shared__ uint32_t tail_cover;
    constexpr int kHead = kRouteHead;
    __shared__ Pair head_s[kHead];  // the first probes' table entries, for the threshold seed below
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    SelectArgs s;
    s.dist = a.dense;
    s.q_cnt = nullptr;
    s.stride = a.nlist;
    s.cnt_all = a.nlist;
    s.nq = a.nq;
    s.k = a.nprobe;
    s.wpq = kNWave;
    s.vec4 = 0;
    s.out_ord = nullptr;  // the probed lists stay in LDS: the probe table below is all that leaves
    s.out_dist = nullptr;
    const uint64_t *keys = nullptr;
    HG_IVF_STAMP(a.dbg, 17, qi == 0 && threadIdx.x == 0);  // tail begins
    // (the query for the threshold seed at the end: on its way while the lists are picked)
    float4 q[NCH];
    if (a.tau) load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
    if (a.nprobe <= kWave) keys = select_small_wg<COH>(a.dense + static_cast<int64_t>(qi) * a.nlist, a.nlist, a.nprobe, smem);
    else select_topk_wg<COH>(s, qi, kNWave, smem, &keys);
    HG_IVF_STAMP(a.dbg, 18, qi == 0 && threadIdx.x == 0);  // nprobe nearest centroids selected
    // (the tail launch: the query's int8 codes by an otherwise idle wave WHILE wave 0 writes the probe table -- ahead of the
    // selection the encoding wave kept the other three waiting at the selection's barrier)
    if (encode_here && a.qcodes && a.tau && wave == kNWave - 1) route_encode<NCH>(a, qi, q, lane);
    if (wave == 0) {
        route_probe_table(a, qi, keys, lane, head_s, &tail_cover, &tail_qcnt);
        HG_IVF_STAMP(a.dbg, 19, qi == 0 && threadIdx.x == 0);  // probe table written, pairs filed
    }
    if (!a.tau) return;
    // survivor stream: the first threshold, from the head of the query's candidate stream
    __syncthreads();
    if (a.home && head_s[0].row_end - head_s[0].row_begin >= a.k) {
        // large batches: every row of the nearest list goes through the matrix cores in half precision next, and the k-th
        // smallest upper bound found there -- over the whole list, not a sample of it -- is the query's threshold before the
        // bounds pass reads it (ivf_home_kernel, then ivf_mid_kernel): 64 f32 rows per query are not fetched here (0.8 GB at
        // batch 4096)
        if (threadIdx.x == 0) a.tau[qi] = 0xffffffffu;
        return;
    }
    const float qn = (a.metric == METRIC_COS || (a.half && a.metric == METRIC_DOT)) ? query_norm<NCH>(q) : 0.0f;
    // the head's table entries are in LDS (no dependent global reads) when they cover the rows the seed looks at
    const bool head_ok = a.nprobe <= kHead || tail_cover >= static_cast<uint32_t>(kSeedMax);
    const Pair *pp = head_ok ? head_s : a.pairs + static_cast<int64_t>(qi) * a.nprobe;
    seed_tau_wg<NCH, RB, L2>(q, qn, a.metric, pp, head_ok ? (a.nprobe < kHead ? a.nprobe : kHead) : a.nprobe, tail_qcnt, a.k,
                             a.rows, a.row_norms, a.ld, reinterpret_cast<float *>(smem), a.tau + qi, a.seed_rows, a.half, a.hmeta);
    HG_IVF_STAMP(a.dbg, 20, qi == 0 && threadIdx.x == 0);  // threshold seeded
}

// int8 codes + bound scalars of query qi (the survivor stream's bounds pass): one wave
template <int NCH>
__device__ __forceinline__ void route_encode(const RouteArgs &a, int qi, const float4 (&q)[NCH], int lane) {
    QueryCode<NCH> qc;
    encode_query<NCH>(q, qc);
#pragma unroll
    for (int c = 0; c < NCH; c++) a.qcodes[(static_cast<int64_t>(qi) * NCH + c) * kWave + lane] = qc.a[c];
    if (lane == 0) a.qscal[qi] = qc.sc;
}

template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_route_kernel(RouteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int tail_last;
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int qi = blockIdx.x / a.blocks_per_query, bx = blockIdx.x % a.blocks_per_query;
    HG_IVF_STAMP(a.dbg, 16, blockIdx.x == 0 && threadIdx.x == 0);  // first workgroup of the routing kernel starts
    if (qi >= a.nq) {  // the extra workgroups behind the queries' own: the bounds pass's work list
        if (a.wl_on) worklist_part_wg(a.wl, static_cast<int>(blockIdx.x) - a.nq * a.blocks_per_query);
        return;
    }
    const int64_t r0 = static_cast<int64_t>(bx) * a.rows_per_block;
    const int64_t r1 = r0 + a.rows_per_block < a.nlist ? r0 + a.rows_per_block : a.nlist;
    if (r0 < r1) {
        float4 q[NCH];
        load_query<NCH>(q, a.Q + qi * a.qld, a.dim, lane);
        const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
        if (a.qcodes && bx == 0 && wave == kNWave - 1) route_encode<NCH>(a, qi, q, lane);  // once per query
        const int nvec = static_cast<int>(a.ld / 4);
        for (int64_t base = r0 + wave * RB; base < r1; base += kNWave * RB) {
            float4 r[RB][NCH];
            const float myrn = (a.metric == METRIC_COS && lane < RB && base + lane < r1) ? a.cnorms[base + lane] : 0.0f;
#pragma unroll
            for (int b = 0; b < RB; b++) load_row<NCH>(r[b], a.cent + (base + b) * a.ld, nvec, lane, base + b < r1);
            float sm[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) sm[b] = lane_partial<NCH, L2>(q, r[b]);
            // lane b: row b's sum and (above) its norm: the distance is finished and stored by the lane that holds both
            const float mine = rows_sum_to_lane<RB>(sm, lane);
            if (lane < RB && base + lane < r1)
                coherent_store(a.dense + static_cast<int64_t>(qi) * a.nlist + base + lane, finish_dist(a.metric, mine, qn, myrn));
        }
    }
    wait_stores_acked();  // the slice of distances is at the point of coherence before this workgroup counts itself
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(a.done + qi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tail_last = prev == static_cast<uint32_t>(a.blocks_per_query) - 1 ? 1 : 0;
        if (tail_last) __hip_atomic_store(a.done + qi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!tail_last) return;
    route_tail_wg<NCH, RB, L2, true>(a, qi, smem);
}

// Centroid distances of a batch, [nq][nlist] in the GEMV order (same bits as above), for the tail launch below: a
// workgroup takes a slice of the centroid table and a group of `qgroup` queries; every wave fetches its eight centroid
// rows ONCE and walks the group's queries over them, the next query on its way while the current one is scored.  (One
// GEMV per query reads the whole table per query: 805 MB out of L2 for 256 queries x 1024 centroids x 768.)
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_route_dist_kernel(RouteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    float4 *qs = reinterpret_cast<float4 *>(smem);  // [qgroup][NCH][64]: the group's queries in the lane layout
    __shared__ float qn_s[16];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int grp = blockIdx.x / a.blocks_per_query, bx = blockIdx.x % a.blocks_per_query;
    const int q0 = grp * a.qgroup;
    const int qn_here = a.nq - q0 < a.qgroup ? a.nq - q0 : a.qgroup;
    const int64_t r0 = static_cast<int64_t>(bx) * a.rows_per_block;
    const int64_t r1 = r0 + a.rows_per_block < a.nlist ? r0 + a.rows_per_block : a.nlist;
    const int nvec = static_cast<int>(a.ld / 4);
    int64_t base = r0 + wave * RB;
    float4 r[RB][NCH];
    float myrn = (a.metric == METRIC_COS && lane < RB && base + lane < r1) ? a.cnorms[base + lane] : 0.0f;
#pragma unroll
    for (int b = 0; b < RB; b++) load_row<NCH>(r[b], a.cent + (base + b) * a.ld, nvec, lane, base + b < r1);
    // the group's queries (and their norms) into LDS once per workgroup, while the first rows are on their way: read from
    // global memory inside the loop -- one query ahead -- every step waited out an L2 round trip (batch 1024: 106 us for
    // what is 50 us of arithmetic)
    for (int g = wave; g < qn_here; g += kNWave) {
        float4 qq[NCH];
        load_query<NCH>(qq, a.Q + static_cast<int64_t>(q0 + g) * a.qld, a.dim, lane);
#pragma unroll
        for (int c = 0; c < NCH; c++) qs[(g * NCH + c) * kWave + lane] = qq[c];
        if (a.metric == METRIC_COS) {
            const float n = query_norm<NCH>(qq);
            if (lane == 0) qn_s[g] = n;
        }
    }
    __syncthreads();
    for (; base < r1; base += kNWave * RB) {
        for (int g = 0; g < qn_here; g++) {
            const int qi = q0 + g;
            float4 q[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) q[c] = qs[(g * NCH + c) * kWave + lane];
            const float qn = a.metric == METRIC_COS ? qn_s[g] : 0.0f;
            float sm[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) sm[b] = lane_partial<NCH, L2>(q, r[b]);
            // lane b takes row b's distance: RB consecutive floats of the query's row of the matrix
            const float mine = rows_sum_to_lane<RB>(sm, lane);  // (eight rows: one halving exchange, the butterflies' bits)
            if (lane < RB && base + lane < r1)
                a.dense[static_cast<int64_t>(qi) * a.nlist + base + lane] = finish_dist(a.metric, mine, qn, myrn);
        }
        const int64_t nb = base + kNWave * RB;
        if (nb < r1) {
            myrn = (a.metric == METRIC_COS && lane < RB && nb + lane < r1) ? a.cnorms[nb + lane] : 0.0f;
#pragma unroll
            for (int b = 0; b < RB; b++) load_row<NCH>(r[b], a.cent + (nb + b) * a.ld, nvec, lane, nb + b < r1);
        }
    }
}

// The same distances ON THE F32 MATRIX CORES, in the GEMV order, bit for bit (cosine / dot, rows of 256 / 512 / 768 elements).
// The GEMV order is 64 independent chains -- lane l accumulates elements 256 c + 4 l + j, one fmaf each -- whose results are
// added in the xor butterfly's tree.  An f32 matrix instruction IS an fmaf chain over its k-slots (v_mfma_f32_32x32x2_f32:
// slot 0, 1; v_mfma_f32_16x16x4_f32: slot 0, 1, 2, 3 -- tools/micro/mfma16_order.hip), so a chain of them over the elements of
// lane partial l computes that partial for a whole tile of (centroid, query) pairs at once, and the 64 partial tiles are added
// pairwise in the butterfly's association: tile l joins the pending tile of its level like a carry in a binary counter (at
// most six pending).  tools/micro/mfma_gemv_order.hip checks the scheme against lane_partial + wave_sum on the hardware --
// rows of scales e^+-12, denormal products, signed zeros: equal in every bit.  The VALU form is bound by its issue rate (4096 x
// 1024 x 768: 150 us); this one by the f32 matrix rate.
//
// First built on v_mfma_f32_32x32x2_f32 (32 queries per workgroup in 99 KB of LDS: one wave per SIMD, 2 NCH instructions and
// four operand selects per lane partial, 16-register tiles): 127 us at 4096 queries -- the tree's additions and the matrix
// instructions took turns instead of overlapping.  Lessons that stay: a conditional load becomes a branch and every wait a
// vmcnt(0); the compiler re-loads "prefetched" operands in the iteration that uses them unless a memory clobber pins them;
// operand arrays behind such a clobber go to scratch memory, named registers do not; lane (i, h) fetching its own 16 bytes of
// 32 rows per instruction makes the CU's address unit the bound (eight rows x 128 B per instruction into an LDS slab instead).
__host__ __device__ constexpr int route_mfma_ldq(int nch) { return 256 * nch + 4; }   // floats per staged query (conflict-free reads)
__host__ __device__ constexpr int route_mfma_astr(int nch) { return 32 * nch + 4; }   // floats per row of a wave's operand slab

// On v_mfma_f32_16x16x4_f32 the four k-slots are exactly the four elements 256 c + 4 l + 0..3 of a lane partial: ONE
// instruction per (lane partial, chunk), operands one float per lane (lane (i, s) reads element s of row i's four: no
// selects), tiles of 16 centroids x 16 queries in four registers (the tree's additions cost a quarter), 16 queries per
// workgroup = 49 KB of LDS + a 6 KB operand slab per wave (eight lane partials of 16 centroids, fetched eight rows x 128 B per
// instruction, the next block on its way under this one's matrix work): two workgroups per CU, two waves per SIMD, and one's
// additions run under the other's matrix instructions.  4096 x 1024 x 768: 99 us (65 TFLOP/s), 1024: 33, 256: 13.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int kRoute16Q = 16;
__host__ __device__ constexpr size_t route_mfma16_lds(int nch) {
    return sizeof(float) * (kRoute16Q * route_mfma_ldq(nch) + kNWave * 16 * route_mfma_astr(nch));
}

template <int NCH>
__global__ __launch_bounds__(kWG) void ivf_route_mfma16_kernel(RouteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int LDQ = route_mfma_ldq(NCH);
    constexpr int ASTR = route_mfma_astr(NCH);
    float *qs = reinterpret_cast<float *>(smem);  // [16][LDQ]
    float *as_all = qs + kRoute16Q * LDQ;         // [waves][16][ASTR]: the A operands of one block of eight lane partials
    __shared__ float qn_s[kRoute16Q];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int grp = blockIdx.x / a.blocks_per_query, bx = blockIdx.x % a.blocks_per_query;
    const int q0 = grp * kRoute16Q;
    const int i = lane & 15, ks = lane >> 4;
    const int64_t r0 = static_cast<int64_t>(bx) * a.rows_per_block;
    const int64_t r1 = r0 + a.rows_per_block < a.nlist ? r0 + a.rows_per_block : a.nlist;
    float *aslab = as_all + wave * 16 * ASTR;
    const int prow = lane >> 3, ppiece = lane & 7;
    {
        float4 qq[kRoute16Q / kNWave][NCH];
#pragma unroll
        for (int t = 0; t < kRoute16Q / kNWave; t++) {
            const int g = wave + t * kNWave;
            if (q0 + g < a.nq && a.dim == 256 * NCH && (a.qld & 3) == 0) {  // whole 16-byte pieces (load_query checks every element)
#pragma unroll
                for (int c = 0; c < NCH; c++) qq[t][c] = *reinterpret_cast<const float4 *>(a.Q + static_cast<int64_t>(q0 + g) * a.qld + 256 * c + 4 * lane);
            } else if (q0 + g < a.nq) load_query<NCH>(qq[t], a.Q + static_cast<int64_t>(q0 + g) * a.qld, a.dim, lane);
            else {
#pragma unroll
                for (int c = 0; c < NCH; c++) qq[t][c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
#pragma unroll
        for (int t = 0; t < kRoute16Q / kNWave; t++) {
            const int g = wave + t * kNWave;
#pragma unroll
            for (int c = 0; c < NCH; c++) *reinterpret_cast<float4 *>(qs + g * LDQ + 256 * c + 4 * lane) = qq[t][c];
            const float n = a.metric == METRIC_COS ? query_norm<NCH>(qq[t]) : 0.0f;
            if (lane == 0) qn_s[g] = n;
        }
    }
    __syncthreads();
    const float *qrow = qs + i * LDQ + ks;      // B operand: lane (j = i, s) reads element s of query j's four
    const float *arow = aslab + i * ASTR + ks;  // A operand: lane (i, s) of centroid i's
#define R16_FOR6(M) M(0) M(1) M(2) M(3) M(4) M(5)
#define R16_ROWPTR(k, tb_)                                                                                 \
    (a.cent + ((tb_) + ((k) & 1) * 8 + prow < a.nlist ? (tb_) + ((k) & 1) * 8 + prow : a.nlist - 1) * a.ld + 256 * ((k) >> 1) + 4 * ppiece)
    for (int64_t tb = r0 + 16 * wave; tb < r1; tb += 16 * kNWave) {
#define R16_DECL(k) float4 ac##k = make_float4(0.0f, 0.0f, 0.0f, 0.0f), an##k = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        R16_FOR6(R16_DECL)
#undef R16_DECL
#define R16_FIRST(k) \
    if constexpr (k < 2 * NCH) ac##k = *reinterpret_cast<const float4 *>(R16_ROWPTR(k, tb));
        R16_FOR6(R16_FIRST)
#undef R16_FIRST
        f32x4_t s3, s4, s5, fin;  // pending tiles of levels 3, 4, 5 of the tree; the sum
#pragma unroll 1
        for (int lb = 0; lb < 8; lb++) {
#define R16_PARK(k) \
    if constexpr (k < 2 * NCH) *reinterpret_cast<float4 *>(aslab + ((k & 1) * 8 + prow) * ASTR + (k >> 1) * 32 + 4 * ppiece) = ac##k;
            R16_FOR6(R16_PARK)
#undef R16_PARK
            const int nlb = lb + 1 < 8 ? lb + 1 : 7;  // (no branch around the loads: their wait counts stay exact)
#define R16_NEXT(k) \
    if constexpr (k < 2 * NCH) an##k = *reinterpret_cast<const float4 *>(R16_ROWPTR(k, tb) + 32 * nlb);
            R16_FOR6(R16_NEXT)
#undef R16_NEXT
            asm volatile("" ::: "memory");  // (or the compiler forgets the prefetch)
            f32x4_t st[3], T;
            const float *qb = qrow + 32 * lb;
#pragma unroll
            for (int u = 0; u < 8; u++) {
                f32x4_t P = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
                for (int c = 0; c < NCH; c++) P = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[32 * c + 4 * u], qb[256 * c + 4 * u], P, 0, 0, 0);
                int t = u, lvl = 0;
                while (t & 1) {  // (compile-time after unrolling)
                    P = st[lvl] + P;
                    t >>= 1;
                    lvl++;
                }
                if (lvl < 3) st[lvl] = P;
                else T = P;
            }
            if (lb & 1) {
                T = s3 + T;
                if (lb & 2) {
                    T = s4 + T;
                    if (lb & 4) fin = s5 + T;
                    else s5 = T;
                } else {
                    s4 = T;
                }
            } else {
                s3 = T;
            }
#define R16_ROLL(k) ac##k = an##k;
            R16_FOR6(R16_ROLL)
#undef R16_ROLL
        }
        // C/D: lane holds column j = i (the query); register g is row 4 s + g of the tile
        const int qi = q0 + i;
        if (qi < a.nq) {
            const float qn = qn_s[i];
            float *orow = a.dense + static_cast<int64_t>(qi) * a.nlist;
            float rn[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t cc = tb + 4 * ks + g;
                rn[g] = (a.metric == METRIC_COS && cc < r1) ? a.cnorms[cc] : 0.0f;
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int64_t cc = tb + 4 * ks + g;
                if (cc < r1) orow[cc] = finish_dist(a.metric, fin[g], qn, rn[g]);
            }
        }
    }
#undef R16_ROWPTR
#undef R16_FOR6
}

// The same tail as a launch of its own, one workgroup per query, behind a distance pass that serves many queries per
// centroid row (large batches): select, probe table, pairs filed by list, query codes, first threshold.
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_route_tail_kernel(RouteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    if (qi >= a.nq) {  // the extra workgroups behind the queries' own: the bounds pass's work list
        if (a.wl_on) worklist_part_wg(a.wl, qi - a.nq);
        return;
    }
    if (a.qcodes && !a.tau && wave == kNWave - 1) {  // (no threshold wanted: the tail does not load the query itself)
        float4 q[NCH];
        load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
        route_encode<NCH>(a, qi, q, lane);
    }
    route_tail_wg<NCH, RB, L2, false>(a, qi, smem, true);
}

template <int NCH>
__global__ void ivf_route_tail_wave_kernel(RouteArgs a);
static int launch_route_tail_wave(const RouteArgs &a, int nch, hipStream_t st);

// The (query, list) pairs of a LARGE batch filed under their lists (round 5).  The tail's own way is one agent-scope atomicAdd
// per pair on its list's counter: at batch 16384 that is 524k atomics on 1024 addresses, 512 in a row per address -- the tail
// kernel took 160 - 177 us of which ~25 are its work.  Here a workgroup takes ~8192 consecutive pairs, ranks them per list in
// an LDS histogram (LDS atomics), reserves each list's run with ONE global atomicAdd per (workgroup, list) -- 64 per address
// instead of 512 -- and scatters the members.  The members of a list end up in another order than the tail would have filed
// them in (results do not depend on it: survivors carry order keys); a full bucket marks its query for the finish kernel's
// fallback as before.
struct BucketArgs {
    const int32_t *probes;  // [npairs] list of pair i = query i / nprobe, probe i % nprobe (-1: none)
    const Pair *pairs;      // [npairs]
    const int64_t *listoff;
    int32_t npairs, nprobe, nlist, per_wg;
    uint32_t *bk_cnt;
    uint2 *bk_mem;
    int32_t bk_cap;
    uint32_t *surv_cnt;
};
constexpr int kBucketPairs = 8;  // pairs per thread
__global__ __launch_bounds__(1024) void ivf_bucket_fill_kernel(BucketArgs a) {
    extern __shared__ uint32_t bhist[];  // [nlist]: pairs of this workgroup per list, then the base of their run in the bucket
    const int tid = threadIdx.x;
    const int p0 = static_cast<int>(blockIdx.x) * a.per_wg, p1 = p0 + a.per_wg < a.npairs ? p0 + a.per_wg : a.npairs;
    for (int l = tid; l < a.nlist; l += 1024) bhist[l] = 0u;
    __syncthreads();
    int32_t myl[kBucketPairs];
    uint32_t myrank[kBucketPairs];
#pragma unroll
    for (int u = 0; u < kBucketPairs; u++) {
        const int i = p0 + u * 1024 + tid;
        myl[u] = -1;
        myrank[u] = 0;
        if (i < p1) {
            const int32_t l = a.probes[i];
            if (l >= 0 && a.listoff[l + 1] > a.listoff[l]) {
                myl[u] = l;
                myrank[u] = atomicAdd(&bhist[l], 1u);
            }
        }
    }
    __syncthreads();
    for (int l = tid; l < a.nlist; l += 1024) {
        const uint32_t c = bhist[l];
        bhist[l] = c ? __hip_atomic_fetch_add(a.bk_cnt + l, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBucketPairs; u++) {
        if (myl[u] < 0) continue;
        const int i = p0 + u * 1024 + tid;
        const uint32_t slot = bhist[myl[u]] + myrank[u];
        const int q = i / a.nprobe;
        if (slot < static_cast<uint32_t>(a.bk_cap))
            a.bk_mem[static_cast<int64_t>(myl[u]) * a.bk_cap + slot] = make_uint2(static_cast<uint32_t>(q), a.pairs[i].ord_base);
        else
            a.surv_cnt[q] = 0x80000000u;  // (the bucket is full: the query takes the finish kernel's plain f32 scan)
    }
}

// The same tail with ONE WAVE per query, four queries per workgroup (round 5: large home-list batches -- a query's threshold comes
// from the home-list pass --, nprobe and k <= 64).  A workgroup per query is a chain of ~15 us of which a CU holds eight:
// 16384 queries took 170 us.  A wave takes 16 distances per lane and chunk, picks the nprobe nearest by the bisection of the key
// space started from its lane minima (wave_topk_sorted), writes the probe table, files the pairs, encodes the query.
template <int NCH>
__global__ __launch_bounds__(kWG) void ivf_route_tail_wave_kernel(RouteArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];  // [kNWave][2][nprobe] keys: a wave's list and its scratch
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int qi = static_cast<int>(blockIdx.x) * kNWave + wave;
    if (qi >= a.nq) return;
    uint64_t *mylist = reinterpret_cast<uint64_t *>(smem) + static_cast<size_t>(wave) * 2 * a.nprobe, *myscr = mylist + a.nprobe;
    float4 q[NCH];
    load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
    constexpr int S = 16;
    const float *in = a.dense + static_cast<int64_t>(qi) * a.nlist;
    uint64_t carry = ~0ull;
    for (int base = 0; base < a.nlist; base += S * kWave) {
        uint64_t key[S];
#pragma unroll
        for (int s2 = 0; s2 < S; s2++) {
            const int i = base + s2 * kWave + lane;
            key[s2] = i < a.nlist ? make_key(in[i], static_cast<uint32_t>(i)) : ~0ull;
        }
        wave_topk_sorted<S>(key, carry, a.nprobe, myscr, mylist, lane);
        carry = lane < a.nprobe ? mylist[lane] : ~0ull;
    }
    route_probe_table(a, qi, mylist, lane, nullptr, nullptr, nullptr);
    if (a.qcodes) route_encode<NCH>(a, qi, q, lane);
    if (!a.tau) return;
    // the threshold comes from the home-list pass -- unless the nearest list is shorter than k (rare: a seed by this wave alone,
    // from the first <= 64 rows of the candidate stream in f32: any k exact distances bound the k-th)
    const uint64_t k0 = mylist[0];
    int64_t len0 = 0;
    if (k0 != ~0ull) {
        const uint32_t l0 = static_cast<uint32_t>(k0);
        len0 = a.listoff[l0 + 1] - a.listoff[l0];
    }
    if (len0 >= a.k) {
        if (lane == 0) a.tau[qi] = 0xffffffffu;
        return;
    }
    const Pair *pp = a.pairs + static_cast<int64_t>(qi) * a.nprobe;  // (this wave's own stores, acknowledged in route_probe_table)
    const Pair last = pp[a.nprobe - 1];
    const int64_t qcnt = static_cast<int64_t>(last.ord_base) + (last.row_end - last.row_begin);
    const int m = static_cast<int>(qcnt < kWave ? qcnt : kWave);
    if (m < a.k) {  // fewer candidates than k: nothing can be excluded
        if (lane == 0) a.tau[qi] = 0xffffffffu;
        return;
    }
    int64_t myrow = 0;
    float myrn = 0.0f;
    if (lane < m) {
        int lo = 0, hi = a.nprobe - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pp[mid].ord_base <= static_cast<uint32_t>(lane)) lo = mid;
            else hi = mid - 1;
        }
        myrow = pp[lo].row_begin + (static_cast<uint32_t>(lane) - pp[lo].ord_base);
        myrn = a.metric == METRIC_COS ? a.row_norms[myrow] : 0.0f;
    }
    const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
    const int nvec = static_cast<int>(a.ld / 4);
    float mine = __builtin_inff();
    for (int j = 0; j < m; j++) {  // (one row per trip: a few dozen rows of the rare query that needs them)
        const int64_t row = (static_cast<int64_t>(__builtin_amdgcn_readlane(static_cast<int>(myrow >> 32), j)) << 32) |
                            static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(myrow), j));
        float4 r[NCH];
        load_row<NCH>(r, a.rows + row * a.ld, nvec, lane, true);
        const float sum = a.metric == METRIC_L2 ? wave_sum(lane_partial<NCH, true>(q, r)) : wave_sum(lane_partial<NCH, false>(q, r));
        const float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), j));
        const float d = finish_dist(a.metric, sum, qn, rn);
        if (lane == j) mine = d == d ? d : __builtin_inff();
    }
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kWave; j++) {
        const float o = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), j));
        rank += (o < mine || (o == mine && j < lane)) ? 1 : 0;
    }
    const uint64_t sel = __ballot(rank == a.k - 1);
    const float kth = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), __ffsll(static_cast<unsigned long long>(sel)) - 1));
    if (lane == 0) a.tau[qi] = kth < __builtin_inff() ? tau_encode(kth) : 0xffffffffu;
}

static int launch_bucket_fill(const RouteArgs &r, hipStream_t st) {
    BucketArgs b;
    memset(&b, 0, sizeof(b));
    b.probes = r.probes;
    b.pairs = r.pairs;
    b.listoff = r.listoff;
    const int64_t npairs = static_cast<int64_t>(r.nq) * r.nprobe;
    HG_REQUIRE(npairs < 2147483647LL && r.probes && r.surv_cnt, HNSWGPU_EINVAL, "bucket fill: too many pairs, or no probe table");
    b.npairs = static_cast<int32_t>(npairs);
    b.nprobe = r.nprobe;
    b.nlist = r.nlist;
    b.per_wg = kBucketPairs * 1024;
    b.bk_cnt = r.bk_cnt;
    b.bk_mem = r.bk_mem;
    b.bk_cap = r.bk_cap;
    b.surv_cnt = r.surv_cnt;
    const unsigned blocks = static_cast<unsigned>((npairs + b.per_wg - 1) / b.per_wg);
    const size_t lds = sizeof(uint32_t) * static_cast<size_t>(r.nlist);
    HG_REQUIRE(lds <= 48 * 1024, HNSWGPU_ELIMIT, "bucket fill: too many lists for its LDS histogram");
    hipLaunchKernelGGL(ivf_bucket_fill_kernel, dim3(blocks), dim3(1024), lds, st, b);
    HG_HIP(hipGetLastError());
    return 0;
}

static int launch_route_tail_wave(const RouteArgs &a, int nch, hipStream_t st) {
    count_launch(HNSWGPU_COUNT_ROUTE_TAIL_WAVES);
    const size_t lds = sizeof(uint64_t) * kNWave * 2 * static_cast<size_t>(a.nprobe);
    const unsigned blocks = static_cast<unsigned>((a.nq + kNWave - 1) / kNWave);
    switch (nch) {
#define CALLW(N) case N: hipLaunchKernelGGL((ivf_route_tail_wave_kernel<N>), dim3(blocks), dim3(kWG), lds, st, a); break
        CALLW(1);
        CALLW(2);
        CALLW(3);
        CALLW(4);
        CALLW(6);
        CALLW(8);
        CALLW(12);
#undef CALLW
        default: set_error("unsupported row length"); return HNSWGPU_ELIMIT;
    }
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_ivf_route(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t nprobe, Pair *pairs, int32_t *probes,
                     int32_t *qcnt, hipStream_t st, const RouteStream *rs, bool two_launches) {
    RouteArgs a;
    memset(&a, 0, sizeof(a));
    if (rs) {
        a.qcodes = rs->qcodes;
        a.qscal = rs->qscal;
        a.tau = rs->tau;
        a.surv_cnt = rs->surv_cnt;
        a.k = rs->k;
        a.seed_rows = stream_seed_rows(nq, idx->ivf_n_global > 0 ? idx->ivf_n_global : idx->n, idx->nlist);
        a.bk_cnt = rs->bk_cnt;
        a.bk_mem = rs->bk_mem;
        a.bk_cap = rs->bk_cap;
        a.home = rs->home;
        if (rs->wl && rs->bk_cnt) {
            a.wl = *rs->wl;
            a.wl_on = 1;
        }
    }
    a.dbg = g_tile_dbg_buf;  // null outside diagnostic sessions
    a.rows = idx->d_lrows;
    a.row_norms = idx->d_lnorms;
    if (idx->d_lhalf && tune(HNSWGPU_TUNE_SEED_HALF, 1) != 0) {
        a.half = idx->d_lhalf;
        a.hmeta = idx->d_lhmeta;
    }
    a.cent = idx->d_cent;
    a.cnorms = idx->d_cnorms;
    a.ld = idx->ld;
    a.nlist = idx->nlist;
    a.Q = d_Q;
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.nq = nq;
    a.nprobe = nprobe;
    // one loop trip of every wave per workgroup for a handful of queries (1024 centroids: 32 workgroups per query),
    // more rows per workgroup as the batch grows
    const int per_iter = scan_rows_per_iter(idx->nch);
    int64_t want_blocks = std::max<int64_t>(1, 2048 / std::max(nq, 1));
    int64_t rpb = (idx->nlist + want_blocks - 1) / want_blocks;
    rpb = std::max<int64_t>(per_iter, (rpb + per_iter - 1) / per_iter * per_iter);
    a.rows_per_block = static_cast<int32_t>(rpb);
    a.blocks_per_query = static_cast<int32_t>((idx->nlist + rpb - 1) / rpb);
    HG_TRY(idx->s_tile.ensure(sizeof(float) * static_cast<size_t>(nq) * idx->nlist));
    HG_TRY(ensure_counters(idx, nq, st));
    a.dense = idx->s_tile.as<float>();
    a.done = idx->s_done.as<uint32_t>() + idx->s_done_n;
    a.wl.filed = idx->s_done.as<uint32_t>() + 2 * idx->s_done_n;  // two words behind the per-query counters
    const unsigned extra = a.wl_on ? kWorklistParts : 0;         // workgroups behind the queries' own
    a.listoff = idx->d_listoff;
    a.glistoff = idx->d_glistoff ? idx->d_glistoff : idx->d_listoff;
    a.pairs = pairs;
    a.probes = probes;
    a.qcnt = qcnt;
    const size_t lds = std::max<size_t>(sizeof(uint64_t) * ((nprobe <= kWave ? 2 * kNWave : kNWave) + 1) * nprobe, sizeof(float) * kSeedMax);
    HG_REQUIRE(lds <= 48 * 1024, HNSWGPU_ELIMIT, "nprobe too large for the fused routing kernel");
    const bool l2 = a.metric == METRIC_L2;
    // the tail with a wave per query: large home-list batches (a query whose nearest list is shorter than k seeds its threshold itself)
    const int64_t pqw = tune(HNSWGPU_TUNE_QUERY_WAVES, -1);
    const bool wave_tail = two_launches && rs && rs->home && !a.wl_on && a.bk_cnt && a.tau && a.probes && a.surv_cnt && nprobe <= kWave &&
                           rs->k <= kWave && idx->nlist <= 12 * 1024 && pqw != 0 && (pqw > 0 || nq >= 2048);
    if (two_launches) {
        // larger batches: the distances by workgroups that share their centroid rows among a group of queries, then the tail
        // as a launch of its own (plain loads: the distances come from an earlier launch)
        // (the group's queries are staged in LDS: 1 KB x NCH each, 48 KB at most)
        // cosine / dot, rows of 256 / 512 / 768 elements, from 256 queries: the GEMV-order distances on the f32 matrix cores
        // (1M x 768 / 1024 centroids, VALU vs matrix cores: 32 queries 6.7 vs 11.8 us, 64: 8.6 vs 11.8, 128: 12.4 vs 11.9, 256:
        // 20 vs 13, 1024: 54 vs 33, 4096: 150 vs 99)
        const int64_t rm = tune(HNSWGPU_TUNE_ROUTE_MFMA, -1);  // -1 that rule, 0 never, 1 whenever possible, > 1: slices per query group
        if (!l2 && idx->nch <= 3 && idx->ld == 256 * idx->nch && rm != 0 && (rm > 0 || nq >= 256)) {
            const int64_t ngroups = (nq + kRoute16Q - 1) / kRoute16Q, tiles = (idx->nlist + 15) / 16;
            int64_t split = std::max<int64_t>(1, std::min<int64_t>((tiles + kNWave - 1) / kNWave, (1024 + ngroups - 1) / ngroups));
            if (rm > 1) split = std::max<int64_t>(1, std::min<int64_t>(rm, tiles));  // (tuning: slices per query group)
            const int64_t tps = (tiles + split - 1) / split;
            a.rows_per_block = static_cast<int32_t>(tps * 16);
            a.blocks_per_query = static_cast<int32_t>((tiles + tps - 1) / tps);
            const size_t mlds = route_mfma16_lds(idx->nch);
#define CALLM(N)                                                                                                                            \
do {                                                                                                                                    \
    static bool attr_done[64] = {};                                                                                                     \
    if (mlds > 48 * 1024 && attr_needed(attr_done))                                                                                     \
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ivf_route_mfma16_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                   static_cast<int>(route_mfma16_lds(N))));                                                            \
    hipLaunchKernelGGL((ivf_route_mfma16_kernel<N>), dim3(static_cast<unsigned>(ngroups * a.blocks_per_query)), dim3(kWG), mlds, st, a); \
} while (0)
            switch (idx->nch) {
                case 1: CALLM(1); break;
                case 2: CALLM(2); break;
                default: CALLM(3); break;
            }
#undef CALLM
            HG_HIP(hipGetLastError());
            if (wave_tail) {
                a.defer_file = 1;
                HG_TRY(launch_route_tail_wave(a, idx->nch, st));
                HG_TRY(launch_bucket_fill(a, st));
                return 0;
            }
#define CALL(N, R, L) hipLaunchKernelGGL((ivf_route_tail_kernel<N, R, L>), dim3(static_cast<unsigned>(nq) + extra), dim3(kWG), lds, st, a)
            HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
            HG_HIP(hipGetLastError());
            return 0;
        }
        a.qgroup = static_cast<int32_t>(std::max(2, std::min(std::min(16, 48 / idx->nch), nq / 16)));
        const int64_t ngroups = (nq + a.qgroup - 1) / a.qgroup;
        const int64_t route_wgs = tune(HNSWGPU_TUNE_ROUTE_WGS, 2048);
        int64_t wb = std::max<int64_t>(1, route_wgs / ngroups);
        int64_t rp = (idx->nlist + wb - 1) / wb;
        rp = std::max<int64_t>(per_iter, (rp + per_iter - 1) / per_iter * per_iter);
        a.rows_per_block = static_cast<int32_t>(rp);
        a.blocks_per_query = static_cast<int32_t>((idx->nlist + rp - 1) / rp);
        const int64_t dblocks = ngroups * a.blocks_per_query;
        const size_t qlds = sizeof(float4) * kWave * idx->nch * a.qgroup;
#define CALL(N, R, L) hipLaunchKernelGGL((ivf_route_dist_kernel<N, R, L>), dim3(static_cast<unsigned>(dblocks)), dim3(kWG), qlds, st, a)
        HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
        HG_HIP(hipGetLastError());
        if (wave_tail) {
            a.defer_file = 1;
            HG_TRY(launch_route_tail_wave(a, idx->nch, st));
            HG_TRY(launch_bucket_fill(a, st));
            return 0;
        }
#define CALL(N, R, L) hipLaunchKernelGGL((ivf_route_tail_kernel<N, R, L>), dim3(static_cast<unsigned>(nq) + extra), dim3(kWG), lds, st, a)
        HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
        HG_HIP(hipGetLastError());
        return 0;
    }
    const int64_t blocks = static_cast<int64_t>(nq) * a.blocks_per_query;
#define CALL(N, R, L) hipLaunchKernelGGL((ivf_route_kernel<N, R, L>), dim3(static_cast<unsigned>(blocks) + extra), dim3(kWG), lds, st, a)
    HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// Every query against ALL `nrows` rows of the scan's table, top-k by a dense pass: the scan stores the distances
// ([nq][nrows] floats) and select_topk_kernel picks the k smallest (key = (distance, row), the same keys the
// partial-list path builds).  For a short table and a large k -- centroid routing: 1024 centroids, k = nprobe = 32 --
// the partial-list path emitted a k-slot list per wave for ~8 rows each and spent 31-38 us merging them.
int scan_dense_topk(hnswgpu_index *idx, ScanArgs a, int32_t nq, int64_t nrows, hipStream_t st) {
    a.mode = MODE_STORE;
    a.npairs = nq;
    a.nchunks = plan_chunks(idx->nch, nrows, nrows, nq, &a.chunk_rows);
    HG_TRY(idx->s_tile.ensure(sizeof(float) * static_cast<size_t>(nq) * nrows));
    HG_TRY(idx->s_ord.ensure(sizeof(uint32_t) * static_cast<size_t>(nq) * a.k));
    HG_TRY(idx->s_dist.ensure(sizeof(float) * static_cast<size_t>(nq) * a.k));
    a.out = idx->s_tile.as<float>();
    a.out_stride = nrows;
    const int k = a.k;
    HG_TRY(launch_scan(idx->nch, a, st));
    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.dist = a.out;
    s.stride = nrows;
    s.cnt_all = nrows;
    s.nq = nq;
    s.k = k;
    s.out_ord = idx->s_ord.as<uint32_t>();
    s.out_dist = idx->s_dist.as<float>();
    return launch_select(s, st);
}

int begin_call(hnswgpu_index *idx, hipStream_t st) {
    if (idx->ev_valid && idx->ev_stream != st) HG_HIP(hipStreamWaitEvent(st, idx->ev_last, 0));
    return 0;
}
int quiesce(hnswgpu_index *idx, hipStream_t st) {
    HG_TRY(begin_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    for (auto &sl : idx->slots)
        if (sl.st) HG_HIP(hipStreamSynchronize(sl.st));
    return 0;
}
int end_call(hnswgpu_index *idx, hipStream_t st) {
    if (!idx->ev_last) HG_HIP(hipEventCreateWithFlags(&idx->ev_last, hipEventDisableTiming));
    HG_HIP(hipEventRecord(idx->ev_last, st));
    idx->ev_stream = st;
    idx->ev_valid = true;
    return 0;
}

int ensure_pinned(hnswgpu_index *idx, size_t bytes) {
    if (bytes <= idx->h_pin_cap) return 0;
    if (idx->h_pin) (void)hipHostFree(idx->h_pin);
    idx->h_pin = nullptr;
    idx->h_pin_cap = 0;
    HG_HIP(hipHostMalloc(&idx->h_pin, bytes + bytes / 2, hipHostMallocDefault));
    idx->h_pin_cap = bytes + bytes / 2;
    return 0;
}

// ---- waiting on one 32-bit word: a short spin, then the kernel's futex (Linux; elsewhere the spin yields) ----------
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}
static void word_wake(std::atomic<uint32_t> *w) {
#if defined(__linux__)
    syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0);
#else
    (void)w;  // the waiter polls
#endif
}
static uint32_t word_wait_nonzero(std::atomic<uint32_t> *w) {
    for (int i = 0; i < 256; i++) {  // a batch usually answers in a few hundred microseconds: brief spin, then sleep
        const uint32_t v = w->load(std::memory_order_acquire);
        if (v) return v;
        cpu_relax();
    }
    for (;;) {
        const uint32_t v = w->load(std::memory_order_acquire);
        if (v) return v;
#if defined(__linux__)
        syscall(SYS_futex, reinterpret_cast<uint32_t *>(w), FUTEX_WAIT_PRIVATE, 0u, nullptr, nullptr, 0);
#else
        std::this_thread::yield();
#endif
    }
}
static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int slot_prepare(hnswgpu_index::Slot &s, size_t bytes) {
    if (!s.st) HG_HIP(hipStreamCreateWithFlags(&s.st, hipStreamNonBlocking));
    if (!s.d_again) {
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_again), sizeof(int32_t) * (kZcMaxQueries + 1)));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&s.d_done), sizeof(uint32_t) * 4));
        HG_HIP(hipMemsetAsync(s.d_again, 0, sizeof(int32_t) * (kZcMaxQueries + 1), s.st));
        HG_HIP(hipMemsetAsync(s.d_done, 0, sizeof(uint32_t) * 4, s.st));
        HG_HIP(hipStreamSynchronize(s.st));
    }
    if (bytes > s.cap) {
        if (s.h) {
            HG_HIP(hipStreamSynchronize(s.st));
            (void)hipHostFree(s.h);
        }
        s.h = s.d = nullptr;
        s.cap = 0;
        const size_t want = bytes + bytes / 2 + 4096;
        // mapped + coherent: the kernels address these bytes directly and the host sees their stores without a copy
        HG_HIP(hipHostMalloc(&s.h, want, hipHostMallocMapped | hipHostMallocCoherent));
        HG_HIP(hipHostGetDevicePointer(&s.d, s.h, 0));
        memset(s.h, 0, 64);
        s.cap = want;
        s.seq = 0;
    }
    return 0;
}

int slot_wait(hnswgpu_index::Slot &s, volatile uint32_t *flag, uint32_t seq) {
    const double t0 = now_us();
    for (uint64_t spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
        cpu_relax();
        if ((spins & 0x3fff) == 0x3fff && now_us() - t0 > 2e6) break;  // 2 s: something is wrong, ask the runtime
    }
    HG_HIP(hipStreamSynchronize(s.st));  // surfaces a kernel fault; a healthy launch has set the flag by now
    HG_REQUIRE(__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq, HNSWGPU_EHIP, "search kernel finished without publishing its results");
    return 0;
}

// Serve `me` through combiner `c`.  At any time at most ONE thread is the collector: it waits a moment for the callers
// a finishing batch has just released (they come straight back with their next query), takes the first queued request
// and everything behind it that `take` lets join, gives the collector role to the next queued thread and only then
// runs its batch -- so up to c.max_inflight batches overlap on the device.  The wait adapts to what is measured: at
// most a quarter of the previous batch's turnaround (and 100 us), over as soon as as many callers are queued as the
// previous batch held, or arrivals have stopped for 10 us.
int combine_search(hnswgpu_index::Combiner &c, hnswgpu_index::SearchReq &me,
                   const std::function<bool(const hnswgpu_index::SearchReq *, const hnswgpu_index::SearchReq *, int64_t)> &take,
                   const std::function<int(const std::vector<hnswgpu_index::SearchReq *> &, int32_t)> &run) {
    using Req = hnswgpu_index::SearchReq;
    bool collect;
    {
        std::lock_guard<std::mutex> cl(c.mu);
        try {
            c.pending.push_back(&me);
        } catch (...) {
            set_error("host allocation failed while queueing a search");
            return HNSWGPU_ENOMEM;
        }
        c.npending.store(static_cast<int>(c.pending.size()), std::memory_order_release);
        collect = !c.collector;
        if (collect) c.collector = true;
    }
    for (;;) {
        if (!collect) {
            const uint32_t st = word_wait_nonzero(&me.state);
            if (st == 1) break;  // served
            me.state.store(0, std::memory_order_relaxed);  // st == 2: the collector role was handed to me
        }
        collect = false;
        // ---- a free device slot
        {
            std::unique_lock<std::mutex> cl(c.mu);
            while (c.inflight >= c.max_inflight) {
                c.slot_waiter = &me;
                cl.unlock();
                word_wait_nonzero(&me.state);  // a finishing batch sets 2 again
                me.state.store(0, std::memory_order_relaxed);
                cl.lock();
            }
            c.slot_waiter = nullptr;
        }
        // ---- linger for the callers on their way back
        const int last_batch = c.last.load(std::memory_order_relaxed);
        if (last_batch > 1) {
            const double limit = std::min(100.0, std::max(10.0, c.last_run_us.load(std::memory_order_relaxed) * 0.25)), t0 = now_us();
            int seen = c.npending.load(std::memory_order_acquire);
            double t_change = t0;
            while (seen < last_batch) {
                const double t = now_us();
                if (t - t0 > limit || (seen > 1 && t - t_change > 10.0)) break;
                cpu_relax();
                const int n = c.npending.load(std::memory_order_acquire);
                if (n != seen) {
                    seen = n;
                    t_change = t;
                }
            }
        }
        // ---- split the queue.  Nothing below may leave the collector role taken or a request unanswered: an exception
        // (bad_alloc from the vectors, the message copies, anything `run` allocates) would otherwise park every current
        // and future caller for good.  Whatever happens, the batch is answered -- with an error if need be.
        std::vector<Req *> batch;
        int64_t total = 0;
        Req *next = nullptr;
        bool split_failed = false;
        {
            std::lock_guard<std::mutex> cl(c.mu);
            try {
                std::vector<Req *> rest;
                batch.reserve(c.pending.size());
                rest.reserve(c.pending.size());
                for (auto *r : c.pending) {
                    if (batch.empty() || take(batch[0], r, total)) {
                        batch.push_back(r);
                        total += r->nq;
                    } else {
                        rest.push_back(r);
                    }
                }
                c.pending.swap(rest);
            } catch (...) {  // nothing was taken: answer me with an error, somebody else collects
                split_failed = true;
                batch.clear();
                c.pending.erase(std::remove(c.pending.begin(), c.pending.end(), &me), c.pending.end());
            }
            c.npending.store(static_cast<int>(c.pending.size()), std::memory_order_release);
            if (!split_failed) {
                c.last.store(static_cast<int>(batch.size()), std::memory_order_relaxed);
                c.inflight++;
            }
            // hand the collector role on before running: the next batch forms while this one is on the device
            bool mine = split_failed || std::find(batch.begin(), batch.end(), &me) != batch.end();
            if (!mine) {
                collect = true;  // my own request is still queued (its k / ef differ from this batch's): I stay collector
            } else if (!c.pending.empty()) {
                next = c.pending.front();
            } else {
                c.collector = false;
            }
        }
        if (next) {
            next->state.store(2, std::memory_order_release);
            word_wake(&next->state);
        }
        if (split_failed) {
            set_error("host allocation failed while forming a combined batch");
            return HNSWGPU_ENOMEM;
        }
        int rc = 0;
        const char *msg = "";
        const double t_run = now_us();
        try {
            rc = run(batch, static_cast<int32_t>(total));
            if (rc) msg = hnswgpu_last_error();
        } catch (const std::bad_alloc &) {
            rc = HNSWGPU_ENOMEM;
            msg = "host allocation failed while serving a combined batch";
        } catch (...) {
            rc = HNSWGPU_EINVAL;
            msg = "unexpected exception while serving a combined batch";
        }
        Req *waiter = nullptr;
        {
            std::lock_guard<std::mutex> cl(c.mu);
            c.inflight--;
            c.last_run_us.store(now_us() - t_run, std::memory_order_relaxed);
            waiter = c.slot_waiter;
            c.slot_waiter = nullptr;
        }
        if (waiter) {  // the collector was parked for a device slot
            waiter->state.store(2, std::memory_order_release);
            word_wake(&waiter->state);
        }
        bool mine = false;
        for (auto *r : batch) {
            r->rc = rc;
            if (rc) {
                try {
                    r->err = msg;
                } catch (...) {  // the code still says what happened
                }
            }
            if (r == &me) {
                mine = true;
                continue;
            }
            r->state.store(1, std::memory_order_release);  // after this store `r` may be gone: only its address is used
            word_wake(&r->state);
        }
        if (mine) break;
    }
    if (me.rc) set_error("%s", me.err.c_str());
    return me.rc;
}

// the batched (query group resident in LDS) path: MFMA tiles for cosine / dot, the register-row VALU kernel for L2
bool tile_path_ok(const hnswgpu_index *idx) {
    return idx->metric == METRIC_L2 ? idx->dim <= kL2MaxDim : idx->dim <= kTileMaxDim;
}

int tile_mode() { return static_cast<int>(tune(HNSWGPU_TUNE_TILE, -1)); }

std::atomic<int64_t> g_tune[HNSWGPU_TUNE_COUNT];
std::atomic<int64_t> g_launch_count[HNSWGPU_COUNT_N];
// The six environment variables include/hnswgpu.h documents, read once when the library is loaded.
static const struct TuneInit {
    TuneInit() {
        for (auto &t : g_tune) t.store(kTuneUnset, std::memory_order_relaxed);
        static const struct {
            const char *name;
            int key;
        } env[] = {{"HNSWGPU_TILE_PAIRS", HNSWGPU_TUNE_TILE_PAIRS},       {"HNSWGPU_PREFILTER", HNSWGPU_TUNE_PREFILTER},
                   {"HNSWGPU_IVF_HALF", HNSWGPU_TUNE_IVF_HALF},           {"HNSWGPU_IVF_CALIBRATE", HNSWGPU_TUNE_IVF_CALIBRATE},
                   {"HNSWGPU_BUILD_THREADS", HNSWGPU_TUNE_BUILD_THREADS}, {"HNSWGPU_PREFETCH", HNSWGPU_TUNE_PREFETCH}};
        for (const auto &e : env)
            if (const char *v = getenv(e.name)) g_tune[e.key].store(atoll(v), std::memory_order_relaxed);
    }
} g_tune_init;
#ifdef HG_DIAG
int g_stream_dbg = 0, g_tile_dbg = 0;  // hnswgpu_debug_set_ablation: kernels with parts cut out, for timing only
#endif

unsigned long long *g_tile_dbg_buf = nullptr;  // set through hnswgpu_debug_set_tile_stamps (diagnostics only)

int launch_tile(const TileArgs &a, int64_t ngroups_bound, int dim, hipStream_t st) {
    int64_t blocks = ngroups_bound * a.nchunks;
    if (blocks <= 0) return 0;
    // explicit work list: workgroup b serves item (b % 8) * ceil(nitems / 8) + b / 8 (one contiguous eighth of the
    // list per XCD), so the grid must cover 8 * ceil(nitems / 8) workgroups, not just nitems
    if (a.members) blocks = (blocks + 7) & ~7LL;
    if (a.work_ctr) {  // persistent: one workgroup per CU (its LDS slot fills a CU) pulling items off the queue
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        blocks = std::min<int64_t>(blocks, std::max(cus, 8));
    }
    HG_REQUIRE(blocks < 2147483647LL, HNSWGPU_ELIMIT, "tile grid too large");
    if (a.metric == METRIC_L2 || a.gemv_order) {  // same groups, work list and outputs; rows in registers instead of MFMA tiles
        const size_t l2lds = l2_group_lds_bytes(a.ld);
        const int nch = static_cast<int>((a.ld / 4 + kWave - 1) / kWave);
        HG_REQUIRE(nch >= 1 && nch <= 4, HNSWGPU_ELIMIT, "the register-row group scan supports dim <= %d", kL2MaxDim);
        TileArgs b2 = a;
        b2.dbg = 0;
        b2.dbg_buf = nullptr;
        HG_REQUIRE(!b2.work_ctr, HNSWGPU_EINVAL, "the group kernel takes one work item per workgroup");
        const bool l2m = a.metric == METRIC_L2;
#define CALL_L2M(N, R, LM)                                                                                       \
    do {                                                                                                         \
        static bool l2_attr_done[64] = {};                                                                       \
        if (attr_needed(l2_attr_done))                                                                           \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&l2_group_kernel<N, R, LM>),               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
        hipLaunchKernelGGL((l2_group_kernel<N, R, LM>), dim3(static_cast<unsigned>(blocks)), dim3(kTileThreads), l2lds, st, b2); \
    } while (0)
#define CALL_L2(N, R)                 \
    do {                              \
        if (l2m) CALL_L2M(N, R, true); \
        else CALL_L2M(N, R, false);    \
    } while (0)
        switch (nch) {
            case 1: CALL_L2(1, 8); break;
            case 2: CALL_L2(2, 8); break;
            case 3: CALL_L2(3, 8); break;
            default: CALL_L2(4, 4); break;
        }
#undef CALL_L2
#undef CALL_L2M
        HG_HIP(hipGetLastError());
        return 0;
    }
    size_t lds = tile_lds_bytes(dim);
    static bool attr_done[64] = {};
    if (attr_needed(attr_done)) {
        HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tile_scan_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    TileArgs b = a;
#ifdef HG_DIAG
    b.dbg = g_tile_dbg;  // ablation timing only (results are wrong on purpose)
#else
    b.dbg = 0;
#endif
    b.dbg_buf = g_tile_dbg_buf;
    hipLaunchKernelGGL(tile_scan_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kTileThreads), lds, st, b);
    HG_HIP(hipGetLastError());
    return 0;
}

int launch_select(const SelectArgs &a0, hipStream_t st) {
    if (a0.nq <= 0) return 0;
    SelectArgs a = a0;
    // waves per query: measured at 1024 queries x 31k candidates (k = 10): 232 / 135 / 84 / 90 / 106 us with
    // 1 / 2 / 4 / 8 / 16 waves (past 4 the second-level merge and the larger workgroups cost more than the shorter
    // streams save); the per-query count is only known on the device, the row stride bounds it
    const int64_t cand = a.q_cnt ? a.stride : a.cnt_all;
    // (a small batch leaves most CUs idle: shorter streams per wave -- 32 queries x 35k candidates: 23 -> 15 us with 8 waves)
    const int64_t per_wave = a.nq <= 128 ? 4096 : 16384;
    int w = 1;
    while (w < 16 && cand >= per_wave * w && static_cast<int64_t>(a.nq) * w * 2 <= 16384) w *= 2;
    if (const int64_t e = tune(HNSWGPU_TUNE_SELECT_W, 0)) w = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(16, e)));  // tuning override
    while (w & (w - 1)) w &= w - 1;
    a.wpq = w;
    a.vec4 = (a.stride % 4 == 0 && reinterpret_cast<uintptr_t>(a.dist) % 16 == 0) ? 1 : 0;
    const size_t lds = sizeof(uint64_t) * (w == 1 ? kNWave : w + 1) * a.k;
    HG_REQUIRE(lds <= 150 * 1024, HNSWGPU_ELIMIT, "k too large for the select kernel");
    if (lds > 48 * 1024) {
        static bool attr_done[64] = {};
        if (attr_needed(attr_done)) {
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&select_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        }
    }
    if (w == 1)
        hipLaunchKernelGGL(select_topk_kernel, dim3((a.nq + kNWave - 1) / kNWave), dim3(kWG), lds, st, a);
    else
        hipLaunchKernelGGL(select_topk_kernel, dim3(a.nq), dim3(w * kWave), lds, st, a);
    HG_HIP(hipGetLastError());
    return 0;
}

int pad_queries(hnswgpu_index *idx, const float *d_Q, int64_t qld, int32_t nq, hipStream_t st) {
    HG_TRY(idx->s_qp.ensure(sizeof(float) * static_cast<size_t>(nq) * idx->ld));
    HG_TRY(idx->s_qn.ensure(sizeof(float) * static_cast<size_t>(nq)));
    if (idx->ld != idx->dim) HG_HIP(hipMemsetAsync(idx->s_qp.p, 0, sizeof(float) * static_cast<size_t>(nq) * idx->ld, st));
    HG_HIP(hipMemcpy2DAsync(idx->s_qp.p, sizeof(float) * idx->ld, d_Q, sizeof(float) * qld, sizeof(float) * idx->dim, nq,
                            hipMemcpyDeviceToDevice, st));
    return launch_norms(idx->nch, idx->s_qp.as<float>(), idx->ld, nq, idx->s_qn.as<float>(), st);
}

__global__ void key_decode_kernel(const unsigned long long *keys, int64_t cnt, uint32_t *ord, float *dist) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const uint64_t key = keys[i];
    const bool ok = key != ~0ull;
    ord[i] = ok ? static_cast<uint32_t>(key) : 0xffffffffu;
    dist[i] = ok ? key_dist(key) : __uint_as_float(0x7f800000u);
}

// k = 1: the nearest row of every query with the argmin fused into the tile kernel (assign-to-nearest-centroid,
// ivf_flat.clj:79-90): no dense distance array, no selection pass, one launch for any number of queries.
static int tile_argmin_all(hnswgpu_index *idx, const float *Qp, const float *q_norms, int32_t nq, const float *rows,
                           const float *row_norms, int64_t nrows, hipStream_t st, int prof_slot) {
    const int tq = tile_tq(idx->dim);
    const size_t kbytes = sizeof(unsigned long long) * static_cast<size_t>(nq);
    HG_TRY(idx->s_tile.ensure(kbytes));
    HG_HIP(hipMemsetAsync(idx->s_tile.p, 0xff, kbytes, st));
    TileArgs t;
    memset(&t, 0, sizeof(t));
    t.rows = rows;
    t.row_norms = row_norms;
    t.ld = idx->ld;
    t.dim = idx->dim;
    t.metric = idx->metric;
    t.Qp = Qp;
    t.q_norms = q_norms;
    t.nrows_all = nrows;
    t.nq = nq;
    t.out_key = idx->s_tile.as<unsigned long long>();
    const int64_t groups = (nq + tq - 1) / tq;
    const int64_t tiles = (nrows + kTileRows - 1) / kTileRows;
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(tiles, (2048 + groups - 1) / groups));
    const int64_t cr = ((tiles + want - 1) / want) * kTileRows;
    t.chunk_rows = static_cast<int32_t>(cr);
    t.nchunks = static_cast<int32_t>(std::max<int64_t>(1, (tiles + cr / kTileRows / 2) / (cr / kTileRows)));
    hipEvent_t e0;
    prof_begin(idx, prof_slot, st, &e0);
    HG_TRY(launch_tile(t, groups, idx->dim, st));
    prof_end(idx, prof_slot, st, e0);
    hipLaunchKernelGGL(key_decode_kernel, dim3(static_cast<unsigned>((nq + 255) / 256)), dim3(256), 0, st, t.out_key,
                       static_cast<int64_t>(nq), idx->s_ord.as<uint32_t>(), idx->s_dist.as<float>());
    HG_HIP(hipGetLastError());
    return 0;
}

int tile_topk_all(hnswgpu_index *idx, const float *Qp, const float *q_norms, int32_t nq, const float *rows,
                  const float *row_norms, int64_t nrows, int32_t k, hipStream_t st, int prof_slot, bool gemv_order) {
    HG_TRY(idx->s_ord.ensure(sizeof(uint32_t) * static_cast<size_t>(nq) * k));
    HG_TRY(idx->s_dist.ensure(sizeof(float) * static_cast<size_t>(nq) * k));
    if (k == 1 && !gemv_order) return tile_argmin_all(idx, Qp, q_norms, nq, rows, row_norms, nrows, st, prof_slot);
    // distance scratch [qb][nrows]; bound it to ~2 GiB by batching the queries
    const int tq = tile_tq(idx->dim);
    int64_t qb = std::max<int64_t>(tq, ((2LL << 30) / (4 * std::max<int64_t>(nrows, 1))) / tq * tq);
    qb = std::min<int64_t>(qb, (nq + tq - 1) / tq * tq);
    HG_TRY(idx->s_tile.ensure(sizeof(float) * static_cast<size_t>(qb) * nrows));
    for (int64_t q0 = 0; q0 < nq; q0 += qb) {
        int32_t nb = static_cast<int32_t>(std::min<int64_t>(qb, nq - q0));
        TileArgs t;
        memset(&t, 0, sizeof(t));
        t.rows = rows;
        t.row_norms = row_norms;
        t.ld = idx->ld;
        t.dim = idx->dim;
        t.metric = idx->metric;
        t.Qp = Qp + q0 * idx->ld;
        t.q_norms = q_norms + q0;
        t.nrows_all = nrows;
        t.nq = nb;
        t.out_stride = nrows;
        t.gemv_order = gemv_order ? 1 : 0;
        int64_t groups = (nb + tq - 1) / tq;
        int64_t tiles = (nrows + kTileRows - 1) / kTileRows;
        int64_t want = std::max<int64_t>(1, std::min<int64_t>(tiles, (2048 + groups - 1) / groups));
        int64_t cr = ((tiles + want - 1) / want) * kTileRows;
        t.chunk_rows = static_cast<int32_t>(cr);
        t.nchunks = static_cast<int32_t>(std::max<int64_t>(1, (tiles + cr / kTileRows / 2) / (cr / kTileRows)));
        t.out = idx->s_tile.as<float>();
        hipEvent_t e0;
        prof_begin(idx, prof_slot, st, &e0);
        HG_TRY(launch_tile(t, groups, idx->dim, st));
        prof_end(idx, prof_slot, st, e0);
        SelectArgs s;
        memset(&s, 0, sizeof(s));
        s.dist = t.out;
        s.stride = nrows;
        s.cnt_all = nrows;
        s.nq = nb;
        s.k = k;
        s.out_ord = idx->s_ord.as<uint32_t>() + q0 * k;
        s.out_dist = idx->s_dist.as<float>() + q0 * k;
        HG_TRY(launch_select(s, st));
    }
    return 0;
}

int upload_queries(hnswgpu_index *idx, const float *Q, int32_t nq, hipStream_t st) {
    size_t bytes = sizeof(float) * static_cast<size_t>(nq) * idx->dim;
    HG_TRY(idx->s_q.ensure(bytes));
    HG_HIP(hipMemcpyAsync(idx->s_q.p, Q, bytes, hipMemcpyHostToDevice, st));
    return 0;
}

__global__ void ord_to_ids_kernel(const uint32_t *ord, int64_t cnt, int32_t *ids) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < cnt) ids[i] = ord[i] == 0xffffffffu ? -1 : static_cast<int32_t>(ord[i]);
}

// [nshard][nq][k] (id, dist) -> [nq][k]; ties keep the lower shard (then the lower rank) first
__global__ __launch_bounds__(kWave) void merge_shards_kernel(const int32_t *ids, const float *dist, int nshard,
                                                             int nq, int kin, int k, int32_t *out_ids,
                                                             float *out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *list = reinterpret_cast<uint64_t *>(smem);
    const int lane = threadIdx.x, q = blockIdx.x;
    int cnt = 0;
    uint64_t thr = ~0ull;
    const int tot = nshard * kin;
    for (int base = 0; base < tot; base += kWave) {
        int i = base + lane;
        uint64_t key = ~0ull;
        if (i < tot) {
            int s = i / kin, r = i % kin;
            int64_t src = (static_cast<int64_t>(s) * nq + q) * kin + r;
            if (ids[src] >= 0) key = make_key(dist[src], static_cast<uint32_t>(i));
        }
        uint64_t mask = __ballot(key < thr);
        while (mask) {
            int b = __ffsll(static_cast<unsigned long long>(mask)) - 1;
            mask &= mask - 1;
            uint64_t kb = lane_bcast(key, b);
            if (kb < thr) {
                wave_insert(list, cnt, k, kb, lane);
                thr = cnt == k ? list[k - 1] : ~0ull;
            }
        }
    }
    for (int i = lane; i < k; i += kWave) {
        bool ok = i < cnt;
        int32_t id = -1;
        float d = __uint_as_float(0x7f800000u);
        if (ok) {
            uint32_t o = static_cast<uint32_t>(list[i]);
            int s = o / kin, r = o % kin;
            int64_t src = (static_cast<int64_t>(s) * nq + q) * kin + r;
            id = ids[src];
            d = dist[src];
        }
        out_ids[static_cast<int64_t>(q) * k + i] = id;
        out_dist[static_cast<int64_t>(q) * k + i] = d;
    }
}

// [nshard][nq][k] (id, dist, order) -> [nq][k] by (distance, order): the shards of ONE inverted-file index number
// their candidates by the position in the candidate stream of the whole index (Pair::gord_base), so this merge equals
// the unsharded search's own stable sort (ivf_flat.clj:291-294) -- ties included.  One wave per query.
__global__ __launch_bounds__(kWave) void merge_keyed_kernel(const int32_t *ids, const float *dist, const uint32_t *ord,
                                                            int nshard, int nq, int k, int32_t *out_ids,
                                                            float *out_dist) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *list = reinterpret_cast<uint64_t *>(smem);
    uint32_t *src = reinterpret_cast<uint32_t *>(list + k);
    const int lane = threadIdx.x, q = blockIdx.x;
    int cnt = 0;
    uint64_t thr = ~0ull;
    const int tot = nshard * k;
    for (int base = 0; base < tot; base += kWave) {
        const int i = base + lane;
        uint64_t key = ~0ull;
        if (i < tot) {
            const int64_t at = (static_cast<int64_t>(i / k) * nq + q) * k + i % k;
            if (ids[at] >= 0) key = make_key(dist[at], ord[at]);
        }
        uint64_t mask = __ballot(key < thr);
        while (mask) {
            const int b = __ffsll(static_cast<unsigned long long>(mask)) - 1;
            mask &= mask - 1;
            const uint64_t kb = lane_bcast(key, b);
            if (kb < thr) {
                wave_insert_kv(list, src, cnt, k, kb, static_cast<uint32_t>(base + b), lane);
                thr = cnt == k ? list[k - 1] : ~0ull;
            }
        }
    }
    for (int i = lane; i < k; i += kWave) {
        int32_t id = -1;
        float d = __uint_as_float(0x7f800000u);
        if (i < cnt) {
            const uint32_t o = src[i];
            const int64_t at = (static_cast<int64_t>(o / k) * nq + q) * k + o % k;
            id = ids[at];
            d = dist[at];
        }
        out_ids[static_cast<int64_t>(q) * k + i] = id;
        out_dist[static_cast<int64_t>(q) * k + i] = d;
    }
}

}  // namespace hg

using namespace hg;

extern "C" {

int hnswgpu_version(void) { return HNSWGPU_VERSION; }
const char *hnswgpu_last_error(void) { return g_err; }

int hnswgpu_device_count(int32_t *count) {
    HG_REQUIRE(count, HNSWGPU_EINVAL, "count is null");
    int c = 0;
    HG_HIP(hipGetDeviceCount(&c));
    *count = c;
    return 0;
}

static int create_common(int64_t n, int32_t dim, int32_t metric, int32_t device, hnswgpu_index **out) {
    HG_REQUIRE(out, HNSWGPU_EINVAL, "out is null");
    HG_REQUIRE(n >= 0 && n < 2147483647LL, HNSWGPU_EINVAL, "n out of range (%lld)", (long long)n);
    HG_REQUIRE(dim >= 1, HNSWGPU_EINVAL, "dim must be >= 1");
    HG_REQUIRE(metric >= 0 && metric <= 2, HNSWGPU_EINVAL, "unknown metric %d", metric);
    int64_t ld = (static_cast<int64_t>(dim) + 3) / 4 * 4;
    int nch = pick_nch(ld);
    HG_REQUIRE(nch > 0, HNSWGPU_ELIMIT, "dim %d > 3072 is not supported", dim);
    HG_HIP(hipSetDevice(device));
    hnswgpu_index *idx = new (std::nothrow) hnswgpu_index();
    HG_REQUIRE(idx, HNSWGPU_ENOMEM, "host allocation failed");
    idx->device = device;
    idx->metric = metric;
    idx->n = n;
    idx->dim = dim;
    idx->ld = ld;
    idx->nch = nch;
    const int64_t rej = tune(HNSWGPU_TUNE_PREFILTER, 1);  // default of hnswgpu_set_rejection_test
    idx->rejection_mode = rej < 0 || rej > 2 ? 1 : static_cast<int>(rej);
    (void)hipDeviceGetAttribute(&idx->cus, hipDeviceAttributeMultiprocessorCount, device);
    if (idx->cus <= 0) idx->cus = 256;
    idx->cmb_hnsw.max_inflight = 2;  // two Slots: small synchronous HNSW batches overlap on the device
    idx->cmb_ivf.max_inflight = 1;   // the IVF path works in the index's shared scratch: one batch at a time
    *out = idx;
    return 0;
}

static int finish_create(hnswgpu_index *idx, hipStream_t st) {
    if (idx->n > 0) {
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_norms), sizeof(float) * idx->n));
        HG_TRY(launch_norms(idx->nch, idx->d_base, idx->ld, idx->n, idx->d_norms, st));
    }
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_create(const float *base, int64_t n, int32_t dim, int32_t metric, int32_t device,
                   hnswgpu_index **out) {
    HG_REQUIRE(out, HNSWGPU_EINVAL, "out is null");
    *out = nullptr;
    HG_REQUIRE(base || n == 0, HNSWGPU_EINVAL, "base is null");
    hnswgpu_index *idx = nullptr;
    HG_TRY(create_common(n, dim, metric, device, &idx));
    int rc = [&]() -> int {
        HG_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
        if (n > 0) {
            HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_base), sizeof(float) * n * idx->ld));
            if (idx->ld == dim) {
                HG_HIP(hipMemcpyAsync(idx->d_base, base, sizeof(float) * n * dim, hipMemcpyHostToDevice, idx->stream));
            } else {
                HG_HIP(hipMemsetAsync(idx->d_base, 0, sizeof(float) * n * idx->ld, idx->stream));
                HG_HIP(hipMemcpy2DAsync(idx->d_base, sizeof(float) * idx->ld, base, sizeof(float) * dim,
                                        sizeof(float) * dim, n, hipMemcpyHostToDevice, idx->stream));
            }
        }
        return finish_create(idx, idx->stream);
    }();
    if (rc != 0) {
        hnswgpu_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int hnswgpu_create_dev(const float *d_base, int64_t n, int32_t dim, int64_t ld, int32_t metric, int32_t device,
                       void *stream, hnswgpu_index **out) {
    HG_REQUIRE(out, HNSWGPU_EINVAL, "out is null");
    *out = nullptr;
    HG_REQUIRE(d_base || n == 0, HNSWGPU_EINVAL, "d_base is null");
    HG_REQUIRE(ld >= dim, HNSWGPU_EINVAL, "ld < dim");
    hnswgpu_index *idx = nullptr;
    HG_TRY(create_common(n, dim, metric, device, &idx));
    int rc = [&]() -> int {
        HG_HIP(hipStreamCreateWithFlags(&idx->stream, hipStreamNonBlocking));
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (n > 0) {
            HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_base), sizeof(float) * n * idx->ld));
            if (idx->ld != dim) HG_HIP(hipMemsetAsync(idx->d_base, 0, sizeof(float) * n * idx->ld, st));
            HG_HIP(hipMemcpy2DAsync(idx->d_base, sizeof(float) * idx->ld, d_base, sizeof(float) * ld,
                                    sizeof(float) * dim, n, hipMemcpyDeviceToDevice, st));
        }
        return finish_create(idx, st);
    }();
    if (rc != 0) {
        hnswgpu_destroy(idx);
        return rc;
    }
    *out = idx;
    return 0;
}

int hnswgpu_destroy(hnswgpu_index *idx) {
    if (!idx) return 0;
    (void)hipSetDevice(idx->device);
    if (idx->stream) (void)hipStreamSynchronize(idx->stream);
    for (auto &sl : idx->slots)  // before anything a traversal in flight may read is freed
        if (sl.st) (void)hipStreamSynchronize(sl.st);
    if (idx->lrows_alias) idx->d_lrows = idx->d_lnorms = nullptr;  // the base rows in place: freed once, below
    void *ptrs[] = {idx->d_base,  idx->d_norms,  idx->d_qrows,  idx->d_qmeta,   idx->d_lcmeta, idx->d_lctile, idx->d_lhalf, idx->d_lhmeta, idx->d_rej_stats, idx->d_levels, idx->d_l0,      idx->d_upadj,  idx->d_upoff,  idx->d_glistoff,
                    idx->d_cent,  idx->d_cnorms, idx->d_lrows,  idx->d_lnorms,  idx->d_listoff, idx->d_listids};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    DevBuf *bufs[] = {&idx->s_q,   &idx->s_partial, &idx->s_ord,   &idx->s_dist, &idx->s_pairs, &idx->s_ids,
                      &idx->s_outd, &idx->s_probes,  &idx->s_stats, &idx->s_misc, &idx->s_misc2, &idx->s_vis, &idx->s_qp, &idx->s_qn, &idx->s_tile, &idx->s_grp, &idx->s_done, &idx->s_pf, &idx->s_solo, &idx->s_bk, &idx->s_heavy, &idx->s_home, &idx->s_dh};
    for (DevBuf *b : bufs) b->release();
    for (int s = 0; s < PROF_N; s++)
        for (auto &pr : idx->prof_ev[s]) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    if (idx->ev_last) (void)hipEventDestroy(idx->ev_last);
    if (idx->ev_hnsw_cal) (void)hipEventDestroy(idx->ev_hnsw_cal);
    if (idx->d_hnsw_cal) (void)hipFree(idx->d_hnsw_cal);
    if (idx->hnsw_cal_host) (void)hipHostFree(idx->hnsw_cal_host);
    if (idx->stream) (void)hipStreamDestroy(idx->stream);
    if (idx->h_pin) (void)hipHostFree(idx->h_pin);
    for (auto &sl : idx->slots) {
        if (sl.st) (void)hipStreamSynchronize(sl.st);
        if (sl.h) (void)hipHostFree(sl.h);
        if (sl.d_again) (void)hipFree(sl.d_again);
        if (sl.d_done) (void)hipFree(sl.d_done);
        if (sl.st) (void)hipStreamDestroy(sl.st);
    }
    delete idx;
    return 0;
}

int hnswgpu_info(const hnswgpu_index *idx, int64_t *n, int32_t *dim, int32_t *metric, int32_t *has_graph,
                 int32_t *nlist) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    if (n) *n = idx->n;
    if (dim) *dim = idx->dim;
    if (metric) *metric = idx->metric;
    if (has_graph) *has_graph = idx->has_graph ? 1 : 0;
    if (nlist) *nlist = idx->nlist;
    return 0;
}

int hnswgpu_sync(hnswgpu_index *idx) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_HIP(hipSetDevice(idx->device));
    HG_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

int hnswgpu_pair_distance(int32_t metric, const float *a, const float *b, int32_t dim, int32_t device, float *out) {
    HG_REQUIRE(a && b && out, HNSWGPU_EINVAL, "null argument");
    hnswgpu_index *idx = nullptr;
    HG_TRY(hnswgpu_create(b, 1, dim, metric, device, &idx));
    int rc = hnswgpu_batch_distances(idx, a, nullptr, 1, out);
    hnswgpu_destroy(idx);
    return rc;
}

int hnswgpu_batch_distances(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out) {
    HG_REQUIRE(idx && q && (out || m == 0), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(m >= 0, HNSWGPU_EINVAL, "m < 0");
    if (m == 0) return 0;
    HG_REQUIRE(ids || m <= idx->n, HNSWGPU_EINVAL, "m > n with implicit ids");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, q, 1, st));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * m));
    GatherArgs g;
    memset(&g, 0, sizeof(g));
    g.rows = idx->d_base;
    g.row_norms = idx->d_norms;
    g.ld = idx->ld;
    g.n = idx->n;
    g.q = idx->s_q.as<float>();
    g.dim = idx->dim;
    g.metric = idx->metric;
    g.ids = nullptr;
    if (ids) {
        HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * m));
        HG_HIP(hipMemcpyAsync(idx->s_ids.p, ids, sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
        g.ids = idx->s_ids.as<int32_t>();
    }
    g.m = m;
    g.out = idx->s_outd.as<float>();
    HG_TRY(launch_gather(idx->nch, g, 1, st));
    HG_HIP(hipMemcpyAsync(out, g.out, sizeof(float) * m, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_norms(hnswgpu_index *idx, float *out_norms) {
    HG_REQUIRE(idx && (out_norms || idx->n == 0), HNSWGPU_EINVAL, "null argument");
    if (idx->n == 0) return 0;
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    HG_HIP(hipMemcpyAsync(out_norms, idx->d_norms, sizeof(float) * idx->n, hipMemcpyDeviceToHost, idx->stream));
    HG_HIP(hipStreamSynchronize(idx->stream));
    return 0;
}

static int exact_knn_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t *d_ids,
                             float *d_dist, hipStream_t st) {
    const int tm = tile_mode();
    if (tile_path_ok(idx) && tm != 0 && (tm == 1 || nq >= 16)) {  // many queries share the rows: MFMA tiles
        HG_TRY(pad_queries(idx, d_Q, idx->dim, nq, st));
        HG_TRY(tile_topk_all(idx, idx->s_qp.as<float>(), idx->s_qn.as<float>(), nq, idx->d_base, idx->d_norms, idx->n, k,
                             st, PROF_IVF_SCAN));
        int64_t cnt = static_cast<int64_t>(nq) * k;
        hipLaunchKernelGGL(ord_to_ids_kernel, dim3(static_cast<unsigned>((cnt + 255) / 256)), dim3(256), 0, st,
                           idx->s_ord.as<uint32_t>(), cnt, d_ids);
        HG_HIP(hipGetLastError());
        HG_HIP(hipMemcpyAsync(d_dist, idx->s_dist.p, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
        return 0;
    }
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = idx->d_base;
    a.row_norms = idx->d_norms;
    a.ld = idx->ld;
    a.nrows_all = idx->n;
    a.Q = d_Q;
    a.qld = idx->dim;
    a.q_norms = nullptr;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.pairs = nullptr;
    a.k = k;
    a.role = ROLE_EXACT;
    HG_TRY(scan_topk(idx, a, nq, 1, idx->n, st, PROF_IVF_SCAN));
    int64_t cnt = static_cast<int64_t>(nq) * k;
    hipLaunchKernelGGL(ord_to_ids_kernel, dim3(static_cast<unsigned>((cnt + 255) / 256)), dim3(256), 0, st,
                       idx->s_ord.as<uint32_t>(), cnt, d_ids);
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpyAsync(d_dist, idx->s_dist.p, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int check_search_args(const hnswgpu_index *idx, const void *Q, int32_t nq, int32_t k, const void *ids,
                             const void *dist) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(nq >= 0, HNSWGPU_EINVAL, "nq < 0");
    HG_REQUIRE(k >= 1, HNSWGPU_EINVAL, "k must be >= 1");
    HG_REQUIRE(k <= 1024, HNSWGPU_ELIMIT, "k > 1024 is not supported");
    HG_REQUIRE(nq == 0 || (Q && ids && dist), HNSWGPU_EINVAL, "null argument");
    return 0;
}

static void fill_empty(int32_t *ids, float *dist, int64_t cnt) {
    for (int64_t i = 0; i < cnt; i++) {
        ids[i] = -1;
        dist[i] = __builtin_inff();
    }
}

int hnswgpu_exact_knn_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t *d_out_ids,
                          float *d_out_dist, void *stream) {
    HG_TRY(check_search_args(idx, d_Q, nq, k, d_out_ids, d_out_dist));
    if (nq == 0) return 0;
    HG_REQUIRE(idx->n > 0, HNSWGPU_ESTATE, "empty index: use the host entry point");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(exact_knn_enqueue(idx, d_Q, nq, k, d_out_ids, d_out_dist, st));
    return end_call(idx, st);
}

int hnswgpu_exact_knn(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t *out_ids,
                      float *out_dist) {
    HG_TRY(check_search_args(idx, Q, nq, k, out_ids, out_dist));
    if (nq == 0) return 0;
    int64_t cnt = static_cast<int64_t>(nq) * k;
    if (idx->n == 0) {
        fill_empty(out_ids, out_dist, cnt);
        return 0;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, Q, nq, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    HG_TRY(exact_knn_enqueue(idx, idx->s_q.as<float>(), nq, k, idx->s_ids.as<int32_t>(), idx->s_outd.as<float>(), st));
    HG_HIP(hipMemcpyAsync(out_ids, idx->s_ids.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(out_dist, idx->s_outd.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_merge_lists_dev(int32_t device, const int32_t *d_ids, const float *d_dist, int32_t nshard, int32_t nq,
                            int32_t k_in, int32_t k_out, int32_t *d_out_ids, float *d_out_dist, void *stream) {
    HG_REQUIRE(nshard >= 1 && nq >= 0 && k_in >= 1 && k_out >= 1, HNSWGPU_EINVAL, "bad sizes");
    HG_REQUIRE(k_out <= 1024, HNSWGPU_ELIMIT, "k > 1024 is not supported");
    HG_REQUIRE(static_cast<int64_t>(nshard) * k_in < 2147483647LL, HNSWGPU_ELIMIT, "nshard * k_in too large");
    if (nq == 0) return 0;
    HG_REQUIRE(d_ids && d_dist && d_out_ids && d_out_dist, HNSWGPU_EINVAL, "null argument");
    HG_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(merge_shards_kernel, dim3(nq), dim3(kWave), sizeof(uint64_t) * k_out,
                       static_cast<hipStream_t>(stream), d_ids, d_dist, nshard, nq, k_in, k_out, d_out_ids,
                       d_out_dist);
    HG_HIP(hipGetLastError());
    return 0;
}

int hnswgpu_merge_topk_dev(int32_t device, const int32_t *d_ids, const float *d_dist, int32_t nshard, int32_t nq,
                           int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream) {
    return hnswgpu_merge_lists_dev(device, d_ids, d_dist, nshard, nq, k, k, d_out_ids, d_out_dist, stream);
}

int hnswgpu_merge_keyed_dev(int32_t device, const int32_t *d_ids, const float *d_dist, const uint32_t *d_order,
                            int32_t nshard, int32_t nq, int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream) {
    HG_REQUIRE(nshard >= 1 && nq >= 0 && k >= 1, HNSWGPU_EINVAL, "bad sizes");
    HG_REQUIRE(k <= 1024, HNSWGPU_ELIMIT, "k > 1024 is not supported");
    HG_REQUIRE(static_cast<int64_t>(nshard) * k < 2147483647LL, HNSWGPU_ELIMIT, "nshard * k must be below 2^31");
    if (nq == 0) return 0;
    HG_REQUIRE(d_ids && d_dist && d_order && d_out_ids && d_out_dist, HNSWGPU_EINVAL, "null argument");
    HG_HIP(hipSetDevice(device));
    hipLaunchKernelGGL(merge_keyed_kernel, dim3(nq), dim3(kWave), (sizeof(uint64_t) + sizeof(uint32_t)) * k,
                       static_cast<hipStream_t>(stream), d_ids, d_dist, d_order, nshard, nq, k, d_out_ids, d_out_dist);
    HG_HIP(hipGetLastError());
    return 0;
}

// ---- re-rank: per query, exact distances to a candidate list, stable ascending sort, take k ----
__global__ void rerank_decode_kernel(const uint32_t *ord, int64_t cnt, int k, const int32_t *cand, int m,
                                     int32_t *out_ids) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    uint32_t o = ord[i];
    out_ids[i] = o == 0xffffffffu ? -1 : cand[(i / k) * m + o];
}

static int rerank_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, const int32_t *d_cand, int32_t m,
                          int32_t k, int32_t *d_ids, float *d_dist, hipStream_t st) {
    const int64_t nkeys = static_cast<int64_t>(nq) * m, cnt = static_cast<int64_t>(nq) * k;
    HG_TRY(idx->s_partial.ensure(sizeof(uint64_t) * nkeys));
    HG_TRY(idx->s_ord.ensure(sizeof(uint32_t) * cnt));
    HG_TRY(idx->s_dist.ensure(sizeof(float) * cnt));
    GatherArgs g;
    memset(&g, 0, sizeof(g));
    g.rows = idx->d_base;
    g.row_norms = idx->d_norms;
    g.ld = idx->ld;
    g.n = idx->n;
    g.q = d_Q;
    g.qld = idx->dim;
    g.dim = idx->dim;
    g.metric = idx->metric;
    g.ids = d_cand;
    g.m = m;
    g.out_keys = idx->s_partial.as<uint64_t>();
    HG_TRY(launch_gather(idx->nch, g, nq, st));
    MergeArgs mg;
    mg.partial = g.out_keys;
    mg.keys_per_query = m;
    mg.nq = nq;
    mg.k = k;
    mg.out_ord = idx->s_ord.as<uint32_t>();
    mg.out_dist = idx->s_dist.as<float>();
    HG_TRY(launch_merge(mg, st));
    hipLaunchKernelGGL(rerank_decode_kernel, dim3(static_cast<unsigned>((cnt + 255) / 256)), dim3(256), 0, st,
                       mg.out_ord, cnt, k, d_cand, m, d_ids);
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpyAsync(d_dist, idx->s_dist.p, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
    return 0;
}

static int check_rerank_args(const hnswgpu_index *idx, const void *Q, int32_t nq, const void *cand, int32_t m,
                             int32_t k, const void *ids, const void *dist) {
    HG_TRY(check_search_args(idx, Q, nq, k, ids, dist));
    HG_REQUIRE(m >= 1, HNSWGPU_EINVAL, "m must be >= 1");
    HG_REQUIRE(nq == 0 || cand, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(static_cast<int64_t>(nq) * m < (1LL << 31), HNSWGPU_ELIMIT, "nq * m too large");
    return 0;
}

int hnswgpu_rerank_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, const int32_t *d_cand, int32_t m,
                       int32_t k, int32_t *d_out_ids, float *d_out_dist, void *stream) {
    HG_TRY(check_rerank_args(idx, d_Q, nq, d_cand, m, k, d_out_ids, d_out_dist));
    if (nq == 0) return 0;
    HG_REQUIRE(idx->n > 0, HNSWGPU_ESTATE, "empty index: use the host entry point");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(rerank_enqueue(idx, d_Q, nq, d_cand, m, k, d_out_ids, d_out_dist, st));
    return end_call(idx, st);
}

int hnswgpu_rerank(hnswgpu_index *idx, const float *Q, int32_t nq, const int32_t *cand, int32_t m, int32_t k,
                   int32_t *out_ids, float *out_dist) {
    HG_TRY(check_rerank_args(idx, Q, nq, cand, m, k, out_ids, out_dist));
    if (nq == 0) return 0;
    const int64_t cnt = static_cast<int64_t>(nq) * k;
    if (idx->n == 0) {
        fill_empty(out_ids, out_dist, cnt);
        return 0;
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, Q, nq, st));
    const size_t cbytes = sizeof(int32_t) * static_cast<size_t>(nq) * m;
    HG_TRY(idx->s_misc.ensure(cbytes));
    HG_HIP(hipMemcpyAsync(idx->s_misc.p, cand, cbytes, hipMemcpyHostToDevice, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    HG_TRY(rerank_enqueue(idx, idx->s_q.as<float>(), nq, idx->s_misc.as<int32_t>(), m, k, idx->s_ids.as<int32_t>(),
                          idx->s_outd.as<float>(), st));
    HG_HIP(hipMemcpyAsync(out_ids, idx->s_ids.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(out_dist, idx->s_outd.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

// ---- dense distances: every query against every row (batch-cosine-distances over a query batch) ----
static int dense_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, float *d_out, hipStream_t st) {
    const int tm = tile_mode();
    if (tile_path_ok(idx) && tm != 0 && (tm == 1 || nq >= 16)) {
        HG_TRY(pad_queries(idx, d_Q, idx->dim, nq, st));
        const int tq = tile_tq(idx->dim);
        TileArgs t;
        memset(&t, 0, sizeof(t));
        t.rows = idx->d_base;
        t.row_norms = idx->d_norms;
        t.ld = idx->ld;
        t.dim = idx->dim;
        t.metric = idx->metric;
        t.Qp = idx->s_qp.as<float>();
        t.q_norms = idx->s_qn.as<float>();
        t.nrows_all = idx->n;
        t.nq = nq;
        t.out_stride = idx->n;
        int64_t groups = (nq + tq - 1) / tq;
        int64_t tiles = (idx->n + kTileRows - 1) / kTileRows;
        int64_t want = std::max<int64_t>(1, std::min<int64_t>(tiles, (2048 + groups - 1) / groups));
        int64_t cr = ((tiles + want - 1) / want) * kTileRows;
        t.chunk_rows = static_cast<int32_t>(cr);
        t.nchunks = static_cast<int32_t>(std::max<int64_t>(1, (tiles + cr / kTileRows / 2) / (cr / kTileRows)));
        t.out = d_out;
        return launch_tile(t, groups, idx->dim, st);
    }
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = idx->d_base;
    a.row_norms = idx->d_norms;
    a.ld = idx->ld;
    a.nrows_all = idx->n;
    a.Q = d_Q;
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.mode = MODE_STORE;
    a.role = ROLE_EXACT;
    a.k = 1;
    a.npairs = nq;
    a.nchunks = plan_chunks(idx->nch, idx->n, idx->n, nq, &a.chunk_rows);
    a.out = d_out;
    a.out_stride = idx->n;
    return launch_scan(idx->nch, a, st);
}

static int check_dense_args(const hnswgpu_index *idx, const void *Q, int32_t nq, const void *out) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(nq >= 0, HNSWGPU_EINVAL, "nq < 0");
    HG_REQUIRE(nq == 0 || idx->n == 0 || (Q && out), HNSWGPU_EINVAL, "null argument");
    return 0;
}

int hnswgpu_dense_distances_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, float *d_out, void *stream) {
    HG_TRY(check_dense_args(idx, d_Q, nq, d_out));
    if (nq == 0 || idx->n == 0) return 0;
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(dense_enqueue(idx, d_Q, nq, d_out, st));
    return end_call(idx, st);
}

int hnswgpu_dense_distances(hnswgpu_index *idx, const float *Q, int32_t nq, float *out) {
    HG_TRY(check_dense_args(idx, Q, nq, out));
    if (nq == 0 || idx->n == 0) return 0;
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, Q, nq, st));
    const size_t bytes = sizeof(float) * static_cast<size_t>(nq) * idx->n;
    HG_TRY(idx->s_tile.ensure(bytes));
    HG_TRY(dense_enqueue(idx, idx->s_q.as<float>(), nq, idx->s_tile.as<float>(), st));
    HG_HIP(hipMemcpyAsync(out, idx->s_tile.p, bytes, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

// Measurement / test entry: the lower bounds the HNSW traversal's rejection test (kernels.hpp: quantize_rows_kernel)
// computes for query q against rows ids[0..m), by the traversal's own device functions.  NaN = "no bound".
int hnswgpu_rejection_bounds(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out) {
    return hnswgpu_distance_bounds(idx, q, ids, m, out, nullptr);
}

int hnswgpu_distance_bounds(hnswgpu_index *idx, const float *q, const int32_t *ids, int32_t m, float *out,
                            float *out_ub) {
    HG_REQUIRE(idx && q && ids && out && m >= 1, HNSWGPU_EINVAL, "null argument");
    for (int32_t i = 0; i < m; i++) HG_REQUIRE(ids[i] >= 0 && ids[i] < idx->n, HNSWGPU_EINVAL, "id out of range");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(ensure_qrows(idx, st));
    HG_REQUIRE(idx->d_qrows, HNSWGPU_EINVAL,
               "this handle has no int8 rows (hnswgpu_set_rejection_test: mode 0, or mode 1 with dim < 128)");
    HG_TRY(upload_queries(idx, q, 1, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * m));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * 2 * static_cast<size_t>(m)));
    HG_HIP(hipMemcpyAsync(idx->s_ids.p, ids, sizeof(int32_t) * m, hipMemcpyHostToDevice, st));
    float *d_ub = out_ub ? idx->s_outd.as<float>() + m : nullptr;
#define CALL(N, R, L)                                                                                                \
    hipLaunchKernelGGL((code_bound_kernel<N>), dim3((m + 7) / 8), dim3(kWave), 0, st, idx->s_q.as<float>(), idx->dim, \
                       idx->metric, idx->d_qrows, idx->d_qmeta, idx->s_ids.as<int32_t>(), m, idx->s_outd.as<float>(), d_ub)
    HG_DISPATCH(idx->nch, false, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpyAsync(out, idx->s_outd.p, sizeof(float) * m, hipMemcpyDeviceToHost, st));
    if (out_ub) HG_HIP(hipMemcpyAsync(out_ub, d_ub, sizeof(float) * m, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

// Diagnostic / test entry: the half-precision bounds (stream_kernels.hpp, step 1b) of `m` LIST rows -- positions in list
// order, as hnswgpu_get_ivf's list_ids numbers them -- against one query, by the kernel the searches run.
int hnswgpu_ivf_half_bounds(hnswgpu_index *idx, const float *q, const int32_t *list_rows, int32_t m, float *out_lb,
                            float *out_ub) {
    HG_REQUIRE(idx && q && list_rows && out_lb && out_ub && m >= 1, HNSWGPU_EINVAL, "null argument");
    for (int32_t i = 0; i < m; i++) HG_REQUIRE(list_rows[i] >= 0 && list_rows[i] < idx->n, HNSWGPU_EINVAL, "row out of range");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    HG_REQUIRE(idx->d_lhalf, HNSWGPU_EINVAL, "this handle has no half-precision list rows (no lists, no int8 rows, or HNSWGPU_IVF_HALF=0)");
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, q, 1, st));
    HG_TRY(idx->s_tile.ensure(sizeof(uint4) * static_cast<size_t>(m) + sizeof(uint32_t)));
    std::vector<uint4> ent(m);
    for (int32_t i = 0; i < m; i++)  // no bounds yet: NaN on both sides (the kernel keeps the tighter of old and new)
        ent[i] = make_uint4(static_cast<uint32_t>(i), static_cast<uint32_t>(list_rows[i]), 0x7fc00000u, 0x7fc00000u);
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(idx->s_tile.as<uint4>() + m);
    const uint32_t cnt = static_cast<uint32_t>(m);
    HG_HIP(hipMemcpyAsync(idx->s_tile.p, ent.data(), sizeof(uint4) * m, hipMemcpyHostToDevice, st));
    HG_HIP(hipMemcpyAsync(d_cnt, &cnt, sizeof(cnt), hipMemcpyHostToDevice, st));
    MidArgs a;
    memset(&a, 0, sizeof(a));
    a.surv = idx->s_tile.as<uint4>();
    a.surv_cnt = d_cnt;
    a.cap = m;
    a.nq = 1;
    a.slices = 1;
    a.half = idx->d_lhalf;
    a.hmeta = idx->d_lhmeta;
    a.ld = idx->ld;
    a.Q = idx->s_q.as<float>();
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    HG_TRY(launch_mid(a, idx->nch, st));
    HG_HIP(hipMemcpyAsync(ent.data(), idx->s_tile.p, sizeof(uint4) * m, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    for (int32_t i = 0; i < m; i++) {
        memcpy(out_lb + i, &ent[i].z, sizeof(float));
        memcpy(out_ub + i, &ent[i].w, sizeof(float));
    }
    return 0;
}

// Diagnostic / test entry: the matrix-core half-precision bounds (stream_kernels.hpp, step 1a: ivf_home_kernel) of the
// LIST rows [row_begin, row_end) -- positions in list order -- against `nq` queries, as if that range were the list all of
// them are nearest to; out_lb / out_ub [nq][row_end - row_begin].
int hnswgpu_ivf_home_bounds(hnswgpu_index *idx, const float *Q, int32_t nq, int64_t row_begin, int64_t row_end, float *out_lb,
                            float *out_ub) {
    HG_REQUIRE(idx && Q && out_lb && out_ub && nq >= 1, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    // (under the lock: hnswgpu_hnsw_add grows idx->n)
    HG_REQUIRE(row_begin >= 0 && row_begin < row_end && row_end <= idx->n && row_end - row_begin < (1 << 24), HNSWGPU_EINVAL, "row range");
    HG_HIP(hipSetDevice(idx->device));
    try {  // (the work lists and the staging below are host vectors: no exception crosses the C boundary)
    HG_REQUIRE(idx->d_lhalf, HNSWGPU_EINVAL, "this handle has no half-precision list rows (no lists, no int8 rows, or HNSWGPU_IVF_HALF=0)");
    HG_REQUIRE(idx->ld % 128 == 0, HNSWGPU_EINVAL, "the home-list pass serves rows of whole 128-element steps");
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(upload_queries(idx, Q, nq, st));
    const int64_t len = row_end - row_begin, hstride = (len + 15) / 16 * 16;
    const int gq = home_group(idx->nch);
    std::vector<HomeDesc> items;
    const int64_t chunk = std::max<int64_t>(64, tune(HNSWGPU_TUNE_HOME_CHUNK, 256) / 64 * 64);
    for (int64_t r0 = 0; r0 < len; r0 += chunk)
        for (int32_t q0 = 0; q0 < nq; q0 += gq) {
            HomeDesc d;
            d.rb0 = row_begin;
            d.r0_off = static_cast<int32_t>(r0);
            d.r1_off = static_cast<int32_t>(std::min<int64_t>(len, r0 + chunk));
            d.q0 = q0;
            d.cnt = std::min<int32_t>(gq, nq - q0);
            d.list = 0;
            d.pad = 0;
            items.push_back(d);
        }
    std::vector<int32_t> order(nq);
    for (int32_t i = 0; i < nq; i++) order[i] = i;
    const int32_t nit = static_cast<int32_t>(items.size());
    HG_TRY(idx->s_home.ensure(sizeof(HomeDesc) * items.size() + 64 + sizeof(int32_t) * nq));
    HG_TRY(idx->s_dh.ensure(sizeof(float2) * static_cast<size_t>(nq) * hstride));
    HomeDesc *d_items = idx->s_home.as<HomeDesc>();
    int32_t *d_nit = reinterpret_cast<int32_t *>(d_items + items.size());
    int32_t *d_order = d_nit + 16;
    HG_HIP(hipMemcpyAsync(d_items, items.data(), sizeof(HomeDesc) * items.size(), hipMemcpyHostToDevice, st));
    HG_HIP(hipMemcpyAsync(d_nit, &nit, sizeof(nit), hipMemcpyHostToDevice, st));
    HG_HIP(hipMemcpyAsync(d_order, order.data(), sizeof(int32_t) * nq, hipMemcpyHostToDevice, st));
    HomeArgs a;
    memset(&a, 0, sizeof(a));
    a.items = d_items;
    a.nitems = d_nit;
    a.qorder = d_order;
    a.half = idx->d_lhalf;
    a.hmeta = idx->d_lhmeta;
    a.ld = idx->ld;
    a.Q = idx->s_q.as<float>();
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.dh = idx->s_dh.as<float2>();
    a.hstride = hstride;
    HG_TRY(launch_home(a, nit, idx->nch, st));
    std::vector<float2> host(static_cast<size_t>(nq) * hstride);
    HG_HIP(hipMemcpyAsync(host.data(), a.dh, sizeof(float2) * host.size(), hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    for (int32_t i = 0; i < nq; i++)
        for (int64_t r = 0; r < len; r++) {
            out_lb[i * len + r] = host[i * hstride + r].x;
            out_ub[i * len + r] = host[i * hstride + r].y;
        }
    return 0;
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed in hnswgpu_ivf_home_bounds");
        return HNSWGPU_ENOMEM;
    }
}

int hnswgpu_set_rejection_test(hnswgpu_index *idx, int32_t mode) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(mode >= 0 && mode <= 2, HNSWGPU_EINVAL, "mode must be 0 (off), 1 (large batches) or 2 (always)");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    idx->rejection_mode = mode;
    // (mode 1 measures again at the next IVF search -- but not a SHARD of a larger index: its verdict is the whole index's
    // (hnswgpu_ivf_set_stream_state); measuring alone it could take another kernel path than its siblings, and the sharded
    // answer would no longer be the unsharded one bit for bit)
    if (idx->d_glistoff == nullptr) idx->ivf_calibrated = idx->ivf_stream_off = false;
    if (mode != 0 && (idx->has_graph || idx->nlist > 0)) {  // int8 rows now for what exists, else with the graph / lists
        hipStream_t st = idx->stream;
        HG_TRY(begin_call(idx, st));
        if (idx->has_graph) HG_TRY(ensure_qrows(idx, st));
        HG_TRY(ensure_list_codes(idx, st));
        HG_TRY(ensure_list_half(idx, st));
        HG_TRY(end_call(idx, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

int hnswgpu_hnsw_rejection_state(hnswgpu_index *idx, int32_t *state, int32_t *off, double *frac) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (state) *state = idx->hnsw_cal_state;
    if (off) *off = idx->hnsw_rej_off ? 1 : 0;
    if (frac) *frac = idx->hnsw_cal_frac;
    return 0;
}

int hnswgpu_launch_count(int32_t which, int64_t *out) {
    HG_REQUIRE(which >= 0 && which < HNSWGPU_COUNT_N && out, HNSWGPU_EINVAL, "no such launch counter: %d", which);
    *out = hg::g_launch_count[which].load(std::memory_order_relaxed);
    return 0;
}

int hnswgpu_set_tuning(int32_t key, int64_t value) {
    HG_REQUIRE(key >= 0 && key < HNSWGPU_TUNE_COUNT, HNSWGPU_EINVAL, "no such tuning key: %d", key);
    hg::g_tune[key].store(value, std::memory_order_relaxed);
    return 0;
}

int hnswgpu_get_tuning(int32_t key, int64_t *value, int32_t *is_set) {
    HG_REQUIRE(key >= 0 && key < HNSWGPU_TUNE_COUNT, HNSWGPU_EINVAL, "no such tuning key: %d", key);
    const int64_t v = hg::g_tune[key].load(std::memory_order_relaxed);
    if (value) *value = v;
    if (is_set) *is_set = v != hg::kTuneUnset;
    return 0;
}

#ifdef HG_DIAG
// -DHG_DIAG builds only: which == 0 the bounds pass (stream_kernels.hpp StreamArgs::dbg), 1 the tile scan (TileArgs::dbg)
int hnswgpu_debug_set_ablation(int32_t which, int32_t value) {
    (which == 0 ? hg::g_stream_dbg : hg::g_tile_dbg) = value;
    return 0;
}
#endif

int hnswgpu_debug_set_tile_stamps(void *device_buffer) {
    hg::g_tile_dbg_buf = static_cast<unsigned long long *>(device_buffer);
    return 0;
}

int hnswgpu_set_profiling(hnswgpu_index *idx, int32_t on) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    if (on && !idx->d_rej_stats) {
        HG_HIP(hipSetDevice(idx->device));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_rej_stats), 2 * sizeof(unsigned long long)));
        HG_HIP(hipMemset(idx->d_rej_stats, 0, 2 * sizeof(unsigned long long)));
    }
    idx->prof = on != 0;
    return 0;
}

int hnswgpu_get_profile(hnswgpu_index *idx, int32_t which, double *total_ms, int64_t *launches, int32_t reset) {
    HG_REQUIRE(idx && which >= 0 && which < PROF_N, HNSWGPU_EINVAL, "bad argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    for (auto &pr : idx->prof_ev[which]) {
        HG_HIP(hipEventSynchronize(pr.second));
        float ms = 0.0f;
        HG_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        idx->prof_ms[which] += ms;
        idx->prof_cnt[which] += 1;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    idx->prof_ev[which].clear();
    if (total_ms) *total_ms = idx->prof_ms[which];
    if (launches) *launches = idx->prof_cnt[which];
    if (reset) {
        idx->prof_ms[which] = 0;
        idx->prof_cnt[which] = 0;
    }
    return 0;
}

int hnswgpu_get_rejection_stats(hnswgpu_index *idx, int64_t *f32_rows, int64_t *neighbours, int32_t reset) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    unsigned long long v[2] = {0, 0};
    if (idx->d_rej_stats) {
        hipStream_t st = idx->stream;
        HG_TRY(begin_call(idx, st));  // every earlier call on this handle, whatever its stream, is ordered before st
        HG_HIP(hipMemcpyAsync(v, idx->d_rej_stats, sizeof(v), hipMemcpyDeviceToHost, st));
        if (reset) HG_HIP(hipMemsetAsync(idx->d_rej_stats, 0, sizeof(v), st));
        HG_TRY(end_call(idx, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    if (f32_rows) *f32_rows = static_cast<int64_t>(v[0]);
    if (neighbours) *neighbours = static_cast<int64_t>(v[1]);
    return 0;
}

}  // extern "C"
