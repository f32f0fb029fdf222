// engine.hpp -- host-side state of one index handle + helpers shared by the translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <string>
#include <utility>
#include <vector>

#include "tuning.hpp"

#include "../../include/hnswgpu.h"
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

void set_error(const char *fmt, ...);
extern std::atomic<int64_t> g_launch_count[HNSWGPU_COUNT_N];  // hnswgpu_launch_count
inline void count_launch(int which) { g_launch_count[which].fetch_add(1, std::memory_order_relaxed); }

#define HG_HIP(expr)                                                                             \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            hg::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return _e == hipErrorOutOfMemory ? HNSWGPU_ENOMEM : HNSWGPU_EHIP;                    \
        }                                                                                        \
    } while (0)

#define HG_TRY(expr)             \
    do {                         \
        int _rc = (expr);        \
        if (_rc != 0) return _rc; \
    } while (0)

#define HG_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            hg::set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) {
            HG_HIP(hipFree(p));
            p = nullptr;
            cap = 0;
        }
        size_t want = bytes + bytes / 4 + 256;
        HG_HIP(hipMalloc(&p, want));
        cap = want;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T>
    T *as() const {
        return static_cast<T *>(p);
    }
};

enum { PROF_IVF_SCAN = 0, PROF_HNSW = 1, PROF_ASSIGN = 2, PROF_N = 3 };

}  // namespace hg

struct hnswgpu_index {
    int device = 0;
    int metric = 0;
    int64_t n = 0;
    int dim = 0;
    int64_t ld = 0;  // row stride in floats (multiple of 4)
    int nch = 0;     // 256-float chunks per row (template parameter NCH)
    float *d_base = nullptr;
    float *d_norms = nullptr;
    // int8 copy of the rows + per-row (scale, error bound) for the HNSW traversal's rejection test (kernels.hpp:
    // quantize_rows_kernel); made when a graph is installed or built, null while rejection_mode is 0
    uint32_t *d_qrows = nullptr;
    float4 *d_qmeta = nullptr;
    // int8 copy of the IVF list rows (list order) in the MFMA tile layout + per-row bound terms, for the bounds pass of
    // the list scan (stream_kernels.hpp); made with the lists, null while rejection_mode is 0
    uint32_t *d_lctile = nullptr;
    float4 *d_lcmeta = nullptr;
    uint2 *d_lhalf = nullptr;    // list rows in fp16, row-major (four halves per element), + per-row (scale, E, 0, 1/|v|)
    float4 *d_lhmeta = nullptr;
    unsigned long long *d_rej_stats = nullptr;  // [2], counted by the traversal while profiling is on
    int rejection_mode = 1;  // 0 = off, 1 = batches that fill the chip (launch_hnsw_idx), 2 = every launch
    // HNSW, mode 1: the first large launch on a graph counts what the int8 test decides (f32 rows fetched / neighbours
    // evaluated, launch_hnsw_idx); where it rejects too little to pay for its own bytes -- rows whose neighbours all lie
    // within the bounds' width of the list's worst -- the later large launches evaluate every neighbour in f32.  0 = not
    // measured, 1 = measured (the counters are on their way to hnsw_cal_host), 2 = decided; reset with the graph.
    int hnsw_cal_state = 0;
    bool hnsw_rej_off = false;
    double hnsw_cal_frac = 0.0;  // f32 rows / neighbours of the measured launch
    unsigned long long *d_hnsw_cal = nullptr, *hnsw_cal_host = nullptr;  // [2] device counters, pinned host copy
    hipEvent_t ev_hnsw_cal = nullptr;
    // IVF, mode 1: the first search measures what the int8 bounds separate on THIS data (ivf_calibrate) and switches the
    // survivor stream off for the handle when they leave more than a quarter of the candidates
    bool ivf_calibrated = false, ivf_calibrating = false, ivf_stream_off = false;
    int cus = 256;
    hipStream_t stream = nullptr;
    std::mutex mu;
    // Combining of concurrent synchronous searches (hnswgpu_hnsw_search from many host threads, the reference's
    // parallel-search-futures pattern): callers queue their request; one of them -- the collector -- takes everything
    // that is queued with the same (k, ef) as ONE batch, launches it and hands the results back.  Twenty threads issuing
    // single queries then share a launch instead of queueing twenty of them one after the other.
    struct SearchReq {
        const float *Q;
        int32_t nq, k, ef;  // ef: layer-0 breadth of an HNSW request, nprobe of an IVF request
        int32_t *out_ids;
        float *out_dist;
        int64_t *stats;
        int rc = 0;
        std::string err;
        // 0 = queued, 1 = served (rc / err / outputs are final), 2 = "you collect the next batch".  Every caller sleeps on
        // ITS OWN word (futex): serving a batch of 200 wakes 200 threads one by one, none of which then fights the
        // other 199 for a shared mutex (the condition-variable broadcast of round 1 collapsed at 200 callers).
        std::atomic<uint32_t> state{0};
    };
    struct Combiner {
        std::mutex mu;
        std::vector<SearchReq *> pending;
        std::atomic<int> npending{0};  // pending.size(), readable without the lock (a lingering collector polls it)
        bool collector = false;        // a thread is forming the next batch (at most one at a time)
        int inflight = 0;              // batches launched and not yet answered
        int max_inflight = 1;          // batches that may overlap on the device (own stream + staging each)
        SearchReq *slot_waiter = nullptr;  // the collector, parked until a batch in flight completes
        // (atomics: a lingering collector reads both while a finishing batch of the other slot writes them)
        std::atomic<int> last{0};              // requests in the batch launched last
        std::atomic<double> last_run_us{0.0};  // how long that batch took from launch to results (its callers return after that)
    };
    Combiner cmb_hnsw, cmb_ivf;
    // One batch in flight of the small synchronous searches: its own stream, one block of MAPPED pinned host memory
    // that the kernels read the queries from and write the results to (no copy calls at all), and a flag in that
    // block the last workgroup sets and the host thread spins on (no interrupt, no hipStreamSynchronize).
    struct Slot {
        std::mutex mu;
        hipStream_t st = nullptr;
        void *h = nullptr;      // host address of the block
        void *d = nullptr;      // device address of the same bytes
        size_t cap = 0;
        int32_t *d_again = nullptr;    // [1 + kZcMaxQueries]: queries that ran out of ghost slots (see hnsw.hip)
        uint32_t *d_done = nullptr;    // workgroups that have finished the current launch
        uint32_t seq = 0;              // value the flag takes when the current launch has finished
    };
    Slot slots[2];
    // a small synchronous IVF call in flight (ivf.hip: ivf_search_batch_slot): the flag word its last kernel sets, taken by the
    // finish kernel's launch when the search goes through it (zc_taken), by a one-thread launch behind the search otherwise
    uint32_t *zc_flag = nullptr;
    uint32_t zc_val = 0;
    bool zc_taken = false;
    void *h_pin = nullptr;    // pinned host staging of a large combined batch (queries in, results out)
    size_t h_pin_cap = 0;
    // cross-stream ordering of the shared scratch buffers: the last call's completion event
    hipEvent_t ev_last = nullptr;
    hipStream_t ev_stream = nullptr;
    bool ev_valid = false;

    // HNSW graph (device + host mirror for export)
    bool has_graph = false;
    uint64_t graph_gen = 0;  // bumped whenever the graph is replaced: a search in flight on a slot stream notices

    int M = 0, M0 = 0, entry = -1, max_level = 0;
    int build_flags = 0;  // HNSWGPU_BUILD_* the graph was built with (hnswgpu_hnsw_add inserts the same way); 0 for an installed graph
    int64_t up_blocks = 0;
    int32_t *d_levels = nullptr, *d_l0 = nullptr, *d_upadj = nullptr;
    int64_t *d_upoff = nullptr;
    std::vector<int32_t> h_levels, h_l0, h_upadj;
    std::vector<int64_t> h_upoff;

    // IVF-FLAT (device: centroids + rows re-ordered so every list is contiguous)
    int nlist = 0;
    int64_t max_list_len = 0;
    int64_t min_list_len = 0;  // (of this handle's lists; a shard: of the lists it holds -- the wave-per-query routing tail needs k rows in every list)
    float *d_cent = nullptr, *d_cnorms = nullptr, *d_lrows = nullptr, *d_lnorms = nullptr;
    int64_t *d_listoff = nullptr;
    int32_t *d_listids = nullptr;
    // a SHARD of a larger IVF index (hnswgpu_set_ivf_shard): prefix sums of the list lengths of the WHOLE index, so
    // that every shard numbers its candidates as the unsharded search would; nullptr on an ordinary index
    int64_t *d_glistoff = nullptr;
    // rows of the WHOLE index the lists belong to (= n on an ordinary index; the sum of the global list lengths on a shard):
    // everything that chooses a kernel path or sizes a sample by the MEAN LIST LENGTH uses this, so that a shard decides
    // what the unsharded index decides (a shard holds n / ndev rows over the same nlist)
    int64_t ivf_n_global = 0;
    bool lrows_alias = false;  // lists are the base rows in place (list_ids = 0..n-1): d_lrows / d_lnorms alias d_base / d_norms
    std::vector<float> h_cent;
    std::vector<int64_t> h_listoff, h_glistlen;
    std::vector<int32_t> h_listids;

    // scratch (grown on demand, reused across calls; calls are serialised by `mu`)
    hg::DevBuf s_q, s_partial, s_ord, s_dist, s_pairs, s_ids, s_outd, s_probes, s_stats, s_misc, s_misc2, s_vis, s_qp, s_qn, s_tile, s_grp, s_done, s_pf, s_solo, s_bk, s_heavy, s_home, s_dh;
    // s_bk: the per-list pair counters of the IVF survivor stream -- zero between searches (the work-list kernel clears
    // them behind its last read); bk_dirty = a search was enqueued past the point that fills them but not past the
    // work-list kernel (an error in between): the next search clears them itself
    bool bk_dirty = false;
    uint32_t pf_seq = 0;  // launch number of the last prefetch-helper launch (24 bits, never 0)
    size_t s_done_n = 0;  // counters per half of s_done (scan tails | route tails)
    uint32_t vis_gen = 0;  // last generation number handed to an HBM visited slab

    // profiling
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[hg::PROF_N];
    double prof_ms[hg::PROF_N] = {0, 0, 0};
    int64_t prof_cnt[hg::PROF_N] = {0, 0, 0};
};

namespace hg {

inline int64_t ivf_mean_len(const hnswgpu_index *idx) {
    return std::max<int64_t>(1, (idx->ivf_n_global > 0 ? idx->ivf_n_global : idx->n) / std::max(idx->nlist, 1));
}

int pick_nch(int64_t ld);  // 0 if unsupported
// hipFuncSetAttribute is per device: true the first time a call site asks on the current device
bool attr_needed(bool (&done)[64]);

// launch wrappers (engine.hip / hnsw.hip)
// CALL(NCH, RB, L2) for the row-loader width `nch` (pick_nch) -- every kernel template's dispatch
#define HG_DISPATCH(nch, l2, CALL)                                        \
    do {                                                                  \
        if (l2) {                                                         \
            switch (nch) {                                                \
                case 1: CALL(1, 8, true); break;                          \
                case 2: CALL(2, 8, true); break;                          \
                case 3: CALL(3, 8, true); break;                          \
                case 4: CALL(4, 4, true); break;                          \
                case 6: CALL(6, 4, true); break;                          \
                case 8: CALL(8, 2, true); break;                          \
                case 12: CALL(12, 2, true); break;                        \
                default: set_error("unsupported row length"); return HNSWGPU_ELIMIT; \
            }                                                             \
        } else {                                                          \
            switch (nch) {                                                \
                case 1: CALL(1, 8, false); break;                         \
                case 2: CALL(2, 8, false); break;                         \
                case 3: CALL(3, 8, false); break;                         \
                case 4: CALL(4, 4, false); break;                         \
                case 6: CALL(6, 4, false); break;                         \
                case 8: CALL(8, 2, false); break;                         \
                case 12: CALL(12, 2, false); break;                       \
                default: set_error("unsupported row length"); return HNSWGPU_ELIMIT; \
            }                                                             \
        }                                                                 \
    } while (0)

// small HNSW launches, one query over several CUs (solo.hip)
bool solo_enabled(int ef);
int launch_hnsw_solo(hnswgpu_index *idx, HnswArgs a, hipStream_t st);
// large HNSW launches, one wave per query on a main list + admission buffer (wave.hip)
int launch_hnsw_wave(hnswgpu_index *idx, const HnswArgs &a, int grid, size_t lds, bool vg, hipStream_t st);
int launch_norms(int nch, const float *rows, int64_t ld, int64_t n, float *out, hipStream_t st);
int ensure_qrows(hnswgpu_index *idx, hipStream_t st);
// int8 codes + per-row bound terms of `n` rows into freshly allocated *crows / *cmeta (the caller owns them)
int quantize_rows(hnswgpu_index *idx, const float *rows, int64_t n, uint32_t **crows, float4 **cmeta, hipStream_t st);
int ensure_list_codes(hnswgpu_index *idx, hipStream_t st);
int ensure_list_half(hnswgpu_index *idx, hipStream_t st);
int launch_scan(int nch, const ScanArgs &a, hipStream_t st);
int launch_merge(const MergeArgs &a, hipStream_t st);
// Serve `me` through combiner `c`: queue it, lead one batch at a time while it is not done.  `take(first, r, total)`
// says whether request r may join a batch that starts with `first` and already holds `total` queries; `run` launches
// a batch and returns its error code.
constexpr int kZcMaxQueries = 256;  // largest combined batch served through a Slot (one workgroup per query and CU)
int slot_prepare(hnswgpu_index::Slot &s, size_t bytes);  // stream, counters, mapped block of at least `bytes`
// Spin on the completion flag of a slot launch (value `seq`); falls back to hipStreamSynchronize if it does not show.
int slot_wait(hnswgpu_index::Slot &s, volatile uint32_t *flag, uint32_t seq);
int combine_search(hnswgpu_index::Combiner &c, hnswgpu_index::SearchReq &me,
                   const std::function<bool(const hnswgpu_index::SearchReq *, const hnswgpu_index::SearchReq *, int64_t)> &take,
                   const std::function<int(const std::vector<hnswgpu_index::SearchReq *> &, int32_t)> &run);
// pinned staging block of a combined batch (grown on demand)
int ensure_pinned(hnswgpu_index *idx, size_t bytes);
int scan_dense_topk(hnswgpu_index *idx, ScanArgs a, int32_t nq, int64_t nrows, hipStream_t st);
// small IVF batches: routing in one launch, list scan with the merge / decode folded into its last workgroups
// what the routing step prepares for the survivor stream of the list scan (stream_kernels.hpp), or null
struct RouteStream {
    uint32_t *qcodes;
    QueryScal *qscal;
    uint32_t *tau, *surv_cnt;
    int32_t k;
    uint32_t *bk_cnt;  // per-list buckets of (query, list) pairs, or null (ungrouped bounds pass)
    uint2 *bk_mem;
    int32_t bk_cap;
    int32_t home;  // the home-list pass follows (stream_kernels.hpp, step 1a): a query whose nearest list holds k rows gets its threshold there
    const struct WorklistArgs *wl;  // optional: the bounds pass's work list by extra workgroups of the tail's launch (filed: set by the launch)
};
int launch_ivf_route(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t nprobe, Pair *pairs, int32_t *probes,
                     int32_t *qcnt, hipStream_t st, const RouteStream *rs = nullptr, bool two_launches = false);
// the survivor stream of the IVF list scan (stream_kernels.hpp)
struct StreamArgs;
struct FinishArgs;
struct PrepArgs;
int launch_query_prep(const PrepArgs &a, int nch, hipStream_t st);
// narrow: the epilogue for few queries per probed list (lane = row); otherwise lane = query
struct MidArgs;
struct HeavyArgs;
struct HomeArgs;
int launch_mid(const MidArgs &a, int nch, hipStream_t st);
int launch_heavy(const HeavyArgs &a, hipStream_t st);
int launch_home(const HomeArgs &a, int64_t blocks, int nch, hipStream_t st);
int launch_stream_bounds(const StreamArgs &a, int64_t blocks, int nch, bool narrow, hipStream_t st, int qblocks = 1);
int launch_finish(const FinishArgs &a, int nch, hipStream_t st);
// zero-initialised per-query counters of the fused tails (s_done: [n] scan / finish tails | [n] route tails)
int ensure_counters(hnswgpu_index *idx, size_t n, hipStream_t st);
int scan_fused(hnswgpu_index *idx, ScanArgs a, int32_t nq, int32_t pairs_per_query, int64_t max_rows, int64_t mean_rows,
               hipStream_t st, int prof_slot);
extern unsigned long long *g_tile_dbg_buf;
#ifdef HG_DIAG
extern int g_stream_dbg, g_tile_dbg;
#endif
int launch_gather(int nch, GatherArgs a, int32_t nq, hipStream_t st);
int scan_rows_per_iter(int nch);  // kNWave * RB

// pick chunking for a scan: returns nchunks, sets chunk_rows
int plan_chunks(int nch, int64_t max_rows, int64_t mean_rows, int64_t npairs, int32_t *chunk_rows,
                bool list_pairs = false);

// scan + merge: per-query ascending top-k of (ord, dist) into s_ord / s_dist ([nq][k])
int scan_topk(hnswgpu_index *idx, ScanArgs a, int32_t nq, int32_t pairs_per_query, int64_t max_rows,
              hipStream_t st, int prof_slot, int64_t mean_rows = 0);

void prof_begin(hnswgpu_index *idx, int slot, hipStream_t st, hipEvent_t *e0);
void prof_end(hnswgpu_index *idx, int slot, hipStream_t st, hipEvent_t e0);

int upload_queries(hnswgpu_index *idx, const float *Q, int32_t nq, hipStream_t st);

// --- tiled (MFMA) scan path -------------------------------------------------------------------------
bool tile_path_ok(const hnswgpu_index *idx);  // metric != L2 and dim fits the LDS-resident query group
int tile_mode();                               // HNSWGPU_TUNE_TILE: -1 auto, 0 never, 1 whenever possible
int launch_tile(const TileArgs &a, int64_t ngroups_bound, int dim, hipStream_t st);
int launch_select(const SelectArgs &a, hipStream_t st);
// queries (nq x dim, row stride qld) -> s_qp (nq x ld, zero padded) + s_qn (device-order norms)
int pad_queries(hnswgpu_index *idx, const float *d_Q, int64_t qld, int32_t nq, hipStream_t st);
// every query against rows [0, nrows): per-query ascending top-k into s_ord / s_dist
int tile_topk_all(hnswgpu_index *idx, const float *Qp, const float *q_norms, int32_t nq, const float *rows,
                  const float *row_norms, int64_t nrows, int32_t k, hipStream_t st, int prof_slot, bool gemv_order = false);

// Every entry point brackets its device work with these: a call on stream B waits for the previous
// call's work on stream A before it may reuse the index's scratch buffers.
int begin_call(hnswgpu_index *idx, hipStream_t st);
int end_call(hnswgpu_index *idx, hipStream_t st);
// Before device memory that searches read is freed (graph, lists, the handle itself): wait, under idx->mu, for the
// main stream AND for the two slot streams -- the small synchronous searches launch on those without begin_call /
// end_call and release idx->mu before their kernel has finished.
int quiesce(hnswgpu_index *idx, hipStream_t st);

}  // namespace hg
