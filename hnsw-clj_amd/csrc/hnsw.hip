// hnsw.hip -- HNSW graph import/export, batched GPU search and batched GPU-assisted build.
// Reference: src/hnsw/ultra_fast.clj (file:line cited per function).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "build_kernels.hpp"
#include "engine.hpp"
#include "javarandom.hpp"

namespace hg {

static size_t hnsw_lds_bytes(int cap, int nwords, int nw) {
    // kernels.hpp: hnsw_search_kernel's layout -- ONE list of 8-byte entries + a 16-bit merged position per slot
    return sizeof(uint2) * cap + sizeof(int32_t) * 3 * kMaxDeg + sizeof(int32_t) * 16 + sizeof(int32_t) * nw * kWave +
           sizeof(uint2) * kPfRing + sizeof(uint32_t) * nwords + sizeof(uint16_t) * cap + 16;
}

constexpr size_t kMaxLds = 160 * 1024;

static size_t wave_lds_bytes(int cap, int nwords) {
    // wave_kernels.hpp: hnsw_wave_kernel's layout -- the main list, the buffer's image, the expansion's candidates, the visited set
    return sizeof(uint2) * cap + sizeof(uint2) * kWave + 2 * sizeof(int32_t) * kMaxDeg + sizeof(uint32_t) * nwords + 16;
}

// tuning override (HNSWGPU_TUNE_HNSW_NW = 1 | 2 | 4); 0 = choose by batch size
static int hnsw_nw() {
    const int64_t v = tune(HNSWGPU_TUNE_HNSW_NW, 0);
    return (v == 1 || v == 2 || v == 4) ? static_cast<int>(v) : 0;
}

// Visited-set placement: an LDS bitset while it leaves room for several workgroups per CU, otherwise
// generation stamps in HBM (4 B per row per resident workgroup; 288 GB makes that cheap).
constexpr int64_t kLdsVisitedMaxRows = 262144;  // 32 KiB bitset
static bool force_vg() { return tune(HNSWGPU_TUNE_VIS_GLOBAL, 0) != 0; }  // testing override: the HBM stamps at every size

// helper workgroups per query that prefetch neighbour rows into the query's XCD L2 (kernels.hpp, HnswArgs::pf_mail);
// HNSWGPU_TUNE_PREFETCH = <G> overrides (0 = off)
static int pf_groups() {
    const int64_t v = tune(HNSWGPU_TUNE_PREFETCH, 4);
    return v < 0 ? 0 : (v > 16 ? 16 : static_cast<int>(v));
}
constexpr int kPfMaxQueries = 128;  // up to half a CU count of queries: every traversal and at least one helper per query get a CU

int launch_hnsw_idx(hnswgpu_index *idx, HnswArgs a, hipStream_t st) {
    a.dbg = g_tile_dbg_buf;  // null outside diagnostic sessions
    a.rej_stats = idx->prof ? idx->d_rej_stats : nullptr;
    if (a.nq <= 0) return 0;
    const int nch = idx->nch;
    // The rejection test on int8 rows (kernels.hpp: quantize_rows_kernel) turns one memory round trip per hop into two
    // shorter ones: 2.2x the throughput once the chip is bandwidth-bound, ~5 % slower while a launch is latency-bound.
    // 31k x 768, ef 100, ms per launch without / with: 256 queries 0.537 / 0.571, 512: 0.607 / 0.590, 768: 0.817 /
    // 0.611, 1024: 1.05 / 0.74, 10000: 9.9 / 4.4 (tools/hnsw_batch_sweep.py) -- on from two queries per CU.
    if (!(idx->rejection_mode == 2 || (idx->rejection_mode == 1 && a.nq >= 2 * idx->cus && idx->dim >= 128))) a.qrows = nullptr;
    // Mode 1 also MEASURES (once per graph, on its first large launch): the int8 stage pays while it keeps enough f32 rows
    // from being fetched -- 31k x 768 clustered, ef 640: half of them; 1.25M x 1536 clustered, ef 256: a tenth, and the
    // launch then requests 1.15x the bytes of the plain traversal (profiles/r04_config5_hnsw_shard.txt).  Results never
    // depend on it.  Nothing here blocks: the counters travel to pinned host memory behind an event a later launch looks at.
    bool calibrate = false;
    if (a.qrows && idx->rejection_mode == 1 && !a.q_rows && !a.q_index && tune(HNSWGPU_TUNE_HNSW_CALIBRATE, 1) != 0) {
        if (idx->hnsw_cal_state == 1 && idx->ev_hnsw_cal && hipEventQuery(idx->ev_hnsw_cal) == hipSuccess) {
            const double f32_rows = static_cast<double>(idx->hnsw_cal_host[0]), nb = static_cast<double>(idx->hnsw_cal_host[1]);
            idx->hnsw_cal_frac = nb > 0 ? f32_rows / nb : 0.0;
            const double keep = static_cast<double>(tune(HNSWGPU_TUNE_HNSW_CALIBRATE_PCT, 65)) / 100.0;
            idx->hnsw_rej_off = nb > 0 && idx->hnsw_cal_frac > keep;  // the test leaves more than that of the rows to fetch
            idx->hnsw_cal_state = 2;
        }
        if (idx->hnsw_cal_state == 2 && idx->hnsw_rej_off) a.qrows = nullptr;
        calibrate = idx->hnsw_cal_state == 0 && !a.rej_stats;
    }
    const bool vg = force_vg() || a.n > kLdsVisitedMaxRows;
    const bool pf = pf_groups() > 0 && !vg && !a.q_rows && !a.q_index && a.nq <= kPfMaxQueries && hnsw_nw() == 0 &&
                    a.n < (1LL << 31) && a.M0 <= kMaxDeg;
    // waves per query.  Measured on 31k x 768, ef 128 (tools/tune_hnsw.py): a query takes 1.1 / 1.4 / 2.0 ms
    // with 4 / 2 / 1 waves, and a CU holds 3 / 6 / 12 such workgroups (VGPR-limited), so once a batch
    // exceeds one residency round fewer waves per query win: 10,000 queries run at 601k / 776k / 802k QPS.
    int nw = hnsw_nw() > 0 ? hnsw_nw() : (a.nq > 1536 ? 1 : (a.nq > 768 ? 2 : 4));
    // with the rejection test (fewer f32 rows in flight, below) a CU holds 4 / 8 / 20 such workgroups up to dim 768:
    // 1,024 queries 0.60 / 0.68 / 0.96 ms with 4 / 2 / 1 waves, 2,048: 1.15 / 0.93 / 1.06, 3,072: 1.68 / 1.51 / 1.36
    if (hnsw_nw() == 0 && a.qrows && nch <= 3) nw = a.nq > 2048 ? 1 : (a.nq > 1024 ? 2 : 4);
    // a long candidate list (large ef) bounds the queries a CU holds by its LDS, not by registers: fewer than ~13 waves per
    // CU cannot hide the hop's dependent round trips, so the queries that fit get more waves each (the int8 and f32 steps
    // of a hop then run side by side).  31k x 768 clustered, 10,000 queries, QPS with 1 / 2 / 4 waves: ef 400 729k / 784k /
    // 513k, ef 800 250k / 343k / 258k, ef 1600 60k / 89k / 116k, ef 3200 10.8k / 17.7k / 25.6k (tools/large_ef_nw.py).
    // Launches that fill the chip with one wave per query, and every launch with a long list, take the kernel whose list is a
    // main list + a register-resident admission buffer (wave_kernels.hpp: no positional merge per expansion, the next
    // candidate out of register windows; 8 bytes of LDS per list slot).  31k x 768 clustered, QPS single-workgroup kernel /
    // this one (tools/wave_sweep.py): 10,000 queries ef 50 4.55M / 4.80M, ef 640 516k / 778k, ef 1600 151k / 263k, ef 3200 51k /
    // 91k; 2,048 queries ef 640 467k / 554k; 256 queries ef 50 459k / 418k (four waves per query are the better shape for a
    // short list on a chip that is not full), ef 640 82k / 92k, ef 3200 13.9k / 22.7k.  HNSWGPU_TUNE_HNSW_WAVE = 0: never (A/B).
    const size_t wlds = wave_lds_bytes(a.cap, vg ? 0 : a.nwords);
    const int64_t wave_mode = tune(HNSWGPU_TUNE_HNSW_WAVE, 1);  // 2 = every launch it can serve (tests)
    const bool use_wave = wave_mode != 0 && !pf && !a.q_index && hnsw_nw() == 0 && wlds <= kMaxLds && a.M0 <= kMaxDeg &&
                          (wave_mode >= 2 || nw == 1 || a.ef >= 640);
    if (hnsw_nw() == 0 && !use_wave)
        while (nw < 4 && static_cast<int64_t>(kMaxLds / hnsw_lds_bytes(a.cap, vg ? 0 : a.nwords, nw)) * nw < 13) nw *= 2;
    int grid = a.nq;
    if (a.q_index) {  // repeat pass: few (usually no) work items, one large-list workgroup per CU at most
        nw = 4;
        grid = std::min(a.nq, 256);
    }
    if (pf && solo_enabled(a.ef)) {
        a.nwords = static_cast<int32_t>((a.n + 31) / 32);
        return launch_hnsw_solo(idx, a, st);  // one query over several CUs (solo_kernels.hpp)
    }
    if (pf) {
        // one mailbox per query; four regions in rotation, so that launches in flight (two Slots) never share one
        nw = 4;
        // as many helper workgroups per query as find a CU of their own beside the traversals (at most the configured number)
        a.pf_groups = std::max(1, std::min(pf_groups(), idx->cus / std::max(a.nq, 1) - 1));
        a.pf_hints = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(32, tune(HNSWGPU_TUNE_PF_HINTS, 4))));
        grid = 8 * ((a.nq + 7) / 8) * (1 + a.pf_groups);
        // per region: the mailboxes, then the helpers' published bounds [query][ring slot][neighbour slot]
        const size_t mail_bytes = sizeof(uint32_t) * kPfMailWords * kPfMaxQueries;
        const size_t res_bytes = sizeof(unsigned long long) * kPfMaxQueries * kPfRing * kMaxDeg;
        const size_t region = mail_bytes + res_bytes;
        // (both zero fills are waited for: the other slot's launch runs on ITS stream and would otherwise see the old bytes
        // -- harmless for results, the words are tagged, but its helpers would poll until their timeout; once per handle
        // and once per 2^24 launches)
        if (idx->s_pf.cap < 4 * region) {
            HG_TRY(idx->s_pf.ensure(4 * region));
            HG_HIP(hipMemsetAsync(idx->s_pf.p, 0, idx->s_pf.cap, st));
            HG_HIP(hipStreamSynchronize(st));
        }
        idx->pf_seq = (idx->pf_seq + 1) & 0xffffff;
        if (idx->pf_seq == 0) {  // the launch numbers start over: no tag of the previous cycle may survive
            for (auto &sl : idx->slots)
                if (sl.st) HG_HIP(hipStreamSynchronize(sl.st));
            HG_HIP(hipMemsetAsync(idx->s_pf.p, 0, idx->s_pf.cap, st));
            HG_HIP(hipStreamSynchronize(st));
            idx->pf_seq = 1;
        }
        a.pf_seq = idx->pf_seq;
        char *reg = static_cast<char *>(idx->s_pf.p) + (idx->pf_seq & 3) * region;
        a.pf_mail = reinterpret_cast<uint32_t *>(reg);
        const bool pf_eval = tune(HNSWGPU_TUNE_PF_EVAL, 1) != 0;  // 0 = the helpers only warm the L2 (A/B)
        if (pf_eval) {
            // the helpers evaluate the neighbours of the hinted nodes and publish the distances (kernels.hpp: pf_res); the
            // traversal's own int8 bounds pass stays off: for a handful of queries it costs what it saves
            a.pf_res = reinterpret_cast<unsigned long long *>(reg + mail_bytes);
            a.qrows = nullptr;
        }
    }
    if (vg) {
        a.nwords = 0;
        // one slab of n stamps per workgroup; a persistent grid bounds the slab count
        const int64_t budget = 16LL << 30;
        int64_t max_wg = budget / (4 * std::max<int64_t>(a.n, 1));
        max_wg = std::min<int64_t>(std::max<int64_t>(max_wg, 256), 256 * 12);
        grid = static_cast<int>(std::min<int64_t>(a.nq, max_wg));
        size_t need = sizeof(uint32_t) * static_cast<size_t>(a.n) * grid;
        if (need > idx->s_vis.cap) {
            HG_TRY(idx->s_vis.ensure(need));
            HG_HIP(hipMemsetAsync(idx->s_vis.p, 0, idx->s_vis.cap, st));
            idx->vis_gen = 0;
        }
        const int64_t gens = (static_cast<int64_t>(a.nq) + grid - 1) / grid * (a.max_level + 1);
        if (static_cast<int64_t>(idx->vis_gen) + gens >= 0xfffffff0LL) {
            HG_HIP(hipMemsetAsync(idx->s_vis.p, 0, idx->s_vis.cap, st));
            idx->vis_gen = 0;
        }
        a.vis = idx->s_vis.as<uint32_t>();
        a.vis_stride = a.n;
        a.gen_base = idx->vis_gen;
        idx->vis_gen += static_cast<uint32_t>(gens);
    }
    if (calibrate && a.qrows) {
        if (!idx->d_hnsw_cal) {
            HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_hnsw_cal), 2 * sizeof(unsigned long long)));
            HG_HIP(hipHostMalloc(reinterpret_cast<void **>(&idx->hnsw_cal_host), 2 * sizeof(unsigned long long), hipHostMallocDefault));
            HG_HIP(hipEventCreateWithFlags(&idx->ev_hnsw_cal, hipEventDisableTiming));
        }
        HG_HIP(hipMemsetAsync(idx->d_hnsw_cal, 0, 2 * sizeof(unsigned long long), st));
        a.rej_stats = idx->d_hnsw_cal;
    } else {
        calibrate = false;
    }
    size_t lds = use_wave ? wave_lds_bytes(a.cap, a.nwords) : hnsw_lds_bytes(a.cap, a.nwords, nw);
    HG_REQUIRE(lds <= kMaxLds, HNSWGPU_ELIMIT,
               "HNSW search state (%zu B: ef=%d, n=%lld) exceeds the 160 KiB LDS of a CU", lds, a.ef, (long long)a.n);
    if (use_wave) {
        HG_TRY(launch_hnsw_wave(idx, a, grid, lds, vg, st));
        count_launch(a.qrows ? HNSWGPU_COUNT_HNSW_REJECTION : HNSWGPU_COUNT_HNSW_PLAIN);
        count_launch(HNSWGPU_COUNT_HNSW_WAVE);
        if (calibrate) {
            HG_HIP(hipMemcpyAsync(idx->hnsw_cal_host, idx->d_hnsw_cal, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
            HG_HIP(hipEventRecord(idx->ev_hnsw_cal, st));
            idx->hnsw_cal_state = 1;
        }
        return 0;
    }
    bool l2 = a.metric == METRIC_L2;
#define CALL_K(N, R, L, W, V)                                                                                \
    do {                                                                                                     \
        if (lds > 48 * 1024)                                                                                 \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_search_kernel<N, R, L, W, V>),  \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))); \
        hipLaunchKernelGGL((hnsw_search_kernel<N, R, L, W, V>), dim3(grid), dim3(W * kWave), lds, st, a);    \
    } while (0)
#define CALL_NW(N, R, L, W)          \
    do {                             \
        if (vg) CALL_K(N, R, L, W, true); \
        else CALL_K(N, R, L, W, false);   \
    } while (0)
#define CALL_PF(N, R, L)                                                                                          \
    do {                                                                                                         \
        if (lds > 48 * 1024)                                                                                     \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_search_kernel<N, R, L, 4, false, true>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));     \
        hipLaunchKernelGGL((hnsw_search_kernel<N, R, L, 4, false, true>), dim3(grid), dim3(4 * kWave), lds, st, a); \
    } while (0)
#define CALL(N, R, L)                        \
    do {                                     \
        if (pf) CALL_PF(N, R, L);            \
        else if (nw == 1) CALL_NW(N, R, L, 1);    \
        else if (nw == 2) CALL_NW(N, R, L, 2); \
        else CALL_NW(N, R, L, 4);            \
    } while (0)
    // f32 rows in flight per wave: with the rejection test a hop fetches f32 rows for a handful of neighbours only, half
    // a quarter of them in flight are plenty, and the kernel then holds more waves per SIMD (dim 768, one wave per query:
    // 159 -> 99 VGPRs, five waves instead of three: 10,000 queries 2.24M -> 2.43M QPS, 4,096 queries 1.89M -> 2.39M,
    // 1,024 queries 0.75 -> 0.68 ms)
#define CALLR(N, R, RF)                                   \
    do {                                                  \
        if (a.qrows) {                                    \
            if (l2) CALL(N, RF, true);                    \
            else CALL(N, RF, false);                      \
        } else {                                          \
            if (l2) CALL(N, R, true);                     \
            else CALL(N, R, false);                       \
        }                                                 \
    } while (0)
    switch (nch) {
        case 1: CALLR(1, 8, 4); break;
        case 2: CALLR(2, 8, 4); break;
        case 3: CALLR(3, 8, 2); break;
        // longer rows keep their count: at dim 1536 (HBM-resident, 1.25M rows) two rows in flight instead of four cost
        // 7 % although a third wave fits per SIMD
        case 4: CALLR(4, 4, 4); break;
        case 6: CALLR(6, 4, 4); break;
        case 8: CALLR(8, 2, 2); break;
        case 12: CALLR(12, 2, 2); break;
        default: set_error("unsupported row length"); return HNSWGPU_ELIMIT;
    }
#undef CALLR
#undef CALL
#undef CALL_PF
#undef CALL_NW
#undef CALL_K
    HG_HIP(hipGetLastError());
    count_launch(pf ? HNSWGPU_COUNT_HNSW_HELPERS : (a.qrows ? HNSWGPU_COUNT_HNSW_REJECTION : HNSWGPU_COUNT_HNSW_PLAIN));
    if (calibrate) {
        HG_HIP(hipMemcpyAsync(idx->hnsw_cal_host, idx->d_hnsw_cal, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HG_HIP(hipEventRecord(idx->ev_hnsw_cal, st));
        idx->hnsw_cal_state = 1;
    }
    return 0;
}

static void free_graph(hnswgpu_index *idx) {
    idx->graph_gen++;  // searches in flight on a slot stream compare it before their repeat pass
    idx->build_flags = 0;  // (an installed graph has none: engine.hpp; hnswgpu_hnsw_build_ex sets them once its graph stands)
    idx->hnsw_cal_state = 0;  // the next graph is measured afresh
    idx->hnsw_rej_off = false;
    void *ptrs[] = {idx->d_levels, idx->d_l0, idx->d_upadj, idx->d_upoff};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    idx->d_levels = idx->d_l0 = idx->d_upadj = nullptr;
    idx->d_upoff = nullptr;
    idx->has_graph = false;
}

static int alloc_graph(hnswgpu_index *idx, int M, int M0, int64_t up_blocks) {
    int64_t n = std::max<int64_t>(idx->n, 1);
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_levels), sizeof(int32_t) * n));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_l0), sizeof(int32_t) * n * M0));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_upoff), sizeof(int64_t) * (n + 1)));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_upadj), sizeof(int32_t) * std::max<int64_t>(up_blocks, 1) * M));
    idx->M = M;
    idx->M0 = M0;
    idx->up_blocks = up_blocks;
    return 0;
}

static void fill_args(const hnswgpu_index *idx, HnswArgs &a) {
    memset(&a, 0, sizeof(a));
    a.rows = idx->d_base;
    a.row_norms = idx->d_norms;
    a.qrows = idx->d_qrows;
    a.qmeta = idx->d_qmeta;
    a.ld = idx->ld;
    a.n = idx->n;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.l0_adj = idx->d_l0;
    a.M0 = idx->M0;
    a.up_off = idx->d_upoff;
    a.up_adj = idx->d_upadj;
    a.M = idx->M;
    a.entry = idx->entry;
    a.max_level = idx->max_level;
    a.nwords = static_cast<int32_t>((idx->n + 31) / 32);
}

// A slot launch (see hnswgpu_index::Slot): the kernel publishes its completion to the host itself, the repeat pass is
// left to the caller (it knows from host_again whether any query needs one -- on ordinary data none does).
struct SlotSignal {
    int32_t *again;        // device: [0] = count, [1..] = queries to repeat; zero between calls
    uint32_t *done_cnt;    // device: zero between calls
    uint32_t *host_flag;   // device address of the flag word in the mapped block
    int32_t *host_again;   // device address of the repeat count in the mapped block
    uint32_t flag_val;
};

static int search_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t ef,
                          int32_t *d_ids, float *d_dist, int64_t *d_stats, hipStream_t st,
                          const SlotSignal *sig = nullptr, bool repeat_only = false) {
    HnswArgs a;
    fill_args(idx, a);
    a.Q = d_Q;
    a.qld = idx->dim;
    a.nq = nq;
    a.ef = ef;
    a.k = k;
    a.cap = ef + kGhost;
    a.out_ids = d_ids;
    a.out_dist = d_dist;
    a.stats = d_stats;
    int32_t *again_cnt, *again;
    if (sig) {
        again_cnt = sig->again;
        again = again_cnt + 1;
    } else {
        // queries that run out of ghost slots (hundreds of duplicated rows) list themselves in s_probes[1 ..]
        HG_TRY(idx->s_probes.ensure(sizeof(int32_t) * (static_cast<size_t>(nq) + 1)));
        again_cnt = idx->s_probes.as<int32_t>();
        again = again_cnt + 1;
        HG_HIP(hipMemsetAsync(again_cnt, 0, sizeof(int32_t), st));
    }
    int rc = 0;
    if (!repeat_only) {
        a.again = again;
        a.again_cnt = again_cnt;
        if (sig) {
            a.done_cnt = sig->done_cnt;
            a.host_flag = sig->host_flag;
            a.host_again = sig->host_again;
            a.flag_val = sig->flag_val;
        }
        hipEvent_t e0;
        prof_begin(idx, PROF_HNSW, st, &e0);
        rc = launch_hnsw_idx(idx, a, st);
        prof_end(idx, PROF_HNSW, st, e0);
        if (rc || sig) return rc;
        a.done_cnt = nullptr;
        a.host_flag = nullptr;
    }
    // ... and are repeated, on the device and without a host round trip, with the largest candidate list the LDS
    // holds, so that every tie the reference would still expand (ultra_fast.clj:175-178, `<=`) is kept.  The pass
    // finds no work item on ordinary data (a few microseconds).
    const int vgw = (force_vg() || idx->n > kLdsVisitedMaxRows) ? 0 : static_cast<int>((idx->n + 31) / 32);
    const size_t fixed = hnsw_lds_bytes(0, vgw, 4);
    const int64_t cap_max = std::min<int64_t>(65000, static_cast<int64_t>((kMaxLds - fixed) / (sizeof(uint2) + sizeof(uint16_t))));
    const int32_t big = static_cast<int32_t>(std::min<int64_t>(cap_max - ef, idx->n));
    if (big > kGhost) {
        a.cap = ef + big;
        a.again = nullptr;
        a.again_cnt = nullptr;
        a.q_index = again;
        a.nq_dev = again_cnt;
        rc = launch_hnsw_idx(idx, a, st);
    }
    if (repeat_only && rc == 0) HG_HIP(hipMemsetAsync(again_cnt, 0, sizeof(int32_t), st));  // zero between calls
    return rc;
}

struct HostGraph {
    int64_t n;
    int M, M0;
    std::vector<int32_t> levels, l0, l0_cnt, up, up_cnt;
    std::vector<float> l0_d, up_d;
    std::vector<int64_t> up_off;
    int entry = -1, top = -1;

    int32_t *adj(int32_t node, int lc, float **d, int32_t **cnt, int *m) {
        if (lc == 0) {
            *d = &l0_d[static_cast<size_t>(node) * (M0 + 1)];
            *cnt = &l0_cnt[node];
            *m = M0;
            return &l0[static_cast<size_t>(node) * (M0 + 1)];
        }
        int64_t blk = up_off[node] + (lc - 1);
        *d = &up_d[static_cast<size_t>(blk) * (M + 1)];
        *cnt = &up_cnt[blk];
        *m = M;
        return &up[static_cast<size_t>(blk) * (M + 1)];
    }

    // rows whose adjacency changed since the last upload (device copy is patched incrementally)
    std::vector<int32_t> dirty0;
    std::vector<int64_t> dirtyu;
    std::vector<uint8_t> flag0, flagu;
    // 1 once a list has been pruned: it is then sorted by (distance, insertion order) and stays so
    std::vector<uint8_t> sorted0, sortedu;

    // the dirty lists a linking thread appends to (its own: a node's flags are only touched by the thread that owns
    // the node, the lists are concatenated after the batch)
    struct Dirty {
        std::vector<int32_t> d0;
        std::vector<int64_t> du;
    };

    void mark(int32_t node, int lc, Dirty &out) {
        if (lc == 0) {
            if (!flag0[node]) {
                flag0[node] = 1;
                out.d0.push_back(node);
            }
        } else {
            int64_t blk = up_off[node] + (lc - 1);
            if (!flagu[blk]) {
                flagu[blk] = 1;
                out.du.push_back(blk);
            }
        }
    }

    // one direction of "Connect bidirectionally" (ultra_fast.clj:255-266) + prune-connections-ultra
    // (:279-299): an over-full list keeps its m closest by (distance, insertion order) -- (take max-conns
    // (sort-by dist connections)), a stable sort.  Edge distances are stored with the edges, so pruning needs no
    // distance evaluation.  After its first pruning a list IS that sorted order, and appending one edge and
    // stable-sorting again equals inserting it behind the last edge with distance <= its own and dropping the
    // last: a shift instead of a sort (the sort was most of the 0.25 s of host time in a 0.33 s 31k x 768 build).
    void add_edge(int32_t from, int32_t to, int lc, float dist, Dirty &out) {
        float *d;
        int32_t *cnt;
        int m;
        int32_t *a = adj(from, lc, &d, &cnt, &m);
        for (int i = 0; i < *cnt; i++)
            if (a[i] == to) return;
        uint8_t &srt = lc == 0 ? sorted0[from] : sortedu[up_off[from] + (lc - 1)];
        if (*cnt == m && srt) {
            if (!(dist < d[m - 1])) return;  // sorts last (ties go behind the older edges) and is dropped again
            int pos = m - 1;
            while (pos > 0 && dist < d[pos - 1]) pos--;
            for (int i = m - 1; i > pos; i--) {
                a[i] = a[i - 1];
                d[i] = d[i - 1];
            }
            a[pos] = to;
            d[pos] = dist;
            mark(from, lc, out);
            return;
        }
        a[*cnt] = to;
        d[*cnt] = dist;
        (*cnt)++;
        mark(from, lc, out);
        if (*cnt > m) {  // first overflow of this list: the stable sort itself
            int ord[kMaxDeg + 1];
            const int c = *cnt;
            for (int i = 0; i < c; i++) ord[i] = i;
            std::stable_sort(ord, ord + c, [&](int x, int y) { return d[x] < d[y]; });
            int32_t na[kMaxDeg + 1];
            float nd[kMaxDeg + 1];
            for (int i = 0; i < m; i++) {
                na[i] = a[ord[i]];
                nd[i] = d[ord[i]];
            }
            for (int i = 0; i < m; i++) {
                a[i] = na[i];
                d[i] = nd[i];
            }
            a[m] = -1;
            *cnt = m;
            srt = 1;
        }
    }

    // graph.clj's builder: an edge is appended while the list has room; into a full list it is PENDING until the list is
    // re-selected by the heuristic (prune-connections, graph.clj:208-232) -- returns false then
    bool append_edge(int32_t from, int32_t to, int lc, float dist, Dirty &out) {
        float *d;
        int32_t *cnt;
        int m;
        int32_t *a = adj(from, lc, &d, &cnt, &m);
        for (int i = 0; i < *cnt; i++)
            if (a[i] == to) return true;
        if (*cnt >= m) return false;
        a[*cnt] = to;
        d[*cnt] = dist;
        (*cnt)++;
        mark(from, lc, out);
        return true;
    }
    void set_list(int32_t node, int lc, const int32_t *ids, const float *ds, int n, Dirty &out) {
        float *d;
        int32_t *cnt;
        int m;
        int32_t *a = adj(node, lc, &d, &cnt, &m);
        for (int i = 0; i < n; i++) {
            a[i] = ids[i];
            d[i] = ds[i];
        }
        for (int i = n; i <= m; i++) a[i] = -1;
        *cnt = n;
        mark(node, lc, out);
    }
    // one side of a symmetric removal (graph.clj:226-231); true if the edge was there
    bool remove_edge(int32_t from, int32_t to, int lc, Dirty &out) {
        float *d;
        int32_t *cnt;
        int m;
        int32_t *a = adj(from, lc, &d, &cnt, &m);
        for (int i = 0; i < *cnt; i++)
            if (a[i] == to) {
                for (int t = i; t + 1 < *cnt; t++) {
                    a[t] = a[t + 1];
                    d[t] = d[t + 1];
                }
                a[--(*cnt)] = -1;
                mark(from, lc, out);
                return true;
            }
        return false;
    }
};

// Host threads of the linker.  A batch's edges are applied in two sweeps that give every adjacency list exactly the
// sequence of updates the sequential loop gives it: (A) every new node's OWN lists (disjoint between nodes),
// (B) the reverse edges, each thread walking the whole batch in order and applying the edges whose TARGET node it
// owns (node mod T).  No locks, and the graph does not depend on the thread count.
class LinkPool {
public:
    explicit LinkPool(int n) {
        for (int i = 1; i < n; i++) th_.emplace_back([this, i] { worker(i); });
    }
    ~LinkPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return static_cast<int>(th_.size()) + 1; }
    void run(const std::function<void(int)> &f) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &f;
            pending_ = static_cast<int>(th_.size());
            gen_++;
        }
        cv_.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void worker(int id) {
        int seen = 0;
        for (;;) {
            const std::function<void(int)> *f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                f = job_;
            }
            (*f)(id);
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *job_ = nullptr;
    int gen_ = 0, pending_ = 0;
    bool stop_ = false;
};

// dst[ids[i] * width + j] = packed[i * width + j]
__global__ void scatter_rows_kernel(int32_t *dst, int width, const int64_t *ids, const int32_t *packed, int64_t count) {
    int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= count * width) return;
    int64_t i = t / width;
    int j = static_cast<int>(t % width);
    dst[ids[i] * width + j] = packed[t];
}

// pinned host staging buffer (grown on demand): pageable vectors made the adjacency patches of a 1.25M-row build
// 2.6 s of its 13 s
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() {
        if (p) (void)hipHostFree(p);
    }
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 2 + 4096;
        HG_HIP(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
        return 0;
    }
};

// patch the device adjacency with the rows that changed (full arrays are never re-sent)
static int upload_dirty(hnswgpu_index *idx, HostGraph &g, hipStream_t st, PinnedBuf &pin, LinkPool &pool) {
    for (int pass = 0; pass < 2; pass++) {
        const int width = pass == 0 ? g.M0 : g.M;
        const int64_t cnt = pass == 0 ? static_cast<int64_t>(g.dirty0.size()) : static_cast<int64_t>(g.dirtyu.size());
        if (cnt == 0) continue;
        const size_t ids_bytes = sizeof(int64_t) * static_cast<size_t>(cnt);
        const size_t pack_bytes = sizeof(int32_t) * static_cast<size_t>(cnt) * width;
        HG_TRY(pin.ensure(ids_bytes + pack_bytes));
        int64_t *ids = static_cast<int64_t *>(pin.p);
        int32_t *pack = reinterpret_cast<int32_t *>(ids + cnt);
        auto pack_range = [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; i++) {
                int64_t row = pass == 0 ? g.dirty0[i] : g.dirtyu[i];
                ids[i] = row;
                const int32_t *src = pass == 0 ? &g.l0[row * (g.M0 + 1)] : &g.up[row * (g.M + 1)];
                const int have = pass == 0 ? g.l0_cnt[row] : g.up_cnt[row];
                for (int j = 0; j < width; j++) pack[i * width + j] = j < have ? src[j] : -1;
                if (pass == 0) g.flag0[row] = 0;
                else g.flagu[row] = 0;
            }
        };
        const int nt = cnt >= 4096 ? pool.size() : 1;
        if (nt == 1) pack_range(0, cnt);
        else pool.run([&](int me) { pack_range(cnt * me / nt, cnt * (me + 1) / nt); });
        HG_TRY(idx->s_partial.ensure(ids_bytes + pack_bytes + 64));
        int64_t *d_ids = idx->s_partial.as<int64_t>();
        int32_t *d_pack = reinterpret_cast<int32_t *>(d_ids + cnt);
        HG_HIP(hipMemcpyAsync(d_ids, ids, ids_bytes + pack_bytes, hipMemcpyHostToDevice, st));  // ids | pack, one copy
        int64_t total = cnt * width;
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, st,
                           pass == 0 ? idx->d_l0 : idx->d_upadj, width, d_ids, d_pack, cnt);
        HG_HIP(hipGetLastError());
        HG_HIP(hipStreamSynchronize(st));  // the staging buffers are reused by the next pass
        if (pass == 0) g.dirty0.clear();
        else g.dirtyu.clear();
    }
    return 0;
}

static int upload_graph(hnswgpu_index *idx, const HostGraph &g, hipStream_t st, std::vector<int32_t> &tmp0,
                        std::vector<int32_t> &tmpu) {
    const int64_t n = g.n;
    tmp0.assign(static_cast<size_t>(n) * g.M0, -1);
    for (int64_t i = 0; i < n; i++)
        for (int j = 0; j < g.l0_cnt[i] && j < g.M0; j++) tmp0[i * g.M0 + j] = g.l0[i * (g.M0 + 1) + j];
    const int64_t blocks = g.up_off[n];
    tmpu.assign(static_cast<size_t>(std::max<int64_t>(blocks, 1)) * g.M, -1);
    for (int64_t b = 0; b < blocks; b++)
        for (int j = 0; j < g.up_cnt[b] && j < g.M; j++) tmpu[b * g.M + j] = g.up[b * (g.M + 1) + j];
    HG_HIP(hipMemcpyAsync(idx->d_l0, tmp0.data(), sizeof(int32_t) * tmp0.size(), hipMemcpyHostToDevice, st));
    HG_HIP(hipMemcpyAsync(idx->d_upadj, tmpu.data(), sizeof(int32_t) * tmpu.size(), hipMemcpyHostToDevice, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // namespace hg

using namespace hg;

extern "C" {

int hnswgpu_set_graph(hnswgpu_index *idx, const int32_t *levels, const int32_t *l0_adj, int32_t M0,
                      const int64_t *up_off, const int32_t *up_adj, int32_t M, int32_t entry, int32_t max_level) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    const int64_t n = idx->n;
    HG_REQUIRE(n == 0 || (levels && l0_adj && up_off), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(M >= 1 && M0 >= 1 && M0 <= kMaxDeg && M <= kMaxDeg, HNSWGPU_ELIMIT, "need 1 <= M, M0 <= %d", kMaxDeg);
    int64_t blocks = n > 0 ? up_off[n] : 0;
    if (n > 0) {
        HG_REQUIRE(entry >= 0 && entry < n, HNSWGPU_EINVAL, "entry out of range");
        HG_REQUIRE(max_level >= 0 && levels[entry] >= max_level, HNSWGPU_EINVAL, "entry level < max_level");
        HG_REQUIRE(up_off[0] == 0, HNSWGPU_EINVAL, "up_off[0] != 0");
        for (int64_t i = 0; i < n; i++) {
            HG_REQUIRE(levels[i] >= 0 && up_off[i + 1] - up_off[i] == levels[i], HNSWGPU_EINVAL,
                       "up_off is not the prefix sum of levels at node %lld", (long long)i);
            HG_REQUIRE(levels[i] <= max_level, HNSWGPU_EINVAL, "node %lld level > max_level", (long long)i);
        }
        HG_REQUIRE(blocks == 0 || up_adj, HNSWGPU_EINVAL, "up_adj is null");
        // every edge must point at a node that exists on that layer: the kernel trusts this
        for (int64_t i = 0; i < n * M0; i++)
            HG_REQUIRE(l0_adj[i] >= -1 && l0_adj[i] < n, HNSWGPU_EINVAL, "l0_adj entry out of range");
        for (int64_t i = 0; i < n; i++)
            for (int lv = 1; lv <= levels[i]; lv++)
                for (int j = 0; j < M; j++) {
                    int32_t nb = up_adj[(up_off[i] + lv - 1) * M + j];
                    HG_REQUIRE(nb >= -1 && nb < n, HNSWGPU_EINVAL, "up_adj entry out of range");
                    HG_REQUIRE(nb < 0 || levels[nb] >= lv, HNSWGPU_EINVAL,
                               "edge %lld->%d on layer %d: target has no such layer", (long long)i, nb, lv);
                }
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    // every earlier call on this handle is ordered before `st` by begin_call, except the small synchronous searches on
    // the slot streams: once `st` and both slot streams are idle nothing can still be traversing the graph that is about
    // to be freed (and no other handle on this GPU is stalled)
    HG_TRY(quiesce(idx, st));
    free_graph(idx);
    HG_TRY(alloc_graph(idx, M, M0, blocks));
    HG_TRY(ensure_qrows(idx, st));
    if (n > 0) {
        HG_HIP(hipMemcpyAsync(idx->d_levels, levels, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(idx->d_l0, l0_adj, sizeof(int32_t) * n * M0, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(idx->d_upoff, up_off, sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
        if (blocks > 0)
            HG_HIP(hipMemcpyAsync(idx->d_upadj, up_adj, sizeof(int32_t) * blocks * M, hipMemcpyHostToDevice, st));
        HG_HIP(hipStreamSynchronize(st));
        idx->h_levels.assign(levels, levels + n);
        idx->h_l0.assign(l0_adj, l0_adj + n * M0);
        idx->h_upoff.assign(up_off, up_off + n + 1);
        idx->h_upadj.assign(up_adj, up_adj + blocks * M);
    } else {
        idx->h_levels.clear();
        idx->h_l0.clear();
        idx->h_upoff.assign(1, 0);
        idx->h_upadj.clear();
    }
    idx->entry = n > 0 ? entry : -1;
    idx->max_level = n > 0 ? max_level : 0;
    idx->has_graph = true;
    return 0;
}

int hnswgpu_graph_sizes(const hnswgpu_index *idx, int32_t *M, int32_t *M0, int64_t *up_blocks, int32_t *entry,
                        int32_t *max_level) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(idx->has_graph, HNSWGPU_ESTATE, "index has no graph");
    if (M) *M = idx->M;
    if (M0) *M0 = idx->M0;
    if (up_blocks) *up_blocks = idx->up_blocks;
    if (entry) *entry = idx->entry;
    if (max_level) *max_level = idx->max_level;
    return 0;
}

int hnswgpu_get_graph(const hnswgpu_index *idx, int32_t *levels, int32_t *l0_adj, int64_t *up_off,
                      int32_t *up_adj) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(idx->has_graph, HNSWGPU_ESTATE, "index has no graph");
    if (levels && !idx->h_levels.empty()) memcpy(levels, idx->h_levels.data(), sizeof(int32_t) * idx->h_levels.size());
    if (l0_adj && !idx->h_l0.empty()) memcpy(l0_adj, idx->h_l0.data(), sizeof(int32_t) * idx->h_l0.size());
    if (up_off) memcpy(up_off, idx->h_upoff.data(), sizeof(int64_t) * idx->h_upoff.size());
    if (up_adj && !idx->h_upadj.empty()) memcpy(up_adj, idx->h_upadj.data(), sizeof(int32_t) * idx->h_upadj.size());
    return 0;
}

static int check_hnsw_args(const hnswgpu_index *idx, const void *Q, int32_t nq, int32_t k, int32_t *ef,
                           const void *ids, const void *dist) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(nq >= 0 && k >= 1, HNSWGPU_EINVAL, "need nq >= 0 and k >= 1");
    HG_REQUIRE(nq == 0 || (Q && ids && dist), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(idx->has_graph, HNSWGPU_ESTATE, "index has no graph (call hnswgpu_hnsw_build / hnswgpu_set_graph)");
    if (*ef <= 0) *ef = k > 50 ? k : 50;  // ef = (max k 50), ultra_fast.clj:355
    if (*ef < k) *ef = k;
    HG_REQUIRE(*ef <= 4096, HNSWGPU_ELIMIT, "ef > 4096 is not supported");
    return 0;
}

static bool zero_copy() { return tune(HNSWGPU_TUNE_ZEROCOPY, 1) != 0; }  // 0 = always stage through copies (A/B measurements)

// Small combined batches (at most one workgroup per CU): the queries are written into a block of mapped pinned host
// memory, the traversal kernel reads them from there and writes ids / distances / counters back into the same block,
// and its last workgroup sets a flag the calling thread spins on.  One kernel launch is the only runtime call on the
// path: no copy, no counter reset, no second launch, no hipStreamSynchronize (measured before: 0.49 ms per
// single-query call for a 0.35 ms kernel).  Two such batches may be in flight (own stream, block and counters each).
static int hnsw_search_batch_slot(hnswgpu_index *idx, const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
    const int32_t k = batch[0]->k, ef = batch[0]->ef;
    const size_t cnt = static_cast<size_t>(total) * k;
    // block layout: [flag u32 | repeat count i32 | pad to 64 B][queries][ids][distances][stats]
    const size_t qb = sizeof(float) * static_cast<size_t>(total) * idx->dim, ib = sizeof(int32_t) * cnt, db = sizeof(float) * cnt,
                 sb = sizeof(int64_t) * 2 * total;
    const size_t o_q = 64, o_s = (o_q + qb + 63) & ~size_t(63), o_i = o_s + sb, o_d = o_i + ib, bytes = o_d + db;
    hnswgpu_index::Slot *slot = nullptr;
    std::unique_lock<std::mutex> sl;
    for (auto &s : idx->slots) {
        sl = std::unique_lock<std::mutex>(s.mu, std::try_to_lock);
        if (sl.owns_lock()) {
            slot = &s;
            break;
        }
    }
    if (!slot) {  // both busy (more callers than the combiner's two batches in flight: cannot happen, but be safe)
        slot = &idx->slots[0];
        sl = std::unique_lock<std::mutex>(slot->mu);
    }
    HG_HIP(hipSetDevice(idx->device));
    HG_TRY(slot_prepare(*slot, bytes));
    char *hp = static_cast<char *>(slot->h), *dp = static_cast<char *>(slot->d);
    size_t o = 0;
    float *hq = reinterpret_cast<float *>(hp + o_q);
    for (auto *r : batch) {
        memcpy(hq + o, r->Q, sizeof(float) * static_cast<size_t>(r->nq) * idx->dim);
        o += static_cast<size_t>(r->nq) * idx->dim;
    }
    SlotSignal sig;
    sig.again = slot->d_again;
    sig.done_cnt = slot->d_done;
    sig.host_flag = reinterpret_cast<uint32_t *>(dp);
    sig.host_again = reinterpret_cast<int32_t *>(dp + 4);
    sig.flag_val = ++slot->seq;
    volatile uint32_t *h_flag = reinterpret_cast<volatile uint32_t *>(hp);
    uint64_t gen;
    {
        std::lock_guard<std::mutex> lk(idx->mu);  // the index state is read (and the launch enqueued) under its lock
        HG_REQUIRE(idx->has_graph && idx->n > 0, HNSWGPU_ESTATE, "index has no graph (call hnswgpu_hnsw_build / hnswgpu_set_graph)");
        gen = idx->graph_gen;
        HG_TRY(search_enqueue(idx, reinterpret_cast<const float *>(dp + o_q), total, k, ef, reinterpret_cast<int32_t *>(dp + o_i),
                              reinterpret_cast<float *>(dp + o_d), reinterpret_cast<int64_t *>(dp + o_s), slot->st, &sig));
    }
    HG_TRY(slot_wait(*slot, h_flag, sig.flag_val));
    if (*reinterpret_cast<volatile int32_t *>(hp + 4) != 0) {
        // some query met more tied candidates than the ghost slots hold (hundreds of duplicated rows): the repeat pass --
        // against the graph the first pass ran on, or not at all (set_graph / hnsw_build wait for this slot's stream
        // before they free the old graph, but may have installed a new one since the first pass finished)
        std::lock_guard<std::mutex> lk(idx->mu);
        HG_REQUIRE(idx->has_graph && idx->graph_gen == gen, HNSWGPU_ESTATE,
                   "the graph was replaced while a search on it was in flight");
        HG_TRY(search_enqueue(idx, reinterpret_cast<const float *>(dp + o_q), total, k, ef, reinterpret_cast<int32_t *>(dp + o_i),
                              reinterpret_cast<float *>(dp + o_d), reinterpret_cast<int64_t *>(dp + o_s), slot->st, &sig, true));
        HG_HIP(hipStreamSynchronize(slot->st));
    }
    const int32_t *hi = reinterpret_cast<const int32_t *>(hp + o_i);
    const float *hd = reinterpret_cast<const float *>(hp + o_d);
    const int64_t *hs = reinterpret_cast<const int64_t *>(hp + o_s);
    int64_t q0 = 0;
    for (auto *r : batch) {
        const size_t c = static_cast<size_t>(r->nq) * k;
        memcpy(r->out_ids, hi + q0 * k, sizeof(int32_t) * c);
        memcpy(r->out_dist, hd + q0 * k, sizeof(float) * c);
        if (r->stats) memcpy(r->stats, hs + 2 * q0, sizeof(int64_t) * 2 * r->nq);
        q0 += r->nq;
    }
    return 0;
}

// One launch for a set of queued synchronous requests with the same (k, ef): queries concatenated on the host,
// results scattered back (see hnswgpu_index::SearchReq).
static int hnsw_search_batch(hnswgpu_index *idx, const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
    if (zero_copy() && total <= kZcMaxQueries && !force_vg() && idx->n <= kLdsVisitedMaxRows)
        return hnsw_search_batch_slot(idx, batch, total);
    const int32_t k = batch[0]->k, ef = batch[0]->ef;
    const int64_t cnt = static_cast<int64_t>(total) * k;
    std::lock_guard<std::mutex> lk(idx->mu);
    // under the lock: a concurrent set_graph / hnsw_build may have replaced the graph since the caller's argument check
    HG_REQUIRE(idx->has_graph && idx->n > 0, HNSWGPU_ESTATE, "index has no graph (call hnswgpu_hnsw_build / hnswgpu_set_graph)");
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    HG_TRY(idx->s_stats.ensure(sizeof(int64_t) * 2 * total));
    // one pinned staging block [queries | stats | ids | distances], four transfers per batch whatever the number of
    // callers in it.  (Replaying the single-query sequence -- upload, counter reset, two launches, downloads -- as one
    // captured hipGraph was tried and changed nothing: 0.483 ms per call either way, the time is on the device.)
    const size_t qb = sizeof(float) * static_cast<size_t>(total) * idx->dim, ib = sizeof(int32_t) * cnt,
                 db = sizeof(float) * cnt, sb = sizeof(int64_t) * 2 * total;
    HG_TRY(ensure_pinned(idx, qb + ib + db + sb + 64));
    char *hp = static_cast<char *>(idx->h_pin);
    float *hq = reinterpret_cast<float *>(hp);
    int64_t *hs = reinterpret_cast<int64_t *>(hp + ((qb + 7) & ~size_t(7)));
    int32_t *hi = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(hs) + sb);
    float *hd = reinterpret_cast<float *>(reinterpret_cast<char *>(hi) + ib);
    size_t o = 0;
    for (auto *r : batch) {
        memcpy(hq + o, r->Q, sizeof(float) * static_cast<size_t>(r->nq) * idx->dim);
        o += static_cast<size_t>(r->nq) * idx->dim;
    }
    HG_TRY(upload_queries(idx, hq, total, st));
    HG_TRY(search_enqueue(idx, idx->s_q.as<float>(), total, k, ef, idx->s_ids.as<int32_t>(), idx->s_outd.as<float>(),
                          idx->s_stats.as<int64_t>(), st));
    HG_HIP(hipMemcpyAsync(hi, idx->s_ids.p, ib, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(hd, idx->s_outd.p, db, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(hs, idx->s_stats.p, sb, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    int64_t q0 = 0;
    for (auto *r : batch) {
        const size_t c = static_cast<size_t>(r->nq) * k;
        memcpy(r->out_ids, hi + q0 * k, sizeof(int32_t) * c);
        memcpy(r->out_dist, hd + q0 * k, sizeof(float) * c);
        if (r->stats) memcpy(r->stats, hs + 2 * q0, sizeof(int64_t) * 2 * r->nq);
        q0 += r->nq;
    }
    return 0;
}

int hnswgpu_hnsw_search_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t ef,
                            int32_t *d_out_ids, float *d_out_dist, int64_t *d_stats, void *stream) {
    HG_TRY(check_hnsw_args(idx, d_Q, nq, k, &ef, d_out_ids, d_out_dist));
    if (nq == 0) return 0;
    HG_REQUIRE(idx->n > 0, HNSWGPU_ESTATE, "empty index: use the host entry point");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(search_enqueue(idx, d_Q, nq, k, ef, d_out_ids, d_out_dist, d_stats, st));
    return end_call(idx, st);
}

int hnswgpu_hnsw_search(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t ef, int32_t *out_ids,
                        float *out_dist, int64_t *stats) {
    HG_TRY(check_hnsw_args(idx, Q, nq, k, &ef, out_ids, out_dist));
    if (nq == 0) return 0;
    int64_t cnt = static_cast<int64_t>(nq) * k;
    if (idx->n == 0) {  // (if (or (nil? entry-point) (zero? size)) [] ...), ultra_fast.clj:349-351
        for (int64_t i = 0; i < cnt; i++) {
            out_ids[i] = -1;
            out_dist[i] = __builtin_inff();
        }
        if (stats) memset(stats, 0, sizeof(int64_t) * 2 * nq);
        return 0;
    }
    // queue the request; whoever finds no leader becomes one and serves the queue, batch by batch
    hnswgpu_index::SearchReq me;
    me.Q = Q;
    me.nq = nq;
    me.k = k;
    me.ef = ef;
    me.out_ids = out_ids;
    me.out_dist = out_dist;
    me.stats = stats;
    return combine_search(
        idx->cmb_hnsw, me,
        [](const hnswgpu_index::SearchReq *first, const hnswgpu_index::SearchReq *r, int64_t total) {
            return r->k == first->k && r->ef == first->ef && total + r->nq <= 16384;  // one traversal arithmetic: any mix
        },
        [idx](const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
            return hnsw_search_batch(idx, batch, total);
        });
}

}  // extern "C"

namespace hg {

// Distances of the edges of `nlists` adjacency lists (list t belongs to node owner[t], `width` slots, -1 padded), by the
// traversal's own arithmetic: d(node, neighbour) carries the bits the build-mode search reported for that edge (the lane
// sums are symmetric in their two operands).  One wave per list.  insert-single prunes an over-full list by the stored
// edge distances (ultra_fast.clj:279-299); a graph that arrives through hnswgpu_set_graph / hnswgpu_load has none.
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void edge_dist_kernel(const float *rows, const float *norms, int64_t ld, int dim, int metric,
                                                        const int32_t *owner, const int32_t *adj, int width, int64_t nlists,
                                                        float *out) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t t = static_cast<int64_t>(blockIdx.x) * kNWave + (threadIdx.x >> 6);
    if (t >= nlists) return;
    const int64_t node = owner ? owner[t] : t;
    const int nvec = static_cast<int>(ld / 4);
    float4 q[NCH];
    load_row<NCH>(q, rows + node * ld, nvec, lane, true);
    const float qn = metric == METRIC_COS ? norms[node] : 0.0f;
    for (int j0 = 0; j0 < width; j0 += RB) {
        float4 r[RB][NCH];
        int32_t nb[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            nb[b] = j0 + b < width ? adj[t * width + j0 + b] : -1;
            load_row<NCH>(r[b], rows + static_cast<int64_t>(nb[b] >= 0 ? nb[b] : 0) * ld, nvec, lane, nb[b] >= 0);
        }
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const float sum = wave_sum(lane_partial<NCH, L2>(q, r[b]));
            if (lane == 0 && j0 + b < width)
                out[t * width + j0 + b] = nb[b] >= 0 ? finish_dist(metric, sum, qn, metric == METRIC_COS ? norms[nb[b]] : 0.0f) : 0.0f;
        }
    }
}

static int launch_edge_dist(hnswgpu_index *idx, const int32_t *d_owner, const int32_t *d_adj, int width, int64_t nlists,
                            float *d_out, hipStream_t st) {
    if (nlists <= 0) return 0;
    const unsigned grid = static_cast<unsigned>((nlists + kNWave - 1) / kNWave);
    const bool l2 = idx->metric == METRIC_L2;
#define CALL(N, R, L)                                                                                                  \
    hipLaunchKernelGGL((edge_dist_kernel<N, R, L>), dim3(grid), dim3(kWG), 0, st, idx->d_base, idx->d_norms, idx->ld, idx->dim, \
                       idx->metric, d_owner, d_adj, width, nlists, d_out)
    HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// RAII for the builder's own device buffers
struct OwnedBuf : DevBuf {
    ~OwnedBuf() { release(); }
};

static int launch_heur(hnswgpu_index *idx, HeurArgs a, hipStream_t st) {
    if (a.ntasks <= 0) return 0;
    a.rows = idx->d_base;
    a.row_norms = idx->d_norms;
    a.ld = idx->ld;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.keep_rows = tune(HNSWGPU_TUNE_BUILD_KEEP_ROWS, 1) != 0 ? 1 : 0;
    const bool l2 = idx->metric == METRIC_L2;
#define CALL(N, R, L) hipLaunchKernelGGL((heuristic_select_kernel<N, R, L>), dim3(a.ntasks), dim3(kWG), 0, st, a)
    HG_DISPATCH(idx->nch, l2, CALL);
#undef CALL
    HG_HIP(hipGetLastError());
    return 0;
}

// A pending reverse edge of graph.clj's builder: `id` wants into the FULL list of `target` on layer `lc`.
struct PendingEdge {
    int32_t target, lc, id;
    float dist;
    int64_t seq;  // position in the batch's sequential order of reverse edges
};

// The re-selection of the over-full lists of one batch (prune-connections, graph.clj:208-232): per (target, layer) the
// candidates are its m present edges plus the pending ones, ascending by (distance, id); heuristic_select_kernel keeps
// at most m of them (extend-candidates? false); with `sym` every dropped edge leaves the other node's list as well
// (:226-231).  Batched: every task sees the lists as they stood when the batch's appends were done (the batched build's
// own semantics); `seq_exact` (one insert per batch, the reference's order): a task whose list an earlier task of the
// same insert has shortened is selected again from the list as it then stands -- the sequential loop's result.
struct Pruner {
    hnswgpu_index *idx;
    HostGraph &g;
    hipStream_t st;
    bool sym, seq_exact;
    OwnedBuf d_task, d_out;
    std::vector<int64_t> off;
    std::vector<int32_t> cid, tm, oid, ocnt;
    std::vector<float> cd, od;
    struct Task {
        int32_t target, lc;
        int64_t first;  // its pending edges: pend[first, first + npend)
        int32_t npend;
    };
    std::vector<Task> tasks;
    int64_t n_tasks = 0, n_removed = 0;

    Pruner(hnswgpu_index *i, HostGraph &gr, hipStream_t s, bool sy, bool sq) : idx(i), g(gr), st(s), sym(sy), seq_exact(sq) {}

    // candidates of task t from the CURRENT list + its pending edges, ascending by (distance, id)
    void gather(const Task &t, const std::vector<PendingEdge> &pend, std::vector<std::pair<float, int32_t>> &c) {
        float *d;
        int32_t *cnt;
        int m;
        int32_t *a = g.adj(t.target, t.lc, &d, &cnt, &m);
        c.clear();
        for (int i = 0; i < *cnt; i++) c.emplace_back(d[i], a[i]);
        for (int i = 0; i < t.npend; i++) c.emplace_back(pend[t.first + i].dist, pend[t.first + i].id);
        std::sort(c.begin(), c.end());
        if (c.size() > static_cast<size_t>(kSelMaxCand)) c.resize(kSelMaxCand);
    }

    int select(const std::vector<std::vector<std::pair<float, int32_t>>> &cands, const std::vector<int32_t> &ms) {
        const int nt = static_cast<int>(cands.size());
        off.assign(nt + 1, 0);
        for (int i = 0; i < nt; i++) off[i + 1] = off[i] + static_cast<int64_t>(cands[i].size());
        const int64_t tot = off[nt];
        cid.resize(std::max<int64_t>(tot, 1));
        cd.resize(std::max<int64_t>(tot, 1));
        for (int i = 0; i < nt; i++)
            for (size_t j = 0; j < cands[i].size(); j++) {
                cid[off[i] + j] = cands[i][j].second;
                cd[off[i] + j] = cands[i][j].first;
            }
        return select_flat(nt, ms);
    }

    // off / cid / cd hold `nt` tasks; ms their m.  Results in oid / od / ocnt (stride M0).
    int select_flat(int nt, const std::vector<int32_t> &ms) {
        const int64_t tot = off[nt];
        const int stride = g.M0;
        const size_t b_off = sizeof(int64_t) * (nt + 1), b_id = sizeof(int32_t) * tot, b_d = sizeof(float) * tot, b_m = sizeof(int32_t) * nt;
        HG_TRY(d_task.ensure(b_off + b_id + b_d + b_m + 64));
        char *dp = static_cast<char *>(d_task.p);
        HG_HIP(hipMemcpyAsync(dp, off.data(), b_off, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(dp + b_off, cid.data(), b_id, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(dp + b_off + b_id, cd.data(), b_d, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(dp + b_off + b_id + b_d, ms.data(), b_m, hipMemcpyHostToDevice, st));
        const size_t o_id = sizeof(int32_t) * static_cast<size_t>(nt) * stride, o_d = sizeof(float) * static_cast<size_t>(nt) * stride,
                     o_c = sizeof(int32_t) * nt;
        HG_TRY(d_out.ensure(o_id + o_d + o_c + 64));
        char *op = static_cast<char *>(d_out.p);
        HeurArgs a;
        memset(&a, 0, sizeof(a));
        a.ntasks = nt;
        a.off = reinterpret_cast<const int64_t *>(dp);
        a.cand_id = reinterpret_cast<const int32_t *>(dp + b_off);
        a.cand_d = reinterpret_cast<const float *>(dp + b_off + b_id);
        a.m = reinterpret_cast<const int32_t *>(dp + b_off + b_id + b_d);
        a.extend = 0;  // prune-connections passes extend-candidates? false (graph.clj:222)
        a.out_stride = stride;
        a.out_id = reinterpret_cast<int32_t *>(op);
        a.out_d = reinterpret_cast<float *>(op + o_id);
        a.out_cnt = reinterpret_cast<int32_t *>(op + o_id + o_d);
        HG_TRY(launch_heur(idx, a, st));
        oid.resize(static_cast<size_t>(nt) * stride);
        od.resize(static_cast<size_t>(nt) * stride);
        ocnt.resize(nt);
        HG_HIP(hipMemcpyAsync(oid.data(), a.out_id, o_id, hipMemcpyDeviceToHost, st));
        HG_HIP(hipMemcpyAsync(od.data(), a.out_d, o_d, hipMemcpyDeviceToHost, st));
        HG_HIP(hipMemcpyAsync(ocnt.data(), a.out_cnt, o_c, hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
        return 0;
    }

    // The batched build's form of run(): every task sees the lists as they stood when the batch's appends were done, so
    // the tasks are independent -- candidates gathered and sorted, lists replaced and (SYMMETRIC) removals collected by the
    // linker's threads, flat arrays throughout (a 1.25M-row build has millions of tasks of ~33 candidates).
    int run_batched(std::vector<PendingEdge> &pend, LinkPool &pool, std::vector<HostGraph::Dirty> &dirties) {
        if (pend.empty()) return 0;
        std::sort(pend.begin(), pend.end(), [](const PendingEdge &x, const PendingEdge &y) {
            return x.target != y.target ? x.target < y.target : (x.lc != y.lc ? x.lc < y.lc : x.seq < y.seq);
        });
        tasks.clear();
        for (size_t i = 0; i < pend.size();) {
            size_t j = i;
            while (j < pend.size() && pend[j].target == pend[i].target && pend[j].lc == pend[i].lc) j++;
            tasks.push_back({pend[i].target, pend[i].lc, static_cast<int64_t>(i), static_cast<int32_t>(j - i)});
            i = j;
        }
        const int nt = static_cast<int>(tasks.size());
        off.assign(nt + 1, 0);
        std::vector<int32_t> ms(nt);
        for (int i = 0; i < nt; i++) {
            const Task &t = tasks[i];
            const int have = t.lc == 0 ? g.l0_cnt[t.target] : g.up_cnt[g.up_off[t.target] + (t.lc - 1)];
            off[i + 1] = off[i] + std::min<int64_t>(kSelMaxCand, static_cast<int64_t>(have) + t.npend);
            ms[i] = t.lc == 0 ? g.M0 : g.M;
        }
        cid.resize(std::max<int64_t>(off[nt], 1));
        cd.resize(std::max<int64_t>(off[nt], 1));
        const int nth = nt >= 1024 ? pool.size() : 1;
        auto fill = [&](int me) {
            std::vector<std::pair<float, int32_t>> c;
            for (int i = static_cast<int>(static_cast<int64_t>(nt) * me / nth); i < static_cast<int>(static_cast<int64_t>(nt) * (me + 1) / nth); i++) {
                gather(tasks[i], pend, c);
                for (size_t j = 0; j < c.size(); j++) {
                    cid[off[i] + j] = c[j].second;
                    cd[off[i] + j] = c[j].first;
                }
            }
        };
        if (nth == 1) fill(0);
        else pool.run(fill);
        HG_TRY(select_flat(nt, ms));
        n_tasks += nt;
        const int stride = g.M0;
        std::vector<std::vector<std::pair<int32_t, std::pair<int32_t, int32_t>>>> removals(nth);
        auto apply = [&](int me) {
            for (int i = static_cast<int>(static_cast<int64_t>(nt) * me / nth); i < static_cast<int>(static_cast<int64_t>(nt) * (me + 1) / nth); i++) {
                const Task &t = tasks[i];
                const int32_t *keep = &oid[static_cast<size_t>(i) * stride];
                const int nk = ocnt[i];
                if (sym)  // (before the list is replaced: the candidates are the present edges + the parked ones)
                    for (int64_t j = off[i]; j < off[i + 1]; j++) {
                        bool kept = false;
                        for (int r = 0; r < nk; r++) kept |= keep[r] == cid[j];
                        if (!kept) removals[me].push_back({cid[j], {t.target, t.lc}});
                    }
                g.set_list(t.target, t.lc, keep, &od[static_cast<size_t>(i) * stride], nk, dirties[me]);
            }
        };
        if (nth == 1) apply(0);
        else pool.run(apply);
        for (auto &rv : removals)  // (a removal touches another node's list: serial, and commutative -- the order does not matter)
            for (auto &r : rv)
                if (g.remove_edge(r.first, r.second.first, r.second.second, dirties[0])) n_removed++;
        return 0;
    }

    // pend: the batch's pending edges in sequential order (seq ascending)
    int run(std::vector<PendingEdge> &pend, HostGraph::Dirty &dirty) {
        if (pend.empty()) return 0;
        // group by (target, lc), a group's edges in sequential order; the groups in the order of their first edge
        std::stable_sort(pend.begin(), pend.end(), [](const PendingEdge &x, const PendingEdge &y) {
            return x.target != y.target ? x.target < y.target : x.lc < y.lc;
        });
        tasks.clear();
        for (size_t i = 0; i < pend.size();) {
            size_t j = i;
            while (j < pend.size() && pend[j].target == pend[i].target && pend[j].lc == pend[i].lc) j++;
            tasks.push_back({pend[i].target, pend[i].lc, static_cast<int64_t>(i), static_cast<int32_t>(j - i)});
            i = j;
        }
        std::sort(tasks.begin(), tasks.end(), [&](const Task &x, const Task &y) { return pend[x.first].seq < pend[y.first].seq; });
        const int nt = static_cast<int>(tasks.size());
        std::vector<std::vector<std::pair<float, int32_t>>> cands(nt);
        std::vector<int32_t> ms(nt);
        for (int i = 0; i < nt; i++) {
            gather(tasks[i], pend, cands[i]);
            ms[i] = tasks[i].lc == 0 ? g.M0 : g.M;
        }
        HG_TRY(select(cands, ms));
        n_tasks += nt;
        const int stride = g.M0;
        // keep-lists of the launch above (select() is reused for the re-selections of the sequential mode)
        std::vector<int32_t> kid(oid), kcnt(ocnt);
        std::vector<float> kd(od);
        std::vector<std::pair<int32_t, int32_t>> touched;  // (node, layer) lists shortened by a removal of this run
        auto is_touched = [&](int32_t node, int32_t lc) {
            for (auto &tl : touched)
                if (tl.first == node && tl.second == lc) return true;
            return false;
        };
        std::vector<std::pair<int32_t, std::pair<int32_t, int32_t>>> removals;  // batched: applied after every list is set
        for (int i = 0; i < nt; i++) {
            const Task &t = tasks[i];
            const int32_t *keep = &kid[static_cast<size_t>(i) * stride];
            const float *keepd = &kd[static_cast<size_t>(i) * stride];
            int nk = kcnt[i];
            std::vector<std::pair<float, int32_t>> *cv = &cands[i];
            std::vector<std::vector<std::pair<float, int32_t>>> one(1);
            if (seq_exact && is_touched(t.target, t.lc)) {
                // an earlier pruning of this insert took an edge out of this list: the sequential loop meets it shorter
                float *d;
                int32_t *cnt;
                int m;
                (void)g.adj(t.target, t.lc, &d, &cnt, &m);
                int used = 0;
                while (used < t.npend && *cnt < m) {  // room again: plain appends
                    g.append_edge(t.target, pend[t.first + used].id, t.lc, pend[t.first + used].dist, dirty);
                    used++;
                }
                if (used == t.npend) continue;
                Task rest = {t.target, t.lc, t.first + used, t.npend - used};
                gather(rest, pend, one[0]);
                std::vector<int32_t> m1(1, t.lc == 0 ? g.M0 : g.M);
                HG_TRY(select(one, m1));
                n_tasks++;
                keep = oid.data();
                keepd = od.data();
                nk = ocnt[0];
                cv = &one[0];
            }
            g.set_list(t.target, t.lc, keep, keepd, nk, dirty);
            if (sym)
                for (auto &c : *cv) {
                    bool kept = false;
                    for (int r = 0; r < nk; r++) kept |= keep[r] == c.second;
                    if (kept) continue;
                    if (seq_exact) {
                        if (g.remove_edge(c.second, t.target, t.lc, dirty)) {
                            n_removed++;
                            touched.emplace_back(c.second, t.lc);
                        }
                    } else {
                        removals.push_back({c.second, {t.target, t.lc}});
                    }
                }
        }
        for (auto &r : removals)
            if (g.remove_edge(r.first, r.second.first, r.second.second, dirty)) n_removed++;
        return 0;
    }
};

// insert-batch (ultra_fast.clj:303-344) over rows [done, g.n): every node of a batch runs the reference's per-level
// search (search-layer-ultra with ef-construction at layer 0 and 1 above, :250-251) on the GPU against the graph as it
// stood when the batch started, then the host links the batch in row order (:255-266) and prunes over-full lists
// (:279-299).  The device adjacency (idx->d_l0 / d_upadj, sized for g.n) holds the graph of rows [0, done) on entry and
// of all rows on return; g.entry / g.top are updated (:271-273).
// `flags` (include/hnswgpu.h, HNSWGPU_BUILD_*): SEQUENTIAL = one insert per batch, the walk starting at
// min(level, entry-level) with the entry point (:247-248) -- insert-single itself; HEURISTIC = links chosen by
// get-neighbors-heuristic (graph.clj:162-198) on the device (heuristic_select_kernel) instead of the m closest, over-full
// lists re-selected the same way (prune-connections, graph.clj:208-232); SYMMETRIC = a dropped edge leaves both lists
// (:226-231); EXTEND = extend-candidates? for the new node's own selection (:191-195).
static int insert_batches(hnswgpu_index *idx, HostGraph &g, int64_t done, int ef, hipStream_t st, int flags) {
    const bool seq = flags & HNSWGPU_BUILD_SEQUENTIAL, heur = flags & HNSWGPU_BUILD_HEURISTIC;
    const bool sym = heur && (flags & HNSWGPU_BUILD_SYMMETRIC), extend = flags & HNSWGPU_BUILD_EXTEND;
    const int64_t n = g.n;
    const int M0 = g.M0;
    const int64_t blocks = g.up_off[n];
    int maxlv = 1;
    for (int64_t i = done; i < n; i++) maxlv = std::max(maxlv, g.levels[i]);
    // scratch sizing; the batch actually used grows with the graph (see below)
    const int64_t maxB = seq ? 1 : std::max<int64_t>(1, std::min<int64_t>(16384, tune(HNSWGPU_TUNE_BUILD_BATCH, 16384)));
    const int kk = heur ? ef : M0;         // layer-0 candidates a search hands back: all of them for the heuristic
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * maxB * kk));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * maxB * kk));
    HG_TRY(idx->s_misc.ensure(sizeof(int32_t) * maxB * (2 + maxlv)));   // q_rows, q_levels, up_out_ids
    HG_TRY(idx->s_misc2.ensure(sizeof(float) * maxB * maxlv));           // up_out_dist
    int32_t *d_qrows = idx->s_misc.as<int32_t>();
    int32_t *d_qlev = d_qrows + maxB;
    int32_t *d_upids = d_qlev + maxB;
    // the new nodes' own selections (heuristic mode): [maxB][M0] ids, distances, [maxB] counts, [maxB + 1] offsets
    OwnedBuf d_sel;
    int32_t *d_selid = nullptr, *d_selcnt = nullptr;
    float *d_seld = nullptr;
    int64_t *d_seloff = nullptr;
    if (heur) {
        HG_TRY(d_sel.ensure((sizeof(int32_t) + sizeof(float)) * maxB * M0 + sizeof(int32_t) * maxB + sizeof(int64_t) * (maxB + 1) + 64));
        d_seloff = d_sel.as<int64_t>();
        d_selid = reinterpret_cast<int32_t *>(d_seloff + maxB + 1);
        d_seld = reinterpret_cast<float *>(d_selid + maxB * M0);
        d_selcnt = reinterpret_cast<int32_t *>(d_seld + maxB * M0);
        std::vector<int64_t> h_off(maxB + 1);
        for (int64_t b = 0; b <= maxB; b++) h_off[b] = b * kk;
        HG_HIP(hipMemcpyAsync(d_seloff, h_off.data(), sizeof(int64_t) * (maxB + 1), hipMemcpyHostToDevice, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    std::vector<int32_t> h_ids(maxB * M0), h_up(maxB * maxlv), h_qrows(maxB), h_qlev(maxB);
    std::vector<float> h_d(maxB * M0), h_upd(maxB * maxlv);
    g.flag0.assign(n, 0);
    g.flagu.assign(std::max<int64_t>(blocks, 1), 0);
    // linker threads: at most 16 (a GPU box's CPU share per GPU), HNSWGPU_BUILD_THREADS overrides (1 = sequential)
    int nthreads = static_cast<int>(std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
    if (const int64_t e = tune(HNSWGPU_TUNE_BUILD_THREADS, 0)) nthreads = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>(64, e)));
    if (n - done < 256 || seq) nthreads = 1;  // a handful of rows: no pool to start
    LinkPool pool(nthreads);
    std::vector<HostGraph::Dirty> dirties(pool.size());
    std::vector<std::vector<PendingEdge>> pendings(pool.size());
    std::vector<PendingEdge> pend;
    Pruner pruner(idx, g, st, sym, seq);
    PinnedBuf pin;
    const bool timing = tune(HNSWGPU_TUNE_BUILD_TIMING, 0) != 0;  // developer switch: where a build spends its time
    double t_gpu = 0.0, t_link = 0.0, t_up = 0.0, t_prune = 0.0;
    int64_t nbatch = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    while (done < n) {
        const double tb0 = timing ? now() : 0.0;
        // a batch never exceeds 1/8 of the graph it is searched against; beyond 2048 it grows slowly (1/64)
        // so that big indexes amortise the per-batch round trip without starving early quality
        int64_t B = std::min<int64_t>({maxB, std::max<int64_t>(1, done / 8), 2048 + done / 64, n - done});
        HG_TRY(upload_dirty(idx, g, st, pin, pool));
        if (timing) t_up += now() - tb0;
        for (int64_t b = 0; b < B; b++) {
            h_qrows[b] = static_cast<int32_t>(done + b);
            h_qlev[b] = g.levels[done + b];
        }
        HG_HIP(hipMemcpyAsync(d_qrows, h_qrows.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(d_qlev, h_qlev.data(), sizeof(int32_t) * B, hipMemcpyHostToDevice, st));
        HnswArgs a;
        fill_args(idx, a);
        a.entry = g.entry;
        a.max_level = g.top;
        a.Q = idx->d_base;
        a.qld = idx->ld;
        a.q_rows = d_qrows;
        a.q_levels = d_qlev;
        a.ref_start = seq ? 1 : 0;
        a.nq = static_cast<int32_t>(B);
        a.ef = ef;
        a.k = kk;
        a.cap = ef + kGhost;
        a.out_ids = idx->s_ids.as<int32_t>();
        a.out_dist = idx->s_outd.as<float>();
        a.stats = nullptr;
        a.up_out_ids = d_upids;
        a.up_out_dist = idx->s_misc2.as<float>();
        a.up_stride = maxlv;
        HG_TRY(launch_hnsw_idx(idx, a, st));
        int32_t *h_cnt = nullptr;
        std::vector<int32_t> h_selcnt;
        if (heur) {  // the new nodes' own links: get-neighbors-heuristic over the search's ef candidates
            HeurArgs ha;
            memset(&ha, 0, sizeof(ha));
            ha.ntasks = static_cast<int32_t>(B);
            ha.off = d_seloff;
            ha.cand_id = a.out_ids;
            ha.cand_d = a.out_dist;
            ha.m = nullptr;
            ha.m_all = M0;
            ha.extend = extend ? 1 : 0;
            ha.out_stride = M0;
            ha.out_id = d_selid;
            ha.out_d = d_seld;
            ha.out_cnt = d_selcnt;
            HG_TRY(launch_heur(idx, ha, st));
            h_selcnt.resize(B);
            HG_HIP(hipMemcpyAsync(h_ids.data(), d_selid, sizeof(int32_t) * B * M0, hipMemcpyDeviceToHost, st));
            HG_HIP(hipMemcpyAsync(h_d.data(), d_seld, sizeof(float) * B * M0, hipMemcpyDeviceToHost, st));
            HG_HIP(hipMemcpyAsync(h_selcnt.data(), d_selcnt, sizeof(int32_t) * B, hipMemcpyDeviceToHost, st));
            h_cnt = h_selcnt.data();
        } else {
            HG_HIP(hipMemcpyAsync(h_ids.data(), a.out_ids, sizeof(int32_t) * B * M0, hipMemcpyDeviceToHost, st));
            HG_HIP(hipMemcpyAsync(h_d.data(), a.out_dist, sizeof(float) * B * M0, hipMemcpyDeviceToHost, st));
        }
        HG_HIP(hipMemcpyAsync(h_up.data(), a.up_out_ids, sizeof(int32_t) * B * maxlv, hipMemcpyDeviceToHost, st));
        HG_HIP(hipMemcpyAsync(h_upd.data(), a.up_out_dist, sizeof(float) * B * maxlv, hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
        const double tb1 = timing ? now() : 0.0;
        const int top_at_start = g.top;
        const int take = std::min(M0, ef);
        // (A) own lists of node done + b; (B) reverse edges into the lists of nodes owned by thread `me` (of `nt`)
        auto link_own = [&](int64_t b, HostGraph::Dirty &dirty) {
            const int32_t id = static_cast<int32_t>(done + b);
            const int L = g.levels[id];
            for (int lc = std::min(L, top_at_start); lc >= 1; lc--) {
                const int32_t nb = h_up[b * maxlv + (lc - 1)];
                if (nb < 0 || nb == id) continue;
                g.add_edge(id, nb, lc, h_upd[b * maxlv + (lc - 1)], dirty);
            }
            const int mine = h_cnt ? h_cnt[b] : take;
            for (int t = 0; t < mine; t++) {
                const int32_t nb = h_ids[b * M0 + t];
                if (nb < 0) break;
                if (nb == id) continue;
                g.add_edge(id, nb, 0, h_d[b * M0 + t], dirty);
            }
        };
        // the batch's reverse edges are numbered in the sequential loop's order: node by node, layers top down,
        // a layer's links in selection order
        auto link_reverse = [&](int64_t b, int me, int nt, HostGraph::Dirty &dirty, std::vector<PendingEdge> &pq) {
            const int32_t id = static_cast<int32_t>(done + b);
            const int L = g.levels[id];
            int64_t sq = b * (maxlv + M0);
            auto rev = [&](int32_t nb, int lc, float dist) {
                if (heur) {
                    if (!g.append_edge(nb, id, lc, dist, dirty)) pq.push_back({nb, lc, id, dist, sq});
                } else {
                    g.add_edge(nb, id, lc, dist, dirty);
                }
            };
            for (int lc = std::min(L, top_at_start); lc >= 1; lc--, sq++) {
                const int32_t nb = h_up[b * maxlv + (lc - 1)];
                if (nb < 0 || nb == id || nb % nt != me) continue;
                rev(nb, lc, h_upd[b * maxlv + (lc - 1)]);
            }
            sq = b * (maxlv + M0) + maxlv;
            const int mine = h_cnt ? h_cnt[b] : take;
            for (int t = 0; t < mine; t++, sq++) {
                const int32_t nb = h_ids[b * M0 + t];
                if (nb < 0) break;
                if (nb == id || nb % nt != me) continue;
                rev(nb, 0, h_d[b * M0 + t]);
            }
        };
        const int nt = B >= 256 ? pool.size() : 1;  // small batches: the hand-over costs more than it saves
        if (nt == 1) {
            for (int64_t b = 0; b < B; b++) link_own(b, dirties[0]);
            for (int64_t b = 0; b < B; b++) link_reverse(b, 0, 1, dirties[0], pendings[0]);
        } else {
            pool.run([&](int me) {
                const int64_t lo = B * me / nt, hi = B * (me + 1) / nt;
                for (int64_t b = lo; b < hi; b++) link_own(b, dirties[me]);
            });
            pool.run([&](int me) {
                for (int64_t b = 0; b < B; b++) link_reverse(b, me, nt, dirties[me], pendings[me]);
            });
        }
        const double tb2 = timing ? now() : 0.0;
        if (heur) {  // the over-full lists of this batch: prune-connections (graph.clj:208-232) on the device
            pend.clear();
            for (auto &pq : pendings) {
                pend.insert(pend.end(), pq.begin(), pq.end());
                pq.clear();
            }
            std::sort(pend.begin(), pend.end(), [](const PendingEdge &x, const PendingEdge &y) { return x.seq < y.seq; });
            if (seq) HG_TRY(pruner.run(pend, dirties[0]));
            else HG_TRY(pruner.run_batched(pend, pool, dirties));
        }
        for (auto &dd : dirties) {
            g.dirty0.insert(g.dirty0.end(), dd.d0.begin(), dd.d0.end());
            g.dirtyu.insert(g.dirtyu.end(), dd.du.begin(), dd.du.end());
            dd.d0.clear();
            dd.du.clear();
        }
        for (int64_t b = 0; b < B; b++) {  // :271-273
            const int32_t id = static_cast<int32_t>(done + b);
            if (g.levels[id] > g.top) {
                g.entry = id;
                g.top = g.levels[id];
            }
        }
        done += B;
        if (timing) {
            t_gpu += tb1 - tb0;
            t_link += tb2 - tb1;
            t_prune += now() - tb2;
            nbatch++;
        }
    }
    if (timing)
        fprintf(stderr, "hnsw build: %lld batches, %.2f s upload + search + download (%.2f s of it packing and uploading "
                        "changed adjacency rows), %.2f s host linking, %.2f s re-selection of %lld over-full lists (%lld edges "
                        "removed from the other side)\n",
                static_cast<long long>(nbatch), t_gpu, t_up, t_link, t_prune, static_cast<long long>(pruner.n_tasks),
                static_cast<long long>(pruner.n_removed));
    return 0;
}

// the finished host graph -> device adjacency + the host mirrors of the handle
static int publish_graph(hnswgpu_index *idx, HostGraph &g, hipStream_t st) {
    std::vector<int32_t> tmp0, tmpu;
    const int64_t blocks = g.up_off[g.n];
    HG_TRY(upload_graph(idx, g, st, tmp0, tmpu));
    idx->h_levels = g.levels;
    idx->h_upoff = g.up_off;
    idx->h_l0.swap(tmp0);
    tmpu.resize(static_cast<size_t>(blocks) * g.M);
    idx->h_upadj.swap(tmpu);
    idx->entry = g.entry;
    idx->max_level = std::max(g.top, 0);
    idx->has_graph = true;
    return 0;
}

// levels of rows [from, to): floor(ml * -ln U), ml = 1/ln 2 (ultra_fast.clj:133,143-147), U the row's draw of
// java.util.Random(seed) -- row i takes the i-th nextDouble, so rows added later continue the build's sequence
static void draw_levels(int64_t seed, int64_t from, int64_t to, std::vector<int32_t> &levels, std::vector<int64_t> &up_off) {
    JavaRandom rng(seed);
    const double ml = 1.0 / log(2.0);
    for (int64_t i = 0; i < to; i++) {
        const double u = rng.next_double();
        if (i < from) continue;
        const int lv = u > 0.0 ? static_cast<int>(ml * (-log(u))) : 30;
        levels[i] = std::min(lv, 30);
        up_off[i + 1] = up_off[i] + levels[i];
    }
}

}  // namespace hg

extern "C" {

// build-index / insert-batch (ultra_fast.clj:303-344) as batched insertion (insert_batches above).
// Level draw: floor(ml * -ln U), ml = 1/ln 2 (:133,:143-147), U from java.util.Random(seed).
// flags = 0, the fast default -- differences from the reference's sequential insert-single, stated in DESIGN.md: nodes
// of one batch do not see each other (batches grow from 1 to at most 1/8 of the graph, capped); the walk starts at the
// top layer with ef = 1 instead of at min(level, entry-level) (:247-248); each node links to its m CLOSEST candidates
// (the reference's `(take m candidates)` takes PriorityQueue array order, which is unspecified).
// HNSWGPU_BUILD_SEQUENTIAL removes the first two (insert-single itself: the graph equals oracle.c's orc_hnsw_build_ex
// edge for edge); HNSWGPU_BUILD_HEURISTIC / _SYMMETRIC / _EXTEND select links as src/hnsw/graph.clj:162-232 does.
int hnswgpu_hnsw_build(hnswgpu_index *idx, int32_t M, int32_t ef_construction, int64_t seed) {
    return hnswgpu_hnsw_build_ex(idx, M, ef_construction, seed, 0);
}

int hnswgpu_hnsw_build_ex(hnswgpu_index *idx, int32_t M, int32_t ef_construction, int64_t seed, int32_t flags) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE((flags & ~(HNSWGPU_BUILD_SEQUENTIAL | HNSWGPU_BUILD_HEURISTIC | HNSWGPU_BUILD_SYMMETRIC | HNSWGPU_BUILD_EXTEND)) == 0,
               HNSWGPU_EINVAL, "unknown build flags 0x%x", flags);
    HG_REQUIRE(!(flags & (HNSWGPU_BUILD_SYMMETRIC | HNSWGPU_BUILD_EXTEND)) || (flags & HNSWGPU_BUILD_HEURISTIC), HNSWGPU_EINVAL,
               "HNSWGPU_BUILD_SYMMETRIC / _EXTEND qualify HNSWGPU_BUILD_HEURISTIC");
    HG_REQUIRE(M >= 1 && 2 * M <= kMaxDeg, HNSWGPU_ELIMIT, "need 1 <= M <= %d", kMaxDeg / 2);
    HG_REQUIRE(ef_construction >= 1 && ef_construction <= 4096, HNSWGPU_ELIMIT, "need 1 <= ef_construction <= 4096");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    const int64_t n = idx->n;
    const int M0 = 2 * M;
    HostGraph g;
    g.n = n;
    g.M = M;
    g.M0 = M0;
    g.levels.resize(n);
    g.up_off.assign(n + 1, 0);
    draw_levels(seed, 0, n, g.levels, g.up_off);
    const int64_t blocks = n > 0 ? g.up_off[n] : 0;
    HG_TRY(quiesce(idx, st));  // as in hnswgpu_set_graph
    free_graph(idx);
    HG_TRY(alloc_graph(idx, M, M0, blocks));
    HG_TRY(ensure_qrows(idx, st));
    if (n == 0) {
        idx->h_levels.clear();
        idx->h_l0.clear();
        idx->h_upoff.assign(1, 0);
        idx->h_upadj.clear();
        idx->entry = -1;
        idx->max_level = 0;
        idx->has_graph = true;
        return 0;
    }
    g.l0.assign(static_cast<size_t>(n) * (M0 + 1), -1);
    g.l0_d.assign(static_cast<size_t>(n) * (M0 + 1), 0.f);
    g.l0_cnt.assign(n, 0);
    g.up.assign(static_cast<size_t>(std::max<int64_t>(blocks, 1)) * (M + 1), -1);
    g.up_d.assign(static_cast<size_t>(std::max<int64_t>(blocks, 1)) * (M + 1), 0.f);
    g.up_cnt.assign(std::max<int64_t>(blocks, 1), 0);
    g.sorted0.assign(n, 0);
    g.sortedu.assign(std::max<int64_t>(blocks, 1), 0);
    HG_HIP(hipMemcpyAsync(idx->d_levels, g.levels.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    HG_HIP(hipMemcpyAsync(idx->d_upoff, g.up_off.data(), sizeof(int64_t) * (n + 1), hipMemcpyHostToDevice, st));
    HG_HIP(hipMemsetAsync(idx->d_l0, 0xff, sizeof(int32_t) * n * M0, st));
    HG_HIP(hipMemsetAsync(idx->d_upadj, 0xff, sizeof(int32_t) * std::max<int64_t>(blocks, 1) * M, st));
    g.entry = 0;  // first element becomes the entry point (:229-231)
    g.top = g.levels[0];
    HG_TRY(insert_batches(idx, g, 1, ef_construction, st, flags));
    HG_TRY(publish_graph(idx, g, st));
    idx->build_flags = flags;  // (only a graph that stands carries its builder: hnswgpu_hnsw_add inserts the same way)
    return 0;
}

// insert-single on a LIVE index (ultra_fast.clj:216-275, reached by add-vector! src/hnsw/api.clj:30-33 and add!
// src/hnsw/api/simple.clj:31-42): `m` more rows join the base matrix and the graph that is installed.  The new rows'
// levels continue the seeded java.util.Random sequence (row i takes its i-th draw), they are inserted in batches by the
// kernels and the linker of hnswgpu_hnsw_build against the CURRENT graph -- whose edge distances, which the reference's
// pruning sorts by (:279-299), are recomputed on the device by the traversal's own arithmetic -- and the device
// adjacency, the int8 rows and the host mirrors grow with them.  Row ids of the new rows: n, n + 1, ...
int hnswgpu_hnsw_add(hnswgpu_index *idx, const float *rows, int64_t m, int32_t ef_construction, int64_t seed) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(m >= 0, HNSWGPU_EINVAL, "m must be >= 0");
    if (m == 0) return 0;
    HG_REQUIRE(rows, HNSWGPU_EINVAL, "rows is null");
    HG_REQUIRE(ef_construction >= 1 && ef_construction <= 4096, HNSWGPU_ELIMIT, "need 1 <= ef_construction <= 4096");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_REQUIRE(idx->has_graph, HNSWGPU_ESTATE, "index has no graph (call hnswgpu_hnsw_build / hnswgpu_set_graph first)");
    HG_REQUIRE(idx->nlist == 0, HNSWGPU_ESTATE,
               "the index holds IVF lists over its present rows: add rows first, then build / install the lists");
    HG_REQUIRE(idx->M0 == 2 * idx->M, HNSWGPU_ESTATE, "hnswgpu_hnsw_add needs a graph with M0 = 2 M (as hnswgpu_hnsw_build makes)");
    const int64_t n0 = idx->n, n1 = n0 + m;
    HG_REQUIRE(n1 < 2147483647LL, HNSWGPU_ELIMIT, "the index must hold fewer than 2^31 rows");
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(quiesce(idx, st));
    const int M = idx->M, M0 = idx->M0;
    const int64_t ld = idx->ld;
    // Failure-atomic: everything the call makes -- the grown base, norms, int8 rows, device graph -- is STAGED; the handle's
    // fields point at the staged arrays while the insertion searches run against them (idx->mu is held and the streams are
    // quiesced: no caller can see the intermediate state) and are restored, the staged arrays freed, if any step fails: the
    // index then is exactly what it was (n, graph, host mirrors).  On success the old arrays are freed.
    // Cost: O(n) per call, not per row (the base is copied, every row re-quantised, the installed edges' distances
    // recomputed: ~1 ms per million 768-d rows each) -- add rows in batches.
    struct Staged {
        float *base = nullptr, *norms = nullptr;
        uint32_t *qrows = nullptr;
        float4 *qmeta = nullptr;
        int32_t *levels = nullptr, *l0 = nullptr, *upadj = nullptr;
        int64_t *upoff = nullptr;
    } nw, old;
    old.base = idx->d_base;
    old.norms = idx->d_norms;
    old.qrows = idx->d_qrows;
    old.qmeta = idx->d_qmeta;
    old.levels = idx->d_levels;
    old.l0 = idx->d_l0;
    old.upadj = idx->d_upadj;
    old.upoff = idx->d_upoff;
    const int64_t old_blocks = idx->up_blocks;
    bool installed = false;
    auto free_staged = [](Staged &sg) {
        void *ptrs[] = {sg.base, sg.norms, sg.qrows, sg.qmeta, sg.levels, sg.l0, sg.upadj, sg.upoff};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        sg = Staged();
    };
    HostGraph g;
    const int rc = [&]() -> int {
        // ---- 1. the base matrix, its norms and int8 rows grow (staged)
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.base), sizeof(float) * static_cast<size_t>(n1) * ld));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.norms), sizeof(float) * static_cast<size_t>(n1)));
        if (n0 > 0) {
            HG_HIP(hipMemcpyAsync(nw.base, idx->d_base, sizeof(float) * static_cast<size_t>(n0) * ld, hipMemcpyDeviceToDevice, st));
            HG_HIP(hipMemcpyAsync(nw.norms, idx->d_norms, sizeof(float) * static_cast<size_t>(n0), hipMemcpyDeviceToDevice, st));
        }
        if (ld != idx->dim) HG_HIP(hipMemsetAsync(nw.base + n0 * ld, 0, sizeof(float) * static_cast<size_t>(m) * ld, st));
        HG_HIP(hipMemcpy2DAsync(nw.base + n0 * ld, sizeof(float) * ld, rows, sizeof(float) * idx->dim, sizeof(float) * idx->dim,
                                static_cast<size_t>(m), hipMemcpyHostToDevice, st));
        HG_TRY(launch_norms(idx->nch, nw.base + n0 * ld, ld, m, nw.norms + n0, st));
        if (old.qrows) HG_TRY(quantize_rows(idx, nw.base, n1, &nw.qrows, &nw.qmeta, st));  // (all rows again: one pass over the base)
        // ---- 2. the host graph: the installed adjacency + its edge distances, then room for the new rows
        g.n = n1;
        g.M = M;
        g.M0 = M0;
        g.levels = idx->h_levels;
        g.levels.resize(n1, 0);
        g.up_off = idx->h_upoff;
        g.up_off.resize(n1 + 1, 0);
        draw_levels(seed, n0, n1, g.levels, g.up_off);
        const int64_t blocks0 = idx->h_upoff[n0], blocks1 = g.up_off[n1];
        g.l0.assign(static_cast<size_t>(n1) * (M0 + 1), -1);
        g.l0_d.assign(static_cast<size_t>(n1) * (M0 + 1), 0.f);
        g.l0_cnt.assign(n1, 0);
        g.up.assign(static_cast<size_t>(std::max<int64_t>(blocks1, 1)) * (M + 1), -1);
        g.up_d.assign(static_cast<size_t>(std::max<int64_t>(blocks1, 1)) * (M + 1), 0.f);
        g.up_cnt.assign(std::max<int64_t>(blocks1, 1), 0);
        g.sorted0.assign(n1, 0);
        g.sortedu.assign(std::max<int64_t>(blocks1, 1), 0);
        {
            // edge distances of the installed graph, by the traversal's arithmetic (rows [0, n0) of the old base: still in place)
            std::vector<float> d0(static_cast<size_t>(n0) * M0), du(static_cast<size_t>(std::max<int64_t>(blocks0, 1)) * M);
            std::vector<int32_t> owner(static_cast<size_t>(std::max<int64_t>(blocks0, 1)));
            for (int64_t i = 0; i < n0; i++)
                for (int64_t b = idx->h_upoff[i]; b < idx->h_upoff[i + 1]; b++) owner[b] = static_cast<int32_t>(i);
            HG_TRY(idx->s_outd.ensure(sizeof(float) * (d0.size() + du.size())));
            HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * owner.size()));
            float *dd0 = idx->s_outd.as<float>(), *ddu = dd0 + d0.size();
            HG_HIP(hipMemcpyAsync(idx->s_ids.p, owner.data(), sizeof(int32_t) * owner.size(), hipMemcpyHostToDevice, st));
            HG_TRY(launch_edge_dist(idx, nullptr, idx->d_l0, M0, n0, dd0, st));
            HG_TRY(launch_edge_dist(idx, idx->s_ids.as<int32_t>(), idx->d_upadj, M, blocks0, ddu, st));
            HG_HIP(hipMemcpyAsync(d0.data(), dd0, sizeof(float) * d0.size(), hipMemcpyDeviceToHost, st));
            if (blocks0 > 0) HG_HIP(hipMemcpyAsync(du.data(), ddu, sizeof(float) * static_cast<size_t>(blocks0) * M, hipMemcpyDeviceToHost, st));
            HG_HIP(hipStreamSynchronize(st));
            // a full list whose distances ascend IS the pruned order (prune-connections-ultra's stable sort is the identity on
            // it); any other list is still in insertion order and gets its first sort when it overflows
            auto adopt = [](const int32_t *src, const float *dsrc, int width, int32_t *dst, float *ddst, int32_t &cnt, uint8_t &srt) {
                int c = 0;
                while (c < width && src[c] >= 0) c++;
                bool asc = true;
                for (int j = 0; j < c; j++) {
                    dst[j] = src[j];
                    ddst[j] = dsrc[j];
                    if (j > 0 && dsrc[j] < dsrc[j - 1]) asc = false;
                }
                cnt = c;
                srt = (c == width && asc) ? 1 : 0;
            };
            for (int64_t i = 0; i < n0; i++)
                adopt(&idx->h_l0[static_cast<size_t>(i) * M0], &d0[static_cast<size_t>(i) * M0], M0, &g.l0[static_cast<size_t>(i) * (M0 + 1)],
                      &g.l0_d[static_cast<size_t>(i) * (M0 + 1)], g.l0_cnt[i], g.sorted0[i]);
            for (int64_t b = 0; b < blocks0; b++)
                adopt(&idx->h_upadj[static_cast<size_t>(b) * M], &du[static_cast<size_t>(b) * M], M, &g.up[static_cast<size_t>(b) * (M + 1)],
                      &g.up_d[static_cast<size_t>(b) * (M + 1)], g.up_cnt[b], g.sortedu[b]);
        }
        g.entry = idx->entry;
        g.top = idx->max_level;
        // ---- 3. the device graph grows: old rows copied, new rows empty (staged)
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.levels), sizeof(int32_t) * n1));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.l0), sizeof(int32_t) * n1 * M0));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.upoff), sizeof(int64_t) * (n1 + 1)));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&nw.upadj), sizeof(int32_t) * std::max<int64_t>(blocks1, 1) * M));
        HG_HIP(hipMemcpyAsync(nw.levels, g.levels.data(), sizeof(int32_t) * n1, hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(nw.upoff, g.up_off.data(), sizeof(int64_t) * (n1 + 1), hipMemcpyHostToDevice, st));
        HG_HIP(hipMemsetAsync(nw.l0, 0xff, sizeof(int32_t) * n1 * M0, st));
        HG_HIP(hipMemsetAsync(nw.upadj, 0xff, sizeof(int32_t) * std::max<int64_t>(blocks1, 1) * M, st));
        if (n0 > 0) HG_HIP(hipMemcpyAsync(nw.l0, idx->d_l0, sizeof(int32_t) * n0 * M0, hipMemcpyDeviceToDevice, st));
        if (blocks0 > 0) HG_HIP(hipMemcpyAsync(nw.upadj, idx->d_upadj, sizeof(int32_t) * blocks0 * M, hipMemcpyDeviceToDevice, st));
        HG_HIP(hipStreamSynchronize(st));
        // ---- 4. the insertion searches run against the staged arrays
        idx->d_base = nw.base;
        idx->d_norms = nw.norms;
        idx->d_qrows = nw.qrows;
        idx->d_qmeta = nw.qmeta;
        idx->d_levels = nw.levels;
        idx->d_l0 = nw.l0;
        idx->d_upoff = nw.upoff;
        idx->d_upadj = nw.upadj;
        idx->up_blocks = blocks1;
        idx->n = n1;
        idx->graph_gen++;
        installed = true;
        if (n0 == 0) {  // an empty index with an (empty) graph: the first new row becomes the entry point (:229-231)
            g.entry = 0;
            g.top = g.levels[0];
            HG_TRY(insert_batches(idx, g, 1, ef_construction, st, idx->build_flags & ~HNSWGPU_BUILD_SEQUENTIAL));
        } else {
            HG_TRY(insert_batches(idx, g, n0, ef_construction, st, idx->build_flags & ~HNSWGPU_BUILD_SEQUENTIAL));
        }
        return publish_graph(idx, g, st);  // (host mirrors, entry, max_level: assigned only once the uploads succeeded)
    }();
    if (rc != 0) {
        (void)hipStreamSynchronize(st);
        if (installed) {
            idx->d_base = old.base;
            idx->d_norms = old.norms;
            idx->d_qrows = old.qrows;
            idx->d_qmeta = old.qmeta;
            idx->d_levels = old.levels;
            idx->d_l0 = old.l0;
            idx->d_upoff = old.upoff;
            idx->d_upadj = old.upadj;
            idx->up_blocks = old_blocks;
            idx->n = n0;
            idx->graph_gen++;
        }
        free_staged(nw);
        return rc;
    }
    free_staged(old);
    return 0;
}

}  // extern "C"
