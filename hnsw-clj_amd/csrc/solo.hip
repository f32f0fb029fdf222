// solo.hip -- launcher of the small-launch HNSW traversal that spreads ONE query over several CUs (solo_kernels.hpp).
// Reference: search-layer-ultra / search-knn, src/hnsw/ultra_fast.clj:151-212, 346-374 (the same traversal, same results).
#include "engine.hpp"
#include "solo_kernels.hpp"

namespace hg {

// helper workgroups per query: HNSWGPU_TUNE_PREFETCH (0 = the solo path is off as well), default 8
static int solo_groups() {
    const int64_t v = tune(HNSWGPU_TUNE_PREFETCH, 8);
    return v < 0 ? 0 : (v > 31 ? 31 : static_cast<int>(v));
}

// HNSWGPU_TUNE_SOLO: 1 (default) = launches whose list is long enough for the split to pay (ef >= kSoloMinEf: a search of a few
// dozen expansions is a descent whose every step waits for the step before it, and the round-2 helpers serve it as well --
// 31k x 768 clustered, one query, round-2 helpers / this kernel: ef 50 249 / 255 us, ef 100 378 / 350, ef 200 612 / 524, ef 640
// 1,900 / 1,210), 2 = every small launch, 0 = never
constexpr int kSoloMinEf = 96;
bool solo_enabled(int ef) {
    const int64_t m = tune(HNSWGPU_TUNE_SOLO, 1);
    return solo_groups() > 0 && (m >= 2 || (m == 1 && ef >= kSoloMinEf));
}

int launch_hnsw_solo(hnswgpu_index *idx, HnswArgs a, hipStream_t st) {
    const int nch = idx->nch;
    a.qrows = nullptr;  // the owner's own int8 pass stays off: for a handful of queries it costs what it saves
    a.pf_groups = std::max(1, std::min(solo_groups(), idx->cus / std::max(a.nq, 1) - 1));
    // the window the fetchers keep evaluated ahead of the sequencer: 8 entries for short searches (ef 100: 350 us against 382
    // with 16), 16 from ef 256 (ef 640: 1.21 ms against 1.25 with 8, 1.28 with 32)
    a.pf_hints = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(32, tune(HNSWGPU_TUNE_PF_HINTS, a.ef < 256 ? 8 : 16))));
    a.solo_chase = tune(HNSWGPU_TUNE_SOLO_CHASE, 1) != 0 ? 1 : 0;
    const int grid = 8 * ((a.nq + 7) / 8) * (1 + a.pf_groups);
    // slots per query of the node-keyed tables: one per row while that stays small (no collisions), else ~32 per list
    // entry (a search claims ~1.5 ef nodes; a record overwritten before the owner has used it costs the owner a gather),
    // and the whole region within 512 MB
    int log2s = static_cast<int>(tune(HNSWGPU_TUNE_SOLO_SLOTS, 0));
    if (log2s <= 0) {
        const int64_t want = std::min<int64_t>(a.n, 32LL * a.ef);
        log2s = 11;
        while ((1LL << log2s) < want && log2s < 18) log2s++;
        while (log2s > 11 && (static_cast<int64_t>(a.nq) << log2s) * a.M0 * 8 > (512LL << 20)) log2s--;
    }
    log2s = std::max(8, std::min(log2s, 18));
    a.solo_log2s = log2s;
    const size_t S = static_cast<size_t>(1) << log2s;
    // (the mailboxes at a fixed place whatever the batch: a word there never holds an older launch's table entry)
    const size_t mail_bytes = (sizeof(uint32_t) * kSoloMailWords * kSoloMaxQueries + 255) & ~static_cast<size_t>(255);
    const size_t claim_bytes = (sizeof(uint32_t) * S * a.nq + 255) & ~static_cast<size_t>(255);
    const size_t rec_bytes = sizeof(unsigned long long) * S * a.M0 * a.nq;
    const size_t region = mail_bytes + claim_bytes + rec_bytes;
    // four regions in rotation, so that launches in flight (two Slots) never share one.  Growing the buffer frees the old
    // one: every stream that may still run a launch on it is waited for first.  Fresh memory is zeroed (tag 0 = nothing).
    if (idx->s_solo.cap < 4 * region) {
        for (auto &sl : idx->slots)
            if (sl.st) HG_HIP(hipStreamSynchronize(sl.st));
        if (idx->stream) HG_HIP(hipStreamSynchronize(idx->stream));
        HG_HIP(hipStreamSynchronize(st));
        HG_TRY(idx->s_solo.ensure(4 * region));
        HG_HIP(hipMemsetAsync(idx->s_solo.p, 0, idx->s_solo.cap, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    idx->pf_seq = (idx->pf_seq + 1) & 0xffffff;
    if ((idx->pf_seq & 0x3fff) == 0) {
        // the 14 bits of the launch number inside a record's tag start over: no word of the previous cycle may survive
        for (auto &sl : idx->slots)
            if (sl.st) HG_HIP(hipStreamSynchronize(sl.st));
        HG_HIP(hipMemsetAsync(idx->s_solo.p, 0, idx->s_solo.cap, st));
        if (idx->s_pf.p) HG_HIP(hipMemsetAsync(idx->s_pf.p, 0, idx->s_pf.cap, st));
        HG_HIP(hipStreamSynchronize(st));
        idx->pf_seq = (idx->pf_seq + 1) & 0xffffff;
        if (idx->pf_seq == 0) idx->pf_seq = 1;
    }
    a.pf_seq = idx->pf_seq;
    char *reg = static_cast<char *>(idx->s_solo.p) + (idx->pf_seq & 3) * (idx->s_solo.cap / 4 & ~static_cast<size_t>(255));
    a.pf_mail = reinterpret_cast<uint32_t *>(reg);
    a.solo_claim = reinterpret_cast<uint32_t *>(reg + mail_bytes);
    a.solo_rec = reinterpret_cast<unsigned long long *>(reg + mail_bytes + claim_bytes);
    const size_t lds = solo_lds_bytes(a.cap, a.nwords);
    HG_REQUIRE(lds <= 160 * 1024, HNSWGPU_ELIMIT, "HNSW search state (%zu B: ef=%d, n=%lld) exceeds the 160 KiB LDS of a CU", lds,
               a.ef, (long long)a.n);
    const bool l2 = a.metric == METRIC_L2;
#define CALL_S(N, R, RH, L)                                                                                     \
    do {                                                                                                        \
        if (lds > 48 * 1024)                                                                                    \
            HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&hnsw_solo_kernel<N, R, RH, L>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));     \
        hipLaunchKernelGGL((hnsw_solo_kernel<N, R, RH, L>), dim3(grid), dim3(kWG), lds, st, a);                 \
    } while (0)
#define CALL_SL(N, R, RH)             \
    do {                              \
        if (l2) CALL_S(N, R, RH, true); \
        else CALL_S(N, R, RH, false);   \
    } while (0)
    switch (nch) {
        case 1: CALL_SL(1, 8, 4); break;
        case 2: CALL_SL(2, 8, 4); break;
        case 3: CALL_SL(3, 8, 4); break;
        case 4: CALL_SL(4, 4, 4); break;
        case 6: CALL_SL(6, 4, 4); break;
        case 8: CALL_SL(8, 2, 2); break;
        case 12: CALL_SL(12, 2, 2); break;
        default: set_error("unsupported row length"); return HNSWGPU_ELIMIT;
    }
#undef CALL_SL
#undef CALL_S
    HG_HIP(hipGetLastError());
    count_launch(HNSWGPU_COUNT_HNSW_SOLO);
    return 0;
}

}  // namespace hg
