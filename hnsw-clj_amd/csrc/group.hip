// group.hip -- ONE index over several GPUs behind the C ABI (include/hnswgpu.h: hnswgpu_group_*).
//
// The reference shards inside one process: search-partitioned scatters a query to the partitions, takes a top-k per
// partition, concatenates, sorts and takes k (src/hnsw/ann/partition/partitioned_hnsw.clj:149-196).  A JVM host that
// binds this library gets the same shape from one call: a group owns one engine handle per device,
//   * IVF: centroids replicated, WHOLE inverted lists dealt to the devices balanced by row count (the rule of
//     hnsw-clj_amd/sharded.py: deal_lists), every device routes the batch identically and scans the probed lists it holds
//     (hnswgpu_set_ivf_shard / hnswgpu_ivf_search_shard_dev: results labelled with their position in the candidate stream
//     of the WHOLE index), the per-device top-k moved to the first device by peer copies over xGMI (nq * k * 12 bytes per
//     device) and merged by (distance, that position): bit for bit the unsharded answer;
//   * HNSW: contiguous row ranges, one independent sub-graph per device (= PartitionedHNSWIndex, :23-27), each searched
//     with the full k, merged by distance with ties to the lower device (the reference's stable sort).
// Every device works on its own stream; nothing on the data path synchronises the devices with each other except the
// events the merge waits for.  `devices` may name one GPU several times: N handles on one card, which is how the parity
// tests run this path on a one-GPU box.  (hnsw-clj_amd/sharded.py keeps the one-process-per-GPU variant over RCCL.)
#include <string.h>

#include <algorithm>
#include <memory>
#include <numeric>
#include <vector>

#include "engine.hpp"

using namespace hg;

namespace hg {

__global__ void remap_ids_kernel(int32_t *ids, int64_t cnt, const int32_t *map, int32_t offset) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int32_t l = ids[i];
    if (l >= 0) ids[i] = map ? map[l] : l + offset;
}

struct GroupBuf {  // device scratch of one member / of the merge, grown on demand
    void *p = nullptr;
    size_t cap = 0;
    int device = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        HG_HIP(hipSetDevice(device));
        if (p) HG_HIP(hipFree(p));
        p = nullptr;
        cap = 0;
        HG_HIP(hipMalloc(&p, bytes + bytes / 4 + 256));
        cap = bytes + bytes / 4 + 256;
        return 0;
    }
    void release() {
        if (p) {
            (void)hipSetDevice(device);
            (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
    }
};

}  // namespace hg

struct hnswgpu_group {
    int32_t dim = 0, metric = 0;
    std::vector<int32_t> devices;
    std::mutex mu;
    struct Member {
        hnswgpu_index *idx = nullptr;
        hipStream_t st = nullptr;
        hipEvent_t ev = nullptr;
        int64_t n = 0, row0 = 0;       // rows held; first global row (HNSW: contiguous ranges)
        int32_t *d_map = nullptr;      // IVF: shard row -> global row
        GroupBuf q, ids, dist, ord;
    };
    std::vector<Member> m;
    int64_t n = 0;
    int kind = 0;  // 0 = empty, 1 = IVF lists dealt to the devices, 2 = HNSW sub-graphs
    int32_t nlist = 0;
    hipStream_t st0 = nullptr;  // merge stream on devices[0]
    GroupBuf g_ids, g_dist, g_ord, o_ids, o_dist;
    void *h_pin = nullptr;      // pinned staging of the results
    size_t h_pin_cap = 0;
};

static void group_clear_members(hnswgpu_group *g) {
    for (auto &mm : g->m) {
        if (mm.idx) (void)hnswgpu_destroy(mm.idx);
        mm.idx = nullptr;
        if (mm.d_map) {
            (void)hipSetDevice(g->devices[&mm - g->m.data()]);
            (void)hipFree(mm.d_map);
            mm.d_map = nullptr;
        }
        mm.n = mm.row0 = 0;
    }
    g->kind = 0;
    g->n = 0;
}

extern "C" {

int hnswgpu_group_create(const int32_t *devices, int32_t ndev, int32_t dim, int32_t metric, hnswgpu_group **out) {
    HG_REQUIRE(devices && out && ndev >= 1 && ndev <= 64, HNSWGPU_EINVAL, "need 1..64 devices");
    HG_REQUIRE(dim >= 1 && dim <= 3072, HNSWGPU_ELIMIT, "dim must be 1..3072");
    HG_REQUIRE(metric >= 0 && metric <= 2, HNSWGPU_EINVAL, "metric must be 0 (cosine), 1 (l2) or 2 (dot)");
    int count = 0;
    HG_HIP(hipGetDeviceCount(&count));
    for (int i = 0; i < ndev; i++) HG_REQUIRE(devices[i] >= 0 && devices[i] < count, HNSWGPU_EINVAL, "device %d does not exist", devices[i]);
    // (a failure below goes through hnswgpu_group_destroy: the streams and events made so far are released with the struct)
    struct Deleter {
        void operator()(hnswgpu_group *p) const { (void)hnswgpu_group_destroy(p); }
    };
    std::unique_ptr<hnswgpu_group, Deleter> g(new (std::nothrow) hnswgpu_group);
    HG_REQUIRE(g, HNSWGPU_ENOMEM, "host allocation failed");
    g->dim = dim;
    g->metric = metric;
    g->devices.assign(devices, devices + ndev);
    g->m.resize(ndev);
    for (int i = 0; i < ndev; i++) {
        HG_HIP(hipSetDevice(devices[i]));
        HG_HIP(hipStreamCreateWithFlags(&g->m[i].st, hipStreamNonBlocking));
        HG_HIP(hipEventCreateWithFlags(&g->m[i].ev, hipEventDisableTiming));
        g->m[i].q.device = g->m[i].ids.device = g->m[i].dist.device = g->m[i].ord.device = devices[i];
        // peers: the merge device reads the members' partial lists over xGMI
        if (devices[i] != devices[0]) {
            int can = 0;
            HG_HIP(hipSetDevice(devices[0]));
            (void)hipDeviceCanAccessPeer(&can, devices[0], devices[i]);
            if (can) {
                const hipError_t e = hipDeviceEnablePeerAccess(devices[i], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();  // copies still work, staged
            }
        }
    }
    HG_HIP(hipSetDevice(devices[0]));
    HG_HIP(hipStreamCreateWithFlags(&g->st0, hipStreamNonBlocking));
    g->g_ids.device = g->g_dist.device = g->g_ord.device = g->o_ids.device = g->o_dist.device = devices[0];
    *out = g.release();
    return 0;
}

int hnswgpu_group_destroy(hnswgpu_group *g) {
    if (!g) return 0;
    group_clear_members(g);
    for (size_t i = 0; i < g->m.size(); i++) {
        auto &mm = g->m[i];
        (void)hipSetDevice(g->devices[i]);
        if (mm.st) (void)hipStreamSynchronize(mm.st);
        mm.q.release();
        mm.ids.release();
        mm.dist.release();
        mm.ord.release();
        if (mm.ev) (void)hipEventDestroy(mm.ev);
        if (mm.st) (void)hipStreamDestroy(mm.st);
    }
    (void)hipSetDevice(g->devices[0]);
    if (g->st0) (void)hipStreamSynchronize(g->st0);
    g->g_ids.release();
    g->g_dist.release();
    g->g_ord.release();
    g->o_ids.release();
    g->o_dist.release();
    if (g->st0) (void)hipStreamDestroy(g->st0);
    if (g->h_pin) (void)hipHostFree(g->h_pin);
    delete g;
    return 0;
}

int hnswgpu_group_info(const hnswgpu_group *g, int32_t *ndev, int64_t *n, int32_t *kind, int64_t *rows_per_device) {
    HG_REQUIRE(g, HNSWGPU_EINVAL, "group is null");
    if (ndev) *ndev = static_cast<int32_t>(g->devices.size());
    if (n) *n = g->n;
    if (kind) *kind = g->kind;
    if (rows_per_device)
        for (size_t i = 0; i < g->m.size(); i++) rows_per_device[i] = g->m[i].n;
    return 0;
}

// IVF: the lists of ONE index dealt to the devices (whole lists, longest first onto the least loaded device; ties: the
// lower list index, the lower device -- hnsw-clj_amd/sharded.py: deal_lists), centroids replicated.
int hnswgpu_group_set_ivf(hnswgpu_group *g, const float *base, int64_t n, const float *centroids, int32_t nlist,
                          const int64_t *list_off, const int32_t *list_ids) {
    HG_REQUIRE(g && base && centroids && list_off && list_ids, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(n >= 1 && n < 2147483647LL && nlist >= 1, HNSWGPU_EINVAL, "need 1 <= n < 2^31 rows and nlist >= 1");
    HG_REQUIRE(list_off[0] == 0 && list_off[nlist] == n, HNSWGPU_EINVAL, "list_off must start at 0 and end at n");
    for (int l = 0; l < nlist; l++) HG_REQUIRE(list_off[l] <= list_off[l + 1], HNSWGPU_EINVAL, "list_off not monotone");
    for (int64_t i = 0; i < n; i++) HG_REQUIRE(list_ids[i] >= 0 && list_ids[i] < n, HNSWGPU_EINVAL, "list_ids[%lld] out of range", (long long)i);
    std::lock_guard<std::mutex> lk(g->mu);
    const int nd = static_cast<int>(g->devices.size());
    // (whatever fails in there: the group is left EMPTY, not with fresh members beside the old n / nlist / kind)
    auto deal = [&]() -> int {
        group_clear_members(g);
        g->n = 0;
        g->nlist = 0;
        g->kind = 0;
        std::vector<int64_t> glen(nlist);
        for (int l = 0; l < nlist; l++) glen[l] = list_off[l + 1] - list_off[l];
        std::vector<int32_t> order(nlist), owner(nlist);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return glen[a] > glen[b]; });
        std::vector<int64_t> load(nd, 0);
        for (int32_t l : order) {
            const int r = static_cast<int>(std::min_element(load.begin(), load.end()) - load.begin());  // first minimum
            owner[l] = r;
            load[r] += glen[l];
        }
        for (int r = 0; r < nd; r++) {
            auto &mm = g->m[r];
            // this device's rows: list by list, index order inside a list -- the shard then scans its base in place
            std::vector<int64_t> off(static_cast<size_t>(nlist) + 1, 0);
            std::vector<int32_t> map;
            map.reserve(static_cast<size_t>(load[r]));
            for (int l = 0; l < nlist; l++) {
                if (owner[l] == r)
                    for (int64_t i = list_off[l]; i < list_off[l + 1]; i++) map.push_back(list_ids[i]);
                off[l + 1] = static_cast<int64_t>(map.size());
            }
            const int64_t rn = static_cast<int64_t>(map.size());
            std::vector<float> rows(static_cast<size_t>(std::max<int64_t>(rn, 1)) * g->dim);
            for (int64_t i = 0; i < rn; i++)
                memcpy(rows.data() + static_cast<size_t>(i) * g->dim, base + static_cast<size_t>(map[i]) * g->dim, sizeof(float) * g->dim);
            std::vector<int32_t> ident(static_cast<size_t>(rn));
            std::iota(ident.begin(), ident.end(), 0);
            HG_TRY(hnswgpu_create(rows.data(), rn, g->dim, g->metric, g->devices[r], &mm.idx));
            HG_TRY(hnswgpu_set_ivf_shard(mm.idx, centroids, nlist, off.data(), ident.data(), glen.data()));
            HG_HIP(hipSetDevice(g->devices[r]));
            HG_HIP(hipMalloc(reinterpret_cast<void **>(&mm.d_map), sizeof(int32_t) * std::max<int64_t>(rn, 1)));
            if (rn > 0) HG_HIP(hipMemcpy(mm.d_map, map.data(), sizeof(int32_t) * rn, hipMemcpyHostToDevice));
            mm.n = rn;
        }
        // ONE verdict for the whole index (include/hnswgpu.h: hnswgpu_ivf_stream_state): off if any shard with rows says so
        int32_t off_any = 0;
        for (int r = 0; r < nd; r++) {
            int32_t off = 0;
            if (g->m[r].n > 0) HG_TRY(hnswgpu_ivf_stream_state(g->m[r].idx, &off));
            off_any |= off;
        }
        for (int r = 0; r < nd; r++) HG_TRY(hnswgpu_ivf_set_stream_state(g->m[r].idx, off_any));
        return 0;
    };
    int rc;
    try {
        rc = deal();
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed while dealing the lists");
        rc = HNSWGPU_ENOMEM;
    }
    if (rc != 0) {
        group_clear_members(g);
        return rc;
    }
    g->n = n;
    g->nlist = nlist;
    g->kind = 1;
    return 0;
}

// HNSW: contiguous row ranges, one sub-graph per device, built on the device (hnswgpu_hnsw_build; device r's levels
// are drawn from seed + r).
int hnswgpu_group_hnsw_build(hnswgpu_group *g, const float *base, int64_t n, int32_t M, int32_t ef_construction,
                             int64_t seed) {
    HG_REQUIRE(g && base, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(n >= 1 && n < 2147483647LL, HNSWGPU_EINVAL, "need 1 <= n < 2^31 rows");
    std::lock_guard<std::mutex> lk(g->mu);
    const int nd = static_cast<int>(g->devices.size());
    group_clear_members(g);
    g->n = 0;
    g->nlist = 0;
    g->kind = 0;
    auto build = [&]() -> int {
        for (int r = 0; r < nd; r++) {
            auto &mm = g->m[r];
            const int64_t r0 = n * r / nd, r1 = n * (r + 1) / nd;
            HG_TRY(hnswgpu_create(base + static_cast<size_t>(r0) * g->dim, r1 - r0, g->dim, g->metric, g->devices[r], &mm.idx));
            if (r1 > r0) HG_TRY(hnswgpu_hnsw_build(mm.idx, M, ef_construction, seed + r));
            mm.n = r1 - r0;
            mm.row0 = r0;
        }
        return 0;
    };
    if (const int rc = build(); rc != 0) {
        group_clear_members(g);  // (a failed build leaves the group empty)
        return rc;
    }
    g->n = n;
    g->kind = 2;
    return 0;
}

// per-device results -> devices[0] -> merged -> host
static int group_search_impl(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t param, int32_t *out_ids,
                        float *out_dist) {
    const int nd = static_cast<int>(g->devices.size());
    const size_t cnt = static_cast<size_t>(nq) * k;
    const size_t qbytes = sizeof(float) * static_cast<size_t>(nq) * g->dim;
    const bool ivf = g->kind == 1;
    // 1. every device: queries up, its own search on its own stream, local rows -> global rows
    for (int r = 0; r < nd; r++) {
        auto &mm = g->m[r];
        HG_HIP(hipSetDevice(g->devices[r]));
        HG_TRY(mm.q.ensure(qbytes));
        HG_TRY(mm.ids.ensure(sizeof(int32_t) * cnt));
        HG_TRY(mm.dist.ensure(sizeof(float) * cnt));
        if (ivf) HG_TRY(mm.ord.ensure(sizeof(uint32_t) * cnt));
        HG_HIP(hipMemcpyAsync(mm.q.p, Q, qbytes, hipMemcpyHostToDevice, mm.st));
        if (mm.n > 0) {
            if (ivf)
                HG_TRY(hnswgpu_ivf_search_shard_dev(mm.idx, static_cast<const float *>(mm.q.p), nq, k, param, static_cast<int32_t *>(mm.ids.p),
                                                    static_cast<float *>(mm.dist.p), static_cast<uint32_t *>(mm.ord.p), mm.st));
            else
                HG_TRY(hnswgpu_hnsw_search_dev(mm.idx, static_cast<const float *>(mm.q.p), nq, k, param, static_cast<int32_t *>(mm.ids.p),
                                               static_cast<float *>(mm.dist.p), nullptr, mm.st));
            hipLaunchKernelGGL(remap_ids_kernel, dim3(static_cast<unsigned>((cnt + 255) / 256)), dim3(256), 0, mm.st,
                               static_cast<int32_t *>(mm.ids.p), static_cast<int64_t>(cnt), mm.d_map, static_cast<int32_t>(mm.row0));
            HG_HIP(hipGetLastError());
        } else {  // a device without rows contributes nothing
            HG_HIP(hipMemsetAsync(mm.ids.p, 0xff, sizeof(int32_t) * cnt, mm.st));
            HG_HIP(hipMemsetAsync(mm.dist.p, 0x7f, sizeof(float) * cnt, mm.st));  // 0x7f7f7f7f: a huge finite distance, id -1
            if (ivf) HG_HIP(hipMemsetAsync(mm.ord.p, 0xff, sizeof(uint32_t) * cnt, mm.st));
        }
        HG_HIP(hipEventRecord(mm.ev, mm.st));
    }
    // 2. the partial lists to devices[0] (peer copies: xGMI between GPUs), one merge launch, results down
    HG_HIP(hipSetDevice(g->devices[0]));
    HG_TRY(g->g_ids.ensure(sizeof(int32_t) * cnt * nd));
    HG_TRY(g->g_dist.ensure(sizeof(float) * cnt * nd));
    if (ivf) HG_TRY(g->g_ord.ensure(sizeof(uint32_t) * cnt * nd));
    HG_TRY(g->o_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(g->o_dist.ensure(sizeof(float) * cnt));
    for (int r = 0; r < nd; r++) {
        auto &mm = g->m[r];
        HG_HIP(hipStreamWaitEvent(g->st0, mm.ev, 0));
        HG_HIP(hipMemcpyPeerAsync(static_cast<int32_t *>(g->g_ids.p) + cnt * r, g->devices[0], mm.ids.p, g->devices[r], sizeof(int32_t) * cnt, g->st0));
        HG_HIP(hipMemcpyPeerAsync(static_cast<float *>(g->g_dist.p) + cnt * r, g->devices[0], mm.dist.p, g->devices[r], sizeof(float) * cnt, g->st0));
        if (ivf)
            HG_HIP(hipMemcpyPeerAsync(static_cast<uint32_t *>(g->g_ord.p) + cnt * r, g->devices[0], mm.ord.p, g->devices[r], sizeof(uint32_t) * cnt, g->st0));
    }
    if (ivf)
        HG_TRY(hnswgpu_merge_keyed_dev(g->devices[0], static_cast<const int32_t *>(g->g_ids.p), static_cast<const float *>(g->g_dist.p),
                                       static_cast<const uint32_t *>(g->g_ord.p), nd, nq, k, static_cast<int32_t *>(g->o_ids.p),
                                       static_cast<float *>(g->o_dist.p), g->st0));
    else
        HG_TRY(hnswgpu_merge_topk_dev(g->devices[0], static_cast<const int32_t *>(g->g_ids.p), static_cast<const float *>(g->g_dist.p), nd,
                                      nq, k, static_cast<int32_t *>(g->o_ids.p), static_cast<float *>(g->o_dist.p), g->st0));
    HG_HIP(hipSetDevice(g->devices[0]));
    HG_HIP(hipMemcpyAsync(out_ids, g->o_ids.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, g->st0));
    HG_HIP(hipMemcpyAsync(out_dist, g->o_dist.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, g->st0));
    HG_HIP(hipStreamSynchronize(g->st0));
    return 0;
}

// An error part-way must not return while other members' copies from the caller's (pageable) Q or into its outputs are
// still in flight: every stream of the group is drained first.
static int group_search(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t param, int32_t *out_ids,
                        float *out_dist) {
    const int rc = group_search_impl(g, Q, nq, k, param, out_ids, out_dist);
    if (rc != 0) {
        for (size_t r = 0; r < g->m.size(); r++) {
            (void)hipSetDevice(g->devices[r]);
            if (g->m[r].st) (void)hipStreamSynchronize(g->m[r].st);
        }
        (void)hipSetDevice(g->devices[0]);
        if (g->st0) (void)hipStreamSynchronize(g->st0);
    }
    return rc;
}

int hnswgpu_group_ivf_search(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t nprobe, int32_t *out_ids,
                             float *out_dist) {
    HG_REQUIRE(g, HNSWGPU_EINVAL, "group is null");
    HG_REQUIRE(nq >= 0 && k >= 1 && nprobe >= 1, HNSWGPU_EINVAL, "need nq >= 0, k >= 1, nprobe >= 1");
    HG_REQUIRE(k <= 1024 && nprobe <= 1024, HNSWGPU_ELIMIT, "k / nprobe > 1024 is not supported");
    if (nq == 0) return 0;
    HG_REQUIRE(Q && out_ids && out_dist, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    HG_REQUIRE(g->kind == 1, HNSWGPU_ESTATE, "the group holds no IVF index (call hnswgpu_group_set_ivf)");
    return group_search(g, Q, nq, k, std::min(nprobe, g->nlist), out_ids, out_dist);
}

int hnswgpu_group_hnsw_search(hnswgpu_group *g, const float *Q, int32_t nq, int32_t k, int32_t ef, int32_t *out_ids,
                              float *out_dist) {
    HG_REQUIRE(g, HNSWGPU_EINVAL, "group is null");
    HG_REQUIRE(nq >= 0 && k >= 1 && ef >= 1, HNSWGPU_EINVAL, "need nq >= 0, k >= 1, ef >= 1");
    HG_REQUIRE(k <= 1024, HNSWGPU_ELIMIT, "k > 1024 is not supported");
    if (nq == 0) return 0;
    HG_REQUIRE(Q && out_ids && out_dist, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(g->mu);
    HG_REQUIRE(g->kind == 2, HNSWGPU_ESTATE, "the group holds no HNSW sub-graphs (call hnswgpu_group_hnsw_build)");
    return group_search(g, Q, nq, k, ef, out_ids, out_dist);
}

// the sub-index a device holds (read-only use: info, get_graph, save of an HNSW part ...); NULL if out of range
hnswgpu_index *hnswgpu_group_member(hnswgpu_group *g, int32_t i) {
    if (!g || i < 0 || i >= static_cast<int32_t>(g->m.size())) return nullptr;
    return g->m[i].idx;
}

}  // extern "C"
