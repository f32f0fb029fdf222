// ivf.hip -- IVF-FLAT on the device: k-means++ / Lloyd build, list layout, batched nprobe search.
// Reference: src/hnsw/ann/partition/ivf_flat.clj (file:line cited per function).
#include <float.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "engine.hpp"
#include "stream_kernels.hpp"
#include "javarandom.hpp"

namespace hg {


// probes[q][p] = p-th nearest centroid; pairs[q*nprobe+p] = that list's row range + the offset of
// its rows in the query's concatenated candidate stream (ties are broken in that order, which is
// the order search-ivf-flat concatenates partitions in, ivf_flat.clj:281-294)
__global__ __launch_bounds__(kWG) void probe_pairs_kernel(const uint32_t *ord, int nq, int nprobe, const int64_t *listoff,
                                                          const int64_t *glistoff, Pair *pairs, int32_t *probes,
                                                          int32_t *qcnt, uint32_t *bk_cnt, uint2 *bk_mem, int32_t bk_cap,
                                                          uint32_t *surv_cnt) {
    // one wave per query: lane p owns probe p (+64, +128, ...); the offsets of the probes in the query's
    // concatenated candidate stream are an exclusive prefix sum over the list lengths (wave scan)
    const int lane = threadIdx.x & (kWave - 1);
    const int q = blockIdx.x * kNWave + (threadIdx.x >> 6);
    if (q >= nq) return;
    uint32_t carry = 0, gcarry = 0;
    bool over = false;
    for (int p0 = 0; p0 < nprobe; p0 += kWave) {
        const int p = p0 + lane;
        uint32_t l = p < nprobe ? ord[static_cast<int64_t>(q) * nprobe + p] : 0xffffffffu;
        Pair pr;
        pr.q = q;
        pr.pad = 0;
        pr.row_begin = pr.row_end = 0;
        uint32_t glen = 0;
        if (l != 0xffffffffu) {
            pr.row_begin = listoff[l];
            pr.row_end = listoff[l + 1];
            glen = static_cast<uint32_t>(glistoff[l + 1] - glistoff[l]);
        }
        const uint32_t len = static_cast<uint32_t>(pr.row_end - pr.row_begin);
        uint32_t incl = len, gincl = glen;  // inclusive scans across the wave
        for (int off = 1; off < kWave; off <<= 1) {
            uint32_t o = __shfl_up(incl, off, kWave), go = __shfl_up(gincl, off, kWave);
            if (lane >= off) {
                incl += o;
                gincl += go;
            }
        }
        pr.ord_base = carry + incl - len;
        pr.gord_base = gcarry + gincl - glen;
        if (p < nprobe) {
            pairs[static_cast<int64_t>(q) * nprobe + p] = pr;
            if (probes) probes[static_cast<int64_t>(q) * nprobe + p] = l == 0xffffffffu ? -1 : static_cast<int32_t>(l);
            if (bk_cnt && len > 0) {  // survivor stream: the pair filed under its list (stream_kernels.hpp)
                const uint32_t slot = __hip_atomic_fetch_add(bk_cnt + l, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (slot < static_cast<uint32_t>(bk_cap)) bk_mem[static_cast<int64_t>(l) * bk_cap + slot] = make_uint2(static_cast<uint32_t>(q), pr.ord_base);
                else over = true;
            }
        }
        carry += __shfl(incl, kWave - 1, kWave);
        gcarry += __shfl(gincl, kWave - 1, kWave);
    }
    if (qcnt && lane == 0) qcnt[q] = static_cast<int32_t>(carry);
    // an empty survivor list -- or, when a bucket was full, the mark that sends the query through the finish kernel's fallback
    const bool any_over = __ballot(over) != 0;
    if (surv_cnt && lane == 0) surv_cnt[q] = any_over ? 0x80000000u : 0u;
}

// ---- execution order of the GEMV list scan: pairs sorted by the list they probe (counting sort in one workgroup;
// the order inside a list is whatever the atomics give -- it only decides WHEN a pair runs, not what it computes)
constexpr int kOrderMaxLists = 16384;  // LDS histogram: 64 KiB
__device__ __forceinline__ void pair_order_wg(const int32_t *probes, int npairs, int nlist, int32_t *order, int stride) {
    extern __shared__ int32_t ocnt[];  // [nlist + 1] (bucket nlist: pairs without a list), then [1024] scan scratch
    int32_t *part = ocnt + nlist + 1;
    const int tid = threadIdx.x, nb = nlist + 1;
    for (int i = tid; i < nb; i += 1024) ocnt[i] = 0;
    __syncthreads();
    for (int i = tid; i < npairs; i += 1024) {
        const int l = probes[static_cast<int64_t>(i) * stride];
        atomicAdd(&ocnt[l >= 0 && l < nlist ? l : nlist], 1);
    }
    __syncthreads();
    // exclusive scan over the buckets: every thread a contiguous range, then the 1024 range sums
    const int per = (nb + 1023) / 1024, lo = tid * per, hi = lo + per < nb ? lo + per : nb;
    int sum = 0;
    for (int i = lo; i < hi; i++) sum += ocnt[i];
    part[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int add = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int run = part[tid] - sum;
    for (int i = lo; i < hi; i++) {
        const int c = ocnt[i];
        ocnt[i] = run;
        run += c;
    }
    __syncthreads();
    for (int i = tid; i < npairs; i += 1024) {
        const int l = probes[static_cast<int64_t>(i) * stride];
        order[atomicAdd(&ocnt[l >= 0 && l < nlist ? l : nlist], 1)] = i;
    }
}
__global__ __launch_bounds__(1024) void pair_order_kernel(const int32_t *probes, int npairs, int nlist, int32_t *order,
                                                          int stride = 1) {  // item i probes list probes[i * stride]
    pair_order_wg(probes, npairs, nlist, order, stride);
}

// ---- work list of the grouped bounds pass (stream_kernels.hpp) --------------------------------------------------
// The routing step has filed every (query, list) pair under its list (bk_cnt / bk_mem); this turns the per-list counts
// into the dense list of work items (list, group of <= 32 members, chunk of rows), chunk-major inside a list so that the
// groups reading the same rows are neighbours.  One workgroup: a scan over the lists' item counts, then every thread
// fills items by bisection into the scanned offsets (a hot list's hundreds of items are not one thread's job).
// A second workgroup, when launched, is pair_order_kernel for the query order of the later kernels (ord_*): two one-workgroup
// jobs of a large batch side by side instead of one behind the other.
__global__ __launch_bounds__(1024) void ivf_worklist_kernel(uint32_t *bk_cnt, int32_t bk_cap, int nlist,
                                                            const int64_t *list_off, int64_t chunk_rows, int max_chunks,
                                                            WorkDesc *desc, int32_t *nitems, const int32_t *ord_probes,
                                                            int ord_nq, int32_t *ord_out, int ord_stride,
                                                            HomeDesc *home_desc, int32_t *home_nitems, int home_gq,
                                                            int home_chunk, int tq) {  // tq: members per group (32, or 64: two column blocks)
    if (blockIdx.x == 1) {
        pair_order_wg(ord_probes, ord_nq, nlist, ord_out, ord_stride);
        if (home_desc) {
            // the work list of ivf_home_kernel: after the counting sort ocnt[l] is where list l's queries END in the order
            // (they start where list l - 1's end); every list that is some query's nearest gives (groups of <= home_gq of
            // them) x (chunks of home_chunk rows) items, in whatever order the threads reserve their ranges
            extern __shared__ int32_t ocnt[];
            __shared__ int32_t hn_s;
            if (threadIdx.x == 0) hn_s = 0;
            __syncthreads();
            for (int l = threadIdx.x; l < nlist; l += 1024) {
                const int end = ocnt[l], start = l > 0 ? ocnt[l - 1] : 0, c = end - start;
                const int64_t rb0 = list_off[l], rows = list_off[l + 1] - rb0;
                if (c <= 0 || rows <= 0) continue;
                const int ng = (c + home_gq - 1) / home_gq, nch = static_cast<int>((rows + home_chunk - 1) / home_chunk);
                int at = atomicAdd(&hn_s, ng * nch);
                for (int ch = 0; ch < nch; ch++)
                    for (int g = 0; g < ng; g++, at++) {
                        HomeDesc d;
                        d.rb0 = rb0;
                        d.r0_off = ch * home_chunk;
                        d.r1_off = static_cast<int32_t>(rows < static_cast<int64_t>(ch + 1) * home_chunk ? rows : static_cast<int64_t>(ch + 1) * home_chunk);
                        d.q0 = start + g * home_gq;
                        d.cnt = c - g * home_gq < home_gq ? c - g * home_gq : home_gq;
                        d.list = l;
                        d.pad = 0;
                        home_desc[at] = d;
                    }
            }
            __syncthreads();
            if (threadIdx.x == 0) *home_nitems = hn_s;
        }
        return;
    }
    __shared__ int32_t off_s[1025];  // exclusive offsets of this pass's lists
    __shared__ int32_t part[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int l0 = 0; l0 < nlist; l0 += 1024) {
        const int l = l0 + tid;
        int c = 0, nch = 0;
        int64_t rows = 0;
        if (l < nlist) {
            const uint32_t filed = bk_cnt[l];
            c = filed < static_cast<uint32_t>(bk_cap) ? static_cast<int>(filed) : bk_cap;
            rows = list_off[l + 1] - list_off[l];
            nch = (c > 0 && rows > 0) ? static_cast<int>(tile_nchunks(rows, chunk_rows, max_chunks)) : 0;
        }
        const int ng = (c + tq - 1) / tq;
        const int nw = ng * nch;
        // inclusive scan: within the wave, then across the 16 waves
        int incl = nw;
        for (int o = 1; o < kWave; o <<= 1) {
            const int v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) part[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += part[w];
        const int base = carry_s;
        off_s[tid] = before + incl - nw;
        if (tid == 1023) off_s[1024] = before + incl;
        __syncthreads();
        const int total = off_s[1024];
        for (int w = tid; w < total; w += 1024) {
            int lo = 0, hi = 1023;  // the last list of this pass whose offset <= w
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (off_s[mid] <= w) lo = mid;
                else hi = mid - 1;
            }
            const int ll = l0 + lo, k = w - off_s[lo];
            const uint32_t filed = bk_cnt[ll];
            const int cc = filed < static_cast<uint32_t>(bk_cap) ? static_cast<int>(filed) : bk_cap;
            const int ngl = (cc + tq - 1) / tq;
            const int ch = k / ngl, g = k - ch * ngl;  // chunk-major
            const int64_t rb0 = list_off[ll], rws = list_off[ll + 1] - rb0;
            const int64_t tiles = (rws + kTileRows - 1) / kTileRows;
            const int64_t nchl = tile_nchunks(rws, chunk_rows, max_chunks);
            const int64_t per = (tiles + nchl - 1) / nchl * kTileRows;
            const int64_t a0 = ch * per, a1 = a0 + per < rws ? a0 + per : rws;
            WorkDesc d;
            d.rb0 = rb0;
            d.r0_off = static_cast<int32_t>(a0);
            d.r1_off = static_cast<int32_t>(a1 > a0 ? a1 : a0);
            d.list = ll;
            d.mem0 = g * tq;
            d.cnt = cc - g * tq < tq ? cc - g * tq : tq;
            d.pad = 0;
            desc[base + w] = d;
        }
        __syncthreads();
        if (tid == 0) carry_s = base + total;
        __syncthreads();
    }
    if (tid == 0) *nitems = carry_s;
    // the counters are zero again for the next search (every read of them is behind the barriers above)
    for (int l = tid; l < nlist; l += 1024) bk_cnt[l] = 0;
}

// ---- grouping of (query, probed list) pairs by list, for the tiled scan --------------------------------
__global__ void ivf_hist_kernel(const int32_t *probes, int64_t npairs, int32_t *cnt) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < npairs && probes[i] >= 0) atomicAdd(&cnt[probes[i]], 1);
}

// One workgroup: exclusive scans over the lists (members, groups of <= 32) in chunks of 1024 lists, then
// every thread writes the groups of its own list.
// With `probes` set (batches of a few thousand pairs) the kernel also does the histogram before and the scatter of the
// pairs into their groups after: one launch instead of a memset and three kernels.
__global__ __launch_bounds__(1024) void ivf_plan_kernel(int32_t *cnt, int nlist, int32_t *list_mem_begin,
                                                        int32_t *fill, int32_t *grp_seg, int32_t *grp_mem_begin,
                                                        int32_t *grp_mem_cnt, int32_t *ngroups, int tq,
                                                        const int64_t *list_off, int64_t chunk_rows, int max_chunks,
                                                        int32_t *wi_group, int32_t *wi_chunk, int32_t *nitems,
                                                        int32_t *work_ctr, const int32_t *probes, int64_t npairs,
                                                        const Pair *pairs, int64_t stride, GroupMember *members) {
    __shared__ int32_t sm[1024], sg[1024], sw[1024];
    __shared__ int32_t carry_m, carry_g, carry_w;
    const int tid = threadIdx.x;
    if (tid == 0) carry_m = carry_g = carry_w = 0;
    if (tid < 8 && work_ctr) work_ctr[tid] = 0;  // the persistent tile scan's per-XCD item counters
    if (probes) {  // histogram of the probed lists (ivf_hist_kernel's job)
        for (int l = tid; l < nlist; l += 1024) cnt[l] = 0;
        __syncthreads();
        for (int64_t i = tid; i < npairs; i += 1024)
            if (probes[i] >= 0) atomicAdd(&cnt[probes[i]], 1);
    }
    __syncthreads();
    for (int l0 = 0; l0 < nlist; l0 += 1024) {
        const int l = l0 + tid;
        const int c = l < nlist ? cnt[l] : 0;
        const int ng = (c + tq - 1) / tq;
        const int64_t rows = l < nlist ? list_off[l + 1] - list_off[l] : 0;
        const int nch = (ng > 0 && rows > 0) ? static_cast<int>(tile_nchunks(rows, chunk_rows, max_chunks)) : 0;
        const int nw = ng * nch;  // work items of this list: every group x every chunk
        sm[tid] = c;
        sg[tid] = ng;
        sw[tid] = nw;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
            int am = tid >= off ? sm[tid - off] : 0, ag = tid >= off ? sg[tid - off] : 0, aw = tid >= off ? sw[tid - off] : 0;
            __syncthreads();
            sm[tid] += am;
            sg[tid] += ag;
            sw[tid] += aw;
            __syncthreads();
        }
        const int mem = carry_m + sm[tid] - c, g0 = carry_g + sg[tid] - ng;
        int w = carry_w + sw[tid] - nw;
        if (l < nlist) {
            list_mem_begin[l] = mem;
            fill[l] = 0;
            for (int b = 0, g = g0; b < c; b += tq, g++) {
                grp_seg[g] = l;
                grp_mem_begin[g] = mem + b;
                grp_mem_cnt[g] = c - b < tq ? c - b : tq;
            }
            // work items chunk-major: the groups of one chunk (same rows) are neighbours -- see tile_scan_kernel
            for (int ch = 0; ch < nch; ch++)
                for (int g = g0; g < g0 + ng; g++, w++) {
                    wi_group[w] = g;
                    wi_chunk[w] = ch;
                }
        }
        __syncthreads();
        if (tid == 1023) {
            carry_m += sm[1023];
            carry_g += sg[1023];
            carry_w += sw[1023];
        }
        __syncthreads();
    }
    if (tid == 0) {
        *ngroups = carry_g;
        *nitems = carry_w;
    }
    if (probes) {  // ivf_scatter_kernel's job: every pair into a slot of its list's member range
        __syncthreads();
        for (int64_t i = tid; i < npairs; i += 1024) {
            const int l = probes[i];
            if (l < 0) continue;
            const int slot = list_mem_begin[l] + atomicAdd(&fill[l], 1);
            GroupMember m;
            m.q = pairs[i].q;
            m.pad = 0;
            m.out_base = static_cast<int64_t>(pairs[i].q) * stride + pairs[i].ord_base;
            members[slot] = m;
        }
    }
}

__global__ void ivf_scatter_kernel(const Pair *pairs, const int32_t *probes, int64_t npairs, int64_t stride,
                                   const int32_t *list_mem_begin, int32_t *fill, GroupMember *members) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    int l = probes[i];
    if (l < 0) return;
    int slot = list_mem_begin[l] + atomicAdd(&fill[l], 1);
    GroupMember m;
    m.q = pairs[i].q;
    m.pad = 0;
    m.out_base = static_cast<int64_t>(pairs[i].q) * stride + pairs[i].ord_base;
    members[slot] = m;
}

__global__ void ivf_decode_kernel(const uint32_t *ord, int nq, int k, const Pair *pairs, int nprobe,
                                  const int32_t *listids, int32_t *out_ids, uint32_t *out_gord) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<int64_t>(nq) * k) return;
    int q = static_cast<int>(i / k);
    uint32_t o = ord[i], go = 0xffffffffu;
    int32_t id = -1;
    if (o != 0xffffffffu) {
        const Pair *pp = pairs + static_cast<int64_t>(q) * nprobe;
        int p = 0;
        while (p + 1 < nprobe && pp[p + 1].ord_base <= o) p++;
        // lists of length 0 share an ord_base with their successor: take the last one that fits
        id = listids[pp[p].row_begin + (o - pp[p].ord_base)];
        go = pp[p].gord_base + (o - pp[p].ord_base);  // position in the candidate stream of the whole index
    }
    out_ids[i] = id;
    if (out_gord) out_gord[i] = go;
}

// dst row pos <- src row listids[pos]; one wave per row, float4 lanes
__global__ __launch_bounds__(kWG) void permute_rows_kernel(const float *src, const float *src_norms, int64_t ld,
                                                           const int32_t *listids, int64_t n, float *dst,
                                                           float *dst_norms) {
    int lane = threadIdx.x & (kWave - 1);
    int64_t pos = static_cast<int64_t>(blockIdx.x) * kNWave + (threadIdx.x >> 6);
    if (pos >= n) return;
    int64_t s = listids[pos];
    const float4 *sp = reinterpret_cast<const float4 *>(src + s * ld);
    float4 *dp = reinterpret_cast<float4 *>(dst + pos * ld);
    for (int i = lane; i < ld / 4; i += kWave) dp[i] = sp[i];
    if (lane == 0) dst_norms[pos] = src_norms[s];
}

// compute-centroid (ivf_flat.clj:66-77): f64 sum of the member rows in index order, divided by the
// count, rounded to f32 for storage; an empty list keeps its previous centroid (:112-114).
// One workgroup per list; thread t owns columns t, t+256, ...
__global__ __launch_bounds__(kWG) void centroid_mean_kernel(const float *rows, int64_t ld, int dim,
                                                            const int64_t *listoff, const int32_t *listids,
                                                            float *cent) {
    int l = blockIdx.x;
    int64_t b = listoff[l], e = listoff[l + 1];
    if (e <= b) return;
    for (int c0 = threadIdx.x; c0 < dim; c0 += kWG) {
        double s = 0.0;
        for (int64_t i = b; i < e; i++) s = s + static_cast<double>(rows[static_cast<int64_t>(listids[i]) * ld + c0]);
        cent[static_cast<int64_t>(l) * ld + c0] = static_cast<float>(s / static_cast<double>(e - b));
    }
}

// The f64 column sums of compute-centroid (ivf_flat.clj:70-75) without the division: what one shard of a row-sharded
// index contributes to a Lloyd update (the sums of all shards are added, then divided by the global count).
__global__ __launch_bounds__(kWG) void list_sum_kernel(const float *rows, int64_t ld, int dim, const int64_t *listoff,
                                                       const int32_t *listids, double *sums) {
    int l = blockIdx.x;
    int64_t b = listoff[l], e = listoff[l + 1];
    for (int c0 = threadIdx.x; c0 < dim; c0 += kWG) {
        double s = 0.0;
        for (int64_t i = b; i < e; i++) s = s + static_cast<double>(rows[static_cast<int64_t>(listids[i]) * ld + c0]);
        sums[static_cast<int64_t>(l) * dim + c0] = s;
    }
}

// sum of d^2 over blocks of kSampleBlock elements in f64 (fixed reduction order): the D^2 sampling of
// k-means++ then needs only these partial sums plus ONE block of distances on the host
constexpr int kSampleBlock = 4096;
__global__ __launch_bounds__(kWG) void block_sumsq_kernel(const float *d, int64_t n, double *out) {
    __shared__ double sm[kWG];
    const int64_t b0 = static_cast<int64_t>(blockIdx.x) * kSampleBlock;
    double s = 0.0;
    for (int i = 0; i < kSampleBlock / kWG; i++) {
        int64_t j = b0 + static_cast<int64_t>(threadIdx.x) * (kSampleBlock / kWG) + i;
        if (j < n) s = s + static_cast<double>(d[j]) * static_cast<double>(d[j]);
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int off = kWG / 2; off > 0; off >>= 1) {
        if (threadIdx.x < off) sm[threadIdx.x] = sm[threadIdx.x] + sm[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sm[0];
}

__global__ void fill_kernel(float *p, int64_t n, float v) {
    int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

static int validate_lists(int64_t n, int32_t nlist, const int64_t *off, const int32_t *ids) {
    HG_REQUIRE(off[0] == 0 && off[nlist] == n, HNSWGPU_EINVAL, "list_off must start at 0 and end at n");
    for (int l = 0; l < nlist; l++) HG_REQUIRE(off[l] <= off[l + 1], HNSWGPU_EINVAL, "list_off not monotone");
    std::vector<uint8_t> seen(static_cast<size_t>(n), 0);
    for (int64_t i = 0; i < n; i++) {
        HG_REQUIRE(ids[i] >= 0 && ids[i] < n, HNSWGPU_EINVAL, "list_ids[%lld] out of range", (long long)i);
        HG_REQUIRE(!seen[ids[i]], HNSWGPU_EINVAL, "row %d is in two lists", ids[i]);
        seen[ids[i]] = 1;
    }
    return 0;
}

static void free_ivf(hnswgpu_index *idx) {
    if (idx->lrows_alias) idx->d_lrows = idx->d_lnorms = nullptr;  // the base rows themselves: not ours to free
    void *ptrs[] = {idx->d_cent, idx->d_cnorms, idx->d_lrows, idx->d_lnorms, idx->d_listoff, idx->d_listids, idx->d_glistoff,
                    idx->d_lcmeta, idx->d_lctile, idx->d_lhalf, idx->d_lhmeta};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    idx->d_cent = idx->d_cnorms = idx->d_lrows = idx->d_lnorms = nullptr;
    idx->d_lcmeta = nullptr;
    idx->d_lctile = nullptr;
    idx->d_lhalf = nullptr;
    idx->d_lhmeta = nullptr;
    idx->ivf_calibrated = idx->ivf_stream_off = false;
    idx->d_listoff = idx->d_glistoff = nullptr;
    idx->d_listids = nullptr;
    idx->lrows_alias = false;
    idx->h_glistlen.clear();
    idx->ivf_n_global = 0;
    idx->nlist = 0;
}

// device centroids (nlist x ld) + host lists -> the searchable layout
static int install_lists(hnswgpu_index *idx, int32_t nlist, const int64_t *off, const int32_t *ids, hipStream_t st) {
    int64_t n = idx->n;
    // rows that already arrive in list order (a shard filled list by list, hnsw-clj_amd/sharded.py) are scanned in
    // place: no second copy of the base
    bool identity = n > 0;
    for (int64_t i = 0; i < n && identity; i++) identity = ids[i] == i;
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_listoff), sizeof(int64_t) * (nlist + 1)));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_listids), sizeof(int32_t) * std::max<int64_t>(n, 1)));
    HG_HIP(hipMemcpyAsync(idx->d_listoff, off, sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
    if (identity) {
        idx->lrows_alias = true;
        idx->d_lrows = idx->d_base;
        idx->d_lnorms = idx->d_norms;
        HG_HIP(hipMemcpyAsync(idx->d_listids, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    } else {
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_lrows), sizeof(float) * std::max<int64_t>(n, 1) * idx->ld));
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_lnorms), sizeof(float) * std::max<int64_t>(n, 1)));
    }
    if (n > 0 && !identity) {
        HG_HIP(hipMemcpyAsync(idx->d_listids, ids, sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(permute_rows_kernel, dim3(static_cast<unsigned>((n + kNWave - 1) / kNWave)), dim3(kWG), 0,
                           st, idx->d_base, idx->d_norms, idx->ld, idx->d_listids, n, idx->d_lrows, idx->d_lnorms);
        HG_HIP(hipGetLastError());
    }
    idx->h_listoff.assign(off, off + nlist + 1);
    idx->h_listids.assign(ids, ids + n);
    idx->max_list_len = 0;
    idx->min_list_len = nlist > 0 ? off[1] - off[0] : 0;
    for (int l = 0; l < nlist; l++) {
        idx->max_list_len = std::max(idx->max_list_len, off[l + 1] - off[l]);
        idx->min_list_len = std::min(idx->min_list_len, off[l + 1] - off[l]);
    }
    idx->nlist = nlist;
    HG_TRY(ensure_list_codes(idx, st));
    HG_TRY(ensure_list_half(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

static int alloc_centroids(hnswgpu_index *idx, int32_t nlist) {
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_cent), sizeof(float) * nlist * idx->ld));
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_cnorms), sizeof(float) * nlist));
    return 0;
}

static int download_centroids(hnswgpu_index *idx, hipStream_t st) {
    idx->h_cent.resize(static_cast<size_t>(idx->nlist) * idx->dim);
    HG_HIP(hipMemcpy2DAsync(idx->h_cent.data(), sizeof(float) * idx->dim, idx->d_cent, sizeof(float) * idx->ld,
                            sizeof(float) * idx->dim, idx->nlist, hipMemcpyDeviceToHost, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

// assignment of every base row to its nearest centroid (ivf_flat.clj:79-90) -> s_ord / s_dist [n]
static int assign_enqueue(hnswgpu_index *idx, const float *d_cent, const float *d_cnorms, int32_t nlist,
                          hipStream_t st) {
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    a.rows = d_cent;
    a.row_norms = d_cnorms;
    a.ld = idx->ld;
    a.nrows_all = nlist;
    a.Q = idx->d_base;
    a.qld = idx->ld;
    a.q_norms = idx->d_norms;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.pairs = nullptr;
    a.k = 1;
    a.role = ROLE_ASSIGN;
    if (tile_path_ok(idx) && tile_mode() != 0)  // dense X . C^T: every base row against every centroid
        return tile_topk_all(idx, idx->d_base, idx->d_norms, static_cast<int32_t>(idx->n), d_cent, d_cnorms, nlist, 1, st,
                             PROF_ASSIGN);
    return scan_topk(idx, a, static_cast<int32_t>(idx->n), 1, nlist, st, PROF_ASSIGN);
}

static void lists_from_assign(const uint32_t *assign, int64_t n, int32_t nlist, std::vector<int64_t> &off,
                              std::vector<int32_t> &ids) {
    off.assign(nlist + 1, 0);
    for (int64_t i = 0; i < n; i++) off[assign[i] + 1]++;
    for (int l = 0; l < nlist; l++) off[l + 1] += off[l];
    ids.resize(static_cast<size_t>(n));
    std::vector<int64_t> cur(off.begin(), off.end() - 1);
    for (int64_t i = 0; i < n; i++) ids[cur[assign[i]]++] = static_cast<int32_t>(i);  // index order inside a list
}

// kmeans-plus-plus-init (ivf_flat.clj:32-60), incremental: one streaming min-update pass per new
// centroid instead of the reference's O(N k^2 D) recomputation (same minima).  The D^2 sampling
// itself (sequential f64 prefix sum + Random draw, :51-58) runs on the host over the device-computed
// distances so that its order of additions is the reference's.
static int kmeanspp_device(hnswgpu_index *idx, int32_t nlist, int64_t seed, std::vector<int32_t> &chosen,
                           hipStream_t st) {
    int64_t n = idx->n;
    JavaRandom rng(seed);
    HG_TRY(idx->s_misc.ensure(sizeof(float) * n));
    float *mind = idx->s_misc.as<float>();
    hipLaunchKernelGGL(fill_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, st, mind, n, FLT_MAX);
    HG_HIP(hipGetLastError());
    const int64_t nb = (n + kSampleBlock - 1) / kSampleBlock;
    HG_TRY(idx->s_misc2.ensure(sizeof(double) * nb));
    double *d_bs = idx->s_misc2.as<double>();
    std::vector<double> bs(static_cast<size_t>(nb));
    std::vector<float> h(kSampleBlock);
    chosen.resize(nlist);
    int32_t cur = rng.next_int(static_cast<int32_t>(n));
    chosen[0] = cur;
    // with int8 rows a round reads them and fetches f32 rows only where the new centre may be nearest (seed_update_kernel):
    // the same minima (1M x 768 / 1024 lists: 0.47 -> ~0.12 ms per round, the whole build 0.78 -> 0.42 s)
    const int seed_bounds = static_cast<int>(tune(HNSWGPU_TUNE_SEED_BOUNDS, 1));  // 0 = every round a full f32 pass (A/B)
    if (seed_bounds) HG_TRY(ensure_qrows(idx, st));
    for (int c = 1; c < nlist; c++) {
        if (seed_bounds && idx->d_qrows) {
            SeedArgs sa;
            sa.rows = idx->d_base;
            sa.row_norms = idx->d_norms;
            sa.ld = idx->ld;
            sa.n = n;
            sa.dim = idx->dim;
            sa.metric = idx->metric;
            sa.cur = cur;
            sa.qrows = idx->d_qrows;
            sa.qmeta = idx->d_qmeta;
            sa.out = mind;
            sa.rows_per_wg = 512;
            const unsigned sgrid = static_cast<unsigned>((n + sa.rows_per_wg - 1) / sa.rows_per_wg);
#define CALL(N, R, L) \
    hipLaunchKernelGGL((seed_update_kernel<N, (N <= 3 ? 4 : (N <= 6 ? 2 : 1)), L>), dim3(sgrid), dim3(kWG), 0, st, sa)
            HG_DISPATCH(idx->nch, idx->metric == METRIC_L2, CALL);
#undef CALL
            HG_HIP(hipGetLastError());
        } else {
        ScanArgs a;
        memset(&a, 0, sizeof(a));
        a.rows = idx->d_base;
        a.row_norms = idx->d_norms;
        a.ld = idx->ld;
        a.nrows_all = n;
        a.Q = idx->d_base + static_cast<int64_t>(cur) * idx->ld;
        a.qld = idx->ld;
        a.q_norms = idx->d_norms + cur;
        a.dim = idx->dim;
        a.metric = idx->metric;
        a.mode = MODE_MINUPD;
        a.role = ROLE_SEED;
        a.pairs = nullptr;
        a.npairs = 1;
        a.nchunks = plan_chunks(idx->nch, n, n, 1, &a.chunk_rows);
        a.k = 1;
        a.out = mind;
        HG_TRY(launch_scan(idx->nch, a, st));
        }
        hipLaunchKernelGGL(block_sumsq_kernel, dim3(static_cast<unsigned>(nb)), dim3(kWG), 0, st, mind, n, d_bs);
        HG_HIP(hipGetLastError());
        HG_HIP(hipMemcpyAsync(bs.data(), d_bs, sizeof(double) * nb, hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
        // (loop [i 0 cumsum 0.0] (if (>= (+ cumsum dist-sq) r) pick i ...)) :54-58, two-level: whole blocks are
        // skipped by their partial sums, the block that holds r is walked element by element
        double sum = 0.0;
        for (int64_t b = 0; b < nb; b++) sum = sum + bs[b];
        const double r = rng.next_double() * sum;
        double cum = 0.0;
        int64_t i = n - 1;  // clamped against round-off; the reference would throw
        bool found = false;
        for (int64_t b = 0; b < nb && !found; b++) {
            if (cum + bs[b] < r && b + 1 < nb) {
                cum = cum + bs[b];
                continue;
            }
            const int64_t b0 = b * kSampleBlock, cnt = std::min<int64_t>(kSampleBlock, n - b0);
            HG_HIP(hipMemcpyAsync(h.data(), mind + b0, sizeof(float) * cnt, hipMemcpyDeviceToHost, st));
            HG_HIP(hipStreamSynchronize(st));
            for (int64_t j = 0; j < cnt; j++) {
                double dsq = static_cast<double>(h[j]) * static_cast<double>(h[j]);
                if (cum + dsq >= r) {
                    i = b0 + j;
                    found = true;
                    break;
                }
                cum = cum + dsq;
            }
        }
        cur = static_cast<int32_t>(i);
        chosen[c] = cur;
    }
    return 0;
}

// The (query, probed list) pairs of a batch grouped by list on the device: groups of <= 32 queries, every group cut into
// chunks of whole tiles -- the dense work list the tile scan, the register-row group kernel and the bounds pass share.
struct GroupPlan {
    int64_t stride;      // candidates per query, upper bound (a multiple of 4)
    int64_t gbound;      // upper bound of the number of groups
    int64_t cr;          // rows per chunk (whole tiles)
    int32_t max_chunks;
    int64_t wbound;      // upper bound of the number of work items
    int32_t *gseg, *gmb, *gmc, *wig, *wic, *ngr, *nit, *wctr;
    GroupMember *members;
};

// ptiles > 0: persistent workgroups pulling items of up to `ptiles` tiles off a queue (tile_scan_kernel); 0 = one
// workgroup per item, items sized to fill the chip.  member_stride: a member's output base is q * member_stride + the
// pair's offset in the query's candidate stream (0 = the offset alone: the survivor stream's order keys).
static int plan_groups(hnswgpu_index *idx, int32_t nq, int32_t nprobe, const int32_t *d_probes, int64_t ptiles,
                       int64_t member_stride, hipStream_t st, GroupPlan &g) {
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    const int nlist = idx->nlist;
    // a multiple of 4 so that every query's array is 16-B aligned (float4 select)
    g.stride = (static_cast<int64_t>(nprobe) * idx->max_list_len + 3) / 4 * 4;
    const int tq = tile_tq(idx->dim);
    g.gbound = npairs / tq + nlist;  // sum_l ceil(cnt_l / tq) <= this
    // rows per workgroup: whole tiles, enough workgroups to fill the chip; the device splits every list
    const int64_t mean = std::max<int64_t>(1, idx->n / std::max(nlist, 1));
    const int64_t est_groups = std::max<int64_t>(1, npairs / tq + nlist / 2);
    const int64_t mean_tiles = (mean + kTileRows - 1) / kTileRows;
    const int64_t tgt = tune(HNSWGPU_TUNE_TILE_WGS, 2048);
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(mean_tiles, (tgt + est_groups - 1) / est_groups));
    int64_t cr = ((mean_tiles + want - 1) / want) * kTileRows;
    if (ptiles > 0) cr = std::max<int64_t>(1, std::min<int64_t>(ptiles, mean_tiles)) * kTileRows;
    const int64_t max_tiles = (idx->max_list_len + kTileRows - 1) / kTileRows, tpc = cr / kTileRows;
    g.cr = cr;
    g.max_chunks = static_cast<int32_t>(std::max<int64_t>(1, (max_tiles + tpc / 2) / tpc));
    g.wbound = g.gbound * g.max_chunks;
    // int32 scratch: cnt | list_mem_begin | fill [nlist each] | grp_seg | grp_mem_begin | grp_mem_cnt [gbound each] |
    //                wi_group | wi_chunk [wbound each] | ngroups | nitems | work counters
    const size_t ints = 3 * static_cast<size_t>(nlist) + 3 * static_cast<size_t>(g.gbound) + 2 * static_cast<size_t>(g.wbound) + 4 + 8 + 8;
    HG_TRY(idx->s_misc.ensure(sizeof(int32_t) * ints));
    HG_TRY(idx->s_misc2.ensure(sizeof(GroupMember) * static_cast<size_t>(npairs)));
    int32_t *cnt = idx->s_misc.as<int32_t>();
    int32_t *lmb = cnt + nlist, *fill = lmb + nlist;
    g.gseg = fill + nlist;
    g.gmb = g.gseg + g.gbound;
    g.gmc = g.gmb + g.gbound;
    g.wig = g.gmc + g.gbound;
    g.wic = g.wig + g.wbound;
    g.ngr = g.wic + g.wbound;
    g.nit = g.ngr + 1;
    g.wctr = g.ngr + 4;
    g.members = idx->s_misc2.as<GroupMember>();
    if (npairs <= 8192) {  // small batches are launch-bound: histogram, plan and scatter in the one-workgroup kernel
        hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(1024), 0, st, cnt, nlist, lmb, fill, g.gseg, g.gmb, g.gmc, g.ngr, tq,
                           idx->d_listoff, cr, g.max_chunks, g.wig, g.wic, g.nit, ptiles > 0 ? g.wctr : nullptr, d_probes,
                           npairs, idx->s_pairs.as<Pair>(), member_stride, g.members);
    } else {
        HG_HIP(hipMemsetAsync(cnt, 0, sizeof(int32_t) * nlist, st));
        hipLaunchKernelGGL(ivf_hist_kernel, dim3(static_cast<unsigned>((npairs + 255) / 256)), dim3(256), 0, st, d_probes,
                           npairs, cnt);
        hipLaunchKernelGGL(ivf_plan_kernel, dim3(1), dim3(1024), 0, st, cnt, nlist, lmb, fill, g.gseg, g.gmb, g.gmc, g.ngr, tq,
                           idx->d_listoff, cr, g.max_chunks, g.wig, g.wic, g.nit, ptiles > 0 ? g.wctr : nullptr,
                           static_cast<const int32_t *>(nullptr), static_cast<int64_t>(0), static_cast<const Pair *>(nullptr),
                           static_cast<int64_t>(0), static_cast<GroupMember *>(nullptr));
        hipLaunchKernelGGL(ivf_scatter_kernel, dim3(static_cast<unsigned>((npairs + 255) / 256)), dim3(256), 0, st,
                           idx->s_pairs.as<Pair>(), d_probes, npairs, member_stride, lmb, fill, g.members);
    }
    HG_HIP(hipGetLastError());
    return 0;
}

// Large batches: group the (query, list) pairs by list, keep each group of <= 32 queries resident in LDS
// and stream the list through the MFMA tile kernel once per group; distances land in a dense
// per-query candidate array (position = the pair's order key), then one select pass per query.
// gemv_order: the register-row group kernel (GEMV summation order) instead of the MFMA tiles.
static int ivf_tile_scan(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                         const int32_t *d_probes, const int32_t *d_qcnt, hipStream_t st, bool gemv_order) {
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    const int tq = tile_tq(idx->dim);
    // persistent workgroups pulling items off a queue (tile_scan_kernel): items of up to `ptiles` tiles, however many
    // there are -- the queue balances them.  0 = one workgroup per item, items sized to fill the chip (round 1)
    const int64_t ptiles_env = tune(HNSWGPU_TUNE_TILE_PERSIST, 4);
    // the L2 group kernel takes one item per workgroup; few groups: shorter items, so that every CU has several
    const int64_t est_groups = std::max<int64_t>(1, npairs / tq + idx->nlist / 2);
    const int64_t ptiles = idx->metric == METRIC_L2 || gemv_order ? 0 : (ptiles_env == 4 && est_groups < 1200 ? 2 : ptiles_env);
    GroupPlan g;
    const int64_t stride0 = (static_cast<int64_t>(nprobe) * idx->max_list_len + 3) / 4 * 4;
    HG_TRY(plan_groups(idx, nq, nprobe, d_probes, ptiles, stride0, st, g));
    const int64_t stride = g.stride;
    HG_TRY(idx->s_tile.ensure(sizeof(float) * static_cast<size_t>(nq) * stride));  // queries already padded
    TileArgs t;
    memset(&t, 0, sizeof(t));
    t.rows = idx->d_lrows;
    t.row_norms = idx->d_lnorms;
    t.ld = idx->ld;
    t.dim = idx->dim;
    t.metric = idx->metric;
    t.gemv_order = gemv_order ? 1 : 0;
    t.Qp = idx->s_qp.as<float>();
    t.q_norms = idx->s_qn.as<float>();
    t.grp_seg = g.gseg;
    t.grp_mem_begin = g.gmb;
    t.grp_mem_cnt = g.gmc;
    t.ngroups = g.ngr;
    t.members = g.members;
    t.seg_off = idx->d_listoff;
    t.nq = nq;
    t.wi_group = g.wig;
    t.wi_chunk = g.wic;
    t.nitems = g.nit;
    t.work_ctr = ptiles > 0 ? g.wctr : nullptr;
    t.chunk_rows = static_cast<int32_t>(g.cr);
    t.nchunks = g.max_chunks;
    t.out = idx->s_tile.as<float>();
    hipEvent_t e0;
    HG_TRY(idx->s_ord.ensure(sizeof(uint32_t) * static_cast<size_t>(nq) * k));
    HG_TRY(idx->s_dist.ensure(sizeof(float) * static_cast<size_t>(nq) * k));
    SelectArgs s;
    memset(&s, 0, sizeof(s));
    s.dist = t.out;
    s.q_cnt = d_qcnt;
    s.stride = stride;
    s.nq = nq;
    s.k = k;
    s.out_ord = idx->s_ord.as<uint32_t>();
    s.out_dist = idx->s_dist.as<float>();
    prof_begin(idx, PROF_IVF_SCAN, st, &e0);
    HG_TRY(launch_tile(t, g.gbound, idx->dim, st));
    prof_end(idx, PROF_IVF_SCAN, st, e0);
    return launch_select(s, st);
}

// scratch of the survivor stream, carved out of s_qn: [QueryScal x nq | tau x nq | survivor count x nq]
// The tuning values ONE search decides its passes by, read once (ivf_search_enqueue): the table is process-wide and another
// thread may change it between the routing and the scan of the same search -- with the home-list decision flipping in between
// the routing would seed no threshold for a scan that expects one (every query through the overflow fallback).
struct StreamTune {
    int64_t stream_mid, finish_order, stream_home, mid_slices, mid_compact, home_strays;
    void read() {
        stream_mid = tune(HNSWGPU_TUNE_STREAM_MID, -1);
        finish_order = tune(HNSWGPU_TUNE_FINISH_ORDER, 512);  // 0 = never (A/B)
        stream_home = tune(HNSWGPU_TUNE_STREAM_HOME, -1);
        mid_slices = tune(HNSWGPU_TUNE_MID_SLICES, 0);
        mid_compact = tune(HNSWGPU_TUNE_MID_COMPACT, 1);
        home_strays = tune(HNSWGPU_TUNE_HOME_STRAYS, 32);
    }
};

struct StreamScratch {
    StreamTune tn;
    uint32_t *qcodes;
    QueryScal *qscal;
    uint32_t *tau, *surv_cnt;
    // the (query, list) pairs filed by list for the grouped bounds pass (s_misc): counters [nlist], members [nlist][bk_cap]
    uint32_t *bk_cnt;
    uint2 *bk_mem;
    int32_t bk_cap;
    bool wl_folded;  // the routing tail's launch has built the bounds pass's work list (worklist_part_wg)
};

// Buckets for the pairs of a batch: room for the whole batch under every list (a list may be probed by every query) as
// long as that stays under 256 MB -- 32M / nlist members per list otherwise.  A hotter list files what fits; the queries
// beyond take the finish kernel's fallback (exact, just slower).
static int stream_buckets(hnswgpu_index *idx, int32_t nq, int32_t nprobe, StreamScratch &s, hipStream_t st) {
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    (void)npairs;
    int64_t cap = std::max<int64_t>(64, (32LL << 20) / std::max(idx->nlist, 1) / 32 * 32);  // <= 256 MB of members
    cap = std::min<int64_t>(cap, (static_cast<int64_t>(nq) + 31) / 32 * 32);
    const int64_t cap_env = tune(HNSWGPU_TUNE_STREAM_BUCKET, 0);  // tests: tiny buckets force the fallback
    if (cap_env > 0) cap = cap_env;
    // members in s_misc; the counters in a buffer of their own that is zero between searches (a 4 KB memset is a 6 us
    // launch: a thirtieth of a batch-32 search)
    HG_TRY(idx->s_misc.ensure(sizeof(uint2) * static_cast<size_t>(idx->nlist) * cap));
    const size_t cnt_bytes = sizeof(uint32_t) * static_cast<size_t>(idx->nlist);
    if (idx->s_bk.cap < cnt_bytes) {
        HG_TRY(idx->s_bk.ensure(cnt_bytes));
        idx->bk_dirty = true;
    }
    if (idx->bk_dirty) HG_HIP(hipMemsetAsync(idx->s_bk.p, 0, idx->s_bk.cap, st));
    idx->bk_dirty = true;  // until the work-list kernel of this search has been enqueued
    s.bk_cnt = idx->s_bk.as<uint32_t>();
    s.bk_mem = idx->s_misc.as<uint2>();
    s.bk_cap = static_cast<int32_t>(cap);
    return 0;
}
static int stream_scratch(hnswgpu_index *idx, int32_t nq, StreamScratch &s) {
    HG_TRY(idx->s_qp.ensure(sizeof(uint32_t) * kWave * idx->nch * static_cast<size_t>(nq)));
    HG_TRY(idx->s_qn.ensure((sizeof(QueryScal) + 2 * sizeof(uint32_t)) * static_cast<size_t>(nq)));
    s.qcodes = idx->s_qp.as<uint32_t>();
    s.qscal = idx->s_qn.as<QueryScal>();
    s.tau = reinterpret_cast<uint32_t *>(s.qscal + nq);
    s.surv_cnt = s.tau + nq;
    return 0;
}

// The half-precision pass of a batch (stream_kernels.hpp, step 1b).  Once the exact pass would be the largest kernel (its
// rows are 3 KB each and every query fetches its own) the survivors first meet their half-precision rows.  The survivors
// are a few per cent of the candidates: from ~1.5 M candidates per batch (48 queries x 32 lists x 977 rows; 5 queries at 10 M
// rows) the pass saves more than its launch costs -- measured at 1M x 768: batch 32 0.183 ms without vs 0.195 with, 64:
// 0.250 vs 0.243, 128: 0.298 vs 0.277.  HNSWGPU_TUNE_STREAM_MID=<queries> overrides (tests: 1 = always; 0 = never).
static bool ivf_mid_mode(const hnswgpu_index *idx, int32_t nq, int32_t nprobe, const StreamTune &tn) {
    const int64_t mid_env = tn.stream_mid;
    const int64_t cand = static_cast<int64_t>(nq) * nprobe * ivf_mean_len(idx);
    return idx->d_lhalf != nullptr && !idx->ivf_calibrating && (mid_env >= 0 ? (mid_env > 0 && nq >= mid_env) : cand >= 1500000);
}
// are the queries of a batch served in the order of their nearest list (grouped bounds pass, large batches)?
static bool ivf_ordered_mode(const hnswgpu_index *idx, int32_t nq, bool grouped, const StreamTune &tn) {
    const int64_t order_min = tn.finish_order;
    return grouped && order_min > 0 && nq >= order_min && idx->nlist <= kOrderMaxLists;
}
// The home-list pass (stream_kernels.hpp, step 1a): batches in which the lists are home to about a query each or more --
// every home list once through the matrix cores in half precision for all of its queries (ivf_home_kernel) instead of a
// half row per (query, survivor), and the queries' thresholds from there instead of from 64 sampled f32 rows.  Rows of whole
// 128-element steps; from 512 queries and half a query per list (measured on the bench index, on / off: batch
// 128 0.265 / 0.266 ms, 256 0.321 / 0.318, 512 0.383 / 0.399, 1024 0.480 / 0.558); HNSWGPU_TUNE_STREAM_HOME: -1 this rule, 0
// never, 1 whenever the queries are ordered.  Decided BEFORE the routing (whose tail then skips the threshold seed) and again by the scan: one rule.
static bool ivf_home_mode(const hnswgpu_index *idx, int32_t nq, int32_t nprobe, bool grouped, const StreamTune &tn) {
    const int64_t home_env = tn.stream_home;
    const int64_t hstride = (idx->max_list_len + 15) / 16 * 16;
    return ivf_mid_mode(idx, nq, nprobe, tn) && ivf_ordered_mode(idx, nq, grouped, tn) && idx->ld % 128 == 0 &&
           home_env != 0 && tn.mid_slices <= 1 && tn.mid_compact != 0 &&
           (home_env > 0 || (nq >= 512 && 2LL * nq >= idx->nlist)) && static_cast<int64_t>(nq) * hstride * 8 <= (2LL << 30) &&
           (static_cast<int64_t>(nq) / home_group(idx->nch) + std::min<int64_t>(nq, idx->nlist)) * ((idx->max_list_len + 4095) / 4096) < (1LL << 30);
}

// The bounds pass's cut of the lists into work items -- by the routing (which builds the work list of a small batch inside
// its tail's launch) and by the scan: one rule.
struct WorkPlan {
    int64_t chunk_rows;  // rows per work item: whole tiles; enough working workgroups to fill the chip a few times over
    int32_t nchunks;
    bool narrow;         // which epilogue: few queries per probed list -> lane = row
    int qblocks;         // 32-query column blocks per group
    int64_t wbound;      // grouped: items <= sum over lists of ceil(members / 32) * chunks <= (npairs / 32 + nlist) * nchunks
};
static WorkPlan stream_work_plan(const hnswgpu_index *idx, int32_t nq, int32_t nprobe, bool grouped, const StreamTune &tn) {
    WorkPlan p;
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    const bool mid = ivf_mid_mode(idx, nq, nprobe, tn);
    // The largest batches (a list probed by 256 queries and more on average, entries appended without bounds): two 32-query
    // column blocks per group -- every staged row operand meets 64 queries, the lists leave L2 half as often
    const int64_t wide2 = tune(HNSWGPU_TUNE_STREAM_WIDE2, -1);  // -1 that rule, 0 never, 1 whenever the epilogue allows
    const int64_t narrow_env = tune(HNSWGPU_TUNE_STREAM_NARROW, -1);  // A/B: 0 / 1 force
    p.narrow = narrow_env >= 0 ? narrow_env != 0 : npairs < 6LL * idx->nlist;
    // (round 5 built FOUR blocks as well -- the list's rows out of the L2 once per 128 members -- and measured them slower at every
    // batch size, 4096: 0.324 vs 0.281 ms, 16384: 0.862 vs 0.800 (two) / 0.864 (one): it is not the re-streamed rows that bound the
    // pass at these sizes but a matrix instruction's 1 KB query operand out of LDS -- four SIMDs at one instruction per 32 cycles
    // ARE the LDS's 128 bytes per cycle.  1 forces two blocks, 4 four, wherever the wide deferring epilogue runs.)
    p.qblocks = 1;
    if (grouped && mid && !p.narrow && idx->nch <= 4 /* (128 queries' codes of 1024 bytes: 128 KB of LDS) */ && wide2 != 0) {
        if (wide2 == 4) p.qblocks = 4;
        else if (wide2 > 0 || npairs >= 256LL * idx->nlist) p.qblocks = 2;
    }
    // (Also built in round 5 and removed again: the members' codes in REGISTERS -- a 32-member column block per wave, four or
    // eight waves walking the same rows, no LDS operand at all -- on the hope that the CU's L1 would serve the second to eighth
    // reader of a row operand: it does not (batch 4096: 0.386 vs 0.241 ms, 16384: 1.02 vs 0.67).)
    const int64_t mean = ivf_mean_len(idx);
    const int64_t mean_tiles = (mean + kTileRows - 1) / kTileRows, max_tiles = (idx->max_list_len + kTileRows - 1) / kTileRows;
    // (grouped small batches -- a query or two per probed list -- in finer items: batch 32 on the bench index, target 1024 /
    // 2048 / 4096 / 8192: 145.3 / 144.7 / 142.3 / 142.5 us per search)
    const int64_t tgt = tune(HNSWGPU_TUNE_STREAM_WGS, grouped && npairs < 6LL * idx->nlist ? 4096 : 2048);
    // units of work the target is spread over: the lists (or pairs) that have work -- times the GROUPS a list's members form (round
    // 5: a list probed by 128 queries is four items per chunk already; cut into two chunks as well, every item staged its 24 KB of
    // query codes for two row blocks per wave -- whole lists: batch 4096 0.389 -> 0.313 ms, 16384 0.819 -> 0.655 in the diagnostic build)
    int64_t units = grouped ? std::min<int64_t>(npairs, idx->nlist) : npairs;
    if (grouped) {  // (rounded up: 48 members per list are two groups -- batch 1536: 0.169 -> 0.155 ms with whole lists)
        const int64_t per_group = std::max<int64_t>(units, 1) * kTileQ * p.qblocks;
        units *= std::max<int64_t>(1, (npairs + per_group - 1) / per_group);
    }
    const int64_t want = std::max<int64_t>(1, std::min<int64_t>(mean_tiles, (tgt + units - 1) / std::max<int64_t>(units, 1)));
    const int64_t cr = ((mean_tiles + want - 1) / want) * kTileRows, tpc = cr / kTileRows;
    p.chunk_rows = cr;
    p.nchunks = static_cast<int32_t>(std::max<int64_t>(1, (max_tiles + tpc / 2) / tpc));
    p.wbound = (npairs / kTileQ + idx->nlist) * p.nchunks;
    return p;
}

// The list scan as a survivor stream (stream_kernels.hpp): int8 bounds with a running threshold -> compact survivor
// lists -> f32 distances (GEMV order), top-k, ids and distances written by the finish kernel.  The query codes, tau = none
// and empty survivor lists are set up by the routing step.
static int ivf_stream_scan(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe, const int32_t *d_qcnt,
                           const int32_t *d_probes, int32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_gord,
                           const StreamScratch &sc, hipStream_t st) {
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    StreamArgs b;
    memset(&b, 0, sizeof(b));
    int64_t blocks;
    const int32_t *qorder = nullptr;  // large batches: the queries in the order of their nearest list (below)
    const int64_t stride = (static_cast<int64_t>(nprobe) * idx->max_list_len + 3) / 4 * 4;
    const bool grouped = sc.bk_cnt != nullptr;
    const StreamTune &tn = sc.tn;  // (one reading of the tuning table per search: the routing decided by the same values)
    const WorkPlan plan = stream_work_plan(idx, nq, nprobe, grouped, tn);
    const int64_t cr = plan.chunk_rows;
    b.chunk_rows = static_cast<int32_t>(cr);
    b.nchunks = plan.nchunks;
    const bool mid = ivf_mid_mode(idx, nq, nprobe, tn);
    const int64_t order_min = tn.finish_order;
    const bool ordered = d_probes != nullptr && ivf_ordered_mode(idx, nq, grouped, tn);
    const int64_t hstride = (idx->max_list_len + 15) / 16 * 16;
    const int home_gq = home_group(idx->nch);
    bool home = d_probes != nullptr && ivf_home_mode(idx, nq, nprobe, grouped, tn);  // (the routing took the same decision)
    int64_t home_chunk = 256, home_bound = 0;
    if (const int64_t hc = tune(HNSWGPU_TUNE_HOME_CHUNK, 0); hc >= 64) home_chunk = hc / 64 * 64;
    if (home) {
        const int64_t groups = nq / home_gq + std::min<int64_t>(nq, idx->nlist);
        while (home_chunk < 4096 && groups * ((idx->max_list_len + home_chunk - 1) / home_chunk) > 16384) home_chunk *= 2;
        home_bound = groups * ((idx->max_list_len + home_chunk - 1) / home_chunk);
        HG_REQUIRE(home_bound < 2147483647LL, HNSWGPU_ELIMIT, "home-list pass: work list too large");
    }
    HomeDesc *home_desc = nullptr;
    int32_t *home_nit = nullptr;
    uint32_t *home_first = nullptr;  // [nq] entries of a query's list behind the home-list launch
    if (home) {
        HG_TRY(idx->s_home.ensure(sizeof(HomeDesc) * static_cast<size_t>(home_bound) + 64 + sizeof(uint32_t) * static_cast<size_t>(nq)));
        HG_TRY(idx->s_dh.ensure(sizeof(float2) * static_cast<size_t>(nq) * hstride));
        home_desc = idx->s_home.as<HomeDesc>();
        home_nit = reinterpret_cast<int32_t *>(home_desc + home_bound);
        home_first = reinterpret_cast<uint32_t *>(home_nit + 16);
    }
    const bool narrow = plan.narrow;
    const int qblocks = plan.qblocks;
    if (grouped) {
        const int64_t wbound = plan.wbound;
        HG_REQUIRE(wbound < 2147483647LL, HNSWGPU_ELIMIT, "bounds pass work list too large");
        HG_TRY(idx->s_misc2.ensure(sizeof(WorkDesc) * static_cast<size_t>(wbound) + 64));
        WorkDesc *desc = idx->s_misc2.as<WorkDesc>();
        int32_t *nit = reinterpret_cast<int32_t *>(desc + wbound);
        // (large batches: the query order of the half-precision pass and the finish kernel by a second workgroup of this launch)
        size_t olds = 0;
        if (ordered) {
            HG_TRY(idx->s_stats.ensure(sizeof(int32_t) * static_cast<size_t>(nq)));  // (s_ids / s_outd may be the caller's outputs)
            olds = sizeof(int32_t) * (idx->nlist + 1 + 1024);
            if (olds > 32 * 1024) {  // (+ the work list's static 4 KB)
                static bool attr_done[64] = {};
                if (attr_needed(attr_done))
                    HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&ivf_worklist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(sizeof(int32_t) * (kOrderMaxLists + 1 + 1024))));
            }
            qorder = idx->s_stats.as<int32_t>();
        }
        if (!sc.wl_folded) {  // (a small batch's list was built inside the routing tail's launch: worklist_part_wg)
            hipLaunchKernelGGL(ivf_worklist_kernel, dim3(qorder ? 2 : 1), dim3(1024), olds, st, sc.bk_cnt, sc.bk_cap, idx->nlist,
                               idx->d_listoff, cr, b.nchunks, desc, nit, d_probes, nq, idx->s_stats.as<int32_t>(), nprobe,
                               home_desc, home_nit, home_gq, static_cast<int>(home_chunk), kTileQ * qblocks);
            HG_HIP(hipGetLastError());
        }
        idx->bk_dirty = false;  // (the kernel leaves the counters zero)
        b.wi_desc = desc;
        b.nitems = nit;
        b.bk_mem = sc.bk_mem;
        b.bk_cap = sc.bk_cap;
        blocks = (wbound + 7) & ~7LL;
    } else {  // a handful of queries: every (query, list) pair is its own item
        b.pairs = idx->s_pairs.as<Pair>();
        b.npairs = static_cast<int32_t>(npairs);
        blocks = npairs * b.nchunks;
    }
    // survivors per query that fit; a query with more takes the finish kernel's fallback (the plain f32 scan)
    const int64_t cap_env = tune(HNSWGPU_TUNE_STREAM_CAP, 0);  // tests: a tiny list forces the fallback
    int64_t cap = std::min<int64_t>(stride, std::max<int64_t>(4096, 4 * idx->max_list_len));
    if (cap_env > 0) cap = cap_env;
    cap = std::max<int64_t>(cap, 1);
    HG_TRY(idx->s_tile.ensure(sizeof(uint4) * static_cast<size_t>(nq) * cap));
    b.metric = idx->metric;
    b.k = k;
    b.ctile = idx->d_lctile;
    b.cmeta = idx->d_lcmeta;
    b.qcodes = sc.qcodes;
    b.qscal = sc.qscal;
    b.tau = sc.tau;
    b.surv_cnt = sc.surv_cnt;
    b.surv = idx->s_tile.as<uint4>();
    b.cap = cap;
#ifdef HG_DIAG
    b.dbg = g_stream_dbg;  // ablation timing only (results are wrong on purpose): hnswgpu_debug_set_ablation
#else
    b.dbg = 0;
#endif
    b.stamps = g_tile_dbg_buf;  // null outside diagnostic sessions
    b.defer = mid ? 1 : 0;
    if (home) {
        // the home lists first: every row of a query's nearest list through the matrix cores in half precision
        // (ivf_home_kernel), then one workgroup per query takes the k-th smallest upper bound found there as the query's
        // threshold and starts its survivor list with the rows it does not exclude (ivf_home_select_kernel).  The
        // bounds pass below appends nothing for a (query, home list) pair and meets every other list with that threshold;
        // what survives there (the part of a query's cluster that k-means put into a second list, above all) goes through the
        // per-survivor half-precision pass behind it, held against the same threshold.
        HG_REQUIRE(qorder != nullptr, HNSWGPU_EINVAL, "home-list pass without the query order");
        b.home_pairs = idx->s_pairs.as<Pair>();
        b.home_nprobe = nprobe;
        HomeArgs ho;
        memset(&ho, 0, sizeof(ho));
        ho.items = home_desc;
        ho.nitems = home_nit;
        ho.qorder = qorder;
        ho.half = idx->d_lhalf;
        ho.hmeta = idx->d_lhmeta;
        ho.ld = idx->ld;
        ho.Q = d_Q;
        ho.qld = idx->dim;
        ho.dim = idx->dim;
        ho.metric = idx->metric;
        ho.dh = idx->s_dh.as<float2>();
        ho.hstride = hstride;
        HG_TRY(launch_home(ho, home_bound, idx->nch, st));
        HomeSelectArgs hs;
        memset(&hs, 0, sizeof(hs));
        hs.dh = ho.dh;
        hs.hstride = hstride;
        hs.pairs = idx->s_pairs.as<Pair>();
        hs.nq = nq;
        hs.nprobe = nprobe;
        hs.k = k;
        hs.surv = b.surv;
        hs.cap = cap;
        hs.surv_cnt = b.surv_cnt;
        hs.first = home_first;
        hs.tau = b.tau;
        // (large batches: a wave per query, four queries per workgroup -- the per-query chain four times as often per CU)
        const int64_t pqw = tune(HNSWGPU_TUNE_QUERY_WAVES, -1);  // -1 from 2048 queries, 0 never, 1 wherever a wave can serve a query
        if (k <= kWave && pqw != 0 && (pqw > 0 || nq >= 2048))
            hipLaunchKernelGGL(ivf_home_select_wave_kernel, dim3((nq + kNWave - 1) / kNWave), dim3(kWG), 0, st, hs);
        else
            hipLaunchKernelGGL(ivf_home_select_kernel, dim3(nq), dim3(kWG), 0, st, hs);
        HG_HIP(hipGetLastError());
    }
    hipEvent_t e0;
    prof_begin(idx, PROF_IVF_SCAN, st, &e0);
    // which epilogue: few queries per probed list -> lane = row (a list probed by more takes several passes); many -> lane = query
    HG_TRY(launch_stream_bounds(b, blocks, idx->nch, narrow, st, qblocks));
    prof_end(idx, PROF_IVF_SCAN, st, e0);
    // Large batches: a query's survivors are, above all, its nearest list -- and several queries share one.  The queries are
    // taken in the order of their nearest list, a contiguous eighth of that order per XCD, so that the queries which read
    // the same rows run side by side on ONE L2 (each XCD otherwise fetches the list for itself).
    if (!qorder && d_probes && order_min > 0 && nq >= order_min && idx->nlist <= kOrderMaxLists) {  // (ungrouped launches)
        HG_TRY(idx->s_stats.ensure(sizeof(int32_t) * static_cast<size_t>(nq)));  // (s_ids / s_outd may be the caller's outputs)
        const size_t olds = sizeof(int32_t) * (idx->nlist + 1 + 1024);
        if (olds > 48 * 1024) {
            static bool attr_done[64] = {};
            if (attr_needed(attr_done))
                HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_order_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           static_cast<int>(sizeof(int32_t) * (kOrderMaxLists + 1 + 1024))));
        }
        hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(1024), olds, st, d_probes, nq, idx->nlist, idx->s_stats.as<int32_t>(), nprobe);
        HG_HIP(hipGetLastError());
        qorder = idx->s_stats.as<int32_t>();
    }
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    f.prepass = mid ? 1 : 0;  // (lists of more than 128 entries only: what a compacting pass has left is evaluated in one step)
    f.adapt = static_cast<int32_t>(tune(HNSWGPU_TUNE_FINISH_ADAPT, 1));        // A/B
    f.bisect_min = static_cast<int32_t>(tune(HNSWGPU_TUNE_FINISH_BISECT, 1));  // A/B
    f.dbg = g_tile_dbg_buf;
    f.surv = b.surv;
    f.surv_cnt = b.surv_cnt;
    f.tau = b.tau;
    f.cap = cap;
    f.q_cnt = d_qcnt;
    f.pairs = idx->s_pairs.as<Pair>();
    f.nq = nq;
    f.nprobe = nprobe;
    f.k = k;
    // workgroups per query: the chip filled a few times over for small batches; one or two for large ones
    // (behind the half-precision pass a query has little more than k rows left to fetch: fewer workgroups, each of which
    // reads the whole survivor list for the threshold)
    // (measured at batch 32, slices 16 / 32 / 64 / 128: 0.176 / 0.169 / 0.176 / 0.184 ms; one query: 64 is best)
    f.slices = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(64, (mid ? 512 : (nq <= 64 ? 1024 : 2048)) / nq)));
    f.span = nq <= 32 ? 16 : 64;
    if (const int64_t sl = tune(HNSWGPU_TUNE_FINISH_SLICES, 0)) f.slices = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(sl, 256)));  // tuning
    if (const int64_t sp = tune(HNSWGPU_TUNE_FINISH_SPAN, 0)) f.span = sp >= 64 ? 64 : (sp >= 32 ? 32 : 16);
    f.rows = idx->d_lrows;
    f.row_norms = idx->d_lnorms;
    f.ld = idx->ld;
    f.Q = d_Q;
    f.qld = idx->dim;
    f.dim = idx->dim;
    f.metric = idx->metric;
    // Queries one workgroup cannot serve (see ivf_heavy_kernel): when the half-precision pass or the finish kernel gives a
    // query fewer than eight, the queries with more than 4096 survivors -- and the overflowed ones, whose finish is the
    // plain f32 scan of every candidate -- are listed, and 32 x 64 extra workgroups per launch take them in 64 slices.
    const int64_t mid_slices = !mid ? 0 : (qorder ? 1 : std::max<int64_t>(1, std::min<int64_t>(16, 4096 / nq)));
    const bool heavy = tune(HNSWGPU_TUNE_STREAM_HEAVY, 1) != 0 && ((mid && mid_slices < 8) || f.slices < 8);
    constexpr int kHeavySlices = 64;
    // keys per query handed to its last workgroup: the slices' lists -- or, for a short survivor list spread over many
    // workgroups (small batches), a key per survivor (FinishArgs::direct; measured at batch 32: see DESIGN)
    f.direct = f.slices >= 16 && !heavy && k <= kWave ? static_cast<int32_t>(std::max<int64_t>(0, std::min<int64_t>(1024, tune(HNSWGPU_TUNE_FINISH_DIRECT, 1024)))) : 0;
    f.pstride = std::max<int64_t>(static_cast<int64_t>(heavy ? std::max(f.slices, kHeavySlices) : f.slices) * (k <= kWave ? 1 : kNWave) * k, f.direct);
    const size_t keys = static_cast<size_t>(nq) * f.pstride;
    HG_TRY(idx->s_partial.ensure(sizeof(uint64_t) * keys));
    if (heavy) {
        HG_TRY(idx->s_heavy.ensure(sizeof(int32_t) * (2 * static_cast<size_t>(nq) + 4)));
        HeavyArgs ha;
        memset(&ha, 0, sizeof(ha));
        ha.surv_cnt = b.surv_cnt;
        ha.nq = nq;
        ha.cap = static_cast<uint32_t>(cap);
        ha.times_mean = static_cast<uint32_t>(tune(HNSWGPU_TUNE_STREAM_HEAVY_MEAN, 4));
        ha.thr = static_cast<uint32_t>(std::min<int64_t>(cap, tune(HNSWGPU_TUNE_STREAM_HEAVY_MIN, 4096)));
        ha.cnt = idx->s_heavy.as<uint32_t>();
        ha.list = idx->s_heavy.as<int32_t>() + 4;
        if (home) {  // ... and the queries the bounds pass appended more than a few candidates to
            ha.first = home_first;
            ha.few = static_cast<uint32_t>(std::max<int64_t>(0, tn.home_strays));
            ha.todo_cnt = idx->s_heavy.as<uint32_t>() + 1;
            ha.todo = ha.list + nq;
        }
        HG_TRY(launch_heavy(ha, st));
        f.heavy_cnt = ha.cnt;
        f.heavy_list = ha.list;
        f.heavy_slices = kHeavySlices;
    }
    HG_TRY(ensure_counters(idx, nq, st));
    f.partial = idx->s_partial.as<uint64_t>();
    f.done = idx->s_done.as<uint32_t>();
    f.listids = idx->d_listids;
    f.out_ids = d_out_ids;
    f.out_dist = d_out_dist;
    f.out_gord = d_out_gord;
    f.stats = (idx->prof || idx->ivf_calibrating) ? idx->d_rej_stats : nullptr;
    if (idx->zc_flag && !heavy && !idx->ivf_calibrating) {  // a flagged synchronous call: the last query's workgroup tells the caller
        f.host_flag = idx->zc_flag;
        f.flag_val = idx->zc_val;
        f.done_q = idx->s_done.as<uint32_t>() + 2 * idx->s_done_n + 2;
        idx->zc_taken = true;
    }
    f.qorder = f.slices == 1 ? qorder : nullptr;
    f.main_blocks = static_cast<int32_t>(f.qorder ? (static_cast<int64_t>(nq) + 7) / 8 * 8 : static_cast<int64_t>(nq) * f.slices);
    if (mid) {
        MidArgs ma;
        memset(&ma, 0, sizeof(ma));
        ma.surv = b.surv;
        ma.surv_cnt = b.surv_cnt;
        ma.cap = cap;
        ma.nq = nq;
        ma.slices = qorder ? 1 : static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(16, 4096 / nq)));
        if (const int64_t sl = tn.mid_slices) ma.slices = static_cast<int32_t>(std::max<int64_t>(1, std::min<int64_t>(sl, 64)));  // tuning
        ma.qorder = ma.slices == 1 ? qorder : nullptr;
        // one workgroup per query sees the whole list: it also applies the threshold of the upper bounds and compacts the
        // list (up to 4096 entries: 16 KB of LDS -- more would cost the kernel its occupancy; a longer list is left to the finish kernel's own pass)
        ma.tau = b.tau;
        ma.k = k;
        ma.compact = ma.slices == 1 && tn.mid_compact ? static_cast<int32_t>(std::min<int64_t>(cap, 4096)) : 0;
        ma.half = idx->d_lhalf;
        ma.hmeta = idx->d_lhmeta;
        ma.ld = idx->ld;
        ma.Q = d_Q;
        ma.qld = idx->dim;
        ma.dim = idx->dim;
        ma.metric = idx->metric;
        if (heavy) {
            ma.heavy_cnt = f.heavy_cnt;
            ma.heavy_list = f.heavy_list;
            ma.heavy_slices = kHeavySlices;
            ma.main_blocks = static_cast<int32_t>(ma.qorder ? (static_cast<int64_t>(nq) + 7) / 8 * 8 : static_cast<int64_t>(nq) * ma.slices);
        }
        ma.first = home_first;  // (home-list batches: half rows only for what the bounds pass appended)
        if (home && heavy) {
            ma.todo_cnt = idx->s_heavy.as<uint32_t>() + 1;
            ma.todo = f.heavy_list + nq;
            ma.todo_slices = 8;
        }
        ma.first_few = static_cast<int32_t>(tn.home_strays);
        HG_TRY(launch_mid(ma, idx->nch, st));
    }
    return launch_finish(f, idx->nch, st);
}

// (query, list) pairs per list from which the MFMA tile scan serves a cosine / dot batch -- and with it the k-ordered
// summation.  Up to here the GEMV order: the survivor stream (bounds on the int8 rows, f32 distances of the survivors), or
// on a handle without int8 rows one GEMV per pair / the register-row group kernel (same bits all three).
//  * With int8 rows (and k <= kStreamMaxK) the boundary is 48: measured on 1M x 768 / 1024 lists / nprobe 32, survivor
//    stream vs tile scan, end to end: batch 512: 0.55 vs 0.82 ms; 1024: 0.85 vs 0.98; 2048: 1.53 vs 1.53; 4096: 2.92 vs
//    2.67 -- the tile scan reads every probed list in f32 whatever the batch, the stream a quarter of that plus the
//    survivors' rows per query.
//  * Without them it is 12: the f32 GEMV-order scans re-read a list per pair (or per group of queries at the VALU's rate).
//  * With the half-precision rows as well (the default) there is no boundary: the stream's f32 traffic no longer grows
//    with the survivors (~15 rows per query instead of ~970), and it stays ahead at every batch size measured -- same
//    index, stream vs tile scan, end to end: batch 1024: 0.565 vs 1.09 ms; 2048: 0.85 vs 1.51; 4096: 1.33 vs 2.67;
//    8192: 2.26 vs 4.97; 16384: 4.05 vs 9.59.  Per query the tile scan reads 32 lists x 977 rows x 3 KB / 32 queries of a group = 3 MB and is
//    bound by the f32 matrix rate besides; the stream reads a quarter of that in int8 plus 1.5 MB of half rows.
constexpr int64_t kTilePairs = 12, kTilePairsCoded = 48, kTilePairsNever = 1LL << 40;
constexpr int32_t kStreamMaxK = 256;  // largest k the bounds pass serves (larger k: the f32 scans)

// The largest k the stream serves on this handle: its first threshold is the k-th smallest distance of a sample of the
// query's nearest list (stream_seed_rows), and a list of several thousand rows holds several of the data's clusters -- of a
// sample of 256 rows only 256 / (mean length / 1024) can be expected in the query's own.  A larger k gets a threshold from a
// foreign cluster, every survivor list overflows and the search is the f32 scan behind a wasted bounds pass (1M rows in 128
// lists, k = 100, batch 1024: 147 ms against 25): such searches take the f32 paths.
static int32_t ivf_stream_max_k(const hnswgpu_index *idx) {
    const int64_t mean = ivf_mean_len(idx);  // of the WHOLE index: a shard must take the path the unsharded index takes
    const int64_t scale = std::max<int64_t>(1, std::min<int64_t>(8, mean / 1024));
    return static_cast<int32_t>(kStreamMaxK / scale);
}
// can this search go through the survivor stream at all (int8 list rows present and switched on, k within its range)?
static bool ivf_codes_usable(const hnswgpu_index *idx, int32_t k) {
    return idx->d_lctile != nullptr &&
           (idx->rejection_mode == 2 || (idx->rejection_mode == 1 && idx->dim >= 128 && !idx->ivf_stream_off)) &&
           tile_mode() != 0 && k <= ivf_stream_max_k(idx) && tune(HNSWGPU_TUNE_IVF_CODES, 1) > 0;
}
// the boundary of the two summation orders for this handle and k (HNSWGPU_TILE_PAIRS overrides: the parity suite pins it
// at 12 so that its small indexes still reach the tile path)
static int64_t ivf_tile_pairs(const hnswgpu_index *idx, int32_t k) {
    const int64_t e = tune(HNSWGPU_TUNE_TILE_PAIRS, 0);
    if (e > 0) return e;
    if (!ivf_codes_usable(idx, k)) return kTilePairs;
    return idx->d_lhalf ? kTilePairsNever : kTilePairsCoded;
}

static int ivf_search_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                              int32_t *d_out_ids, float *d_out_dist, int32_t *d_out_probes, hipStream_t st,
                              const int32_t *d_given_probes = nullptr, uint32_t *d_out_gord = nullptr);

// What do the int8 bounds separate on THIS handle's rows?  On data with cluster structure they leave a query its home
// cluster (3 % of the candidates on the bench index); on rows without any -- i.i.d. gaussian: distances concentrate, every
// candidate lies within the bounds' width of the k-th -- they leave everything, and the stream is the f32 scan behind a
// wasted bounds pass (1M x 768 gaussian, batch 1024: 17 ms against the tile scan's 1.0).  Mode 1 (automatic) therefore
// measures once per set of lists, at the first search: 32 list rows as queries, k = 10, up to 32 probes, the int8 pass
// alone; more than a quarter of the candidates fetched in f32 switches the stream off for the handle (mode 2 forces it,
// mode 0 never builds it).  The decision depends on the rows and the lists only: the same handle always takes the same path.
static int ivf_calibrate(hnswgpu_index *idx, hipStream_t st) {
    if (idx->rejection_mode != 1 || !idx->d_lctile || idx->dim < 128 || idx->n < 4096 || tune(HNSWGPU_TUNE_IVF_CALIBRATE, 1) == 0) {
        idx->ivf_calibrated = true;  // nothing to measure in this mode
        return 0;
    }
    // one allocation [queries | ids | distances | counters], freed on every path; the handle counts as calibrated only
    // once the measurement has succeeded (a transient failure is retried by the next search); the user's profiling state
    // and events are not touched (the counters go through ivf_calibrating, not through idx->prof).  The first search of a
    // mode-1 handle is therefore synchronous once -- also through the *_dev entry points.
    const int32_t cq = 32, ck = 10, cp = std::min(32, idx->nlist);
    const size_t b_q = sizeof(float) * cq * idx->dim, b_i = sizeof(int32_t) * cq * ck, b_d = sizeof(float) * cq * ck;
    char *buf = nullptr;
    HG_HIP(hipMalloc(reinterpret_cast<void **>(&buf), b_q + b_i + b_d + 2 * sizeof(unsigned long long)));
    float *d_q = reinterpret_cast<float *>(buf);
    int32_t *d_i = reinterpret_cast<int32_t *>(buf + b_q);
    float *d_d = reinterpret_cast<float *>(buf + b_q + b_i);
    unsigned long long *d_st = reinterpret_cast<unsigned long long *>(buf + b_q + b_i + b_d), got[2] = {0, 0};
    unsigned long long *const old_stats = idx->d_rej_stats;
    const int rc = [&]() -> int {
        const int64_t step = idx->n / cq;
        for (int i = 0; i < cq; i++)  // evenly spaced list rows as queries
            HG_HIP(hipMemcpyAsync(d_q + static_cast<int64_t>(i) * idx->dim, idx->d_lrows + (static_cast<int64_t>(i) * step + step / 2) * idx->ld,
                                  sizeof(float) * idx->dim, hipMemcpyDeviceToDevice, st));
        HG_HIP(hipMemsetAsync(d_st, 0, sizeof(got), st));
        idx->d_rej_stats = d_st;
        idx->ivf_calibrating = true;
        const int r = ivf_search_enqueue(idx, d_q, cq, ck, cp, d_i, d_d, nullptr, st);
        idx->ivf_calibrating = false;
        idx->d_rej_stats = old_stats;
        HG_TRY(r);
        HG_HIP(hipMemcpyAsync(got, d_st, sizeof(got), hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
        return 0;
    }();
    idx->ivf_calibrating = false;
    idx->d_rej_stats = old_stats;
    if (rc != 0) (void)hipStreamSynchronize(st);
    (void)hipFree(buf);
    if (rc != 0) return rc;
    // [0] = f32 rows fetched, [1] = candidates
    idx->ivf_stream_off = got[1] >= static_cast<unsigned long long>(cq) * 256 && got[0] * 4 > got[1];
    idx->ivf_calibrated = true;
    return 0;
}

static int ivf_search_enqueue(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                              int32_t *d_out_ids, float *d_out_dist, int32_t *d_out_probes, hipStream_t st,
                              const int32_t *d_given_probes, uint32_t *d_out_gord) {
    if (!idx->ivf_calibrated && !idx->ivf_calibrating) HG_TRY(ivf_calibrate(idx, st));
    const int64_t *glistoff = idx->d_glistoff ? idx->d_glistoff : idx->d_listoff;
    if (nprobe > idx->nlist && !d_given_probes) nprobe = idx->nlist;
    // 1. centroid routing (:261-269): top-nprobe of the centroid table, stable on the centroid index
    ScanArgs a;
    memset(&a, 0, sizeof(a));
    HG_TRY(idx->s_pairs.ensure(sizeof(Pair) * static_cast<size_t>(nq) * nprobe));
    const int64_t npairs = static_cast<int64_t>(nq) * nprobe;
    // tiled (MFMA) list scan once the batch holds more than kTilePairs (query, list) pairs per list: the GEMV scan streams a
    // list once per pair (pairs of one list side by side on one L2, see ScanArgs::order), the tile scan once per
    // group of <= 32 pairs but at ~5 TB/s.  Measured on 1M x 768 / 1024 lists / nprobe 32 (tools/ivf_batch_time.py,
    // GEMV vs tiled, end to end): batch 32: 0.42 vs 0.56 ms; 48: 0.54 vs 0.61; 64: 0.64 vs 0.64; 80: 0.72 vs 0.65;
    // 96: 0.82 vs 0.71; 128: 1.03 vs 0.71.
    const int tm = tile_mode();
    const int code_env = static_cast<int>(tune(HNSWGPU_TUNE_IVF_CODES, 1));  // 0 = never (A/B), N > 0 = from N queries per batch
    // the survivor stream (stream_kernels.hpp): k up to a tile chunk's rows can get a threshold from one chunk
    const bool codes_ok = idx->d_lctile != nullptr &&
                          (idx->rejection_mode == 2 || (idx->rejection_mode == 1 && idx->dim >= 128 && !idx->ivf_stream_off)) &&
                          code_env > 0 && nq >= code_env && tm != 0 && k <= ivf_stream_max_k(idx);
    // (Euclidean has one arithmetic at every batch size -- its "tile" path is the register-row group kernel -- so the
    // bounds pipeline below serves all its batches: batch 1024 at 1M x 768: 4.8 -> 2.8 ms)
    const int64_t tile_pairs = ivf_tile_pairs(idx, k);
    const bool use_tile = tile_path_ok(idx) && tm != 0 && (tm == 1 || npairs > tile_pairs * idx->nlist) &&
                          !(idx->metric == METRIC_L2 && codes_ok && tm != 1);
    // Between the fused small-batch path and the tile scan: bounds on the int8 list rows first, f32 distances -- the
    // GEMV order, so the bits of this regime are unchanged -- only for the candidates that can still be among the k
    // nearest (code_kernels.hpp).  Without the int8 rows (hnswgpu_set_rejection_test mode 0) the same bits come from
    // the f32 scans below: one GEMV per pair, or the register-row group kernel from 1.5 pairs per list.
    const bool use_code = !use_tile && codes_ok;
    int32_t *probes_buf = d_out_probes;
    int32_t *qcnt_buf = nullptr;
    // GEMV scan with enough pairs for lists to be probed twice: run the pairs in list order (see ScanArgs::order).
    // Below half a pair per list there is next to nothing to share and the sort's ~10 us would be all cost.
    const int order_mode = static_cast<int>(tune(HNSWGPU_TUNE_SCAN_ORDER, 1));  // 0 = never (A/B)
    // A handful of queries: the launches around the list scan (select, probe table, merge, decode, copy) cost as much as
    // the scan, so their work is folded into the routing and scan kernels' last workgroups (two launches instead of
    // seven; 1M x 768, one query: 99 -> 88 us per call, 78 -> 74 us back to back).  Larger batches keep the separate
    // launches: their merge runs one workgroup per query in parallel, and a tail would only lengthen the scan kernel.
    const int fused_env = static_cast<int>(tune(HNSWGPU_TUNE_IVF_FUSED, 1));  // 0 = never, 1 = small batches (default), 2 = every GEMV-path batch
    const bool fused_mode = !use_code && (fused_env == 2 || (fused_env == 1 && nq <= 8));
    // Without int8 rows, from 1.5 pairs per list up to that boundary: the register-row group kernel (l2_kernels.hpp) fetches a list once for
    // all the queries probing it and keeps the GEMV summation order, so the results stay bit-identical to the GEMV
    // scan's (the contract up to kTilePairs pairs per list) while the second and third readers of a list cost no traffic.
    // Same index, GEMV vs group, list-scan kernel / end to end: batch 32: 0.356 / 0.423 vs 0.331 / 0.423 ms;
    // 48: 0.464 / 0.531 vs 0.431 / 0.527; 64: 0.572 / 0.643 vs 0.488 / 0.589 (about 885 distinct lists x 3 MB at
    // ~5.5 TB/s).  The three extra launches (histogram, plan, scatter) cost what the kernel gains below 1.5 pairs per list.
    const int group_env = static_cast<int>(tune(HNSWGPU_TUNE_IVF_GROUP, 1));  // 0 = never (A/B), 2 = from half a pair per list
    const bool use_group = !use_tile && !use_code && !fused_mode && group_env && tm != 0 && idx->dim <= kL2MaxDim &&
                           (group_env == 2 ? npairs * 2 >= idx->nlist : npairs * 2 >= 3LL * idx->nlist);
    const bool use_order = !use_tile && !use_group && !use_code && order_mode && idx->nlist <= kOrderMaxLists &&
                           npairs * 2 >= idx->nlist && npairs <= (1 << 22);
    int32_t *order_buf = nullptr;
    if (use_tile || use_group || use_code || use_order) {
        HG_TRY(idx->s_grp.ensure(sizeof(int32_t) * (2 * npairs + nq + 16)));
        if (!probes_buf) probes_buf = idx->s_grp.as<int32_t>();
        qcnt_buf = use_tile || use_group || use_code ? idx->s_grp.as<int32_t>() + npairs : nullptr;
        order_buf = idx->s_grp.as<int32_t>() + npairs + nq + 16;
    }
    if (use_tile || use_group) HG_TRY(pad_queries(idx, d_Q, idx->dim, nq, st));
    // survivor stream: query codes, thresholds and survivor counters live in s_qp / s_qn (sized here: the routing below may
    // still use both for padded queries and norms, and must not move s_qn afterwards)
    StreamScratch sc;
    memset(&sc, 0, sizeof(sc));
    sc.tn.read();
    const int stream_route_max = static_cast<int>(tune(HNSWGPU_TUNE_STREAM_ROUTE, 12));  // largest batch routed by the one-launch routing kernel
    const int stream_group_min = static_cast<int>(tune(HNSWGPU_TUNE_STREAM_GROUP, 5));    // queries from which the bounds pass groups the pairs by list
    if (use_code) {
        HG_TRY(stream_scratch(idx, nq, sc));
        if (nq >= stream_group_min) HG_TRY(stream_buckets(idx, nq, nprobe, sc, st));  // (zeroes the per-list counters)
    }
    bool codes_done = false;
    // the home-list pass of large batches (ivf_home_mode): the routing's tail leaves the thresholds to it
    const bool home_mode = use_code && sc.bk_cnt != nullptr && probes_buf != nullptr && ivf_home_mode(idx, nq, nprobe, true, sc.tn);
    // small grouped batches: the bounds pass's work list by extra workgroups of the routing tail's launch (larger ones order
    // their queries in a second workgroup of ivf_worklist_kernel: they keep that launch)
    WorklistArgs wl;
    memset(&wl, 0, sizeof(wl));
    const bool fold_wl = use_code && sc.bk_cnt != nullptr && !d_given_probes && !ivf_ordered_mode(idx, nq, true, sc.tn) &&
                         tune(HNSWGPU_TUNE_WORKLIST_FOLD, 1) != 0 && static_cast<int64_t>(nq) * idx->nlist <= (64LL << 20);
    if (fold_wl) {
        const WorkPlan plan = stream_work_plan(idx, nq, nprobe, true, sc.tn);
        HG_REQUIRE(plan.wbound < 2147483647LL, HNSWGPU_ELIMIT, "bounds pass work list too large");
        HG_TRY(idx->s_misc2.ensure(sizeof(WorkDesc) * static_cast<size_t>(plan.wbound) + 64));
        HG_TRY(ensure_counters(idx, nq, st));
        wl.bk_cnt = sc.bk_cnt;
        wl.bk_cap = sc.bk_cap;
        wl.nlist = idx->nlist;
        wl.list_off = idx->d_listoff;
        wl.chunk_rows = plan.chunk_rows;
        wl.max_chunks = plan.nchunks;
        wl.tq = kTileQ * plan.qblocks;
        wl.desc = idx->s_misc2.as<WorkDesc>();
        wl.nitems = reinterpret_cast<int32_t *>(wl.desc + plan.wbound);
        wl.surv_cnt = sc.surv_cnt;
        wl.nq = nq;
    }
    // queries from which a GEMV-order batch routes through the group kernel
    // (Euclidean 1M x 768: 512 queries 0.85 vs 0.87 ms (GEMV vs group), 1024: 1.50 vs 1.43, 4096: 5.08 vs 4.70)
    const int route_group_min = static_cast<int>(tune(HNSWGPU_TUNE_ROUTE_GROUP, 1024));
    if (d_given_probes) {  // caller-chosen lists (the :turbo mode's random partitions, :271-272); -1 = none
        hipLaunchKernelGGL(probe_pairs_kernel, dim3((nq + kNWave - 1) / kNWave), dim3(kWG), 0, st,
                           reinterpret_cast<const uint32_t *>(d_given_probes), nq, nprobe, idx->d_listoff, glistoff,
                           idx->s_pairs.as<Pair>(), probes_buf, qcnt_buf, sc.bk_cnt, sc.bk_mem, sc.bk_cap, sc.surv_cnt);
        HG_HIP(hipGetLastError());
    } else {
    a.rows = idx->d_cent;
    a.row_norms = idx->d_cnorms;
    a.ld = idx->ld;
    a.nrows_all = idx->nlist;
    a.Q = d_Q;
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.k = nprobe;
    a.role = ROLE_ROUTE;
    if (!use_tile && (fused_mode || (use_code && nq <= stream_route_max)) && static_cast<int64_t>(nq) * idx->nlist <= (64LL << 20)) {
        // small batches: distances to the centroids, the choice of the nprobe nearest and the probe table in ONE launch
        // (for the survivor stream also the query codes, tau = none and the empty survivor lists)
        RouteStream rs = {sc.qcodes, sc.qscal, sc.tau, sc.surv_cnt, k, sc.bk_cnt, sc.bk_mem, sc.bk_cap, home_mode ? 1 : 0, fold_wl ? &wl : nullptr};
        HG_TRY(launch_ivf_route(idx, d_Q, nq, nprobe, idx->s_pairs.as<Pair>(), probes_buf, qcnt_buf, st, use_code ? &rs : nullptr));
        codes_done = use_code;
        sc.wl_folded = use_code && fold_wl;
    } else if (use_code && static_cast<int64_t>(nq) * idx->nlist <= (64LL << 20)) {
        // survivor stream, larger batches: the centroid distances by a pass that serves a group of queries per fetch of a
        // centroid row (the GEMV order, same bits), then ONE launch for everything else of the routing -- select, probe
        // table, pairs filed by list, query codes, first thresholds
        RouteStream rs = {sc.qcodes, sc.qscal, sc.tau, sc.surv_cnt, k, sc.bk_cnt, sc.bk_mem, sc.bk_cap, home_mode ? 1 : 0, fold_wl ? &wl : nullptr};
        HG_TRY(launch_ivf_route(idx, d_Q, nq, nprobe, idx->s_pairs.as<Pair>(), probes_buf, qcnt_buf, st, &rs, true));
        codes_done = true;
        sc.wl_folded = fold_wl;
    } else {
    if (use_tile)  // every query against the centroid table on the tile kernel as well
        HG_TRY(tile_topk_all(idx, idx->s_qp.as<float>(), idx->s_qn.as<float>(), nq, idx->d_cent, idx->d_cnorms, idx->nlist,
                             nprobe, st, -1));
    else if (nq >= route_group_min && idx->dim <= kL2MaxDim && tm != 0) {
        // a large batch in the GEMV order (Euclidean): the centroid table once per group of 32 queries (register-row group
        // kernel) instead of once per query, the same bits
        if (!use_group) HG_TRY(pad_queries(idx, d_Q, idx->dim, nq, st));
        HG_TRY(tile_topk_all(idx, idx->s_qp.as<float>(), idx->s_qn.as<float>(), nq, idx->d_cent, idx->d_cnorms, idx->nlist,
                             nprobe, st, -1, true));
    } else if (static_cast<int64_t>(nq) * idx->nlist <= (64LL << 20))  // dense [nq][nlist] distances + select
        HG_TRY(scan_dense_topk(idx, a, nq, idx->nlist, st));
    else
        HG_TRY(scan_topk(idx, a, nq, 1, idx->nlist, st, -1));
    hipLaunchKernelGGL(probe_pairs_kernel, dim3((nq + kNWave - 1) / kNWave), dim3(kWG), 0, st, idx->s_ord.as<uint32_t>(), nq,
                       nprobe, idx->d_listoff, glistoff, idx->s_pairs.as<Pair>(), probes_buf, qcnt_buf, sc.bk_cnt, sc.bk_mem, sc.bk_cap,
                       sc.surv_cnt);
    HG_HIP(hipGetLastError());
    }
    }
    if (use_code) {
        // bounds on the int8 list rows with a running threshold, f32 distances -- the GEMV order, so the bits of this regime
        // are unchanged -- only for the survivors; the finish kernel writes ids and distances
        if (!codes_done) {  // what the fused routing kernel does in its tail: codes, empty survivor lists, first thresholds
            HG_TRY(stream_scratch(idx, nq, sc));  // (s_qp may have moved under the padded queries of the routing)
            PrepArgs pa;
            memset(&pa, 0, sizeof(pa));
            pa.Q = d_Q;
            pa.qld = idx->dim;
            pa.dim = idx->dim;
            pa.metric = idx->metric;
            pa.nq = nq;
            pa.nprobe = nprobe;
            pa.k = k;
            pa.seed_rows = stream_seed_rows(nq, idx->ivf_n_global > 0 ? idx->ivf_n_global : idx->n, idx->nlist);
            pa.home = home_mode ? 1 : 0;
            pa.pairs = idx->s_pairs.as<Pair>();
            pa.qcnt = qcnt_buf;
            pa.rows = idx->d_lrows;
            pa.row_norms = idx->d_lnorms;
            pa.ld = idx->ld;
            if (idx->d_lhalf && tune(HNSWGPU_TUNE_SEED_HALF, 1) != 0) {
                pa.half = idx->d_lhalf;
                pa.hmeta = idx->d_lhmeta;
            }
            pa.qcodes = sc.qcodes;
            pa.qscal = sc.qscal;
            pa.tau = sc.tau;
            HG_TRY(launch_query_prep(pa, idx->nch, st));
        }
        return ivf_stream_scan(idx, d_Q, nq, k, nprobe, qcnt_buf, probes_buf, d_out_ids, d_out_dist, d_out_gord, sc, st);
    }
    // 2. scan the probed lists (:217-234) and merge (:291-294)
    memset(&a, 0, sizeof(a));
    a.rows = idx->d_lrows;
    a.row_norms = idx->d_lnorms;
    a.ld = idx->ld;
    a.Q = d_Q;
    a.qld = idx->dim;
    a.dim = idx->dim;
    a.metric = idx->metric;
    a.pairs = idx->s_pairs.as<Pair>();
    a.k = k;
    a.role = ROLE_LIST_SCAN;
    if (use_tile || use_group) {
        HG_TRY(ivf_tile_scan(idx, d_Q, nq, k, nprobe, probes_buf, qcnt_buf, st, use_group));
    } else {
        if (use_order) {
            const size_t olds = sizeof(int32_t) * (idx->nlist + 1 + 1024);
            if (olds > 48 * 1024) {  // once per device, for the largest histogram the kernel supports
                static bool attr_done[64] = {};
                if (attr_needed(attr_done))
                    HG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&pair_order_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               static_cast<int>(sizeof(int32_t) * (kOrderMaxLists + 1 + 1024))));
            }
            hipLaunchKernelGGL(pair_order_kernel, dim3(1), dim3(1024), olds, st, probes_buf, static_cast<int>(npairs),
                               idx->nlist, order_buf);
            HG_HIP(hipGetLastError());
            a.order = order_buf;
            // all the pairs of a list in one run (one XCD) while a run stays a small part of an XCD's share
            a.run = 8;
            while (a.run < 64 && static_cast<int64_t>(a.run) * idx->nlist < npairs) a.run *= 2;
        }
        if (fused_mode) {
            // the scan's last workgroup per query merges the partial lists, maps the winners to row ids and writes the
            // results: no merge, decode or copy launch behind the scan
            a.listids = idx->d_listids;
            a.out_ids = d_out_ids;
            a.out_dist = d_out_dist;
            a.out_gord = d_out_gord;
            return scan_fused(idx, a, nq, nprobe, idx->max_list_len, ivf_mean_len(idx), st,
                              PROF_IVF_SCAN);
        }
        HG_TRY(scan_topk(idx, a, nq, nprobe, idx->max_list_len, st, PROF_IVF_SCAN,
                         ivf_mean_len(idx)));
    }
    int64_t cnt = static_cast<int64_t>(nq) * k;
    hipLaunchKernelGGL(ivf_decode_kernel, dim3(static_cast<unsigned>((cnt + 255) / 256)), dim3(256), 0, st,
                       idx->s_ord.as<uint32_t>(), nq, k, idx->s_pairs.as<Pair>(), nprobe, idx->d_listids, d_out_ids,
                       d_out_gord);
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpyAsync(d_out_dist, idx->s_dist.p, sizeof(float) * cnt, hipMemcpyDeviceToDevice, st));
    return 0;
}

}  // namespace hg

using namespace hg;

extern "C" {

static int set_ivf_impl(hnswgpu_index *idx, const float *centroids, int32_t nlist, const int64_t *list_off,
                        const int32_t *list_ids, const int64_t *global_len) {
    HG_REQUIRE(idx && centroids && list_off && (list_ids || idx->n == 0), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(nlist >= 1, HNSWGPU_EINVAL, "nlist must be >= 1");
    HG_TRY(validate_lists(idx->n, nlist, list_off, list_ids));
    std::vector<int64_t> goff;
    if (global_len) {
        goff.assign(static_cast<size_t>(nlist) + 1, 0);
        for (int l = 0; l < nlist; l++) {
            HG_REQUIRE(global_len[l] >= list_off[l + 1] - list_off[l], HNSWGPU_EINVAL,
                       "list %d: global length %lld < the %lld rows this shard holds", l, (long long)global_len[l],
                       (long long)(list_off[l + 1] - list_off[l]));
            goff[l + 1] = goff[l] + global_len[l];
        }
        HG_REQUIRE(goff[nlist] < 0xffffffffLL, HNSWGPU_ELIMIT, "the whole index must hold fewer than 2^32 rows");
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    // every earlier call on this handle is ordered before `st` by begin_call (the small HNSW searches on the slot streams
    // are waited for as well: lists that alias the base rows share them with the traversal): nothing can still read the
    // lists about to be freed (no device-wide synchronisation: other handles keep running)
    HG_TRY(quiesce(idx, st));
    free_ivf(idx);
    HG_TRY(alloc_centroids(idx, nlist));
    HG_HIP(hipMemsetAsync(idx->d_cent, 0, sizeof(float) * nlist * idx->ld, st));
    HG_HIP(hipMemcpy2DAsync(idx->d_cent, sizeof(float) * idx->ld, centroids, sizeof(float) * idx->dim,
                            sizeof(float) * idx->dim, nlist, hipMemcpyHostToDevice, st));
    HG_TRY(launch_norms(idx->nch, idx->d_cent, idx->ld, nlist, idx->d_cnorms, st));
    if (global_len) {
        HG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_glistoff), sizeof(int64_t) * (nlist + 1)));
        HG_HIP(hipMemcpyAsync(idx->d_glistoff, goff.data(), sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
    }
    HG_TRY(install_lists(idx, nlist, list_off, list_ids, st));  // synchronises: goff may go out of scope
    idx->ivf_n_global = idx->n;
    if (global_len) {
        idx->h_glistlen.assign(global_len, global_len + nlist);
        idx->ivf_n_global = goff[nlist];
    }
    idx->h_cent.assign(centroids, centroids + static_cast<size_t>(nlist) * idx->dim);
    return 0;
}

int hnswgpu_set_ivf(hnswgpu_index *idx, const float *centroids, int32_t nlist, const int64_t *list_off,
                    const int32_t *list_ids) {
    return set_ivf_impl(idx, centroids, nlist, list_off, list_ids, nullptr);
}

int hnswgpu_set_ivf_shard(hnswgpu_index *idx, const float *centroids, int32_t nlist, const int64_t *list_off,
                          const int32_t *list_ids, const int64_t *global_list_len) {
    HG_REQUIRE(global_list_len, HNSWGPU_EINVAL, "global_list_len is null");
    return set_ivf_impl(idx, centroids, nlist, list_off, list_ids, global_list_len);
}

int hnswgpu_get_ivf(const hnswgpu_index *idx, float *centroids, int64_t *list_off, int32_t *list_ids) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists");
    if (centroids) memcpy(centroids, idx->h_cent.data(), sizeof(float) * idx->h_cent.size());
    if (list_off) memcpy(list_off, idx->h_listoff.data(), sizeof(int64_t) * idx->h_listoff.size());
    if (list_ids && !idx->h_listids.empty())
        memcpy(list_ids, idx->h_listids.data(), sizeof(int32_t) * idx->h_listids.size());
    return 0;
}

int hnswgpu_kmeans_assign(hnswgpu_index *idx, const float *centroids, int32_t nlist, int32_t *out_assign,
                          float *out_dist) {
    HG_REQUIRE(idx && centroids && (out_assign || idx->n == 0), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(nlist >= 1, HNSWGPU_EINVAL, "nlist must be >= 1");
    if (idx->n == 0) return 0;
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(idx->s_misc.ensure(sizeof(float) * nlist * idx->ld));
    HG_TRY(idx->s_misc2.ensure(sizeof(float) * nlist));
    HG_HIP(hipMemsetAsync(idx->s_misc.p, 0, sizeof(float) * nlist * idx->ld, st));
    HG_HIP(hipMemcpy2DAsync(idx->s_misc.p, sizeof(float) * idx->ld, centroids, sizeof(float) * idx->dim,
                            sizeof(float) * idx->dim, nlist, hipMemcpyHostToDevice, st));
    HG_TRY(launch_norms(idx->nch, idx->s_misc.as<float>(), idx->ld, nlist, idx->s_misc2.as<float>(), st));
    HG_TRY(assign_enqueue(idx, idx->s_misc.as<float>(), idx->s_misc2.as<float>(), nlist, st));
    HG_HIP(hipMemcpyAsync(out_assign, idx->s_ord.p, sizeof(int32_t) * idx->n, hipMemcpyDeviceToHost, st));
    if (out_dist) HG_HIP(hipMemcpyAsync(out_dist, idx->s_dist.p, sizeof(float) * idx->n, hipMemcpyDeviceToHost, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_list_means(hnswgpu_index *idx, int32_t nlist, const int64_t *list_off, const int32_t *list_ids,
                       float *out_centroids) {
    HG_REQUIRE(idx && list_off && (list_ids || idx->n == 0) && out_centroids, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(nlist >= 1, HNSWGPU_EINVAL, "nlist must be >= 1");
    HG_TRY(validate_lists(idx->n, nlist, list_off, list_ids));
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(idx->s_misc.ensure(sizeof(int64_t) * (nlist + 1)));
    HG_TRY(idx->s_misc2.ensure(sizeof(int32_t) * std::max<int64_t>(idx->n, 1)));
    HG_TRY(idx->s_tile.ensure(sizeof(float) * static_cast<size_t>(nlist) * idx->ld));
    HG_HIP(hipMemsetAsync(idx->s_tile.p, 0, sizeof(float) * static_cast<size_t>(nlist) * idx->ld, st));  // empty list -> zero vector
    HG_HIP(hipMemcpyAsync(idx->s_misc.p, list_off, sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
    if (idx->n > 0)
        HG_HIP(hipMemcpyAsync(idx->s_misc2.p, list_ids, sizeof(int32_t) * idx->n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(centroid_mean_kernel, dim3(nlist), dim3(kWG), 0, st, idx->d_base, idx->ld, idx->dim,
                       idx->s_misc.as<int64_t>(), idx->s_misc2.as<int32_t>(), idx->s_tile.as<float>());
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpy2DAsync(out_centroids, sizeof(float) * idx->dim, idx->s_tile.p, sizeof(float) * idx->ld,
                            sizeof(float) * idx->dim, nlist, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_list_sums(hnswgpu_index *idx, int32_t nlist, const int64_t *list_off, const int32_t *list_ids,
                      double *out_sums) {
    HG_REQUIRE(idx && list_off && (list_ids || idx->n == 0) && out_sums, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(nlist >= 1, HNSWGPU_EINVAL, "nlist must be >= 1");
    HG_TRY(validate_lists(idx->n, nlist, list_off, list_ids));
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    const size_t sbytes = sizeof(double) * static_cast<size_t>(nlist) * idx->dim;
    HG_TRY(idx->s_misc.ensure(sizeof(int64_t) * (nlist + 1)));
    HG_TRY(idx->s_misc2.ensure(sizeof(int32_t) * std::max<int64_t>(idx->n, 1)));
    HG_TRY(idx->s_tile.ensure(sbytes));
    HG_HIP(hipMemcpyAsync(idx->s_misc.p, list_off, sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
    if (idx->n > 0)
        HG_HIP(hipMemcpyAsync(idx->s_misc2.p, list_ids, sizeof(int32_t) * idx->n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(list_sum_kernel, dim3(nlist), dim3(kWG), 0, st, idx->d_base, idx->ld, idx->dim,
                       idx->s_misc.as<int64_t>(), idx->s_misc2.as<int32_t>(), idx->s_tile.as<double>());
    HG_HIP(hipGetLastError());
    HG_HIP(hipMemcpyAsync(out_sums, idx->s_tile.p, sbytes, hipMemcpyDeviceToHost, st));
    HG_TRY(end_call(idx, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

int hnswgpu_kmeanspp(hnswgpu_index *idx, int32_t nlist, int64_t seed, int32_t *out_rows) {
    HG_REQUIRE(idx && out_rows, HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(nlist >= 1 && idx->n >= 1, HNSWGPU_EINVAL, "need nlist >= 1 and a non-empty index");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    std::vector<int32_t> chosen;
    HG_TRY(begin_call(idx, idx->stream));
    HG_TRY(kmeanspp_device(idx, nlist, seed, chosen, idx->stream));
    memcpy(out_rows, chosen.data(), sizeof(int32_t) * nlist);
    return 0;
}

int hnswgpu_ivf_build(hnswgpu_index *idx, int32_t nlist, int32_t max_iter, int64_t seed) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(nlist >= 1 && max_iter >= 0, HNSWGPU_EINVAL, "bad nlist / max_iter");
    HG_REQUIRE(idx->n >= 1, HNSWGPU_ESTATE, "cannot partition an empty index");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    const int64_t n = idx->n;
    HG_TRY(quiesce(idx, st));  // see set_ivf_impl: retires every earlier call on this handle, and only those
    free_ivf(idx);
    std::vector<int32_t> chosen;
    HG_TRY(kmeanspp_device(idx, nlist, seed, chosen, st));
    HG_TRY(alloc_centroids(idx, nlist));
    for (int c = 0; c < nlist; c++)  // centroids start as copies of data rows (:39,:57)
        HG_HIP(hipMemcpyAsync(idx->d_cent + static_cast<int64_t>(c) * idx->ld,
                              idx->d_base + static_cast<int64_t>(chosen[c]) * idx->ld, sizeof(float) * idx->ld,
                              hipMemcpyDeviceToDevice, st));
    HG_TRY(launch_norms(idx->nch, idx->d_cent, idx->ld, nlist, idx->d_cnorms, st));
    std::vector<uint32_t> assign(static_cast<size_t>(n));
    std::vector<int64_t> off;
    std::vector<int32_t> ids;
    HG_TRY(idx->s_misc.ensure(sizeof(int64_t) * (nlist + 1)));
    HG_TRY(idx->s_misc2.ensure(sizeof(int32_t) * n));
    for (int it = 0; it <= max_iter; it++) {  // max_iter Lloyd passes (:100-117) + the final assignment (:120-124)
        HG_TRY(assign_enqueue(idx, idx->d_cent, idx->d_cnorms, nlist, st));
        HG_HIP(hipMemcpyAsync(assign.data(), idx->s_ord.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, st));
        HG_HIP(hipStreamSynchronize(st));
        lists_from_assign(assign.data(), n, nlist, off, ids);
        if (it == max_iter) break;
        HG_HIP(hipMemcpyAsync(idx->s_misc.p, off.data(), sizeof(int64_t) * (nlist + 1), hipMemcpyHostToDevice, st));
        HG_HIP(hipMemcpyAsync(idx->s_misc2.p, ids.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(centroid_mean_kernel, dim3(nlist), dim3(kWG), 0, st, idx->d_base, idx->ld, idx->dim,
                           idx->s_misc.as<int64_t>(), idx->s_misc2.as<int32_t>(), idx->d_cent);
        HG_HIP(hipGetLastError());
        HG_TRY(launch_norms(idx->nch, idx->d_cent, idx->ld, nlist, idx->d_cnorms, st));
        HG_HIP(hipStreamSynchronize(st));
    }
    HG_TRY(install_lists(idx, nlist, off.data(), ids.data(), st));
    HG_TRY(download_centroids(idx, st));
    return 0;
}

static int check_ivf_args(const hnswgpu_index *idx, const void *Q, int32_t nq, int32_t k, int32_t nprobe,
                          const void *ids, const void *dist) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    HG_REQUIRE(nq >= 0 && k >= 1 && nprobe >= 1, HNSWGPU_EINVAL, "need nq >= 0, k >= 1, nprobe >= 1");
    HG_REQUIRE(k <= 1024 && nprobe <= 1024, HNSWGPU_ELIMIT, "k / nprobe > 1024 is not supported");
    HG_REQUIRE(nq == 0 || (Q && ids && dist), HNSWGPU_EINVAL, "null argument");
    HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists (call hnswgpu_ivf_build / hnswgpu_set_ivf)");
    return 0;
}

int hnswgpu_ivf_search_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                           int32_t *d_out_ids, float *d_out_dist, void *stream) {
    HG_TRY(check_ivf_args(idx, d_Q, nq, k, nprobe, d_out_ids, d_out_dist));
    if (nq == 0) return 0;
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(ivf_search_enqueue(idx, d_Q, nq, k, nprobe, d_out_ids, d_out_dist, nullptr, st));
    return end_call(idx, st);
}

int hnswgpu_ivf_search_shard_dev(hnswgpu_index *idx, const float *d_Q, int32_t nq, int32_t k, int32_t nprobe,
                                 int32_t *d_out_ids, float *d_out_dist, uint32_t *d_out_order, void *stream) {
    HG_TRY(check_ivf_args(idx, d_Q, nq, k, nprobe, d_out_ids, d_out_dist));
    if (nq == 0) return 0;
    HG_REQUIRE(d_out_order, HNSWGPU_EINVAL, "d_out_order is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = static_cast<hipStream_t>(stream);
    HG_TRY(begin_call(idx, st));
    HG_TRY(ivf_search_enqueue(idx, d_Q, nq, k, nprobe, d_out_ids, d_out_dist, nullptr, st, nullptr, d_out_order));
    return end_call(idx, st);
}

int hnswgpu_ivf_search_lists(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t nprobe,
                             const int32_t *probes, int32_t *out_ids, float *out_dist) {
    HG_TRY(check_ivf_args(idx, Q, nq, k, nprobe, out_ids, out_dist));
    if (nq == 0) return 0;
    HG_REQUIRE(probes, HNSWGPU_EINVAL, "probes is null");
    for (int64_t i = 0; i < static_cast<int64_t>(nq) * nprobe; i++)
        HG_REQUIRE(probes[i] >= -1 && probes[i] < idx->nlist, HNSWGPU_EINVAL, "probe list id out of range");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    int64_t cnt = static_cast<int64_t>(nq) * k;
    HG_TRY(upload_queries(idx, Q, nq, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    HG_TRY(idx->s_probes.ensure(sizeof(int32_t) * static_cast<size_t>(nq) * nprobe));
    HG_HIP(hipMemcpyAsync(idx->s_probes.p, probes, sizeof(int32_t) * static_cast<size_t>(nq) * nprobe,
                          hipMemcpyHostToDevice, st));
    HG_TRY(ivf_search_enqueue(idx, idx->s_q.as<float>(), nq, k, nprobe, idx->s_ids.as<int32_t>(),
                              idx->s_outd.as<float>(), nullptr, st, idx->s_probes.as<int32_t>()));
    HG_HIP(hipMemcpyAsync(out_ids, idx->s_ids.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(out_dist, idx->s_outd.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

// One launch for a set of queued synchronous IVF requests with the same (k, nprobe); see combine_search.
// The flag a synchronous caller spins on: behind the search's last launch on the same stream, whose results are in the mapped
// block by then.
__global__ void slot_signal_kernel(uint32_t *flag, uint32_t val) {
    __threadfence_system();
    __hip_atomic_store(flag, val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Small synchronous batches (the reference's search-knn seam: one query per call): queries and results in a block of mapped
// pinned host memory -- the routing kernel reads the queries there, the finish kernel writes ids and distances there --, a
// one-thread launch behind them sets a flag the caller spins on.  No upload, no downloads, no hipStreamSynchronize (what the
// hnsw path has done since round 2: hnsw_search_batch_slot): one query, host buffers in and out, 72 -> ~50 us per call.
constexpr int32_t kIvfZcMaxQueries = 16;
static int ivf_search_batch_slot(hnswgpu_index *idx, const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
    const int32_t k = batch[0]->k;
    const size_t cnt = static_cast<size_t>(total) * k;
    const size_t qb = sizeof(float) * static_cast<size_t>(total) * idx->dim, ib = sizeof(int32_t) * cnt, db = sizeof(float) * cnt;
    const size_t o_q = 64, o_i = (o_q + qb + 63) & ~size_t(63), o_d = o_i + ((ib + 63) & ~size_t(63)), bytes = o_d + db;
    hnswgpu_index::Slot *slot = nullptr;
    std::unique_lock<std::mutex> sl;
    for (auto &s : idx->slots) {
        sl = std::unique_lock<std::mutex>(s.mu, std::try_to_lock);
        if (sl.owns_lock()) {
            slot = &s;
            break;
        }
    }
    if (!slot) {
        slot = &idx->slots[0];
        sl = std::unique_lock<std::mutex>(slot->mu);
    }
    HG_HIP(hipSetDevice(idx->device));
    HG_TRY(slot_prepare(*slot, bytes));
    char *hp = static_cast<char *>(slot->h), *dp = static_cast<char *>(slot->d);
    float *hq = reinterpret_cast<float *>(hp + o_q);
    size_t o = 0;
    for (auto *r : batch) {
        memcpy(hq + o, r->Q, sizeof(float) * static_cast<size_t>(r->nq) * idx->dim);
        o += static_cast<size_t>(r->nq) * idx->dim;
    }
    const uint32_t flag_val = ++slot->seq;
    volatile uint32_t *h_flag = reinterpret_cast<volatile uint32_t *>(hp);
    {
        std::lock_guard<std::mutex> lk(idx->mu);  // the index state is read (and the launches enqueued) under its lock
        HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists (call hnswgpu_ivf_build / hnswgpu_set_ivf)");
        const int32_t np = std::min(batch[0]->ef, idx->nlist);
        HG_TRY(begin_call(idx, slot->st));
        idx->zc_flag = reinterpret_cast<uint32_t *>(dp);
        idx->zc_val = flag_val;
        idx->zc_taken = false;
        const int rc = ivf_search_enqueue(idx, reinterpret_cast<const float *>(dp + o_q), total, k, np, reinterpret_cast<int32_t *>(dp + o_i),
                                          reinterpret_cast<float *>(dp + o_d), nullptr, slot->st);
        idx->zc_flag = nullptr;
        if (rc) return rc;
        if (!idx->zc_taken) {  // (a path without the finish kernel: the flag by a launch of its own behind it)
            hipLaunchKernelGGL(slot_signal_kernel, dim3(1), dim3(1), 0, slot->st, reinterpret_cast<uint32_t *>(dp), flag_val);
            HG_HIP(hipGetLastError());
        }
        HG_TRY(end_call(idx, slot->st));
    }
    HG_TRY(slot_wait(*slot, h_flag, flag_val));
    const int32_t *hi = reinterpret_cast<const int32_t *>(hp + o_i);
    const float *hd = reinterpret_cast<const float *>(hp + o_d);
    int64_t q0 = 0;
    for (auto *r : batch) {
        const size_t c = static_cast<size_t>(r->nq) * k;
        memcpy(r->out_ids, hi + q0 * k, sizeof(int32_t) * c);
        memcpy(r->out_dist, hd + q0 * k, sizeof(float) * c);
        q0 += r->nq;
    }
    return 0;
}

static int ivf_search_batch(hnswgpu_index *idx, const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
    if (tune(HNSWGPU_TUNE_ZEROCOPY, 1) != 0 && total <= kIvfZcMaxQueries) return ivf_search_batch_slot(idx, batch, total);
    const int32_t k = batch[0]->k;
    const int64_t cnt = static_cast<int64_t>(total) * k;
    std::lock_guard<std::mutex> lk(idx->mu);
    // under the lock: a concurrent set_ivf / ivf_build may have replaced (or a failed one removed) the lists since the
    // caller's argument check
    HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists (call hnswgpu_ivf_build / hnswgpu_set_ivf)");
    const int32_t np = std::min(batch[0]->ef, idx->nlist);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    const size_t qb = sizeof(float) * static_cast<size_t>(total) * idx->dim, ib = sizeof(int32_t) * cnt, db = sizeof(float) * cnt;
    HG_TRY(ensure_pinned(idx, qb + ib + db + 64));
    char *hp = static_cast<char *>(idx->h_pin);
    float *hq = reinterpret_cast<float *>(hp);
    int32_t *hi = reinterpret_cast<int32_t *>(hp + qb);
    float *hd = reinterpret_cast<float *>(hp + qb + ib);
    size_t o = 0;
    for (auto *r : batch) {
        memcpy(hq + o, r->Q, sizeof(float) * static_cast<size_t>(r->nq) * idx->dim);
        o += static_cast<size_t>(r->nq) * idx->dim;
    }
    HG_TRY(upload_queries(idx, hq, total, st));
    HG_TRY(ivf_search_enqueue(idx, idx->s_q.as<float>(), total, k, np, idx->s_ids.as<int32_t>(), idx->s_outd.as<float>(),
                              nullptr, st));
    HG_HIP(hipMemcpyAsync(hi, idx->s_ids.p, ib, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(hd, idx->s_outd.p, db, hipMemcpyDeviceToHost, st));
    HG_HIP(hipStreamSynchronize(st));
    int64_t q0 = 0;
    for (auto *r : batch) {
        const size_t c = static_cast<size_t>(r->nq) * k;
        memcpy(r->out_ids, hi + q0 * k, sizeof(int32_t) * c);
        memcpy(r->out_dist, hd + q0 * k, sizeof(float) * c);
        q0 += r->nq;
    }
    return 0;
}

int hnswgpu_ivf_stream_state(hnswgpu_index *idx, int32_t *off) {
    HG_REQUIRE(idx && off, HNSWGPU_EINVAL, "null argument");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists");
    if (!idx->ivf_calibrated) {
        HG_HIP(hipSetDevice(idx->device));
        HG_TRY(begin_call(idx, idx->stream));
        HG_TRY(ivf_calibrate(idx, idx->stream));
        HG_TRY(end_call(idx, idx->stream));
    }
    *off = idx->ivf_stream_off ? 1 : 0;
    return 0;
}

int hnswgpu_ivf_set_stream_state(hnswgpu_index *idx, int32_t off) {
    HG_REQUIRE(idx, HNSWGPU_EINVAL, "idx is null");
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_REQUIRE(idx->nlist > 0, HNSWGPU_ESTATE, "index has no IVF lists");
    idx->ivf_stream_off = off != 0;
    idx->ivf_calibrated = true;
    return 0;
}

int hnswgpu_ivf_search(hnswgpu_index *idx, const float *Q, int32_t nq, int32_t k, int32_t nprobe,
                       int32_t *out_ids, float *out_dist, int32_t *out_probes) {
    HG_TRY(check_ivf_args(idx, Q, nq, k, nprobe, out_ids, out_dist));
    if (nq == 0) return 0;
    if (!out_probes) {
        // Concurrent callers are combined into one launch (see hnswgpu_index::SearchReq) -- but only while the
        // combined batch is served by the SAME kernel each request would get alone: the GEMV scan and the MFMA tile
        // scan sum in different orders, and a call's distance bits must not depend on what other threads are doing.
        // (Euclidean: one arithmetic on both paths, so any mix.)
        hnswgpu_index::SearchReq me;
        me.Q = Q;
        me.nq = nq;
        me.k = k;
        me.ef = nprobe;
        me.out_ids = out_ids;
        me.out_dist = out_dist;
        me.stats = nullptr;
        // (the handle's one-off measurement of what the int8 bounds separate decides ivf_tile_pairs: it runs here, under
        // the index lock, before the predicate is built -- concurrent first calls would otherwise combine by the
        // pre-calibration boundary and run on the post-calibration kernel)
        {
            std::lock_guard<std::mutex> lk(idx->mu);
            if (!idx->ivf_calibrated && idx->nlist > 0) {
                HG_HIP(hipSetDevice(idx->device));
                HG_TRY(begin_call(idx, idx->stream));
                HG_TRY(ivf_calibrate(idx, idx->stream));
                HG_TRY(end_call(idx, idx->stream));
            }
        }
        const bool one_arith = idx->metric == METRIC_L2 || !tile_path_ok(idx) || tile_mode() == 0;
        // the kernel a batch of `total` queries gets: ivf_search_enqueue's own predicate, on the BATCH's nprobe (the
        // leader that evaluates this may have asked for another one)
        const int64_t tile_pairs = ivf_tile_pairs(idx, k);  // (requests of one batch share k)
        auto tiled = [idx, tile_pairs](int64_t total, int32_t nprobe_req) {
            const int64_t nl = idx->nlist;
            return total * std::min<int64_t>(nprobe_req, nl) > tile_pairs * nl;
        };
        return combine_search(
            idx->cmb_ivf, me,
            [=](const hnswgpu_index::SearchReq *first, const hnswgpu_index::SearchReq *r, int64_t total) {
                if (r->k != first->k || r->ef != first->ef || total + r->nq > 16384) return false;
                if (one_arith) return true;
                // every member must get the kernel it would get alone (the batch head included)
                const bool batch_tiled = tiled(total + r->nq, first->ef);
                return batch_tiled == tiled(r->nq, r->ef) && batch_tiled == tiled(first->nq, first->ef);
            },
            [idx](const std::vector<hnswgpu_index::SearchReq *> &batch, int32_t total) {
                return ivf_search_batch(idx, batch, total);
            });
    }
    std::lock_guard<std::mutex> lk(idx->mu);
    HG_HIP(hipSetDevice(idx->device));
    hipStream_t st = idx->stream;
    HG_TRY(begin_call(idx, st));
    int32_t np = std::min(nprobe, idx->nlist);
    int64_t cnt = static_cast<int64_t>(nq) * k;
    HG_TRY(upload_queries(idx, Q, nq, st));
    HG_TRY(idx->s_ids.ensure(sizeof(int32_t) * cnt));
    HG_TRY(idx->s_outd.ensure(sizeof(float) * cnt));
    HG_TRY(idx->s_probes.ensure(sizeof(int32_t) * static_cast<size_t>(nq) * np));
    HG_TRY(ivf_search_enqueue(idx, idx->s_q.as<float>(), nq, k, np, idx->s_ids.as<int32_t>(), idx->s_outd.as<float>(),
                              idx->s_probes.as<int32_t>(), st));
    HG_HIP(hipMemcpyAsync(out_ids, idx->s_ids.p, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost, st));
    HG_HIP(hipMemcpyAsync(out_dist, idx->s_outd.p, sizeof(float) * cnt, hipMemcpyDeviceToHost, st));
    if (np < nprobe)
        for (int64_t i = 0; i < static_cast<int64_t>(nq) * nprobe; i++) out_probes[i] = -1;
    HG_HIP(hipMemcpy2DAsync(out_probes, sizeof(int32_t) * nprobe, idx->s_probes.p, sizeof(int32_t) * np,
                            sizeof(int32_t) * np, nq, hipMemcpyDeviceToHost, st));
    HG_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
