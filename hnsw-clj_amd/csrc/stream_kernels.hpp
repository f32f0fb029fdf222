// stream_kernels.hpp -- the IVF list scan as a SURVIVOR STREAM: a candidate passes up to three filters of rising cost --
// its int8 codes, its half-precision row, its f32 row -- and only the last one produces distances.
//
// search-partition scores every row of every probed list (ivf_flat.clj:217-234) and search-ivf-flat keeps the k smallest
// (:281-294).  Every list row also exists as int8 codes with its exact residual (kernels.hpp: quantize_rows_kernel), so
//     lb <= d_f32(q, v) <= ub        from one exact integer dot product
// for every (query, candidate).  Round 2 wrote every lb into a dense [query][candidate] array, selected the k smallest,
// derived a threshold, rewrote the array with f32 distances and selected again: eleven launches, the array touched four
// times.  Here:
//   0. the routing step seeds tau[q], an upper bound of the query's k-th nearest distance D_k: the k-th smallest f32
//      distance among the head of its candidate stream (seed_tau_wg: k candidates are at most that far).
//   1. stream_bounds_kernel (matrix cores, v_mfma_i32_32x32x32_i8: a group of <= 32 queries against 32 list rows per
//      wave step, each probed list read ONCE per group in int8).  A candidate with lb > tau[q] has d >= lb > tau >= D_k:
//      it can neither be among the k nearest nor tie with the k-th, and is dropped on the spot; every other candidate is
//      appended -- (order key, list row, lb, ub), 16 bytes -- to the query's survivor list (one atomic add per query and
//      wave step reserves the slots).
//   2. ivf_mid_kernel (large batches): the survivors' half-precision rows give bounds seventy times narrower; the k-th
//      smallest UPPER bound of the list is a threshold of its own and leaves little more than k entries.
//   3. ivf_finish_kernel: filters the list once more against the threshold (and against the k-th distance found so
//      far), computes the f32 distance of what is left by the GEMV scan's own arithmetic (lane_partial + wave butterfly +
//      finish_dist: the same bits), keeps the k smallest (distance, order key) per wave in registers, and the last
//      workgroup of a query merges the partial lists, maps the winners to row ids and writes ids / distances.
// A threshold only ever decides what is NOT computed; every candidate with d <= D_k survives any valid one, so the result
// is exactly the full f32 scan's whatever order the workgroups run in.  A query whose survivors do not fit its list (data
// on which int8 bounds separate nothing) is flagged by its counter and served by the same finish kernel from the
// candidate stream itself -- the plain f32 scan -- so memory stays bounded and the answer exact.
#pragma once
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// dword address of natural code dword g4 (elements 4 g4 .. 4 g4 + 3) of list row R in the tile layout; S = steps per row.
// Layout: [block of 32 rows][step s of 32 bytes][half h][row r][16 B] -- every MFMA A operand is one contiguous 1 KB
// wave load straight into registers.
__host__ __device__ inline int64_t tile_code_dword(int64_t R, int g4, int S) {
    const int s = g4 >> 3, h = (g4 >> 2) & 1, w = g4 & 3;
    return ((((R >> 5) * S + s) * 2 + h) * 32 + (R & 31)) * 4 + w;
}

// codes of `n` list rows into the tile layout + the per-row bound terms: one wave per row
template <int NCH>
__global__ __launch_bounds__(kWG) void quantize_rows_tile_kernel(const float *rows, int64_t ld, int64_t n, int metric,
                                                                 uint32_t *tile, float4 *meta) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kNWave + (threadIdx.x >> 6);
    if (row >= n) return;
    float4 r[NCH];
    load_row<NCH>(r, rows + row * ld, static_cast<int>(ld / 4), lane, true);
    bool bad;
    const float mx = wave_absmax<NCH>(r, bad);
    uint32_t w[NCH];
    float res;
    int c2;
    encode_lane<NCH>(r, mx, bad, w, res, c2);
#pragma unroll
    for (int c = 0; c < NCH; c++) tile[tile_code_dword(row, c * kWave + lane, NCH * 8)] = w[c];
    if (meta) {
        res = __builtin_sqrtf(wave_sum(res));
        c2 = wave_sum_int(c2);
        const float nv = __builtin_sqrtf(wave_sum(lane_partial<NCH, false>(r, r)));
        if (lane == 0) meta[row] = code_row_meta(metric, mx, res, c2, nv, bad);
    }
}

// ---- the running threshold: orderable bits of a float (make_key's order), 0xffffffff = no bound yet -----------------
__device__ __forceinline__ uint32_t tau_encode(float t) {
    t = t + 0.0f;
    const uint32_t u = __float_as_uint(t);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float tau_decode(uint32_t u) {
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    const float f = __uint_as_float(u);
    return f == f ? f : __builtin_inff();  // the initial all-ones (a NaN pattern): nothing is excluded
}

// ---- the first threshold of a query -----------------------------------------------------------------------------
// The bounds pass drops a candidate against the threshold it knows at that moment, and its workgroups run in list
// order, not in the query's probe order: without a first threshold every list visited before the query's nearest one
// would be appended whole.  So before the pass the query's workgroup evaluates the HEAD of its candidate stream -- the
// first m = max(k, 64) rows of its nearest non-empty lists -- in f32 (the GEMV arithmetic) and takes the k-th smallest of
// those distances: k candidates are at most that far, hence D_k <= tau0.  All threads of a 256-thread workgroup call it.
constexpr int kSeedMax = 256;  // rows evaluated at most (>= the largest k the bounds pass serves)

// Rows of the head sample: 64 (32 for a handful of queries: one gather round) -- at every batch size: with 32 rows for
// batches of 1024 and more, 2 % of the bench index's queries under the dot metric caught fewer than k rows of their own
// cluster, got no threshold, and the batch took 1.0 ms instead of 0.57 -- times the mean list length over 1024, up to eight times: a list of several thousand rows holds
// several of the data's clusters, and a sample that catches fewer than k rows of the query's own yields a threshold from
// another cluster, i.e. none (1M rows in 128 lists, k = 10: every survivor list overflowed and the search took 94 ms instead
// of 23 by the plain f32 scan; with 256 rows sampled the stream is ahead again).
__host__ inline int stream_seed_rows(int nq, int64_t n, int nlist) {
    const int base = nq <= 8 ? 32 : 64;
    const int64_t mean = n / (nlist > 0 ? nlist : 1);
    const int64_t scale = mean / 1024 < 1 ? 1 : (mean / 1024 > 8 ? 8 : mean / 1024);
    const int64_t rows = (base * scale + 15) / 16 * 16;
    return static_cast<int>(rows < 16 ? 16 : (rows > kSeedMax ? kSeedMax : rows));
}

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void half_bounds(int metric, float sum, float qn, float4 mt, float &lb, float &ub);

// With the half-precision copy of the list rows (`half` / `hmeta`, stream step 1b's operands) the head is read from THERE
// and the threshold is the k-th smallest UPPER bound (half_bounds: k candidates are at most that far, as the half-precision
// passes of larger batches argue): half the bytes.  One workgroup reads at what one CU's path to memory carries -- ~21 GB/s
// (profiles/r02_one_cu_gather_microbench.txt) -- so 64 f32 rows (192 KB) were 9.7 us of a batch-32 search's 18-us routing
// tail and 32 rows 5.1 us of a single query's; the bounds are ~1e-4 above the distances, far inside what separates the k-th
// of 64 sampled rows from the k-th of the list.  qn: |q| (cosine and dot), unused for L2.
template <int NCH, int RB, bool L2>
__device__ __forceinline__ void seed_tau_wg(const float4 (&q)[NCH], float qn, int metric, const Pair *pp, int nprobe,
                                            int64_t qcnt, int k, const float *rows, const float *row_norms, int64_t ld,
                                            float *dist_s /* [kSeedMax] LDS */, uint32_t *tau_out, int sample = 64,
                                            const uint2 *half = nullptr, const float4 *hmeta = nullptr) {
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    int m = k > sample ? k : sample;  // (large batches sample fewer rows: thousands of queries x 64 rows is the index again)
    m = (m + 31) & ~31;
    if (m > kSeedMax) m = kSeedMax;
    if (m > qcnt) m = static_cast<int>(qcnt);
    if (m < k) {  // fewer candidates than k: every one of them is a result, nothing can be excluded
        if (tid == 0) *tau_out = 0xffffffffu;
        return;
    }
    const int nvec = static_cast<int>(ld / 4);
    // half rows: twice the f32 path's rows in flight per wave (64 rows: ONE round trip of the workgroup), the meta words
    // fetched behind the rows (their scale is needed only once the rows are in)
    constexpr int HB = RB == 8 ? 16 : RB;
    for (int j0 = wave * HB; half && j0 < m; j0 += kNWave * HB) {
        int64_t myrow = 0;
        const bool mine = lane < HB && j0 + lane < m;
        if (mine) {
            const uint32_t o = static_cast<uint32_t>(j0 + lane);
            int lo = 0, hi = nprobe - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pp[mid].ord_base <= o) lo = mid;
                else hi = mid - 1;
            }
            myrow = pp[lo].row_begin + (o - pp[lo].ord_base);
        }
        const int rlo = static_cast<int>(myrow), rhi = static_cast<int>(myrow >> 32);
        uint2 w[HB][NCH];
#pragma unroll
        for (int b = 0; b < HB; b++) {
            const int64_t row = (static_cast<int64_t>(__builtin_amdgcn_readlane(rhi, b)) << 32) |
                                static_cast<uint32_t>(__builtin_amdgcn_readlane(rlo, b));
            const uint2 *rp = half + row * nvec;
#pragma unroll
            for (int c = 0; c < NCH; c++) w[b][c] = j0 + b < m && c * kWave + lane < nvec ? rp[c * kWave + lane] : make_uint2(0u, 0u);
        }
        float4 mt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (mine) mt = hmeta[myrow];
        float accs[HB];
#pragma unroll
        for (int b = 0; b < HB; b++) {  // (ivf_mid_kernel's arithmetic)
            const float sc = L2 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mt.x), b)) : 0.0f;
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const h4_t h = __builtin_bit_cast(h4_t, w[b][c]);
                const float hv[4] = {static_cast<float>(h[0]), static_cast<float>(h[1]), static_cast<float>(h[2]), static_cast<float>(h[3])};
                const float qv[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (L2) {
                        const float d = qv[j] - hv[j] * sc;
                        acc = __builtin_fmaf(d, d, acc);
                    } else {
                        acc = __builtin_fmaf(qv[j], hv[j], acc);
                    }
                }
            }
            accs[b] = acc;
        }
        float tot;  // lane b < HB: row b's sum, beside its meta word
        if constexpr (HB == 16) {
            float lo8[8], hi8[8];
#pragma unroll
            for (int b = 0; b < 8; b++) {
                lo8[b] = accs[b];
                hi8[b] = accs[8 + b];
            }
            const float t0 = wave_sum8(lo8, lane), t1 = wave_sum8(hi8, lane);  // every lane l: the total of row (l & 7) of its eight
            tot = lane < 8 ? t0 : t1;
        } else {
            tot = rows_sum_to_lane<RB>(accs, lane);
        }
        if (mine) {
            float lb, ub;
            half_bounds(metric, tot, qn, mt, lb, ub);
            dist_s[j0 + lane] = ub == ub ? ub : __builtin_inff();
        }
    }
    for (int j0 = wave * RB; !half && j0 < m; j0 += kNWave * RB) {
        // lane b < RB resolves candidate j0 + b of the stream to its list row (as ivf_finish_kernel does)
        int64_t myrow = 0;
        float myrn = 0.0f;
        if (lane < RB && j0 + lane < m) {
            const uint32_t o = static_cast<uint32_t>(j0 + lane);
            int lo = 0, hi = nprobe - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pp[mid].ord_base <= o) lo = mid;
                else hi = mid - 1;
            }
            myrow = pp[lo].row_begin + (o - pp[lo].ord_base);
            myrn = metric == METRIC_COS ? row_norms[myrow] : 0.0f;
        }
        const int rlo = static_cast<int>(myrow), rhi = static_cast<int>(myrow >> 32);
        float4 r[RB][NCH];
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int64_t row = (static_cast<int64_t>(__builtin_amdgcn_readlane(rhi, b)) << 32) |
                                static_cast<uint32_t>(__builtin_amdgcn_readlane(rlo, b));
            load_row<NCH>(r[b], rows + row * ld, nvec, lane, j0 + b < m);
        }
        float sm[RB];
#pragma unroll
        for (int b = 0; b < RB; b++) sm[b] = lane_partial<NCH, L2>(q, r[b]);
        const float tot = rows_sum_to_lane<RB>(sm, lane);  // lane b: row b's sum (eight rows: one halving exchange)
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), b));
            const float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), b));
            const float d = finish_dist(metric, sum, qn, rn);
            if (lane == 0 && j0 + b < m) dist_s[j0 + b] = d == d ? d : __builtin_inff();
        }
    }
    __syncthreads();
    if (tid < m) {  // (m is a multiple of 32 or the whole stream)
        const float v = dist_s[tid];
        int rank = 0;
        int j = 0;
        for (; j + 4 <= m; j += 4) {  // uniform addresses: LDS broadcasts, four values each
            const float4 o = *reinterpret_cast<const float4 *>(dist_s + j);
            rank += (o.x < v || (o.x == v && j < tid)) ? 1 : 0;
            rank += (o.y < v || (o.y == v && j + 1 < tid)) ? 1 : 0;
            rank += (o.z < v || (o.z == v && j + 2 < tid)) ? 1 : 0;
            rank += (o.w < v || (o.w == v && j + 3 < tid)) ? 1 : 0;
        }
        for (; j < m; j++) {
            const float o = dist_s[j];
            rank += (o < v || (o == v && j < tid)) ? 1 : 0;
        }
        if (rank == k - 1) *tau_out = tau_encode(v);  // ranks are a permutation of 0..m-1: exactly one thread
    }
}

// What the survivor stream needs of every query before the bounds pass, when the routing was not done by the fused routing
// kernel (which does the same in its tail): int8 codes + bound scalars, an empty survivor list, the first threshold.
// One workgroup per query.
struct PrepArgs {
    const float *Q;
    int64_t qld;
    int32_t dim, metric, nq, nprobe, k;
    int32_t seed_rows;  // stream_seed_rows
    int32_t home;       // the home-list pass follows: no seed for a query whose nearest list holds k rows
    const Pair *pairs;
    const int32_t *qcnt;
    const float *rows;
    const float *row_norms;
    int64_t ld;
    const uint2 *half;    // optional: the half-precision copy of the list rows and its meta words (the seed reads those)
    const float4 *hmeta;
    uint32_t *qcodes;
    QueryScal *qscal;
    uint32_t *tau;
};

template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_query_prep_kernel(PrepArgs a) {
    __shared__ __align__(16) float dist_s[kSeedMax];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int qi = blockIdx.x;
    float4 q[NCH];
    load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
    const float qn = (a.metric == METRIC_COS || (a.half && a.metric == METRIC_DOT)) ? query_norm<NCH>(q) : 0.0f;
    if (wave == kNWave - 1) {
        QueryCode<NCH> qc;
        encode_query<NCH>(q, qc);
#pragma unroll
        for (int c = 0; c < NCH; c++) a.qcodes[(static_cast<int64_t>(qi) * NCH + c) * kWave + lane] = qc.a[c];
        if (lane == 0) a.qscal[qi] = qc.sc;
    }
    if (a.home) {
        const Pair hp = a.pairs[static_cast<int64_t>(qi) * a.nprobe];
        if (hp.row_end - hp.row_begin >= a.k) {  // (as route_tail_wg: the threshold comes from the home-list pass)
            if (threadIdx.x == 0) a.tau[qi] = 0xffffffffu;
            return;
        }
    }
    seed_tau_wg<NCH, RB, L2>(q, qn, a.metric, a.pairs + static_cast<int64_t>(qi) * a.nprobe, a.nprobe, a.qcnt[qi], a.k, a.rows,
                             a.row_norms, a.ld, dist_s, a.tau + qi, a.seed_rows, a.half, a.hmeta);
}

// One work item of the grouped bounds pass: rows [rb0 + r0_off, rb0 + r1_off) of inverted list `list` (which starts at row
// rb0) against its members [mem0, mem0 + cnt) (cnt <= 32).  Written by ivf_worklist_kernel from the per-list buckets.
struct WorkDesc {
    int64_t rb0;
    int32_t r0_off, r1_off;
    int32_t list, mem0, cnt, pad;
};

// The work list of a small batch built INSIDE the routing tail's launch (ivf_worklist_kernel is a launch of its own between the
// tail and the bounds pass: ~6 us of a 160-us batch-32 search for ~2 us of work).  The tail launch carries kWorklistParts extra
// workgroups behind the queries' own; they wait until every query has filed its pairs (RouteArgs: the `filed` counter, bumped
// by a query's workgroup between its probe table and its threshold seed -- the seed's ~5 us of f32 rows run beside this), scan
// the per-list counts and fill every kWorklistParts-th item each.  Workgroups are dispatched in index order and the queries'
// workgroups wait for nobody, so the wait ends; should it not within two seconds (it cannot, short of a fault), the list is left
// empty and every query marked for the finish kernel's fallback, the plain f32 scan.  The item order is ivf_worklist_kernel's.
constexpr int kWorklistParts = 4;
struct WorklistArgs {
    uint32_t *bk_cnt;        // [nlist] members filed per list; zero again when the last part is through
    int32_t bk_cap, nlist;
    const int64_t *list_off;
    int64_t chunk_rows;
    int32_t max_chunks, tq;  // tq: members per group
    WorkDesc *desc;
    int32_t *nitems;
    uint32_t *filed;         // [2]: queries that have filed their pairs | parts that are through (both zero between searches)
    uint32_t *surv_cnt;      // [nq] (the fallback mark)
    int32_t nq;
};

__device__ __forceinline__ void worklist_part_wg(const WorklistArgs &a, int part) {
    __shared__ int32_t wl_off[1025];   // exclusive item offsets of this pass's lists
    __shared__ int32_t wl_cnt[1024];   // members per list (capped)
    __shared__ int32_t wl_part[kNWave];
    __shared__ int32_t wl_ok;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        while (coherent_load(a.filed) < static_cast<uint32_t>(a.nq)) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000ull) {  // 2 s at 100 MHz
                ok = 0;
                break;
            }
        }
        wl_ok = ok;
    }
    __syncthreads();
    const bool ok = wl_ok != 0;
    int carry = 0;
    for (int l0 = 0; ok && l0 < a.nlist; l0 += 1024) {
        // four consecutive lists per thread: counts -> items, scanned within the thread, the wave, the workgroup
        int nw[4], run = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int l = l0 + 4 * tid + j;
            int c = 0, nch = 0;
            if (l < a.nlist) {
                const uint32_t filed = coherent_load(a.bk_cnt + l);
                c = filed < static_cast<uint32_t>(a.bk_cap) ? static_cast<int>(filed) : a.bk_cap;
                const int64_t rows = a.list_off[l + 1] - a.list_off[l];
                nch = (c > 0 && rows > 0) ? static_cast<int>(tile_nchunks(rows, a.chunk_rows, a.max_chunks)) : 0;
            }
            wl_cnt[4 * tid + j] = c;
            nw[j] = ((c + a.tq - 1) / a.tq) * nch;
            run += nw[j];
        }
        int incl = run;
        for (int o = 1; o < kWave; o <<= 1) {
            const int v = __shfl_up(incl, o, kWave);
            if (lane >= o) incl += v;
        }
        if (lane == kWave - 1) wl_part[wave] = incl;
        __syncthreads();
        int before = 0;
        for (int w = 0; w < wave; w++) before += wl_part[w];
        int at = before + incl - run;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            wl_off[4 * tid + j] = at;
            at += nw[j];
        }
        if (tid == kWG - 1) wl_off[1024] = at;
        __syncthreads();
        const int total = wl_off[1024];
        for (int w = part * kWG + tid; w < total; w += kWorklistParts * kWG) {
            int lo = 0, hi = 1023;  // the last list of this pass whose offset <= w
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (wl_off[mid] <= w) lo = mid;
                else hi = mid - 1;
            }
            const int ll = l0 + lo, k = w - wl_off[lo];
            const int cc = wl_cnt[lo];
            const int ngl = (cc + a.tq - 1) / a.tq;
            const int ch = k / ngl, g = k - ch * ngl;  // chunk-major
            const int64_t rb0 = a.list_off[ll], rws = a.list_off[ll + 1] - rb0;
            const int64_t tiles = (rws + kTileRows - 1) / kTileRows;
            const int64_t nchl = tile_nchunks(rws, a.chunk_rows, a.max_chunks);
            const int64_t per = (tiles + nchl - 1) / nchl * kTileRows;
            const int64_t a0 = ch * per, a1 = a0 + per < rws ? a0 + per : rws;
            WorkDesc d;
            d.rb0 = rb0;
            d.r0_off = static_cast<int32_t>(a0);
            d.r1_off = static_cast<int32_t>(a1 > a0 ? a1 : a0);
            d.list = ll;
            d.mem0 = g * a.tq;
            d.cnt = cc - g * a.tq < a.tq ? cc - g * a.tq : a.tq;
            d.pad = 0;
            a.desc[carry + w] = d;
        }
        carry += total;
        __syncthreads();  // (wl_off / wl_cnt are rewritten by the next pass)
    }
    if (part == 0 && tid == 0) *a.nitems = ok ? carry : 0;
    if (!ok && part == 0)
        for (int q = tid; q < a.nq; q += kWG) a.surv_cnt[q] = 0x80000000u;
    // the last part through leaves the counters zero for the next search (every part has read them by then)
    __shared__ int32_t wl_last;
    __syncthreads();
    if (tid == 0) {
        const uint32_t prev = __hip_atomic_fetch_add(a.filed + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wl_last = prev == static_cast<uint32_t>(kWorklistParts) - 1 ? 1 : 0;
    }
    __syncthreads();
    if (!wl_last) return;
    for (int l = tid; l < a.nlist; l += kWG) coherent_store(a.bk_cnt + l, 0u);
    if (tid == 0) {
        coherent_store(a.filed, 0u);
        coherent_store(a.filed + 1, 0u);
    }
}

struct StreamArgs {
    // grouped mode: the (query, list) pairs were filed under their list by the routing step (bk_mem); one workgroup per
    // (list, group of <= 32 members, chunk of the list's rows)
    const WorkDesc *wi_desc;  // dense work list built on the device (ivf_worklist_kernel); workgroup b serves item
    const int32_t *nitems;    // (b & 7) * ceil(nitems / 8) + (b >> 3): one contiguous eighth of the list per XCD
    const uint2 *bk_mem;      // [nlist][bk_cap] (query, offset of the list in the query's candidate stream)
    int32_t bk_cap;
    // ungrouped mode (a handful of queries): item = (pair, chunk), chunk-major, one query per item
    const Pair *pairs;
    int32_t npairs;
    int32_t chunk_rows, nchunks;
    int32_t metric, k;
    const uint32_t *ctile;  // list codes, tile layout
    const float4 *cmeta;    // per list row: (scale, E, Z, 1 / |v|)
    const uint32_t *qcodes; // [nq][NCH * 64] natural order
    const QueryScal *qscal;
    uint32_t *tau;       // [nq] running threshold (orderable bits), all-ones at the start
    uint32_t *surv_cnt;  // [nq] survivors appended so far (may exceed cap: the query then takes the fallback)
    uint4 *surv;         // [nq][cap] (order key, list row, lb bits, ub bits)
    int64_t cap;
    int32_t dbg;         // developer ablation switches (-DHG_DIAG builds: hnswgpu_debug_set_ablation); 0 in the product
    int32_t defer;       // the half-precision pass follows: the wide epilogue appends entries without bounds (template DEFER)
    const Pair *home_pairs;  // optional [nq][home_nprobe]: ivf_home_kernel serves every row of a query's nearest list -- the
    int32_t home_nprobe;     // pair (query, that list) appends nothing here
    unsigned long long *stamps;  // -DHG_IVF_STAMPS diagnostic builds only
};


// Latency, not arithmetic, is what this kernel has to manage (a 32-row block is 24 KB of codes: ~1.2 us of a CU's share
// of HBM, against 0.7 us of matrix-core time): every wave keeps eight 1 KB operand loads in flight and issues the first
// loads of its NEXT block before the epilogue of the current one -- and of its first block before the group is set up,
// since the rows to read depend on the work item alone.
//
// Two epilogues turn the 32 x 32 integer tile into decisions.  A group of up to kNarrow queries (small batches: one or two
// queries per probed list) parks its few live columns in LDS and lets every LANE take a ROW -- one pass over 32 rows per
// pair of queries, survivors written side by side -- instead of sixteen passes in which one or two lanes work.  Wider
// groups keep the MFMA's own distribution (lane = query, sixteen rows each) and run the test as packed two-row operations
// against row terms read from LDS four rows at a time.  DEFER (wide groups, the half-precision pass follows): entries are
// appended without bounds -- that pass replaces them.
constexpr int kNarrow = 8;

__host__ __device__ constexpr int stream_epilogue_words(bool narrow) { return narrow ? kNarrow * 33 : 16 * kWave; }
__host__ inline size_t stream_lds_bytes(int nch, bool narrow, int qb = 1, int waves = kTileWaves) {  // qb: 32-query column blocks per group
    const size_t tq = static_cast<size_t>(kTileQ) * qb;
    return static_cast<size_t>(nch) * 256 * tq                              // query codes [block][step][half][query][16 B]
           + sizeof(QueryScal) * tq + sizeof(float4) * 32 * waves           // query scalars, per-wave row terms
           + sizeof(float4) * tq                                            // query terms of the test
           + (sizeof(uint32_t) + sizeof(int32_t)) * tq                      // order bases, query indices
           + (qb > 1 ? 0 : sizeof(int32_t) * stream_epilogue_words(narrow) * waves)  // per-wave tile of the epilogue (two column blocks: entries without bounds, no tile)
           + 16;                                                            // the group's home-list mask
}

// The rejection test of one (query, row) in its cheapest form.  code_bounds' lower bound exceeds the threshold tau
//      cosine     1 - dot s_q s_v / (|q||v|) - W > tau,   W = E + e_q (1 + E)
//      dot       -dot s_q s_v - W > tau,                  W = |q| E + r_q Z
//      Euclidean  d2 - dl > (tau + W)^2 (tau, W >= 0),    d2 = a2 - 2 dot s_q s_v + v2, dl = 2e-6 (a2 + v2), W = w_q + E
// exactly when, with everything that belongs to the row on one side and to the query on the other,
//      float(dot) * r.x  <  P - Q r.y + K r.z
// where (r.x, r.y, r.z) = stream_row_terms (per row and block, by the lane that holds the row) and (P, Q, K) =
// stream_query_terms (per query, once per workgroup).  The rearrangement rounds differently from code_bounds in the
// last bits; P is pulled back by a few 1e-6 of the magnitudes involved (the allowances E carry 1e-4 / 2e-5 / 4e-6 relative,
// the test stays on the safe side of code_bounds' own).  NaN on either side (a row or a query without a bound, no
// threshold yet) fails the comparison: the candidate survives.  The bounds STORED with a survivor are code_bounds'.
__device__ __forceinline__ float4 stream_row_terms(int metric, float4 m) {
    if (metric == METRIC_COS) return make_float4(m.x * m.w, m.y, 0.0f, 0.0f);
    if (metric == METRIC_DOT) return make_float4(m.x, m.y, m.z, 0.0f);
    const float v2 = m.x * m.x * m.z;  // |v'|^2
    return make_float4(m.x, m.y, v2 - m.y * m.y * (1.0f + 4.0e-6f), 0.0f);
}
__device__ __forceinline__ float4 stream_query_terms(int metric, const QueryScal &qc, float tau) {
    if (metric == METRIC_COS) {
        const float ik = 1.0f / (qc.s * qc.iqn);  // NaN for a query without a norm
        return make_float4(((1.0f - qc.eq) - tau - 4.0e-6f * (1.0f + __builtin_fabsf(tau))) * ik, (1.0f + qc.eq) * ik, 0.0f, 0.0f);
    }
    if (metric == METRIC_DOT) {
        const float ik = 1.0f / qc.s, t = -tau * ik;
        return make_float4(t - 2.0e-6f * __builtin_fabsf(t), qc.qn * ik, -(qc.rq * ik), 0.0f);
    }
    // dot x < [(a2 + v2)(1 - 3e-6) - (tw + E)^2 (1 + 1e-6)] / (2 s_q),  tw = tau + w_q:
    //       = k1 (a2 - tw^2 c) + k1 (v2 - E^2 c) - (2 tw c k1) E,   k1 = (1 - 3e-6) / (2 s_q), c = (1 + 1e-6) / (1 - 3e-6)
    const float k1 = (1.0f - 3.0e-6f) / (2.0f * qc.s), c = 1.0f + 4.0e-6f;
    const float tw = tau + (qc.rq + 4.0e-6f * qc.qn);
    return make_float4(k1 * (qc.a2 - tw * tw * c), 2.0f * tw * c * k1, k1, 0.0f);
}
typedef float v2f_t __attribute__((ext_vector_type(2)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool stream_reject(int dot, float4 r, float4 qt) {
    return static_cast<float>(dot) * r.x < __builtin_fmaf(-qt.y, r.y, __builtin_fmaf(r.z, qt.z, qt.x));
}

// QB = 2 (wide deferring launches of the largest batches): TWO 32-query column blocks per group -- a staged row operand meets
// 64 queries (two matrix instructions), so a list probed by 512 queries leaves L2 eight times instead of sixteen: there the
// kernel is bound by the L2s, not by HBM.  Two accumulators are ~150 registers: four-wave workgroups, three per CU (eight-wave
// ones fit once: 358 us against 283 at batch 4096).  At batch 2048 / 4096 it is no faster than one block (194 vs 189, 280 vs 284
// us: what the L2s are spared the lower occupancy takes back), so only batches with 256 and more pairs per list take it.
// QB column blocks of 32 queries per group (1, 2 or 4): every staged row operand meets 32 QB queries -- one matrix instruction per
// block on the same A registers, an accumulator each -- so a list whose members fill several blocks leaves the L2 once per QB
// blocks instead of once per block (batch 4096 on the bench index: 128 members per list, four passes of 0.77 GB out of the L2s
// were 287 us against 144 at batch 1024, where there is one).  Two blocks: four waves; four blocks: eight waves (their codes are
// 96 KB of LDS at 768 dimensions: one workgroup per CU, and eight waves keep 64 KB of row operands in flight).
__host__ __device__ constexpr int stream_waves(int qb) { return qb == 1 ? kTileWaves : (qb == 2 ? 4 : 8); }
template <int NCH, bool NARROW, bool DEFER, int QB = 1>
__global__ __launch_bounds__(stream_waves(QB) * kWave) void stream_bounds_kernel(StreamArgs a) {
    static_assert(QB == 1 || ((QB == 2 || QB == 4) && !NARROW && DEFER), "several column blocks: the wide deferring epilogue only");
    constexpr int S = NCH * 8;  // steps of 32 bytes
    // operand loads in flight per wave.  (Round 5, tools/bounds_pmc.sh at batch 16384: the waves wait on memory 61 % of the time
    // -- SQ_WAIT_ANY / SQ_WAVE_CYCLES --, the matrix cores are 25 % busy, LDS waits 2 %; twelve loads in flight instead of eight
    // made the two-column-block kernel SLOWER, 0.80 vs 0.64 ms: it is the L2s' delivery rate, ~9 TB/s, not the waves' latency.)
    constexpr int PF = 8;
    constexpr int TQ = kTileQ * QB;                     // queries per group
    constexpr int NWV = stream_waves(QB);               // waves per workgroup
    constexpr int NTHR = NWV * kWave;
    static_assert(TQ <= NTHR, "a thread per member sets the group up");
    extern __shared__ __align__(16) unsigned char smem[];
    v4i_t *qb_s = reinterpret_cast<v4i_t *>(smem);                               // [QB][S][2][32]
    QueryScal *qs_s = reinterpret_cast<QueryScal *>(qb_s + S * 64 * QB);         // [TQ]
    float4 *meta_all = reinterpret_cast<float4 *>(qs_s + TQ);                    // [waves][32] row terms
    float4 *qt_s = meta_all + 32 * NWV;                                           // [TQ] query terms
    uint32_t *ob_s = reinterpret_cast<uint32_t *>(qt_s + TQ);                     // [TQ]
    int32_t *qi_s = reinterpret_cast<int32_t *>(ob_s + TQ);                      // [TQ]
    int32_t *narrow_all = qi_s + TQ;                                              // [waves][stream_epilogue_words] (QB == 1)
    uint32_t *home_mask_s = reinterpret_cast<uint32_t *>(narrow_all + (QB > 1 ? 0 : kTileWaves * stream_epilogue_words(NARROW)));  // [QB] slots of the group whose nearest list this is (served by ivf_home_kernel)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    float *terms_w = reinterpret_cast<float *>(meta_all + wave * 32);
    int32_t *tile_w = narrow_all + (QB > 1 ? 0 : wave * stream_epilogue_words(NARROW));  // narrow: [kNarrow][33] dot products; wide: [16][64] (QB == 1)

    const int col = lane & 31, half = lane >> 5;
    constexpr bool narrow = NARROW;
    HG_IVF_STAMP(a.stamps, 22, blockIdx.x == 0 && threadIdx.x == 0);  // first workgroup of the bounds kernel starts
    const v4i_t *tile = reinterpret_cast<const v4i_t *>(a.ctile);
    // ---- work item
    int64_t rb0, r0, r1;
    int cnt, one_q = -1;
    uint32_t one_ob = 0;
    const uint2 *mem = nullptr;
    if (a.pairs) {
        const int pair = static_cast<int>(blockIdx.x % a.npairs), chunk = static_cast<int>(blockIdx.x / a.npairs);
        const Pair p = a.pairs[pair];
        rb0 = p.row_begin;
        const int64_t rows = p.row_end - p.row_begin;
        if (rows <= 0) return;
        const int64_t tiles = (rows + kTileRows - 1) / kTileRows;
        const int64_t nch = tile_nchunks(rows, a.chunk_rows, a.nchunks);
        if (chunk >= nch) return;
        const int64_t per = (tiles + nch - 1) / nch * kTileRows;
        r0 = rb0 + static_cast<int64_t>(chunk) * per;
        r1 = r0 + per < p.row_end ? r0 + per : p.row_end;
        cnt = 1;
        one_q = p.q;
        one_ob = p.ord_base;
    } else {
        const int nitems = *a.nitems;
        const int per_xcd = (nitems + 7) >> 3;
        const int slot = blockIdx.x >> 3;
        if (slot >= per_xcd) return;
        const int item = (blockIdx.x & 7) * per_xcd + slot;
        if (item >= nitems) return;
        const WorkDesc d = a.wi_desc[item];
        rb0 = d.rb0;
        r0 = rb0 + d.r0_off;
        r1 = rb0 + d.r1_off;
        cnt = d.cnt;
        mem = a.bk_mem + static_cast<int64_t>(d.list) * a.bk_cap + d.mem0;
    }
    if (r0 >= r1 || cnt <= 0) return;
    const int rel0 = static_cast<int>(r0 - rb0), rel1 = static_cast<int>(r1 - rb0);

    // blocks of 32 rows of the tile layout that overlap [r0, r1): the layout's blocks are aligned to the WHOLE list array,
    // not to a list, so the first and the last block of a chunk may hold rows of the neighbours -- computed, not used
    const int64_t b0 = r0 >> 5, b1 = (r1 + 31) >> 5;
    int64_t b = b0 + wave;
    v4i_t av[PF];
    float4 metar = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (b < b1) {  // the first block's operands are on their way while the group is set up
        const v4i_t *ap = tile + (b * S) * 64 + lane;
#pragma unroll
        for (int u = 0; u < PF; u++) av[u] = ap[u * 64];
        const int64_t row = b * 32 + (lane & 31);
        if (row >= r0 && row < r1) metar = a.cmeta[row];
    }

    if (tid < TQ) {
        int qi = -1;
        uint32_t ob = 0;
        if (tid < cnt) {
            if (a.pairs) {
                qi = one_q;
                ob = one_ob;
            } else {
                const uint2 m = mem[tid];
                qi = static_cast<int>(m.x);
                ob = m.y;
            }
        }
        qi_s[tid] = qi;
        ob_s[tid] = ob;
        bool home = false;
        if (a.home_pairs && qi >= 0) {
            const Pair hp = a.home_pairs[static_cast<int64_t>(qi) * a.home_nprobe];
            // (the query's FIRST pair, told by its place in the candidate stream, not by its list: hnswgpu_ivf_search_lists may name
            // a list twice, and only the first pair's rows came through the home-list pass)
            home = hp.row_begin == rb0 && hp.row_end > hp.row_begin && ob == hp.ord_base;
        }
        const uint64_t hb = __ballot(home);  // (a wave of this branch = 64 members = two column blocks)
        if (lane == 0) {
            home_mask_s[2 * wave] = static_cast<uint32_t>(hb);
            if (2 * wave + 1 < QB) home_mask_s[2 * wave + 1] = static_cast<uint32_t>(hb >> 32);
        }
    }
    __syncthreads();
    // query codes: natural order in global memory (16-B chunk t of query q = step t / 2, half t & 1); empty slots are zero
    if (tid < cnt) qs_s[tid] = a.qscal[qi_s[tid]];
    for (int f = tid; f < TQ * S * 2; f += NTHR) {
        const int q = f / (S * 2), t = f - q * (S * 2);
        v4i_t v = {0, 0, 0, 0};
        if (q < cnt) v = reinterpret_cast<const v4i_t *>(a.qcodes + static_cast<int64_t>(qi_s[q]) * NCH * kWave)[t];
        qb_s[(q >> 5) * S * 64 + t * 32 + (q & 31)] = v;
    }
    // the query's side of the test (lane = slot of the group): P, Q, K of stream_query_terms, once per workgroup -- the
    // threshold of a query does not move while this kernel runs (it falls again in the finish kernel)
    if (tid < TQ) {
        float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (tid < cnt) t = stream_query_terms(a.metric, qs_s[tid], tau_decode(a.tau[qi_s[tid]]));
        qt_s[tid] = t;
    }
    __syncthreads();

    const uint32_t home_mask = home_mask_s[0];
    const bool live = col < cnt && !((home_mask >> col) & 1u);
    const float4 myqt = qt_s[col];
    const v2f_t P2 = {myqt.x, myqt.x}, nQ2 = {-myqt.y, -myqt.y}, K2 = {myqt.z, myqt.z};
    const int myq = live ? qi_s[col] : 0;
    const uint32_t myob = live ? ob_s[col] : 0;
    uint4 *dst = a.surv + static_cast<int64_t>(myq) * a.cap;
    uint32_t livex = 0;  // bit x - 1: this lane's column of column block x (queries 32 x .. 32 x + 31) is a member and not at home here
#pragma unroll
    for (int x = 1; x < QB; x++)
        if (col + 32 * x < cnt && !((home_mask_s[x] >> col) & 1u)) livex |= 1u << (x - 1);
    const v4i_t *ap = tile + (b * S) * 64 + lane;  // this wave's current block
    const v4i_t *qb_mine = qb_s + half * 32 + col;   // B operand of step s: qb_mine[s * 64]
    for (; b < b1; b += NWV, ap += static_cast<int64_t>(NWV) * S * 64) {
        // the row's side of the test: lane & 31 is the row whose terms these are (rows outside the chunk: zeros)
        const float4 mraw = narrow ? metar : make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // (narrow epilogue: an appended row's bounds come from these)
        const float4 mcur = stream_row_terms(a.metric, metar);
        if (!narrow && lane < 32) {  // [3][32]: a lane reads the terms of its four adjacent rows as one vector each
            terms_w[lane] = mcur.x;
            terms_w[32 + lane] = mcur.y;
            terms_w[64 + lane] = mcur.z;
        }
        v16i_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v16i_t accx[QB > 1 ? QB - 1 : 1];  // (the further column blocks)
#pragma unroll
        for (int x = 0; x < (QB > 1 ? QB - 1 : 1); x++) accx[x] = acc;
        {
            const v4i_t *p = ap;
            const v4i_t *qp = qb_mine;
#pragma unroll 1
            for (int s0 = 0; s0 + PF < S; s0 += PF) {  // a real loop: unrolled, all S query operands leave LDS at once
#pragma unroll
                for (int u = 0; u < PF; u++) {
                    const v4i_t cur = av[u];
                    av[u] = p[(PF + u) * 64];
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, qp[u * 64], acc, 0, 0, 0);
#pragma unroll
                    for (int x = 1; x < QB; x++) accx[x - 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, qp[u * 64 + x * S * 64], accx[x - 1], 0, 0, 0);
                }
                p += PF * 64;
                qp += PF * 64;
            }
#pragma unroll
            for (int u = 0; u < PF; u++) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[u], qp[u * 64], acc, 0, 0, 0);
#pragma unroll
                for (int x = 1; x < QB; x++) accx[x - 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[u], qp[u * 64 + x * S * 64], accx[x - 1], 0, 0, 0);
            }
        }
        metar = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (b + NWV < b1) {  // the next block's first operands and row terms, before this block's epilogue
            const v4i_t *np = ap + static_cast<int64_t>(NWV) * S * 64;
#pragma unroll
            for (int u = 0; u < PF; u++) av[u] = np[u * 64];
            const int64_t row = (b + NWV) * 32 + (lane & 31);
            if (row >= r0 && row < r1) metar = a.cmeta[row];
        }
        if (a.dbg & 1) {  // ablation: no epilogue (results are wrong)
            if (acc[0] == 0x7fffffff) dst[0] = make_uint4(1, 0, 0, 0);
            continue;
        }
        // C/D: lane holds column `col` (the query); register g is row (g & 3) + 8 (g >> 2) + 4 half of the block
        const int rel = static_cast<int>(b * 32 - rb0);  // row 0 of the block, relative to the list
        if (narrow) {
            // ---- eight live columns at a time into LDS; then lane = row, one half of the wave per query
            const int r = rel + col;
            const bool valid = r >= rel0 && r < rel1;
            for (int c0 = 0; c0 < cnt; c0 += kNarrow) {
                if (col >= c0 && col < c0 + kNarrow && live) {
#pragma unroll
                    for (int g = 0; g < 16; g++) tile_w[(col - c0) * 33 + (g & 3) + 8 * (g >> 2) + 4 * half] = acc[g];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int cend = c0 + kNarrow < cnt ? c0 + kNarrow : cnt;
                for (int q0 = c0; q0 < cend; q0 += 2) {
                    const int qx = q0 + half;
                    const bool act = qx < cend;
                    const int qxc = act ? qx : q0;
                    const int dot = tile_w[(qxc - c0) * 33 + col];
                    const bool pass = valid && act && !((home_mask >> qxc) & 1u) && !stream_reject(dot, mcur, qt_s[qxc]);  // NaN (no bound) survives
                    const uint64_t pb = __ballot(pass);
                    if (pb) {  // (most blocks of most lists append nothing)
                        const uint32_t pm = half ? static_cast<uint32_t>(pb >> 32) : static_cast<uint32_t>(pb);
                        const int n = __popc(pm);
                        uint32_t base = 0;
                        if (col == 0 && n > 0)
                            base = __hip_atomic_fetch_add(a.surv_cnt + qi_s[qxc], static_cast<uint32_t>(n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        base = __shfl(base, half * 32, kWave);
                        if (pass) {
                            const uint32_t slot = base + __popc(pm & ((1u << col) - 1u));
                            if (slot < a.cap) {
                                float lb, ub;
                                code_bounds(a.metric, dot, qs_s[qxc], mraw, mraw.w, lb, ub);
                                a.surv[static_cast<int64_t>(qi_s[qxc]) * a.cap + slot] =
                                    make_uint4(ob_s[qxc] + static_cast<uint32_t>(r), static_cast<uint32_t>(rb0) + static_cast<uint32_t>(r),
                                               __float_as_uint(lb), __float_as_uint(ub));
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the next columns overwrite tile_w
            }
            continue;
        }
        // ---- wide groups: lane = query, sixteen rows each: five operations per (query, row) -- convert, multiply, two
        // fused multiply-adds, compare -- and one LDS broadcast of the row's terms
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // meta_s written by lanes 0-31 above is read by every lane below
        uint32_t pmask = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {  // registers 4j .. 4j+3 = rows 8j + 4 half + (0..3): their terms are adjacent in LDS
            const v4f_t X = *reinterpret_cast<const v4f_t *>(terms_w + 8 * j + 4 * half);
            const v4f_t Y = *reinterpret_cast<const v4f_t *>(terms_w + 32 + 8 * j + 4 * half);
            const v4f_t Z = *reinterpret_cast<const v4f_t *>(terms_w + 64 + 8 * j + 4 * half);
#pragma unroll
            for (int h = 0; h < 2; h++) {  // two rows per (packed) operation, written out: left to itself the compiler pairs
                                           // rows across vectors and keeps twenty copies of the three query terms
                const v2f_t d = {static_cast<float>(acc[4 * j + 2 * h]), static_cast<float>(acc[4 * j + 2 * h + 1])};
                const v2f_t x = {X[2 * h], X[2 * h + 1]}, y = {Y[2 * h], Y[2 * h + 1]}, z = {Z[2 * h], Z[2 * h + 1]};
                const v2f_t lhs = d * x;
                const v2f_t rhs = __builtin_elementwise_fma(nQ2, y, __builtin_elementwise_fma(z, K2, P2));
                if (!(lhs[0] < rhs[0])) pmask |= 1u << (4 * j + 2 * h);
                if (!(lhs[1] < rhs[1])) pmask |= 1u << (4 * j + 2 * h + 1);
            }
        }
        if (!live) pmask = 0;
        if (rel < rel0 || rel + 32 > rel1) {  // the first and the last block of a chunk: rows of the neighbours
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int i = rel + (g & 3) + 8 * (g >> 2) + 4 * half;
                if (i < rel0 || i >= rel1) pmask &= ~(1u << g);
            }
        }
        if (a.dbg & 8) pmask = pmask == 0x12345u ? 1u : 0u;  // ablation: the test, no append (results are wrong)
        const int n = __popc(pmask);
        if (__ballot(n > 0)) {  // (most blocks of most lists append nothing)
            const int on = __shfl_xor(n, 32, kWave);
            uint32_t base = 0;
            if (half == 0 && n + on > 0 && !(a.dbg & 4))  // (ablation 4: no counter, every entry lands on slot 0..)
                base = __hip_atomic_fetch_add(a.surv_cnt + myq, static_cast<uint32_t>(n + on), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            base = __shfl(base, col, kWave);
            if (half) base += on;  // half 1 writes behind half 0's `on` entries
            if (DEFER) {
                // the half-precision pass follows and replaces the bounds of every entry: none are computed here (with them
                // -- the products parked in LDS, the row's terms re-read, two square roots per entry -- appending 3 % of the
                // candidates cost as much as the matrix work of all of them: 75 of 214 us at batch 1024)
                for (uint32_t pm = pmask; pm; pm &= pm - 1, base++) {
                    const int g = __ffs(pm) - 1;
                    const uint32_t r = static_cast<uint32_t>(rel + (g & 3) + 8 * (g >> 2) + 4 * half);
                    if (base < a.cap) dst[base] = make_uint4(myob + r, static_cast<uint32_t>(rb0) + r, 0xff800000u, 0x7f800000u);  // [-inf, +inf]
                }
            } else {
                // the products go to LDS and each lane walks ITS set bits: kept in registers and picked by index, the tile
                // would stay live through this path in two copies and cost the kernel half its occupancy
#pragma unroll
                for (int g = 0; g < 16; g++) tile_w[g * kWave + lane] = acc[g];
                if (n > 0) {
                    const QueryScal myqs = qs_s[col];
                    for (uint32_t pm = pmask; pm; pm &= pm - 1, base++) {
                        const int g = __ffs(pm) - 1;
                        const uint32_t r = static_cast<uint32_t>(rel + (g & 3) + 8 * (g >> 2) + 4 * half);
                        if (base < a.cap) {
                            const float4 mt = a.cmeta[rb0 + r];
                            float lb, ub;
                            code_bounds(a.metric, tile_w[g * kWave + lane], myqs, mt, mt.w, lb, ub);
                            dst[base] = make_uint4(myob + r, static_cast<uint32_t>(rb0) + r, __float_as_uint(lb), __float_as_uint(ub));
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int x = 1; x < QB; x++) {  // ---- the further column blocks of the group: the same test on their accumulators, entries without bounds
            const v16i_t &acc1 = accx[x - 1];
            const bool live1 = (livex >> (x - 1)) & 1u;
            const float4 myqt1 = qt_s[col + 32 * x];  // (from LDS per block: held in registers they cost the kernel a workgroup per CU)
            const v2f_t P21 = {myqt1.x, myqt1.x}, nQ21 = {-myqt1.y, -myqt1.y}, K21 = {myqt1.z, myqt1.z};
            uint32_t pm1 = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const v4f_t X = *reinterpret_cast<const v4f_t *>(terms_w + 8 * j + 4 * half);
                const v4f_t Y = *reinterpret_cast<const v4f_t *>(terms_w + 32 + 8 * j + 4 * half);
                const v4f_t Z = *reinterpret_cast<const v4f_t *>(terms_w + 64 + 8 * j + 4 * half);
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const v2f_t d = {static_cast<float>(acc1[4 * j + 2 * h]), static_cast<float>(acc1[4 * j + 2 * h + 1])};
                    const v2f_t x = {X[2 * h], X[2 * h + 1]}, y = {Y[2 * h], Y[2 * h + 1]}, z = {Z[2 * h], Z[2 * h + 1]};
                    const v2f_t lhs = d * x;
                    const v2f_t rhs = __builtin_elementwise_fma(nQ21, y, __builtin_elementwise_fma(z, K21, P21));
                    if (!(lhs[0] < rhs[0])) pm1 |= 1u << (4 * j + 2 * h);
                    if (!(lhs[1] < rhs[1])) pm1 |= 1u << (4 * j + 2 * h + 1);
                }
            }
            if (!live1) pm1 = 0;
            if (rel < rel0 || rel + 32 > rel1) {
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int i = rel + (g & 3) + 8 * (g >> 2) + 4 * half;
                    if (i < rel0 || i >= rel1) pm1 &= ~(1u << g);
                }
            }
            const int n1 = __popc(pm1);
            if (__ballot(n1 > 0)) {
                const int myq1 = live1 ? qi_s[col + 32 * x] : 0;
                const uint32_t myob1 = live1 ? ob_s[col + 32 * x] : 0;
                uint4 *dst1 = a.surv + static_cast<int64_t>(myq1) * a.cap;
                const int on1 = __shfl_xor(n1, 32, kWave);
                uint32_t base = 0;
                if (half == 0 && n1 + on1 > 0)
                    base = __hip_atomic_fetch_add(a.surv_cnt + myq1, static_cast<uint32_t>(n1 + on1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = __shfl(base, col, kWave);
                if (half) base += on1;
                for (uint32_t pm = pm1; pm; pm &= pm - 1, base++) {
                    const int g = __ffs(pm) - 1;
                    const uint32_t r = static_cast<uint32_t>(rel + (g & 3) + 8 * (g >> 2) + 4 * half);
                    if (base < a.cap) dst1[base] = make_uint4(myob1 + r, static_cast<uint32_t>(rb0) + r, 0xff800000u, 0x7f800000u);  // [-inf, +inf]
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // the next block overwrites meta_s
    }
}

// ------------------------------------------------------------------------------------------------
// Step 1b (batches from a few hundred queries up): the survivors once more, against HALF-precision rows.
//
// What the int8 bounds cannot separate is a band of 2 W around the k-th distance (W ~ 0.017 in cosine units at 768
// dimensions: on clustered data ~3 % of a query's candidates, ~1000 rows of 3 KB each -- at batch 1024 the exact pass
// moves 3 GB and is the largest kernel of the search).  A second copy of the list rows in fp16 (1.5 KB, with a power-of-two
// scale per row) narrows the band seventy-fold: |q.v - q.v'| <= |q| |v - v'| with |v - v'| ~ 2^-12 |v| MEASURED per row when
// the copy is made.  This kernel fetches the half row of every survivor, replaces the entry's (lb, ub) by the tighter
// pair, and the finish kernel -- whose first act is the k-th smallest upper bound of the list -- then fetches f32 rows
// for little more than k candidates.  The distances, and with them every result bit, still come from the f32 rows.
//
//   cosine   c = 1 - s (q.h) / (|q||v|)              lb/ub = c -/+ E,      E = 1.01 res / |v| + 4e-5
//   dot      d = -s (q.h)                            lb/ub = d -/+ |q| E,  E = 1.01 res + 2e-5 |v|
//   L2       r = sqrt(sum (q_i - s h_i)^2)           lb/ub = r (1 -/+ 8e-6) -/+ E,  E = 1.01 res    (triangle inequality)
// res = |v - s h| as computed from the stored halves; the absolute terms carry the f32 rounding of BOTH summations (this
// one and the exact pass's: < (4 NCH + 8) 2^-24 relative to |q||v| each, 3.4e-6 at 3072 dimensions) and of the norms.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h4_t __attribute__((ext_vector_type(4)));

template <int NCH>
__global__ __launch_bounds__(kWG) void quantize_rows_half_kernel(const float *rows, int64_t ld, int64_t n, int metric,
                                                                 uint2 *half, float4 *meta) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t row = static_cast<int64_t>(blockIdx.x) * kNWave + (threadIdx.x >> 6);
    if (row >= n) return;
    const int nvec = static_cast<int>(ld / 4);
    float4 r[NCH];
    load_row<NCH>(r, rows + row * ld, nvec, lane, true);
    bool bad;
    const float mx = wave_absmax<NCH>(r, bad);
    // s = 2^(e - 14), e = the exponent of the largest component: the scaled row fills fp16's range from the top
    int field = static_cast<int>((__float_as_uint(mx) >> 23) & 0xffu) - 14;
    field = field < 1 ? 1 : (field > 253 ? 253 : field);
    const float s = __uint_as_float(static_cast<uint32_t>(field) << 23), is = __uint_as_float(static_cast<uint32_t>(254 - field) << 23);
    float res = 0.0f, v2 = 0.0f;
    uint2 *dst = half + row * nvec;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const float e[4] = {r[c].x, r[c].y, r[c].z, r[c].w};
        h4_t h;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            h[j] = static_cast<_Float16>(e[j] * is);
            const float back = static_cast<float>(h[j]) * s;  // (a power-of-two scale: exact)
            const float d = e[j] - back;
            res = __builtin_fmaf(d, d, res);
            v2 = __builtin_fmaf(back, back, v2);
        }
        if (c * kWave + lane < nvec) dst[c * kWave + lane] = __builtin_bit_cast(uint2, h);
    }
    res = __builtin_sqrtf(wave_sum(res));
    const float nv = __builtin_sqrtf(wave_sum(lane_partial<NCH, false>(r, r)));
    float E;
    if (metric == METRIC_COS) E = 1.01f * res / nv + 4.0e-5f;
    else if (metric == METRIC_DOT) E = 1.01f * res + 2.0e-5f * nv;
    else E = 1.01f * res;
    if (bad || !(E >= 0.0f)) E = __uint_as_float(0x7fc00000u);  // NaN: no bound, the exact path
    v2 = wave_sum(v2);  // |v'|^2 of the row as stored (the Euclidean home-list bounds: |q - v'|^2 = |q|^2 - 2 q.v' + |v'|^2)
    if (lane == 0) meta[row] = make_float4(s, E, v2, 1.0f / nv);
}

__device__ __forceinline__ void half_bounds(int metric, float sum, float qn, float4 mt, float &lb, float &ub) {
    if (metric == METRIC_L2) {
        const float r = __builtin_sqrtf(sum);
        lb = r * (1.0f - 8.0e-6f) - mt.y;
        ub = r * (1.0f + 8.0e-6f) + mt.y;
        return;
    }
    const float dh = sum * mt.x;
    if (metric == METRIC_DOT) {
        const float W = qn * mt.y;
        lb = -dh - W;
        ub = -dh + W;
        return;
    }
    const float c = 1.0f - dh * ((1.0f / qn) * mt.w);  // |q| or |v| zero: NaN, no bound
    lb = c - mt.y;
    ub = c + mt.y;
}

// ------------------------------------------------------------------------------------------------
// Step 1a (large batches): the HOME LIST of a query -- its nearest list, where nearly all of its int8 survivors sit (its
// cluster: ~970 of 977 rows on the bench index) -- against half-precision rows ON THE MATRIX CORES, once per list for all
// the queries it is home to.
//
// ivf_mid_kernel fetches a half row per (query, survivor): at batch 4096 four queries on average share a home list and
// each of them reads its ~970 rows for itself -- 6 GB requested, served by the L2s at their limit (591 of 1,337 us).  Here
// a workgroup takes (list, group of <= GQ of the queries it is home to, chunk of rows): the queries' rows are split into
// two fp16 planes (hi + 2^-12 lo: a 22-bit significand, the split's own residual measured) and staged in LDS as B operands,
// the list rows stream from the row-major fp16 copy straight into A operands (lane = (row, 16-byte piece): 64 contiguous
// bytes per row and instruction), v_mfma_f32_16x16x32_f16 does the reduction -- no cross-lane sum at all -- and the
// epilogue writes the pair (lb, ub) of EVERY row of the list for every query of the group into a dense array
// dh[query][row in list].  The bounds pass then appends nothing for a (query, home list) pair and the per-query pass reads
// 8 bytes per home row instead of 1.5 KB.
//
//   q = s_q (hi + 2^-12 lo) + e_q,  v = s h + e_v:   |q.v - s_q s (hi + 2^-12 lo).h| <= (|q| + |e_q|) |e_v| + |e_q| |v|
// and the matrix cores' f32 accumulation of the exact fp16 x fp16 products is within 1e-4 |q||v| of the exact sum (768 +
// 48 roundings of 2^-23 -- twice the unit roundoff, whatever the order or the rounding mode of the adder tree -- times
// sum |a_i b_i| <= |a||b|; measured: 7e-8, test_home_list_bounds_hold_and_are_tight).  So with the row's stored E (1.01 res / |v| + 4e-5 for the cosine):
//   cosine   c = 1 - s_q s S / (|q||v|)     lb/ub = c -/+ (E + 1.01 |e_q| / |q| + 1.1e-4)
//   dot      d = -s_q s S                   lb/ub = d -/+ (|q| E + 1.01 |e_q| |v| + 1.1e-4 |q||v|)
//   L2       d2 = |q|^2 - 2 s_q s S + |v'|^2        lb/ub = sqrt(d2 -/+ eps) (1 -/+ 8e-6) -/+ E,  eps = 2 (1.1e-4 |q| + 1.01 |e_q|) |v'| + 4e-6 (...)
// (|v'|^2 of the stored row sits in the row's meta word; E = 1.01 res: the triangle inequality on |v - v'|.)  The distances,
// and with them every result bit, still come from the f32 rows in the finish kernel.
// ------------------------------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef float v4f32_t __attribute__((ext_vector_type(4)));

// queries per group: what keeps both planes of the group within 48 KB of LDS
__host__ __device__ constexpr int home_group(int nch) { return nch <= 3 ? 16 : (nch <= 6 ? 8 : 4); }
constexpr float kHomeAccum = 1.1e-4f;  // accumulation allowance, relative to |q||v|

struct HomeDesc {  // rows [rb0 + r0_off, rb0 + r1_off) of the list that starts at rb0, queries order[q0 .. q0 + cnt)
    int64_t rb0;
    int32_t r0_off, r1_off;
    int32_t q0, cnt;
    int32_t list, pad;
};

struct HomeArgs {
    const HomeDesc *items;   // written by ivf_worklist_kernel's second workgroup
    const int32_t *nitems;
    const int32_t *qorder;   // the queries in the order of their nearest list
    const uint2 *half;       // [rows][ld / 4] four halves each (row-major)
    const float4 *hmeta;     // (scale, E, |v as stored|^2, 1 / |v|): the squared norm feeds the Euclidean home-list bounds
    int64_t ld;              // elements per row, a multiple of 128
    const float *Q;
    int64_t qld;
    int32_t dim, metric;
    float2 *dh;              // [nq][hstride] (lb, ub) of row r of the query's home list
    int64_t hstride;
};

// bounds from the matrix cores' sum S (in units of s_q s): sq = s_q, eq = 1.01 |e_q|, qn = |q|
__device__ __forceinline__ void home_bounds(int metric, float S, float sq, float eq, float qn, float4 mt, float &lb, float &ub) {
    const float dh = (S * sq) * mt.x;
    if (metric == METRIC_L2) {
        // |q - v'|^2 = |q|^2 - 2 q.v' + |v'|^2 with the dot product from the matrix cores: within 2 (1.1e-4 |q| + |e_q|) |v'| of
        // it, + the f32 rounding of the three terms (4e-6 of their magnitudes); then the triangle inequality on |v - v'| = res
        const float q2 = qn * qn, nv = __builtin_sqrtf(mt.z);
        const float d2 = (q2 - 2.0f * dh) + mt.z;
        const float eps = 2.0f * __builtin_fmaf(kHomeAccum, qn, eq) * nv * (1.0f + 1.0e-6f) + 4.0e-6f * ((q2 + mt.z) + 2.0f * __builtin_fabsf(dh));
        const float lo2 = d2 - eps, hi2 = d2 + eps;
        lb = __builtin_sqrtf(lo2 > 0.0f ? lo2 : 0.0f) * (1.0f - 8.0e-6f) - mt.y;  // (NaN anywhere: a bound that excludes nothing, or NaN)
        ub = __builtin_sqrtf(hi2) * (1.0f + 8.0e-6f) + mt.y;
        if (!(eps >= 0.0f)) lb = ub = __uint_as_float(0x7fc00000u);
        return;
    }
    if (metric == METRIC_DOT) {
        const float nv = 1.0f / mt.w;
        const float W = __builtin_fmaf(qn, mt.y, __builtin_fmaf(eq, nv, kHomeAccum * (qn * nv))) * (1.0f + 1.0e-6f);
        lb = -dh - W;
        ub = -dh + W;
        return;
    }
    const float c = 1.0f - dh * ((1.0f / qn) * mt.w);  // |q| or |v| zero: NaN, no bound
    const float W = (mt.y + eq / qn + kHomeAccum) * (1.0f + 1.0e-6f);
    lb = c - W;
    ub = c + W;
}

template <int NCH, int PF>
__global__ __launch_bounds__(kWG) void ivf_home_kernel(HomeArgs a) {
    constexpr int GQ = home_group(NCH);
    constexpr int SMAX = NCH * 8;  // steps of 32 elements
    __shared__ __align__(16) h8_t qh_s[SMAX * 4 * GQ];  // [step][piece][query]: the B operand of (step, lane) is one 16-byte read
    __shared__ __align__(16) h8_t ql_s[SMAX * 4 * GQ];
    __shared__ float sq_s[16], eq_s[16], qn_s[16];
    __shared__ int32_t qi_s[16];
    const int item = blockIdx.x;
    if (item >= *a.nitems) return;
    const HomeDesc d = a.items[item];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int S = static_cast<int>(a.ld >> 5);
    const int m = lane & 15, kg = lane >> 4;
    const int nrows = d.r1_off - d.r0_off;
    const int nblk = (nrows + 15) >> 4;
    if (nrows <= 0 || d.cnt <= 0) return;
    const h8_t *hbase = reinterpret_cast<const h8_t *>(a.half);
    const int64_t ld8 = a.ld >> 3;

    // ---- the load stream of this wave: blocks wave, wave + 4, ... of 16 rows, S steps each, PF loads in flight
    int ls = 0, lblk = wave;
    auto rowptr = [&](int blk) {
        int r = d.r0_off + blk * 16 + m;
        r = r < d.r1_off ? r : d.r1_off - 1;  // rows past the chunk: the last row again (computed, not used)
        return hbase + (d.rb0 + r) * ld8 + kg;
    };
    const h8_t *lp = rowptr(lblk < nblk ? lblk : 0);
    h8_t av[PF];
    auto issue = [&](h8_t &dst) {
        if (lblk < nblk) dst = lp[ls * 4];
        if (++ls == S) {
            ls = 0;
            lblk += kNWave;
            if (lblk < nblk) lp = rowptr(lblk);
        }
    };
#pragma unroll
    for (int u = 0; u < PF; u++) issue(av[u]);  // on their way while the group is set up

    // ---- the group's queries: two fp16 planes each, in B-operand order
    for (int n = wave; n < GQ; n += kNWave) {
        if (n >= d.cnt) {
            if (lane == 0) qi_s[n] = -1;
            continue;
        }
        const int qi = a.qorder[d.q0 + n];
        float4 q[NCH];
        load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
        bool bad;
        const float mx = wave_absmax<NCH>(q, bad);
        const float qn = query_norm<NCH>(q);  // (|q| in the GEMV order's arithmetic: cosine and dot scale by it, Euclidean squares it)
        // s_q = 2^(e - 14): the scaled query fills fp16's range from the top (as the rows do)
        int field = static_cast<int>((__float_as_uint(mx) >> 23) & 0xffu) - 14;
        field = field < 1 ? 1 : (field > 253 ? 253 : field);
        const float s = __uint_as_float(static_cast<uint32_t>(field) << 23), is = __uint_as_float(static_cast<uint32_t>(254 - field) << 23);
        float res = 0.0f;
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const float e[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
            _Float16 hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float x = e[j] * is;
                hi[j] = static_cast<_Float16>(x);
                const float r1 = x - static_cast<float>(hi[j]);
                lo[j] = static_cast<_Float16>(r1 * 4096.0f);
                const float back = (static_cast<float>(hi[j]) + static_cast<float>(lo[j]) * (1.0f / 4096.0f)) * s;
                const float dd = e[j] - back;
                res = __builtin_fmaf(dd, dd, res);
            }
            // element k = 256 c + 4 lane + j: step 8 c + (lane >> 3), piece (lane >> 1) & 3, halves 4 (lane & 1) + j
            const int step = 8 * c + (lane >> 3);
            if (step < S) {
                const int at = ((step * 4 + ((lane >> 1) & 3)) * GQ + n) * 8 + 4 * (lane & 1);
                typedef _Float16 h4v __attribute__((ext_vector_type(4)));
                *reinterpret_cast<h4v *>(reinterpret_cast<_Float16 *>(qh_s) + at) = h4v{hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<h4v *>(reinterpret_cast<_Float16 *>(ql_s) + at) = h4v{lo[0], lo[1], lo[2], lo[3]};
            }
        }
        res = __builtin_sqrtf(wave_sum(res));
        if (lane == 0) {
            const bool nob = bad || !(mx > 0.0f);
            qi_s[n] = qi;
            sq_s[n] = s;
            eq_s[n] = nob ? __uint_as_float(0x7fc00000u) : 1.01f * res;  // NaN: no bound for this query
            qn_s[n] = qn;
        }
    }
    __syncthreads();

    const int myq = m < GQ ? qi_s[m] : -1;  // C/D: lane & 15 is the column = the query
    const float mysq = myq >= 0 ? sq_s[m] : 0.0f, myeq = myq >= 0 ? eq_s[m] : 0.0f, myqn = myq >= 0 ? qn_s[m] : 0.0f;
    float2 *out = a.dh + static_cast<int64_t>(myq >= 0 ? myq : 0) * a.hstride;
    const h8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int blk = wave; blk < nblk; blk += kNWave) {
        const int rel = d.r0_off + blk * 16 + 4 * kg;  // this lane's four result rows, relative to the list
        float4 mt[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int r = rel + j < d.r1_off ? rel + j : d.r1_off - 1;
            mt[j] = a.hmeta[d.rb0 + r];
        }
        v4f32_t acch = {0.0f, 0.0f, 0.0f, 0.0f}, accl = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int s0 = 0; s0 < S; s0 += PF) {
#pragma unroll
            for (int u = 0; u < PF; u++) {
                const h8_t cur = av[u];
                issue(av[u]);
                const int at = ((s0 + u) * 4 + kg) * GQ + (m < GQ ? m : 0);
                const h8_t bh = m < GQ ? qh_s[at] : zero8, bl = m < GQ ? ql_s[at] : zero8;
                acch = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur, bh, acch, 0, 0, 0);
                accl = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur, bl, accl, 0, 0, 0);
            }
        }
        if (myq >= 0) {
            float lb[4], ub[4];
#pragma unroll
            for (int j = 0; j < 4; j++) home_bounds(a.metric, __builtin_fmaf(accl[j], 1.0f / 4096.0f, acch[j]), mysq, myeq, myqn, mt[j], lb[j], ub[j]);
            float4 *o = reinterpret_cast<float4 *>(out + rel);  // rel is a multiple of 4: 32-byte aligned
            o[0] = make_float4(lb[0], ub[0], lb[1], ub[1]);
            o[1] = make_float4(lb[2], ub[2], lb[3], ub[3]);
        }
    }
}

// One workgroup per query behind ivf_home_kernel: the k-th smallest upper bound over the rows of the query's home list -- k
// candidates are at most that far -- becomes its threshold (tau, which the bounds pass then meets every other list with), and
// the rows that threshold does not exclude (little more than k) start the query's survivor list.  The bounds of up to
// kHomeKeep x 256 rows stay in registers between the two passes (a longer list is read again); the kept rows are placed by
// one block-wide scan of the threads' counts.
constexpr int kHomeKeep = 8;
struct HomeSelectArgs {
    const float2 *dh;
    int64_t hstride;
    const Pair *pairs;  // [nq][nprobe]: pair 0 = the nearest list
    int32_t nq, nprobe, k;
    uint4 *surv;        // [nq][cap]
    int64_t cap;
    uint32_t *surv_cnt; // 0 (or the routing's overflow mark: such a query is left alone) -> entries written
    uint32_t *first;    // [nq] the same count: what a later pass over the list need not look at again
    uint32_t *tau;
};
static __global__ __launch_bounds__(kWG) void ivf_home_select_kernel(HomeSelectArgs a) {
    __shared__ __align__(16) uint32_t ubv_s[kWG];
    __shared__ uint32_t kth_s, wsum_s[kNWave];
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const uint32_t nsv0 = a.surv_cnt[qi];
    const Pair hp = a.pairs[static_cast<int64_t>(qi) * a.nprobe];
    if (tid == 0) a.first[qi] = 0u;
    if (nsv0 > a.cap) return;  // (a bucket overflowed in the routing: the finish kernel walks the candidate stream)
    const int hlen = static_cast<int>(hp.row_end - hp.row_begin);
    if (hlen <= 0) return;
    const float2 *hd = a.dh + static_cast<int64_t>(qi) * a.hstride;
    const float inf = __builtin_inff();
    float2 b[kHomeKeep];
#pragma unroll
    for (int j = 0; j < kHomeKeep; j++) {
        const int r = tid + j * kWG;
        b[j] = r < hlen ? hd[r] : make_float2(inf, inf);
    }
    float ub_min = inf;
#pragma unroll
    for (int j = 0; j < kHomeKeep; j++) ub_min = b[j].y < ub_min ? b[j].y : ub_min;  // (NaN: no upper bound, not counted)
    for (int r = tid + kHomeKeep * kWG; r < hlen; r += kWG) {
        const float u = hd[r].y;
        ub_min = u < ub_min ? u : ub_min;
    }
    const uint32_t v = tau_encode(ub_min);
    ubv_s[tid] = v;
    __syncthreads();
    {
        int rank = 0;
        for (int j = 0; j < kWG; j += 4) {
            const uint4 o = *reinterpret_cast<const uint4 *>(ubv_s + j);  // uniform address: an LDS broadcast
            rank += (o.x < v || (o.x == v && j < tid)) ? 1 : 0;
            rank += (o.y < v || (o.y == v && j + 1 < tid)) ? 1 : 0;
            rank += (o.z < v || (o.z == v && j + 2 < tid)) ? 1 : 0;
            rank += (o.w < v || (o.w == v && j + 3 < tid)) ? 1 : 0;
        }
        if (rank == a.k - 1) kth_s = v;  // ranks are a permutation of 0..255: exactly one thread (k <= 256)
    }
    __syncthreads();
    const float T = tau_decode(kth_s);
    // rows this thread keeps: NaN (no bound) stays
    uint32_t cnt = 0;
#pragma unroll
    for (int j = 0; j < kHomeKeep; j++) cnt += (tid + j * kWG < hlen && !(b[j].x > T)) ? 1u : 0u;
    for (int r = tid + kHomeKeep * kWG; r < hlen; r += kWG) cnt += !(hd[r].x > T) ? 1u : 0u;
    uint32_t incl = cnt;
    for (int o = 1; o < kWave; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o, kWave);
        if (lane >= o) incl += u;
    }
    if (lane == kWave - 1) wsum_s[wave] = incl;
    __syncthreads();
    uint32_t at = incl - cnt, total = 0;
#pragma unroll
    for (int w = 0; w < kNWave; w++) {
        const uint32_t c = wsum_s[w];
        at += w < wave ? c : 0u;
        total += c;
    }
    uint4 *sv = a.surv + static_cast<int64_t>(qi) * a.cap;
    const uint32_t hob = hp.ord_base, hrb = static_cast<uint32_t>(hp.row_begin);
#pragma unroll
    for (int j = 0; j < kHomeKeep; j++) {
        const int r = tid + j * kWG;
        if (r < hlen && !(b[j].x > T)) {
            if (at < a.cap) sv[at] = make_uint4(hob + static_cast<uint32_t>(r), hrb + static_cast<uint32_t>(r), __float_as_uint(b[j].x), __float_as_uint(b[j].y));
            at++;
        }
    }
    for (int r = tid + kHomeKeep * kWG; r < hlen; r += kWG) {
        const float2 e = hd[r];
        if (!(e.x > T)) {
            if (at < a.cap) sv[at] = make_uint4(hob + static_cast<uint32_t>(r), hrb + static_cast<uint32_t>(r), __float_as_uint(e.x), __float_as_uint(e.y));
            at++;
        }
    }
    if (tid == 0) {
        const bool fits = total <= a.cap;
        a.surv_cnt[qi] = fits ? total : static_cast<uint32_t>(a.cap) + 1u;  // (more than fit: the walk of the candidate stream)
        a.first[qi] = fits ? total : 0u;
        if (T < inf) (void)__hip_atomic_fetch_min(a.tau + qi, tau_encode(T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same with ONE WAVE per query, four queries per workgroup (round 5, batches from 2048 queries, k <= 64).  A workgroup per
// query is a chain of dependent round trips -- counter, table entry, the list's bounds, rank, stores: ~16 us whatever the
// batch -- of which a CU holds eight: batch 16384 took 8 rounds = 133 us.  A wave per query is the same chain four times as
// often per CU; nothing in it needs more than 64 lanes (16 bounds per lane at 977 rows, the k-th smallest of the 64 lane
// minima by v_readlane ranks, a DPP prefix sum for the places), and there is no barrier left.
constexpr int kHomeKeepW = 16;
static __global__ __launch_bounds__(kWG) void ivf_home_select_wave_kernel(HomeSelectArgs a) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int qi = static_cast<int>(blockIdx.x) * kNWave + wave;
    if (qi >= a.nq) return;
    const uint32_t nsv0 = a.surv_cnt[qi];
    const Pair hp = a.pairs[static_cast<int64_t>(qi) * a.nprobe];
    if (lane == 0) a.first[qi] = 0u;
    if (nsv0 > a.cap) return;  // (a bucket overflowed in the routing: the finish kernel walks the candidate stream)
    const int hlen = static_cast<int>(hp.row_end - hp.row_begin);
    if (hlen <= 0) return;
    const float2 *hd = a.dh + static_cast<int64_t>(qi) * a.hstride;
    const float inf = __builtin_inff();
    float2 b[kHomeKeepW];
#pragma unroll
    for (int j = 0; j < kHomeKeepW; j++) {
        const int r = lane + j * kWave;
        b[j] = r < hlen ? hd[r] : make_float2(inf, inf);
    }
    float ub_min = inf;
#pragma unroll
    for (int j = 0; j < kHomeKeepW; j++) ub_min = b[j].y < ub_min ? b[j].y : ub_min;  // (NaN: no upper bound, not counted)
    for (int r = lane + kHomeKeepW * kWave; r < hlen; r += kWave) {
        const float u = hd[r].y;
        ub_min = u < ub_min ? u : ub_min;
    }
    // the k-th smallest of the 64 lanes' minima: k candidates are at most that far (ranks are a permutation of 0..63)
    const uint32_t v = tau_encode(ub_min);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kWave; j++) {
        const uint32_t o = __builtin_amdgcn_readlane(v, j);
        rank += (o < v || (o == v && j < lane)) ? 1 : 0;
    }
    const uint32_t kth = __builtin_amdgcn_readlane(v, __ffsll(static_cast<unsigned long long>(__ballot(rank == a.k - 1))) - 1);
    const float T = tau_decode(kth);
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < kHomeKeepW; j++) cnt += (lane + j * kWave < hlen && !(b[j].x > T)) ? 1 : 0;  // NaN (no bound) stays
    for (int r = lane + kHomeKeepW * kWave; r < hlen; r += kWave) cnt += !(hd[r].x > T) ? 1 : 0;
    const int incl = wave_scan_incl(cnt);
    const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(incl, kWave - 1));
    uint32_t at = static_cast<uint32_t>(incl - cnt);
    uint4 *sv = a.surv + static_cast<int64_t>(qi) * a.cap;
    const uint32_t hob = hp.ord_base, hrb = static_cast<uint32_t>(hp.row_begin);
#pragma unroll
    for (int j = 0; j < kHomeKeepW; j++) {
        const int r = lane + j * kWave;
        if (r < hlen && !(b[j].x > T)) {
            if (at < a.cap) sv[at] = make_uint4(hob + static_cast<uint32_t>(r), hrb + static_cast<uint32_t>(r), __float_as_uint(b[j].x), __float_as_uint(b[j].y));
            at++;
        }
    }
    for (int r = lane + kHomeKeepW * kWave; r < hlen; r += kWave) {
        const float2 e = hd[r];
        if (!(e.x > T)) {
            if (at < a.cap) sv[at] = make_uint4(hob + static_cast<uint32_t>(r), hrb + static_cast<uint32_t>(r), __float_as_uint(e.x), __float_as_uint(e.y));
            at++;
        }
    }
    if (lane == 0) {
        const bool fits = total <= a.cap;
        a.surv_cnt[qi] = fits ? total : static_cast<uint32_t>(a.cap) + 1u;  // (more than fit: the walk of the candidate stream)
        a.first[qi] = fits ? total : 0u;
        if (T < inf) (void)__hip_atomic_fetch_min(a.tau + qi, tau_encode(T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// Step 2: survivors -> f32 distances -> the k nearest -> results.
// ------------------------------------------------------------------------------------------------
// ---- queries that are too much for one workgroup -------------------------------------------------------------------
// Large batches give a query ONE workgroup in the half-precision pass and in the finish kernel (its list is short: ~1000
// int8 survivors, then ~15).  A query with far more -- rows the int8 bounds cannot separate from its k-th neighbour by the
// thousand, or a list that overflowed and sends it through the plain f32 scan of all its candidates -- would hold the
// whole launch for milliseconds (measured: one such query of 1024, 5.4 ms).  ivf_heavy_kernel flags them (bit 30 of
// surv_cnt) and lists them; the workgroups of the main grids skip a flagged query, and kHeavySlots x heavy_slices extra
// workgroups appended to the same launches walk the list with many slices per query.
constexpr uint32_t kHeavyBit = 0x40000000u;
constexpr int kHeavySlots = 32;
struct HeavyArgs {
    uint32_t *surv_cnt;
    int32_t nq;
    uint32_t thr;      // fewest survivors of a heavy query (an overflowed list, marked or counted, always is one)
    uint32_t cap;      // survivors that fit a list
    uint32_t times_mean;  // ... and at least this many times the batch's mean (4; tests: 0)
    uint32_t *cnt;     // [1]
    int32_t *list;     // [nq]
    // home-list batches: the queries the bounds pass appended more than `few` candidates to (and that are not heavy) are listed
    // as well -- the only ones the per-survivor half-precision pass behind it has work for (a few per cent of a batch)
    const uint32_t *first;  // optional [nq] entries the home-list launch left
    uint32_t few;
    uint32_t *todo_cnt;     // [1]
    int32_t *todo;          // [nq]
};
static __global__ __launch_bounds__(1024) void ivf_heavy_kernel(HeavyArgs a) {
    __shared__ uint32_t n_s, t_s;
    __shared__ unsigned long long sum_s;
    if (threadIdx.x == 0) {
        n_s = 0;
        t_s = 0;
        sum_s = 0;
    }
    __syncthreads();
    // heavy = far above what this batch's queries typically carry (four times the mean, and at least a.thr), or overflowed:
    // on a 10M-row index every query has ~10k survivors, and one workgroup each is the right share of the chip for them
    unsigned long long mine = 0;
    for (int i = threadIdx.x; i < a.nq; i += 1024) {
        const uint32_t raw = a.surv_cnt[i];
        if (raw <= a.cap) mine += raw;
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0 && mine) atomicAdd(&sum_s, mine);
    __syncthreads();
    const unsigned long long four_mean = static_cast<unsigned long long>(a.times_mean) * sum_s / static_cast<unsigned long long>(a.nq > 0 ? a.nq : 1);
    const uint32_t thr = four_mean > a.thr ? (four_mean < a.cap ? static_cast<uint32_t>(four_mean) : a.cap) : a.thr;
    for (int i = threadIdx.x; i < a.nq; i += 1024) {
        const uint32_t raw = a.surv_cnt[i];
        if (raw > thr) {
            a.list[atomicAdd(&n_s, 1u)] = i;
            a.surv_cnt[i] = raw | kHeavyBit;
        } else if (a.first && raw <= a.cap && raw > a.first[i] + a.few) {
            a.todo[atomicAdd(&t_s, 1u)] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        *a.cnt = n_s;
        if (a.first) *a.todo_cnt = t_s;
    }
}

struct FinishArgs {
    const uint4 *surv;     // (order key, list row, lb bits, ub bits)
    const uint32_t *surv_cnt;
    uint32_t *tau;         // keeps falling while the survivors are evaluated
    int64_t cap;
    const int32_t *q_cnt;  // candidates per query (the fallback walks them all)
    const Pair *pairs;     // [nq][nprobe]
    int32_t nq, nprobe, k;
    int32_t slices;        // workgroups per query
    const int32_t *qorder; // optional (slices == 1): the queries in the order of their nearest list
    int32_t span;          // entries a wave looks at per step: 64, or 16 for a handful of queries
    const float *rows;     // list rows, f32
    const float *row_norms;
    int64_t ld;
    const float *Q;
    int64_t qld;
    int32_t dim, metric;
    uint64_t *partial;     // [nq][pstride]: [slices][k] (k <= 64) or [slices][4][k] per query -- or the keys of a short list, see direct
    int64_t pstride;       // keys per query in there
    int32_t direct;        // survivor lists of up to `direct` entries (<= 1024; 0: never) skip the per-wave and per-workgroup
                           // lists: a key per survivor, written where it was evaluated; the query's last workgroup picks the k smallest
    uint32_t *done;        // [nq], zero between calls
    const int32_t *listids;
    int32_t *out_ids;      // [nq][k]
    float *out_dist;       // [nq][k]
    uint32_t *out_gord;    // optional [nq][k]
    unsigned long long *stats;  // optional: [0] += f32 rows evaluated, [1] += candidates
    int32_t bisect_min;    // smallest k whose final merge bisects the key space (A/B: by insertion below it)
    const uint32_t *heavy_cnt;  // optional (ivf_heavy_kernel): ivf_finish_heavy_kernel walks heavy_list with heavy_slices
    const int32_t *heavy_list;  // workgroups per query; the main kernel skips the queries flagged kHeavyBit
    int32_t heavy_slices, main_blocks;
    int32_t adapt;         // spread a short survivor list over all the waves of the query's workgroups (see span)
    int32_t prepass;       // the upper bounds are worth a look first (ivf_mid_kernel has tightened them)
    uint32_t *host_flag;   // optional (small synchronous calls, results in mapped host memory): the workgroup that writes the LAST
    uint32_t *done_q;      // query's results (counted in done_q, zero between calls) sets *host_flag = flag_val for the spinning caller
    uint32_t flag_val;
    unsigned long long *dbg;  // -DHG_IVF_STAMPS diagnostic builds only
};

// One workgroup's share of query qi: slice sl of its nsv survivors (or, nsv > cap, of its candidate stream), and -- as the
// last workgroup of the query -- the merge and the results.  The body of ivf_finish_kernel.
template <int NCH, int RB, bool L2>
__device__ __forceinline__ void finish_wg(const FinishArgs &a, int qi, int sl, int slices, uint32_t nsv, float tau, unsigned char *smem) {
    __shared__ int tail_last;
    __shared__ __align__(16) uint32_t ub_s[kWG];
    __shared__ uint32_t ub_kth;
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);  // [kNWave][k] (+ the tail's final list and results)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const bool over = nsv > a.cap;  // survivors did not fit: walk the candidate stream itself (the plain f32 scan)
    const int64_t total = over ? a.q_cnt[qi] : nsv;
    // entries a wave looks at per step: no more than spreads the list over every wave of the query's workgroups (a wave
    // fetches four rows at a time: sixteen entries are four dependent round trips, four entries one)
    int span = a.span;
    if (a.adapt) {
        int64_t need = (total + static_cast<int64_t>(slices) * kNWave - 1) / (static_cast<int64_t>(slices) * kNWave);
        need = (need + 3) & ~3LL;
        span = need < 4 ? 4 : (need < span ? static_cast<int>(need) : span);
    }
    const int gran = kNWave * span;
    int64_t per = (total + slices - 1) / slices;
    per = (per + gran - 1) / gran * gran;
    const int64_t i0 = static_cast<int64_t>(sl) * per;
    const int64_t i1 = i0 + per < total ? i0 + per : total;
    // the threshold keeps falling in this kernel too: a wave that holds k exact distances folds its k-th into tau[qi]
    // (k candidates are at most that far), every wave re-reads it once per step
    const bool share = slices <= 8;
    const uint4 *sv = a.surv + static_cast<int64_t>(qi) * a.cap;
    const Pair *pp = a.pairs + static_cast<int64_t>(qi) * a.nprobe;
    const bool regk = a.k <= kWave;
    uint64_t *mylist = lists + wave * a.k;
    int cnt = 0;
    uint64_t thr = ~0ull, mine = ~0ull;
    // A short list spread over many workgroups (a handful of queries: dozens of slices, a step or two per wave): every key
    // straight to partial[qi][i].  The per-wave lists took one insertion per key (~0.15 us each, serial), wave 0 then
    // inserted the other three waves' keys (3 us), and the merge read slices x k keys -- no fewer than the list itself.
    const bool direct = a.direct > 0 && !over && regk && slices > 1 && a.k >= a.bisect_min && nsv <= static_cast<uint32_t>(a.direct);
    uint64_t *dkeys = a.partial + static_cast<int64_t>(qi) * a.pstride;
    // ---- before any f32 row is fetched: the survivors' UPPER bounds give a threshold of their own.  k candidates whose
    // upper bounds are <= T are k candidates at most T away, so D_k <= T; T = the k-th smallest of the 256 threads'
    // minima over a strided share of the list each (the k smallest upper bounds sit in k different shares but for the odd
    // collision, which costs one rank).  Only behind ivf_mid_kernel: the int8 upper bounds are no tighter than the
    // threshold the bounds pass ran with (measured: not one row fewer), the half-precision ones leave little more than k
    // candidates.  (Every workgroup of a query reads the whole list: 16 B per survivor, from L2.)
    if (a.prepass && !over && a.k <= kWG && nsv > 2u * kWave) {  // (a list one step evaluates anyway gains nothing -- e.g. a compacted one)
        float m = __builtin_inff();
        for (uint32_t i = threadIdx.x; i < nsv; i += kWG) {
            const float u = __uint_as_float(sv[i].w);
            m = u < m ? u : m;  // (NaN: no upper bound, not counted)
        }
        const uint32_t v = tau_encode(m);
        ub_s[threadIdx.x] = v;
        __syncthreads();
        int rank = 0;
        for (int j = 0; j < kWG; j += 4) {
            const uint4 o = *reinterpret_cast<const uint4 *>(ub_s + j);  // uniform address: an LDS broadcast
            rank += (o.x < v || (o.x == v && j < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.y < v || (o.y == v && j + 1 < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.z < v || (o.z == v && j + 2 < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.w < v || (o.w == v && j + 3 < static_cast<int>(threadIdx.x))) ? 1 : 0;
        }
        if (rank == a.k - 1) ub_kth = v;  // ranks are a permutation of 0..255: exactly one thread
        __syncthreads();
        const float T = tau_decode(ub_kth);
        tau = T < tau ? T : tau;
    }
    const int nvec = static_cast<int>(a.ld / 4);
    float4 q[NCH];
    load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
    const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
    unsigned long long nsurv = 0;
    for (int64_t base = i0 + wave * span; base < i1; base += gran) {
        const int64_t i = base + lane;
        const bool in = lane < span && i < i1;
        uint32_t o = 0, erow = 0;
        float l = -__builtin_inff();
        if (in) {
            if (over) {
                o = static_cast<uint32_t>(i);
            } else {
                const uint4 e = sv[i];
                o = e.x;
                erow = e.y;
                l = __uint_as_float(e.z);
            }
        }
        // (shared across waves only where a wave has many steps: with dozens of slices per query every wave takes one or
        // two steps, and hundreds of agent-scope atomics on one address queue up, ~1 us each)
        const uint32_t tnext = share ? coherent_load(a.tau + qi) : 0xffffffffu;  // used by the NEXT step: in flight under this one's rows
        const bool s = in && !(l > tau);  // NaN (no bound) survives
        uint64_t m = __ballot(s);
        if (!m) {
            if (direct && in) coherent_store(dkeys + i, static_cast<uint64_t>(~0ull));
            const float tg = tau_decode(tnext);
            tau = tg < tau ? tg : tau;
            continue;
        }
        nsurv += __popcll(m);
        // a survivor carries its list row; the fallback resolves the candidate's position in the stream to one (the last
        // pair whose ord_base <= ord: lists of length 0 share an ord_base with their successor, as in ivf_decode_kernel)
        int64_t myrow = erow;
        if (over) {
            int lo = 0, hi = a.nprobe - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pp[mid].ord_base <= o) lo = mid;
                else hi = mid - 1;
            }
            myrow = s ? pp[lo].row_begin + (o - pp[lo].ord_base) : 0;
        }
        const float myrn = (s && a.metric == METRIC_COS) ? a.row_norms[myrow] : 0.0f;
        const int rlo = static_cast<int>(myrow), rhi = static_cast<int>(myrow >> 32);
        float dmine = 0.0f;  // a surviving lane ends up with ITS candidate's distance
        while (m) {
            float4 r[RB][NCH];
            int js[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) {  // (straight-line: with the loads under `if (m)` and a `break` below, RB = 2 cost 57 registers more than RB = 1)
                const bool has = m != 0;
                const int j = has ? __ffsll(static_cast<unsigned long long>(m)) - 1 : 0;
                js[b] = has ? j : -1;
                m &= m - 1;  // (0 stays 0)
                const int64_t row = (static_cast<int64_t>(__builtin_amdgcn_readlane(rhi, j)) << 32) |
                                    static_cast<uint32_t>(__builtin_amdgcn_readlane(rlo, j));
                load_row<NCH>(r[b], a.rows + (has ? row : 0) * a.ld, nvec, lane, has);
            }
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const int j = js[b] < 0 ? 0 : js[b];
                const float sum = wave_sum(lane_partial<NCH, L2>(q, r[b]));
                const float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), j));
                const float dv = finish_dist(a.metric, sum, qn, rn);
                dmine = lane == js[b] ? dv : dmine;
            }
        }
        // the step's keys into this wave's top-k, one at a time (the insertion code exists once, not once per row in flight)
        const uint64_t key = s ? make_key(dmine, o) : ~0ull;
        if (direct) {
            if (in) coherent_store(dkeys + i, key);
            continue;
        }
        uint64_t mask = __ballot(key < thr);
        while (mask) {
            const int bl = __ffsll(static_cast<unsigned long long>(mask)) - 1;
            mask &= mask - 1;
            const uint64_t kb = lane_bcast(key, bl);
            if (kb < thr) {
                if (regk) {
                    wave_insert_reg(mine, cnt, a.k, kb, lane);
                    thr = wave_kth_reg(mine, a.k);  // ~0 until the list is full
                } else {
                    wave_insert(mylist, cnt, a.k, kb, lane);
                    thr = cnt == a.k ? mylist[a.k - 1] : ~0ull;
                }
            }
        }
        {
            const float tg = tau_decode(tnext);
            tau = tg < tau ? tg : tau;
            const float kd = thr == ~0ull ? __builtin_inff() : key_dist(thr);  // k exact distances at or below kd: D_k <= kd
            if (kd < tau) {
                tau = kd;
                if (lane == 0 && share) (void)__hip_atomic_fetch_min(a.tau + qi, tau_encode(kd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    HG_IVF_STAMP(a.dbg, 25, blockIdx.x == 0 && threadIdx.x == 0);  // ... has evaluated its survivors
    // profiling only; few atomics on purpose (one per workgroup WITH survivors, one per query: they all hit one address)
    if (a.stats) {
        __shared__ unsigned long long nsurv_s[kNWave];
        if (lane == 0) nsurv_s[wave] = nsurv;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long tot = 0;
            for (int w = 0; w < kNWave; w++) tot += nsurv_s[w];
            if (tot) atomicAdd(a.stats, tot);
            if (sl == 0) atomicAdd(a.stats + 1, static_cast<unsigned long long>(a.q_cnt[qi]));
        }
    }
    // ---- this workgroup's partial list (k <= 64: its four lists merged into one first)
    if (direct) {
        // (nothing to hand over but the keys already written)
    } else if (regk) {
        if (wave != 0 && lane < a.k) lists[wave * a.k + lane] = mine;
        __syncthreads();
        if (wave == 0) {
            for (int w = 1; w < kNWave; w++)
                for (int j = 0; j < a.k; j++) {
                    const uint64_t key = lists[w * a.k + j];  // uniform address: an LDS broadcast
                    if (!(key < thr)) break;
                    wave_insert_reg(mine, cnt, a.k, key, lane);
                    thr = wave_kth_reg(mine, a.k);
                }
            if (slices > 1) {
                uint64_t *dstp = dkeys + static_cast<int64_t>(sl) * a.k;
                if (lane < a.k) coherent_store(dstp + lane, mine);
            }
        }
    } else {
        uint64_t *dstp = dkeys + (static_cast<int64_t>(sl) * kNWave + wave) * a.k;
        for (int i = lane; i < a.k; i += kWave) coherent_store(dstp + i, i < cnt ? mylist[i] : static_cast<uint64_t>(~0ull));
    }
    // ---- tail: the last workgroup of query qi merges the partial lists, maps the winners to row ids
    // (ivf_flat.clj:291-294) and writes the results (the hand-over protocol of scan_kernel's fused tail)
    const bool alone = regk && slices == 1;  // the query's only workgroup: wave 0 holds the result, nothing to hand over
    if (!alone) {
        wait_stores_acked();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(a.done + qi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail_last = prev == static_cast<uint32_t>(slices) - 1 ? 1 : 0;
            if (tail_last) __hip_atomic_store(a.done + qi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!tail_last) return;
    }
    HG_IVF_STAMP(a.dbg, 26, qi == 0 && threadIdx.x == 0);  // the last workgroup of query 0 begins the merge
    const int64_t nkeys = direct ? static_cast<int64_t>(nsv) : static_cast<int64_t>(slices) * (regk ? 1 : kNWave) * a.k;
    uint32_t *ord_s = reinterpret_cast<uint32_t *>(smem + sizeof(uint64_t) * (kNWave + 1) * a.k);  // [k]
    float *dist_s = reinterpret_cast<float *>(ord_s + a.k);                                         // [k]
    // (the probe table of the query -- where each probed list starts in the candidate stream and in the list rows -- is
    // fetched side by side while the lists are merged: the winners are then resolved against LDS instead of a chain of
    // dependent global reads)
    int64_t *prb_s = reinterpret_cast<int64_t *>(smem + sizeof(uint64_t) * (kNWave + 2) * a.k);  // [nprobe] row_begin
    const int npe = (a.nprobe + 1) & ~1;
    uint32_t *pob_s = reinterpret_cast<uint32_t *>(prb_s + a.nprobe);                             // [nprobe] ord_base
    uint32_t *pgo_s = pob_s + npe;                                                                // [nprobe] gord_base
    if (wave == kNWave - 1)
        for (int p = lane; p < a.nprobe; p += kWave) {
            const Pair pr = pp[p];
            prb_s[p] = pr.row_begin;
            pob_s[p] = pr.ord_base;
            pgo_s[p] = pr.gord_base;
        }
    if (alone) {
        __syncthreads();  // (the probe table in LDS)
        if (wave != 0) return;
        if (lane < a.k) {
            ord_s[lane] = mine != ~0ull ? static_cast<uint32_t>(mine) : 0xffffffffu;
            dist_s[lane] = mine != ~0ull ? key_dist(mine) : __uint_as_float(0x7f800000u);
        }
    } else if (regk && a.k >= a.bisect_min) {  // the k smallest of the partial lists by bisection (topk_small_wg), not by insertion
        const uint64_t *part = dkeys;
        uint64_t *scr = reinterpret_cast<uint64_t *>(pgo_s + npe);  // [kNWave][k]
        const uint64_t *fin = topk_small_wg(nkeys, a.k, lists, lists + kNWave * a.k, scr, [&](int64_t i) { return coherent_load(part + i); });
        if (wave != 0) return;
        if (lane < a.k) {
            const uint64_t key = fin[lane];
            ord_s[lane] = key != ~0ull ? static_cast<uint32_t>(key) : 0xffffffffu;
            dist_s[lane] = key != ~0ull ? key_dist(key) : __uint_as_float(0x7f800000u);
        }
    } else {
        MergeArgs mg;
        mg.partial = dkeys;  // (this query's keys: "query 0" of the merge)
        mg.keys_per_query = nkeys;
        mg.nq = 0;
        mg.k = a.k;
        mg.out_ord = nullptr;
        mg.out_dist = nullptr;
        merge_topk_wg<true>(mg, 0, kNWave, smem, ord_s, dist_s);  // waves 1..3 return from it after its barrier
    }
    if (wave != 0) return;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < a.k; i += kWave) {
        const uint32_t o = ord_s[i];
        uint32_t go = 0xffffffffu;
        int32_t id = -1;
        if (o != 0xffffffffu) {
            int p = 0, hi = a.nprobe - 1;  // the last pair whose ord_base <= o (as ivf_decode_kernel), by bisection
            while (p < hi) {
                const int mid = (p + hi + 1) >> 1;
                if (pob_s[mid] <= o) p = mid;
                else hi = mid - 1;
            }
            id = a.listids[prb_s[p] + (o - pob_s[p])];
            go = pgo_s[p] + (o - pob_s[p]);
        }
        a.out_ids[static_cast<int64_t>(qi) * a.k + i] = id;
        a.out_dist[static_cast<int64_t>(qi) * a.k + i] = dist_s[i];
        if (a.out_gord) a.out_gord[static_cast<int64_t>(qi) * a.k + i] = go;
    }
    HG_IVF_STAMP(a.dbg, 27, qi == 0 && lane == 0);  // results written
    if (a.host_flag) {
        wait_stores_acked();  // this query's results are out
        if (lane == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(a.done_q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == static_cast<uint32_t>(a.nq) - 1) {
                __hip_atomic_store(a.done_q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(a.host_flag, a.flag_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_finish_kernel(FinishArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    // slice-major: a query's survivors sit in the first slices of its list when it is short, and query-major order would
    // put the busy workgroups of all queries on the same XCDs
    int qi = blockIdx.x % a.nq;
    const int sl = a.qorder ? 0 : blockIdx.x / a.nq;  // (ordered queries: one slice, the grid is padded to whole XCD rounds)
    if (a.qorder) {  // workgroup b runs on XCD b % 8: XCD x takes a contiguous eighth of the ordered queries
        const int per = (a.nq + 7) >> 3;
        const int pos = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (pos >= a.nq || (blockIdx.x >> 3) >= per) return;
        qi = a.qorder[pos];
    }
    HG_IVF_STAMP(a.dbg, 24, blockIdx.x == 0 && threadIdx.x == 0);  // first workgroup of the finish kernel starts
    const uint32_t nsv = a.surv_cnt[qi];
    if (a.heavy_cnt && (nsv & kHeavyBit)) return;  // served by ivf_finish_heavy_kernel
    finish_wg<NCH, RB, L2>(a, qi, sl, a.slices, nsv, tau_decode(a.tau[qi]), smem);
}

// The heavy queries (ivf_heavy_kernel), heavy_slices workgroups each: kHeavySlots x heavy_slices workgroups walk the list.
// (A launch of its own: with the loop around it, the finish body costs 160 registers instead of 95, and the main grid --
// thousands of latency chains -- lives on its occupancy.  When no query is heavy, its workgroups read one word and leave.)
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_finish_heavy_kernel(FinishArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int slot = blockIdx.x % kHeavySlots, sl = blockIdx.x / kHeavySlots;
    const int n = static_cast<int>(*a.heavy_cnt);
    for (int t = slot; t < n; t += kHeavySlots) {
        const int qi = a.heavy_list[t];
        finish_wg<NCH, RB, L2>(a, qi, sl, a.heavy_slices, a.surv_cnt[qi] & ~kHeavyBit, tau_decode(a.tau[qi]), smem);
        __syncthreads();  // (wave 0 may still be writing the results from LDS the next query's waves would overwrite)
    }
}

struct MidArgs {
    uint4 *surv;               // [nq][cap] (order key, list row, lb bits, ub bits): lb / ub are replaced
    uint32_t *surv_cnt;
    uint32_t *tau;
    int64_t cap;
    int32_t nq, slices;
    int32_t k;                 // compact > 0 (slices == 1): the workgroup sees the query's whole list and leaves only the
    int32_t compact;           // entries the k-th smallest upper bound does not exclude (dynamic LDS: 4 B x compact entries)
    const int32_t *qorder;     // optional (slices == 1): the queries in the order of their nearest list (as the finish kernel)
    const uint2 *half;         // [rows][ld / 4] four halves each
    const float4 *hmeta;       // (scale, E, |v as stored|^2, 1 / |v|)
    int64_t ld;
    const float *Q;
    int64_t qld;
    int32_t dim, metric;
    const uint32_t *heavy_cnt;  // optional (ivf_heavy_kernel), as in FinishArgs
    const int32_t *heavy_list;
    int32_t heavy_slices, main_blocks;
    const uint32_t *first;      // optional [nq]: the launch behind the bounds pass of a home-list batch -- entries [0, first) are what
                                // ivf_home_select_kernel left (their bounds stand), [first, nsv) what the bounds pass appended
    int32_t first_few;          // ... and a query it appended no more than this many entries to is left to the finish kernel as it is
    const uint32_t *todo_cnt;   // with first: the queries that have such work, listed by ivf_heavy_kernel: todo_slices workgroups
    const int32_t *todo;        // each, sixteen entries per wave and step (one WORKGROUP walking a few hundred entries is ~45 us of
    int32_t todo_slices;        // dependent gathers); their lists are not compacted -- the finish kernel ranks the upper bounds
};

// NW waves per workgroup: 4, or 8 for the one-workgroup-per-query launches of batches that do not fill the chip otherwise
// (batch 1024: 1024 workgroups of four waves are 16 waves per CU, under half of what fits)
template <int NCH, int RB, bool L2, int NW>
__device__ __forceinline__ void mid_query_wg(const MidArgs &a, int qi, int sl, int slices, uint32_t nsv, bool may_compact,
                                             unsigned char *smem) {
    constexpr int kT = NW * kWave;  // threads
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    if (nsv > a.cap) return;  // the list overflowed: the finish kernel walks the candidate stream instead
    // (the launch behind the bounds pass of a home-list batch: entries [0, f0) are what the home-list launch left -- their
    // bounds stand, no half row is fetched for them; a query the bounds pass appended nothing to is done)
    const uint32_t f0 = a.first ? a.first[qi] : 0u;
    if (a.first && nsv <= f0 + static_cast<uint32_t>(a.first_few)) return;  // (a few strays: the finish kernel fetches their f32 rows outright)
    const int span = a.todo ? 16 : kWave;  // entries a wave looks at per step
    const int gran = NW * span;
    int64_t per = (static_cast<int64_t>(nsv) + slices - 1) / slices;
    per = (per + gran - 1) / gran * gran;
    const int64_t i0 = static_cast<int64_t>(sl) * per;
    const int64_t i1 = i0 + per < nsv ? i0 + per : nsv;
    if (i0 >= i1) return;
    const int nvec = static_cast<int>(a.ld / 4);
    float4 q[NCH];
    load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
    const float qn = L2 ? 0.0f : query_norm<NCH>(q);
    uint4 *sv = a.surv + static_cast<int64_t>(qi) * a.cap;
    float *lb_s = reinterpret_cast<float *>(smem);  // [compact] the entries' new lower bounds
    const bool compact = may_compact && a.compact > 0 && nsv <= static_cast<uint32_t>(a.compact) && a.k <= kT;
    float ub_min = __builtin_inff();                // over this thread's entries
    for (int64_t base = i0 + wave * span; base < i1; base += gran) {
        const int64_t i = base + lane;
        const bool in = lane < span && i < i1;
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        float4 mt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (in) {
            e = sv[i];
            mt = a.hmeta[e.y];
        }
        uint64_t m = __ballot(in && i >= f0);
        float mysum = 0.0f;
        while (m) {
            uint2 w[RB][NCH];
            int js[RB];
            float sc[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) {  // (straight-line, as in finish_wg: conditional loads and a `break` cost registers)
                const bool has = m != 0;
                const int j = has ? __ffsll(static_cast<unsigned long long>(m)) - 1 : 0;
                js[b] = has ? j : -1;
                m &= m - 1;  // (0 stays 0)
                const int64_t row = has ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(e.y), j)) : 0;
                sc[b] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mt.x), j));
                const uint2 *rp = a.half + row * nvec;
#pragma unroll
                for (int c = 0; c < NCH; c++) w[b][c] = has && c * kWave + lane < nvec ? rp[c * kWave + lane] : make_uint2(0u, 0u);
            }
            float accs[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) {
                float acc = 0.0f;
#pragma unroll
                for (int c = 0; c < NCH; c++) {
                    const h4_t h = __builtin_bit_cast(h4_t, w[b][c]);
                    const float hv[4] = {static_cast<float>(h[0]), static_cast<float>(h[1]), static_cast<float>(h[2]), static_cast<float>(h[3])};
                    const float qv[4] = {q[c].x, q[c].y, q[c].z, q[c].w};
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (L2) {
                            const float d = qv[j] - hv[j] * sc[b];
                            acc = __builtin_fmaf(d, d, acc);
                        } else {
                            acc = __builtin_fmaf(qv[j], hv[j], acc);
                        }
                    }
                }
                accs[b] = acc;
            }
            const float tot = rows_sum_to_lane<RB>(accs, lane);  // lane b: row b's sum (eight rows: one halving exchange)
#pragma unroll
            for (int b = 0; b < RB; b++) {
                const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), b));
                mysum = lane == js[b] ? sum : mysum;
            }
        }
        if (in) {
            float lb = __uint_as_float(e.z), ub = __uint_as_float(e.w);
            if (i >= f0) {
                half_bounds(a.metric, mysum, qn, mt, lb, ub);
                // (both pairs hold: the tighter of each -- v_max / v_min return the other operand for a NaN)
                lb = __builtin_fmaxf(lb, __uint_as_float(e.z));
                ub = __builtin_fminf(ub, __uint_as_float(e.w));
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint32_t *>(sv + i) + 2) = make_uint2(__float_as_uint(lb), __float_as_uint(ub));
            }
            if (compact) {
                lb_s[i] = lb;
                ub_min = ub < ub_min ? ub : ub_min;  // (NaN: no upper bound, not counted)
            }
        }
    }
    if (!compact) return;
    // ---- the query's whole list went through this workgroup: the threshold the finish kernel would derive from the upper
    // bounds (the k-th smallest of the 256 threads' minima: k candidates at most that far) is applied here, and the list
    // shrinks to the entries it does not exclude -- little more than k -- before the finish kernel reads it
    __shared__ __align__(16) uint32_t ubv_s[kT];
    __shared__ uint32_t kth_s, wcnt_s[NW];
    const uint32_t v = tau_encode(ub_min);
    ubv_s[threadIdx.x] = v;
    __syncthreads();
    {
        int rank = 0;
        for (int j = 0; j < kT; j += 4) {
            const uint4 o = *reinterpret_cast<const uint4 *>(ubv_s + j);  // uniform address: an LDS broadcast
            rank += (o.x < v || (o.x == v && j < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.y < v || (o.y == v && j + 1 < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.z < v || (o.z == v && j + 2 < static_cast<int>(threadIdx.x))) ? 1 : 0;
            rank += (o.w < v || (o.w == v && j + 3 < static_cast<int>(threadIdx.x))) ? 1 : 0;
        }
        if (rank == a.k - 1) kth_s = v;  // ranks are a permutation of 0..255: exactly one thread
    }
    __syncthreads();
    const float T = tau_decode(kth_s);
    uint32_t nout = 0;  // entries kept so far (uniform)
    for (uint32_t base = 0; base < nsv; base += kT) {
        // in place: a step reads its 256 entries before anything is written, and writes below base + 256
        const uint32_t i = base + threadIdx.x;
        const bool keep = i < nsv && !(lb_s[i] > T);  // NaN (no bound) stays
        uint4 e = make_uint4(0u, 0u, 0u, 0u);
        if (keep) {
            const uint2 head = *reinterpret_cast<const uint2 *>(sv + i);  // (order key, list row): written by the bounds pass
            e = make_uint4(head.x, head.y, __float_as_uint(lb_s[i]), 0x7f800000u);
        }
        const uint64_t m = __ballot(keep);
        if (lane == 0) wcnt_s[wave] = static_cast<uint32_t>(__popcll(m));
        wait_stores_acked();  // (s_waitcnt vmcnt(0): this step's reads have returned before any wave writes)
        __syncthreads();
        uint32_t off = nout, tot = 0;
#pragma unroll
        for (int w = 0; w < NW; w++) {
            const uint32_t c = wcnt_s[w];
            off += w < wave ? c : 0u;
            tot += c;
        }
        if (keep) sv[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
        nout += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.surv_cnt[qi] = nout;
        if (T < __builtin_inff()) (void)__hip_atomic_fetch_min(a.tau + qi, tau_encode(T), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (Tried: finishing the query right here -- finish_wg on the ~15 entries left, no second launch for it.  The finish
    // body needs 132 VGPRs against this kernel's 84; with it inlined the bandwidth-bound half of this kernel lost half its
    // occupancy and batch 4096 went from 1.32 to 2.0 ms; as a real call the compiler reserved 264 registers and scratch.)
}

template <int NCH, int RB, bool L2, int NW>
__global__ __launch_bounds__(NW * kWave) void ivf_mid_kernel(MidArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (a.heavy_cnt && static_cast<int>(blockIdx.x) >= a.main_blocks) {  // the heavy queries, heavy_slices workgroups each
        const int hb = static_cast<int>(blockIdx.x) - a.main_blocks, slot = hb % kHeavySlots, hsl = hb / kHeavySlots;
        const int n = static_cast<int>(*a.heavy_cnt);
        for (int t = slot; t < n; t += kHeavySlots) {
            const int hq = a.heavy_list[t];
            mid_query_wg<NCH, RB, L2, NW>(a, hq, hsl, a.heavy_slices, a.surv_cnt[hq] & ~kHeavyBit, false, smem);
        }
        return;
    }
    int qi = blockIdx.x % a.nq;
    const int sl = a.qorder ? 0 : blockIdx.x / a.nq;
    if (a.todo) {  // home-list batches: only the listed queries have work
        const int n = static_cast<int>(*a.todo_cnt), ts = a.todo_slices;
        const int slot = static_cast<int>(blockIdx.x) / ts, tsl = static_cast<int>(blockIdx.x) % ts, nslots = a.main_blocks / ts;
        for (int t = slot; t < n; t += nslots) {
            const int tq = a.todo[t];
            mid_query_wg<NCH, RB, L2, NW>(a, tq, tsl, ts, a.surv_cnt[tq], false, smem);
        }
        return;
    }
    if (a.qorder) {  // workgroup b runs on XCD b % 8: XCD x takes a contiguous eighth of the ordered queries
        const int per = (a.nq + 7) >> 3;
        const int pos = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (pos >= a.nq || (blockIdx.x >> 3) >= per) return;
        qi = a.qorder[pos];
    }
    const uint32_t nsv = a.surv_cnt[qi];
    if (a.heavy_cnt && (nsv & kHeavyBit)) return;  // served by the workgroups above
    mid_query_wg<NCH, RB, L2, NW>(a, qi, sl, a.slices, nsv, true, smem);
}


// ------------------------------------------------------------------------------------------------
// k-means++ seeding (ivf_flat.clj:43-49) as a bounds pass: a round folds the distances to ONE new centre into every
// row's running minimum (scan_kernel, MODE_MINUPD: `if (d < min) min = d`).  A row whose lower bound is already >= its
// minimum keeps it whatever d is, so only the rows the new centre may actually be nearest to -- its own cluster, and
// early in the seeding the rows that have no centre nearby yet -- fetch their f32 row; the others cost their int8 row.
// Same arithmetic for the rows that are evaluated, hence the same minima, bit for bit.  (Base-order codes in the lane
// layout: the traversal's copy, kernels.hpp.)
// ------------------------------------------------------------------------------------------------
struct SeedArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld, n;
    int32_t dim, metric;
    int64_t cur;             // the new centre is base row `cur`
    const uint32_t *qrows;   // codes of the base rows (lane layout) + meta
    const float4 *qmeta;
    float *out;              // running minima [n]
    int32_t rows_per_wg;     // multiple of 32
};

template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void seed_update_kernel(SeedArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int nvec = static_cast<int>(a.ld / 4);
    float4 q[NCH];
    load_row<NCH>(q, a.rows + a.cur * a.ld, nvec, lane, true);
    const float qn = a.metric == METRIC_COS ? a.row_norms[a.cur] : 0.0f;  // as the f32 pass: the stored norm
    QueryCode<NCH> qc;
    encode_query<NCH>(q, qc);
    const int own = wave_sum8_row(lane);
    const int64_t w0 = static_cast<int64_t>(blockIdx.x) * a.rows_per_wg;
    const int64_t w1 = w0 + a.rows_per_wg < a.n ? w0 + a.rows_per_wg : a.n;
    for (int64_t base = w0 + wave * 8; base < w1; base += kNWave * 8) {
        const int64_t myrow = base + own;
        const bool ok = (lane & 7) == 0 && myrow < w1;
        const float4 mymeta = ok ? a.qmeta[myrow] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const float myold = ok ? a.out[myrow] : 0.0f;
        int acc[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int64_t row = base + b < w1 ? base + b : w1 - 1;
            const uint32_t *rp = a.qrows + (row * kWave + lane) * NCH;
            uint32_t w[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) w[c] = rp[c];
            acc[b] = code_dot<NCH>(qc.a, w);
        }
        const int tot = wave_sum8_int(acc, lane);
        const float lb = code_lower_bound(a.metric, tot, qc.sc, mymeta, mymeta.w);
        const bool need = ok && !(lb >= myold);  // NaN: evaluate
        uint64_t m = __ballot(need);             // bit 8r = row r of the block
        while (m) {
            float4 r[RB][NCH];
            int64_t rw[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) {
                rw[b] = -1;
                if (m) {
                    rw[b] = base + ((__ffsll(static_cast<unsigned long long>(m)) - 1) >> 3);
                    m &= m - 1;
                    load_row<NCH>(r[b], a.rows + rw[b] * a.ld, nvec, lane, true);
                }
            }
#pragma unroll
            for (int b = 0; b < RB; b++) {
                if (rw[b] < 0) break;
                const float s = wave_sum(lane_partial<NCH, L2>(q, r[b]));
                if (lane == 0) {
                    const float d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[rw[b]] : 0.0f);
                    const float o = a.out[rw[b]];
                    if (d < o) a.out[rw[b]] = d;
                }
            }
        }
    }
}

}  // namespace hg
