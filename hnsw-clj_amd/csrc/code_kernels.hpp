// code_kernels.hpp -- the IVF list scan on int8 rows: bounds first, f32 only where it can matter.
//
// search-partition scores every row of every probed list in f32 (ivf_flat.clj:217-234) and search-ivf-flat keeps the k
// smallest (:291-294): 96 MB of rows per query at 1M x 768 / nprobe 32 to find ten of them.  Which ten can be decided
// from a quarter of the bytes.  Every list row also exists as int8 codes with its exact residual (kernels.hpp,
// quantize_rows_kernel -- the same codes and bounds as the HNSW traversal's rejection test), so for every candidate
//     lb <= d_f32(q, v) <= ub        from one exact integer dot product (v_dot4c_i32_i8).
// Pipeline of a batch (ivf.hip: ivf_code_scan):
//   1. code_bounds_kernel (dot4c body code_group_body, or the matrix cores: code_mfma_body): lb of every (query, candidate) into the dense candidate array (position = order key);
//   2. select_topk_kernel on the lb values; ivf_tau_kernel: tau_q = the largest ub among the k smallest lb.  At least
//      k candidates have d <= ub <= tau_q, so the k-th smallest distance D_k <= tau_q;
//   3. ivf_refine_kernel: a candidate with lb > tau_q has d >= lb > tau_q >= D_k and cannot be among the k nearest --
//      nor tie with the k-th -- and becomes +inf; every other candidate ("survivor": NaN bounds included) gets its
//      distance by the GEMV scan's own arithmetic (lane_partial + wave butterfly + finish_dist), bit for bit;
//   4. select_topk_kernel again: the k smallest (distance, order key) -- exactly the result of scanning everything in
//      f32, because every candidate with d <= D_k is a survivor and carries the same bits as scan_kernel would give it.
// The survivors' arithmetic is the GEMV scan's, so this pipeline serves the batch sizes of the GEMV regime (up to 12
// (query, list) pairs per list) with unchanged bits; larger batches keep the MFMA tile scan.
#pragma once
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

// codes + bound scalars of a batch of queries: one wave per query.  Codes in [c][lane] order (conflict-free LDS reads)
template <int NCH>
__global__ __launch_bounds__(kWG) void quantize_queries_kernel(const float *Q, int64_t qld, int dim, int nq,
                                                               uint32_t *qcodes, QueryScal *qscal) {
    const int lane = threadIdx.x & (kWave - 1);
    const int qi = blockIdx.x * kNWave + (threadIdx.x >> 6);
    if (qi >= nq) return;
    float4 q[NCH];
    load_query<NCH>(q, Q + static_cast<int64_t>(qi) * qld, dim, lane);
    QueryCode<NCH> qc;
    encode_query<NCH>(q, qc);
#pragma unroll
    for (int c = 0; c < NCH; c++) qcodes[(static_cast<int64_t>(qi) * NCH + c) * kWave + lane] = qc.a[c];
    if (lane == 0) qscal[qi] = qc.sc;
}

__host__ inline size_t code_group_lds_bytes(int nch) {
    return sizeof(uint32_t) * kTileQ * nch * kWave + sizeof(int64_t) * kTileQ + sizeof(int32_t) * kTileQ +
           sizeof(QueryScal) * kTileQ;
}

// Same groups, work list and output positions as tile_scan_kernel / l2_group_kernel (TileArgs): a workgroup keeps the
// codes of a group of <= 32 queries in LDS, every wave holds eight code rows in registers and walks the group's queries
// over them; a row is fetched once per group.  Per eight (row, query) pairs: 8 * NCH v_dot4c, one halving reduction
// (wave_sum8_int), the bound in the eight lanes that own a row, one store.
template <int NCH>
__device__ __forceinline__ void code_group_body(const TileArgs &a, unsigned char *smem) {
    uint32_t *qc_s = reinterpret_cast<uint32_t *>(smem);                                  // [32][NCH][64]
    int64_t *ob_s = reinterpret_cast<int64_t *>(qc_s + kTileQ * NCH * kWave);             // [32] output bases
    int32_t *qi_s = reinterpret_cast<int32_t *>(ob_s + kTileQ);                           // [32] query index (-1 = empty)
    QueryScal *qs_s = reinterpret_cast<QueryScal *>(qi_s + kTileQ);                       // [32]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    // ---- work item -> (group, chunk): identical to tile_scan_kernel (XCD-contiguous slices of the work list)
    const int nitems = *a.nitems;
    const int per_xcd = (nitems + 7) >> 3;
    const int slot = blockIdx.x >> 3;
    if (slot >= per_xcd) return;
    const int item = (blockIdx.x & 7) * per_xcd + slot;
    if (item >= nitems) return;
    const int g = a.wi_group[item];
    const int chunk = a.wi_chunk[item];
    if (g >= *a.ngroups) return;
    const int seg = a.grp_seg[g];
    const int64_t rb0 = a.seg_off[seg], rb1 = a.seg_off[seg + 1];
    const int cnt = a.grp_mem_cnt[g];
    const int64_t tiles = (rb1 - rb0 + kTileRows - 1) / kTileRows;
    const int64_t nch = tile_nchunks(rb1 - rb0, a.chunk_rows, a.nchunks);
    if (chunk >= nch) return;
    const int64_t per = (tiles + nch - 1) / nch * kTileRows;
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * per;
    const int64_t r1 = r0 + per < rb1 ? r0 + per : rb1;
    if (r0 >= r1 || cnt <= 0) return;

    if (tid < kTileQ) {
        int qi = -1;
        int64_t ob = -1;
        if (tid < cnt) {
            const GroupMember m = a.members[a.grp_mem_begin[g] + tid];
            qi = m.q;
            ob = m.out_base;
            qs_s[tid] = a.qscal[qi];
        }
        qi_s[tid] = qi;
        ob_s[tid] = ob;
    }
    __syncthreads();
    for (int f = tid; f < cnt * NCH * kWave; f += kTileThreads) {
        const int s = f / (NCH * kWave);
        qc_s[f] = a.qcodes[static_cast<int64_t>(qi_s[s]) * NCH * kWave + (f - s * NCH * kWave)];
    }
    __syncthreads();

    const int own = wave_sum8_row(lane);  // the row of a block whose total this lane receives
    const bool owner = (lane & 7) == 0;
    for (int64_t base = r0 + wave * 8; base < r1; base += kTileWaves * 8) {
        const int64_t myrow = base + own;
        const bool mine_ok = owner && myrow < r1;
        const float4 mymeta = mine_ok ? a.cmeta[myrow] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint32_t w[8][NCH];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int64_t row = base + b < r1 ? base + b : r1 - 1;
            const uint32_t *rp = a.crows + (row * kWave + lane) * NCH;
#pragma unroll
            for (int c = 0; c < NCH; c++) w[b][c] = rp[c];
        }
        for (int q = 0; q < cnt; q++) {
            uint32_t qa[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) qa[c] = qc_s[(q * NCH + c) * kWave + lane];
            int acc[8];
#pragma unroll
            for (int b = 0; b < 8; b++) acc[b] = code_dot<NCH>(qa, w[b]);
            const int tot = wave_sum8_int(acc, lane);
            if (mine_ok) a.out[ob_s[q] + (myrow - rb0)] = code_lower_bound(a.metric, tot, qs_s[q], mymeta, mymeta.w);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The bounds pass on the matrix cores.  A group of <= 32 queries against 32 list rows is a 32 x 32 x K integer GEMM
// tile: v_mfma_i32_32x32x32_i8 does in 24 instructions (dim 768) what 8 * 24 v_dot4c + a 17-instruction exchange per 8
// rows and query do on the VALU -- the dot4c kernel above is VALU-bound as soon as lists are probed by several queries
// (Euclidean batch 1024: 1.3 ms against a 0.25 ms memory floor).  The integer dot products are exact either way, so
// both kernels write the same bits.
//
// MFMA operand maps: lane l (r = l & 31, h = l >> 5) supplies 16 bytes of A row r and of B column r per step, the same
// 16 of the step's 32 k for both operands -- WHICH k they are does not matter for a dot product as long as rows and
// queries agree, so a step is simply bytes [32 s, 32 s + 32) of the code vectors in natural element order, half h the
// upper or lower 16.  List rows are stored for that (tile layout): [block of 32 rows][step s][half h][row r][16 B] --
// every A operand is ONE contiguous 1 KB wave load, no LDS on the way.  The group's query codes sit in LDS as
// [step][half][query][16 B] (conflict-free 16-B reads).  C/D: lane l holds column l & 31 (the query), register g row
// (g & 3) + 8 (g >> 2) + 4 (l >> 5).
// ------------------------------------------------------------------------------------------------
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// dword address of natural code dword g4 (elements 4 g4 .. 4 g4 + 3) of list row R in the tile layout; S = steps per row
__host__ __device__ inline int64_t tile_code_dword(int64_t R, int g4, int S) {
    const int s = g4 >> 3, h = (g4 >> 2) & 1, w = g4 & 3;
    return ((((R >> 5) * S + s) * 2 + h) * 32 + (R & 31)) * 4 + w;
}

// codes of `n` rows into the tile layout (+ the same per-row meta as quantize_rows_kernel): one wave per row
template <int NCH>
__global__ __launch_bounds__(kWG) void quantize_rows_tile_kernel(const float *rows, int64_t ld, int64_t n, uint32_t *tile) {
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
}

__host__ inline size_t code_mfma_lds_bytes(int nch, int waves) {
    return static_cast<size_t>(nch) * 256 * kTileQ                      // query codes [step][half][query][16 B]
           + sizeof(int64_t) * kTileQ + sizeof(int32_t) * kTileQ + sizeof(QueryScal) * kTileQ
           + static_cast<size_t>(waves) * (sizeof(float) * 32 * 33 + sizeof(float4) * 32);  // per wave: result tile + row meta
}

constexpr int kCodeMfmaWaves = kTileWaves;  // both bodies run in the same launch shape

template <int NCH>
__device__ __forceinline__ void code_mfma_body(const TileArgs &a, unsigned char *smem) {
    constexpr int S = NCH * 8;  // steps of 32 bytes
    v4i_t *qb_s = reinterpret_cast<v4i_t *>(smem);                                        // [S][2][32]
    int64_t *ob_s = reinterpret_cast<int64_t *>(qb_s + S * 64);                           // [32] output bases
    int32_t *qi_s = reinterpret_cast<int32_t *>(ob_s + kTileQ);                           // [32] query index (-1 = empty)
    QueryScal *qs_s = reinterpret_cast<QueryScal *>(qi_s + kTileQ);                       // [32]
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    float *tile_s = reinterpret_cast<float *>(qs_s + kTileQ) + wave * (32 * 33 + 4 * 32);  // [32 queries][33] bounds
    float4 *meta_s = reinterpret_cast<float4 *>(tile_s + 32 * 33);                        // [32] row meta
    // ---- work item -> (group, chunk): as code_group_body
    const int nitems = *a.nitems;
    const int per_xcd = (nitems + 7) >> 3;
    const int slot = blockIdx.x >> 3;
    if (slot >= per_xcd) return;
    const int item = (blockIdx.x & 7) * per_xcd + slot;
    if (item >= nitems) return;
    const int g = a.wi_group[item];
    const int chunk = a.wi_chunk[item];
    if (g >= *a.ngroups) return;
    const int seg = a.grp_seg[g];
    const int64_t rb0 = a.seg_off[seg], rb1 = a.seg_off[seg + 1];
    const int cnt = a.grp_mem_cnt[g];
    const int64_t tiles = (rb1 - rb0 + kTileRows - 1) / kTileRows;
    const int64_t nch = tile_nchunks(rb1 - rb0, a.chunk_rows, a.nchunks);
    if (chunk >= nch) return;
    const int64_t per = (tiles + nch - 1) / nch * kTileRows;
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * per;
    const int64_t r1 = r0 + per < rb1 ? r0 + per : rb1;
    if (r0 >= r1 || cnt <= 0) return;

    if (tid < kTileQ) {
        int qi = -1;
        int64_t ob = -1;
        if (tid < cnt) {
            const GroupMember m = a.members[a.grp_mem_begin[g] + tid];
            qi = m.q;
            ob = m.out_base;
            qs_s[tid] = a.qscal[qi];
        }
        qi_s[tid] = qi;
        ob_s[tid] = ob;
    }
    __syncthreads();
    // query codes: natural order in global memory (16-B chunk t of query q = step t / 2, half t & 1); empty slots are zero
    for (int f = tid; f < kTileQ * S * 2; f += kCodeMfmaWaves * kWave) {
        const int q = f / (S * 2), t = f - q * (S * 2);
        v4i_t v = {0, 0, 0, 0};
        if (q < cnt) v = reinterpret_cast<const v4i_t *>(a.qcodes + static_cast<int64_t>(qi_s[q]) * NCH * kWave)[t];
        qb_s[t * 32 + q] = v;
    }
    __syncthreads();

    const int col = lane & 31, half = lane >> 5;
    const QueryScal myqs = col < cnt ? qs_s[col] : QueryScal{};
    // blocks of 32 rows of the tile layout that overlap [r0, r1): the layout's blocks are aligned to the WHOLE list array,
    // not to a list, so the first and the last block of a chunk may hold rows of the neighbours -- computed, not stored
    const int64_t b0 = r0 >> 5, b1 = (r1 + 31) >> 5;
    for (int64_t b = b0 + wave; b < b1; b += kCodeMfmaWaves) {
        const v4i_t *ap = reinterpret_cast<const v4i_t *>(a.ctile) + (b * S) * 64 + lane;
        if (lane < 32) {
            const int64_t row = b * 32 + lane;
            meta_s[lane] = (row >= r0 && row < r1) ? a.cmeta[row] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        v16i_t acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        v4i_t av[8];
#pragma unroll
        for (int u = 0; u < 8; u++) av[u] = ap[u * 64];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const v4i_t cur = av[s & 7];
            if (s + 8 < S) av[s & 7] = ap[(s + 8) * 64];  // eight operands (8 KB per wave) in flight
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(cur, qb_s[(s * 2 + half) * 32 + col], acc, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();  // meta_s written by lanes 0-31 above is read by every lane below
        if (col < cnt) {  // (lanes of empty query slots have nothing to say)
#pragma unroll
            for (int gq = 0; gq < 16; gq++) {
                const int i = (gq & 3) + 8 * (gq >> 2) + 4 * half;
                const float4 mt = meta_s[i];
                tile_s[col * 33 + i] = code_lower_bound(a.metric, acc[gq], myqs, mt, mt.w);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // store: lanes 0-31 the 32 rows of one query, lanes 32-63 of the next: whole lines per query
        const int64_t row = b * 32 + col;
        const bool row_ok = row >= r0 && row < r1;
        for (int q2 = 0; q2 < cnt; q2 += 2) {
            const int q = q2 + half;
            if (q < cnt && row_ok) a.out[ob_s[q] + (row - rb0)] = tile_s[q * 33 + col];
        }
        __builtin_amdgcn_wave_barrier();  // the next block overwrites meta_s / tile_s
    }
}

// ONE launch for the bounds pass: the plan kernel's choice (*a.sel, made on the device from the actual number of queries
// per probed list) selects the body -- the matrix cores, or the dot4c path when a.ctile is null or lists are probed by
// few queries.
template <int NCH>
__global__ __launch_bounds__(kTileThreads) void code_bounds_kernel(TileArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (a.ctile != nullptr && a.sel != nullptr && *a.sel != 0) code_mfma_body<NCH>(a, smem);
    else code_group_body<NCH>(a, smem);
}

// ------------------------------------------------------------------------------------------------
// k-means++ seeding (ivf_flat.clj:43-49) as a bounds pass: a round folds the distances to ONE new centre into every
// row's running minimum (scan_kernel, MODE_MINUPD: `if (d < min) min = d`).  A row whose lower bound is already >= its
// minimum keeps it whatever d is, so only the rows the new centre may actually be nearest to -- its own cluster, and
// early in the seeding the rows that have no centre nearby yet -- fetch their f32 row; the others cost their int8 row.
// Same arithmetic for the rows that are evaluated, hence the same minima, bit for bit.
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

struct TauArgs {
    const uint32_t *ord;   // [nq][k] order keys of the k smallest lower bounds (0xffffffff = none)
    const float *lb;       // [nq][k]
    const Pair *pairs;     // [nq][nprobe]
    int32_t nq, k, nprobe;
    int32_t metric;
    const uint32_t *crows;
    const float4 *cmeta;
    const uint32_t *qcodes;
    const QueryScal *qscal;
    float *tau;            // [nq]
};

// tau_q = the largest upper bound among the k candidates with the smallest lower bounds (+inf if there are fewer than k
// candidates with a bound): one wave per query, eight candidates per step, bounds recomputed from the codes.
template <int NCH>
__global__ __launch_bounds__(kWave) void ivf_tau_kernel(TauArgs a) {
    const int lane = threadIdx.x;
    const int qi = blockIdx.x;
    const uint32_t *ord = a.ord + static_cast<int64_t>(qi) * a.k;
    const float *lbv = a.lb + static_cast<int64_t>(qi) * a.k;
    const Pair *pp = a.pairs + static_cast<int64_t>(qi) * a.nprobe;
    uint32_t qa[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) qa[c] = a.qcodes[(static_cast<int64_t>(qi) * NCH + c) * kWave + lane];
    const QueryScal qs = a.qscal[qi];
    const int own = wave_sum8_row(lane);
    float tau = -__builtin_inff();
    bool open = false;  // a slot of the top k without a usable bound: no candidate can be excluded
    for (int j0 = 0; j0 < a.k; j0 += 8) {
        // lane b < 8 resolves candidate j0 + b to its list row
        int64_t myrow = -1;
        if (lane < 8 && j0 + lane < a.k) {
            const uint32_t o = ord[j0 + lane];
            const float l = lbv[j0 + lane];
            if (o != 0xffffffffu && l == l) {
                int p = 0;
                while (p + 1 < a.nprobe && pp[p + 1].ord_base <= o) p++;  // as ivf_decode_kernel
                myrow = pp[p].row_begin + (o - pp[p].ord_base);
            }
        }
        if (__ballot(lane < 8 && j0 + lane < a.k && myrow < 0)) open = true;
        int acc[8];
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const int64_t row = __shfl(myrow, b, kWave);
            const uint32_t *rp = a.crows + ((row >= 0 ? row : 0) * kWave + lane) * NCH;
            uint32_t w[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) w[c] = rp[c];
            acc[b] = code_dot<NCH>(qa, w);
        }
        const int tot = wave_sum8_int(acc, lane);
        const int64_t orow = __shfl(myrow, own, kWave);
        float ub = -__builtin_inff();
        if ((lane & 7) == 0 && orow >= 0) {
            float lb;
            const float4 mt = a.cmeta[orow];
            code_bounds(a.metric, tot, qs, mt, mt.w, lb, ub);
            if (!(ub == ub)) ub = __builtin_inff();
        }
        for (int off = 1; off < kWave; off <<= 1) {
            const float o = __shfl_xor(ub, off, kWave);
            ub = o > ub ? o : ub;
        }
        tau = ub > tau ? ub : tau;
    }
    if (lane == 0) a.tau[qi] = open ? __builtin_inff() : tau;
}

struct RefineArgs {
    float *dist;            // dense candidate array: lower bounds in, distances (or +inf) out
    const int32_t *q_cnt;   // candidates per query
    int64_t stride;
    const float *tau;
    const Pair *pairs;
    int32_t nq, nprobe;
    int32_t chunk;          // candidates per workgroup (4 * span)
    int32_t span;           // candidates per wave step: 64 or 16
    int32_t nchunks;
    const float *rows;      // list rows, f32
    const float *row_norms;
    int64_t ld;
    const float *Q;
    int64_t qld;
    int32_t dim;
    int32_t metric;
    unsigned long long *stats;  // optional: [0] += survivors, [1] += candidates
};

// Step 3 of the pipeline.  Workgroup = (query, slice of its candidates); a wave looks at 64 candidates at a time and
// computes the distance of each survivor with all 64 lanes (RB survivors in flight).
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void ivf_refine_kernel(RefineArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    // slice-major: the survivors of a query sit where its nearest lists are, at the head of its candidate stream, i.e.
    // in slice 0 -- (query-major put every busy workgroup on blockIdx = 0 mod nchunks: ONE XCD did all the work)
    const int qi = blockIdx.x % a.nq, ch = blockIdx.x / a.nq;
    const int cnt = a.q_cnt[qi];
    const int c0 = ch * a.chunk;
    if (c0 >= cnt) return;
    const int c1 = c0 + a.chunk < cnt ? c0 + a.chunk : cnt;
    float *d = a.dist + static_cast<int64_t>(qi) * a.stride;
    const Pair *pp = a.pairs + static_cast<int64_t>(qi) * a.nprobe;
    const float tau = a.tau[qi];
    const int nvec = static_cast<int>(a.ld / 4);
    float4 q[NCH];
    float qn = 0.0f;
    bool have_q = false;  // the query is fetched by the first window that has a survivor (most workgroups have none)
    unsigned long long nsurv = 0;
    // a wave looks at `span` candidates per step (64, or 16 for small batches: a full window is then 16 rows per wave
    // instead of 64, and four times as many waves share a query's survivors)
    const int span = a.span;
    for (int base = c0 + wave * span; base < c1; base += kNWave * span) {
        const int i = base + lane;
        const bool in = lane < span && i < c1;
        const float l = in ? d[i] : __builtin_inff();
        const bool surv = in && !(l > tau);  // NaN (no bound) survives
        uint64_t m = __ballot(surv);
        nsurv += __popcll(m);
        float mine = __builtin_inff();
        if (m) {
            if (!have_q) {
                load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
                qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
                have_q = true;
            }
            // every lane resolves ITS candidate to a list row (the last pair whose ord_base <= ord: lists of length 0
            // share an ord_base with their successor, as in ivf_decode_kernel) and fetches that row's norm: all 64
            // look-ups side by side, nothing dependent left inside the loop below
            int lo = 0, hi = a.nprobe - 1;
            const uint32_t o = static_cast<uint32_t>(i);
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (pp[mid].ord_base <= o) lo = mid;
                else hi = mid - 1;
            }
            const int64_t myrow = surv ? pp[lo].row_begin + (o - pp[lo].ord_base) : 0;
            const float myrn = (surv && a.metric == METRIC_COS) ? a.row_norms[myrow] : 0.0f;
            const int rlo = static_cast<int>(myrow), rhi = static_cast<int>(myrow >> 32);
            while (m) {
                float4 r[RB][NCH];
                int js[RB];
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    js[b] = -1;
                    if (m) {
                        js[b] = __ffsll(static_cast<unsigned long long>(m)) - 1;
                        m &= m - 1;
                        const int64_t row = (static_cast<int64_t>(__builtin_amdgcn_readlane(rhi, js[b])) << 32) |
                                            static_cast<uint32_t>(__builtin_amdgcn_readlane(rlo, js[b]));
                        load_row<NCH>(r[b], a.rows + row * a.ld, nvec, lane, true);
                    }
                }
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    if (js[b] < 0) break;
                    const float s = wave_sum(lane_partial<NCH, L2>(q, r[b]));
                    const float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), js[b]));
                    const float dv = finish_dist(a.metric, s, qn, rn) + 0.0f;
                    mine = lane == js[b] ? dv : mine;
                }
            }
        }
        if (in) d[i] = mine;
    }
    // profiling only; few atomics on purpose (one per wave WITH survivors, one per query): thousands of workgroups adding to
    // one address made the profiled kernel 20x slower than the product's
    if (a.stats && lane == 0) {
        if (nsurv) atomicAdd(a.stats, nsurv);
        if (wave == 0 && ch == 0) atomicAdd(a.stats + 1, static_cast<unsigned long long>(cnt));
    }
}

}  // namespace hg
