// kernels.hpp -- gfx950 (MI355X, CDNA4) device code of the hnsw-clj distance engine.
//
// One numeric contract for every kernel (oracle/oracle.c section 2 mimics it bit for bit):
//   lane l of a 64-lane wavefront accumulates elements {256c + 4l + j} (c ascending, j = 0..3) of a
//   row with one fmaf per element into a single f32 accumulator; the 64 partials are combined by
//   an xor butterfly with offsets 1,2,4,8,16,32 (x = x + shfl_xor(x, off)); cosine is
//   1 - dot / (qnorm * rownorm) with correctly rounded mul / div / sub and both norms
//   sqrt(reduce(v.v)) in the same order; L2 is sqrt(reduce((q-v)^2)); DOT is -dot.
// Rows are streamed/gathered straight into VGPRs as float4 (16 B/lane, 1 KiB per wave
// instruction): every row is used once per query, so an LDS round trip would be pure overhead
// (cdna_hip_programming.md section 5 "GEMV / M <= 16" row).  LDS holds the per-query state instead:
// candidate lists, visited bitset, per-wave top-k lists.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hg {

constexpr int kWave = 64;
constexpr int kWG = 256;
constexpr int kNWave = kWG / kWave;
constexpr int kMaxDeg = 64;  // M0 = 2M <= 64
constexpr int kGhost = 32;   // extra list slots for evicted-but-tied candidates (see hnsw kernel)

constexpr int METRIC_COS = 0, METRIC_L2 = 1, METRIC_DOT = 2;
constexpr int MODE_TOPK = 0, MODE_STORE = 1, MODE_MINUPD = 2;

// xor butterfly 1,2,4,8,16,32.  The first four steps are DPP moves (no LDS crossbar trip): after the
// xor-1 / xor-2 steps every lane of a quad holds the quad's sum, so row_half_mirror / row_mirror pair the
// same partial sums the xor-4 / xor-8 steps would (float add is commutative: bit-identical results).
__device__ __forceinline__ float wave_sum(float x) {
    int v = __float_as_int(x);
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v = __float_as_int(x);
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v = __float_as_int(x);
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = __float_as_int(x);
    x = x + __int_as_float(__builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));  // row_mirror
    x = x + __shfl_xor(x, 16, kWave);
    x = x + __shfl_xor(x, 32, kWave);
    return x;
}

// Eight rows' lane sums reduced TOGETHER, with the association of wave_sum (the xor butterfly 1, 2, 4, 8, 16, 32): at the
// first three levels a lane keeps one of two rows and hands the other to its partner (halving exchange: 8 -> 4 -> 2 -> 1
// values per lane, each level adding exactly the pair x[l] + x[l ^ off] the butterfly adds), then the single value goes
// through the last three levels -- 28 operations and two LDS-crossbar shuffles instead of eight six-step butterflies and
// sixteen.  Lane l ends with the total of row (l & 7): bit for bit wave_sum(x[l & 7]).
__device__ __forceinline__ float wave_sum8(const float (&x)[8], int lane) {
    const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0;
    float y[4], z[2];
#pragma unroll
    for (int p = 0; p < 4; p++) {  // xor 1: even lanes keep rows 0 2 4 6, odd lanes rows 1 3 5 7
        const float keep = b0 ? x[2 * p + 1] : x[2 * p], send = b0 ? x[2 * p] : x[2 * p + 1];
        y[p] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    }
#pragma unroll
    for (int p = 0; p < 2; p++) {  // xor 2: lanes with bit 1 clear keep y[0] / y[2] (rows 0|1, 4|5), the others y[1] / y[3]
        const float keep = b1 ? y[2 * p + 1] : y[2 * p], send = b1 ? y[2 * p] : y[2 * p + 1];
        z[p] = keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    }
    // xor 4: row_shl:4 into banks 0 and 2 (lanes 0-3, 8-11 read lane + 4), row_shr:4 into banks 1 and 3
    const float keep = b2 ? z[1] : z[0], send = b2 ? z[0] : z[1];
    int r = __builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x104, 0xF, 0x5, false);
    r = __builtin_amdgcn_update_dpp(r, __float_as_int(send), 0x114, 0xF, 0xA, false);
    float v = keep + __int_as_float(r);
    v = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8 = lane ^ 8
    v = v + __shfl_xor(v, 16, kWave);
    v = v + __shfl_xor(v, 32, kWave);
    return v;
}

// RB rows' lane sums -> lane b holds the total of row b (lanes >= RB: unspecified).  Eight rows go through wave_sum8, other
// counts through RB butterflies: the same bits either way.
template <int RB>
__device__ __forceinline__ float rows_sum_to_lane(float (&s)[RB], int lane) {
    if constexpr (RB == 8) {
        return wave_sum8(s, lane);
    } else {
        float mine = 0.0f;
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const float t = wave_sum(s[b]);
            mine = lane == b ? t : mine;
        }
        return mine;
    }
}

template <int NCH>
__device__ __forceinline__ void load_query(float4 (&q)[NCH], const float *Q, int dim, int lane) {
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        int e = (c * kWave + lane) * 4;
        q[c].x = e + 0 < dim ? Q[e + 0] : 0.0f;
        q[c].y = e + 1 < dim ? Q[e + 1] : 0.0f;
        q[c].z = e + 2 < dim ? Q[e + 2] : 0.0f;
        q[c].w = e + 3 < dim ? Q[e + 3] : 0.0f;
    }
}

template <int NCH>
__device__ __forceinline__ void load_row(float4 (&r)[NCH], const float *row, int nvec, int lane, bool valid) {
    const float4 *rp = reinterpret_cast<const float4 *>(row);
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        int i = c * kWave + lane;
        r[c] = (valid && i < nvec) ? rp[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <int NCH, bool L2>
__device__ __forceinline__ float lane_partial(const float4 (&q)[NCH], const float4 (&r)[NCH]) {
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        if (L2) {
            float d0 = q[c].x - r[c].x, d1 = q[c].y - r[c].y, d2 = q[c].z - r[c].z, d3 = q[c].w - r[c].w;
            acc = __builtin_fmaf(d0, d0, acc);
            acc = __builtin_fmaf(d1, d1, acc);
            acc = __builtin_fmaf(d2, d2, acc);
            acc = __builtin_fmaf(d3, d3, acc);
        } else {
            acc = __builtin_fmaf(q[c].x, r[c].x, acc);
            acc = __builtin_fmaf(q[c].y, r[c].y, acc);
            acc = __builtin_fmaf(q[c].z, r[c].z, acc);
            acc = __builtin_fmaf(q[c].w, r[c].w, acc);
        }
    }
    return acc;
}

__device__ __forceinline__ float finish_dist(int metric, float s, float qn, float rn) {
    if (metric == METRIC_L2) return __builtin_sqrtf(s);
    if (metric == METRIC_DOT) return -s;
    // IEEE mul / div / sub: built with -ffp-contract=off and correctly rounded f32 divide
    return (qn > 0.0f && rn > 0.0f) ? 1.0f - s / (qn * rn) : 1.0f;
}

template <int NCH>
__device__ __forceinline__ float query_norm(const float4 (&q)[NCH]) {
    return __builtin_sqrtf(wave_sum(lane_partial<NCH, false>(q, q)));
}

// total order on (distance, order) packed in 64 bits; -0 is canonicalised to +0, NaN sorts last
__device__ __forceinline__ uint64_t make_key(float d, uint32_t ord) {
    d = d + 0.0f;
    uint32_t u = __float_as_uint(d);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return (static_cast<uint64_t>(u) << 32) | ord;
}
__device__ __forceinline__ float key_dist(uint64_t key) {
    uint32_t u = static_cast<uint32_t>(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(u);
}

// Data handed from one workgroup to another INSIDE a launch (the fused tails: "last workgroup of a query finishes the
// job"): the eight XCDs' L2s are not coherent with each other, and a device-scope fence costs a write-back plus an
// invalidate of a whole L2 per workgroup (measured: the fused IVF search ran 3-5x slower with __threadfence() pairs).
// Instead the handful of words exchanged are written and read with agent-scope (sc1) accesses, which go to the point
// of coherence themselves; the writer waits for their acknowledgement (vmcnt) before it bumps the counter.
template <class T>
__device__ __forceinline__ void coherent_store(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ T coherent_load(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wait_stores_acked() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Insert `key` (wave-uniform) into the ascending list of `cnt` (<= k) keys kept in LDS by one wave.
__device__ __forceinline__ void wave_insert(uint64_t *list, int &cnt, int k, uint64_t key, int lane) {
    int pos = 0;
    for (int base = 0; base < cnt; base += kWave) {
        int i = base + lane;
        bool lt = (i < cnt) && (list[i] < key);
        pos += __popcll(__ballot(lt));
    }
    if (pos >= k) return;
    int newcnt = cnt + 1 < k ? cnt + 1 : k;
    for (int top = newcnt - 1; top > pos; top -= kWave) {
        int i = top - lane;
        bool act = i > pos;
        uint64_t v = 0;
        if (act) v = list[i - 1];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (act) list[i] = v;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) list[pos] = key;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    cnt = newcnt;
}

// wave_insert with a 32-bit payload carried beside every key (the merge over shards: key = (distance, global order),
// payload = where the candidate sits in the gathered arrays).
__device__ __forceinline__ void wave_insert_kv(uint64_t *list, uint32_t *val, int &cnt, int k, uint64_t key, uint32_t v,
                                               int lane) {
    int pos = 0;
    for (int base = 0; base < cnt; base += kWave) {
        int i = base + lane;
        bool lt = (i < cnt) && (list[i] < key);
        pos += __popcll(__ballot(lt));
    }
    if (pos >= k) return;
    int newcnt = cnt + 1 < k ? cnt + 1 : k;
    for (int top = newcnt - 1; top > pos; top -= kWave) {
        int i = top - lane;
        bool act = i > pos;
        uint64_t kk = 0;
        uint32_t vv = 0;
        if (act) {
            kk = list[i - 1];
            vv = val[i - 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (act) {
            list[i] = kk;
            val[i] = vv;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        list[pos] = key;
        val[pos] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    cnt = newcnt;
}

// Broadcast lane `src`'s 64-bit value (src wave-uniform): two v_readlane instead of two LDS-crossbar permutes.
__device__ __forceinline__ uint64_t lane_bcast(uint64_t v, int src) {
    const uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(v), src);
    const uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(v >> 32), src);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}
// Lane i takes lane i - 1's value across the whole wave (lane 0 keeps its own): one DPP move (wave_shr:1, a GFX9
// control gfx950 still has) instead of an LDS-crossbar permute.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) {
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x138, 0xF, 0xF, false));
}

// The same list kept in registers when k <= 64: lane i holds the i-th smallest key (lanes >= cnt hold
// ~0).  One ballot + one lane shift per insertion instead of an LDS round trip.
__device__ __forceinline__ void wave_insert_reg(uint64_t &mine, int &cnt, int k, uint64_t key, int lane) {
    int pos = __popcll(__ballot(mine < key));  // ascending: the smaller keys form a prefix of the lanes
    if (pos >= k) return;
    const uint32_t ulo = wave_shr1(static_cast<uint32_t>(mine)), uhi = wave_shr1(static_cast<uint32_t>(mine >> 32));
    uint64_t up = (static_cast<uint64_t>(uhi) << 32) | ulo;
    mine = lane < pos ? mine : (lane == pos ? key : up);
    if (lane >= k) mine = ~0ull;
    cnt = cnt + 1 < k ? cnt + 1 : k;
}
__device__ __forceinline__ uint64_t wave_kth_reg(uint64_t mine, int k) {
    uint32_t lo = __builtin_amdgcn_readlane(static_cast<uint32_t>(mine), k - 1);
    uint32_t hi = __builtin_amdgcn_readlane(static_cast<uint32_t>(mine >> 32), k - 1);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// An exclusive upper bound for the k-th smallest key of a set, from 64 per-lane minima of (parts of) that set
// (k <= 64): the k-th smallest of the lane minima has at least k keys at or below it.  Used to start a top-k
// fold with a tight threshold: with the threshold at ~0 every key of the first block of a stream went through
// the one-at-a-time insertion path (512 cross-lane broadcasts before the first rejection).
__device__ __forceinline__ uint64_t kth_bound(uint64_t lane_min, int k, int lane) {
    const uint32_t hi = static_cast<uint32_t>(lane_min >> 32);  // the distance part decides; ties are let through
    int rank = 0;
#pragma unroll
    for (int j = 0; j < kWave; j++) {
        const uint32_t o = __builtin_amdgcn_readlane(hi, j);
        rank += (o < hi || (o == hi && j < lane)) ? 1 : 0;
    }
    const uint64_t sel = __ballot(rank == k - 1);  // ranks are a permutation of 0..63: exactly one lane
    const uint32_t b = __builtin_amdgcn_readlane(hi, __ffsll(static_cast<unsigned long long>(sel)) - 1);
    return b == 0xffffffffu ? ~0ull : (static_cast<uint64_t>(b) + 1) << 32;
}

// The k smallest (k <= 64) of the S + 1 keys every lane of a wave holds (S x 64 fresh keys + `carry`: the wave's best
// so far in lanes < k, all-ones elsewhere), ascending, into out[0..k) (LDS, all-ones padded) -- without walking keys
// through a list one at a time.  The insertion fold above costs a serial ~0.15 us per accepted key: picking 32 probed
// lists out of 1024 centroid distances took 17 us of a 66-us single-query search.  Here the wave bisects the KEY SPACE for
// the k-th smallest key -- a step is S + 1 compares, ballots and popcounts, and it stops as soon as exactly k keys are at
// or below the pivot (~20 steps for float distances) -- compacts the k keys into LDS and sorts them by rank.  Keys are
// distinct (they carry their position), all-ones = none.  scratch: k keys of LDS of this wave's own.
template <int S>
__device__ __forceinline__ void wave_topk_sorted(const uint64_t (&key)[S], uint64_t carry, int k, uint64_t *scratch, uint64_t *out,
                                                 int lane) {
    // the distance words first (32-bit compares at full rate; a 64-bit compare costs four times as much): the smallest dT
    // with count(distance word <= dT) >= k
    uint32_t hw[S + 1];
#pragma unroll
    for (int s = 0; s < S; s++) hw[s] = static_cast<uint32_t>(key[s] >> 32);
    hw[S] = static_cast<uint32_t>(carry >> 32);
    // Where the bisection starts: no key is below the smallest of the 64 lanes' minima, and the k-th smallest of those minima has
    // k keys at or below it (one per lane at least) -- a range that holds a few dozen keys (~8 steps) instead of [0, 2^32) (~20
    // steps of S + 1 dependent ballots: 3.5 of the 7 us in which a query's 32 nearest of 1024 centroids were picked).  The ranks
    // by v_readlane: 64 independent compare-and-adds.
    uint32_t lo, hi;
    {
        uint32_t mh = hw[0];
#pragma unroll
        for (int s = 1; s <= S; s++) mh = hw[s] < mh ? hw[s] : mh;
        int rank = 0;
#pragma unroll
        for (int j = 0; j < kWave; j++) {
            const uint32_t o = __builtin_amdgcn_readlane(mh, j);
            rank += (o < mh || (o == mh && j < lane)) ? 1 : 0;
        }
        // (ranks are a permutation of 0..63: exactly one lane each; k <= 64)
        lo = __builtin_amdgcn_readlane(mh, __ffsll(static_cast<unsigned long long>(__ballot(rank == 0))) - 1);
        hi = __builtin_amdgcn_readlane(mh, __ffsll(static_cast<unsigned long long>(__ballot(rank == k - 1))) - 1);
    }
    bool exact = false;  // exactly k keys at or below hi: no tie to break
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        int c = 0;
#pragma unroll
        for (int s = 0; s <= S; s++) c += __popcll(__ballot(hw[s] <= mid));
        if (c == k) {
            hi = mid;
            exact = true;
            break;
        }
        if (c > k) hi = mid;
        else lo = mid + 1;
    }
    uint64_t T = (static_cast<uint64_t>(hi) << 32) | 0xffffffffu;
    if (!exact) {
        int c_le = 0, c_lt = 0;
#pragma unroll
        for (int s = 0; s <= S; s++) {
            c_le += __popcll(__ballot(hw[s] <= hi));
            c_lt += __popcll(__ballot(hw[s] < hi));
        }
        if (c_le > k) {  // equal distances across the boundary: the k - c_lt smallest positions among them
            const int r = k - c_lt;
            uint32_t plo = 0, phi = 0xffffffffu;
            while (plo < phi) {
                const uint32_t mid = plo + ((phi - plo) >> 1);
                int c = __popcll(__ballot(hw[S] == hi && static_cast<uint32_t>(carry) <= mid));
#pragma unroll
                for (int s = 0; s < S; s++) c += __popcll(__ballot(hw[s] == hi && static_cast<uint32_t>(key[s]) <= mid));
                if (c >= r) phi = mid;
                else plo = mid + 1;
            }
            T = (static_cast<uint64_t>(hi) << 32) | phi;
        }
    }
    if (T == ~0ull) T = ~0ull - 1;  // (fewer than k keys: all of them, never the all-ones fillers)
    if (lane < k) out[lane] = ~0ull;
    // compact the keys <= T (k of them, fewer if fewer exist) ...
    int base = 0;
    {
        const uint64_t m = __ballot(carry <= T);
        if (carry <= T) scratch[base + __popcll(m & ((1ull << lane) - 1ull))] = carry;
        base += __popcll(m);
    }
#pragma unroll
    for (int s = 0; s < S; s++) {
        const uint64_t m = __ballot(key[s] <= T);
        if (key[s] <= T) scratch[base + __popcll(m & ((1ull << lane) - 1ull))] = key[s];
        base += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ... and place each at its rank
    if (lane < base) {
        const uint64_t v = scratch[lane];
        int rank = 0;
        int j = 0;
        for (; j + 8 <= base; j += 8) {  // uniform addresses: LDS broadcasts, eight in flight (one at a time, the loop is 32 LDS latencies)
            uint64_t o[8];
#pragma unroll
            for (int u = 0; u < 8; u++) o[u] = scratch[j + u];
#pragma unroll
            for (int u = 0; u < 8; u++) rank += o[u] < v ? 1 : 0;
        }
        for (; j < base; j++) rank += scratch[j] < v ? 1 : 0;
        out[rank] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Inclusive prefix sum over the 64 lanes of a wave: four row_shr DPP steps inside the rows of 16, then row_bcast:15 / :31 carry
// the row totals on (the GFX9 controls gfx950 still has) -- six VALU adds, no LDS-crossbar trip.
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

// The k <= 64 smallest of up to 1024 keys (four per thread of a 256-thread workgroup, all-ones = none, distinct), ascending
// into fin[0, k) (all-ones padded) -- by HISTOGRAMS of the distance words instead of a bisection of the key space: a pass
// bins every key of the current range into 256 bins (one LDS atomic each), one wave scans the counts and names the bin
// that holds the k-th smallest, and the next pass looks at that bin alone (8 more bits of the distance word) while it
// still holds more than a few dozen keys; then the keys below the boundary bin and the keys in it are compacted and placed
// by rank.  A pass is ~100 instructions per wave where a bisection step is ~20 and a bisection needs ~20 steps on each of two
// levels (one wave executes ~100 dependent instructions per microsecond: picking 32 of 1024 centroid distances took 7 us, the
// finish kernel's merge another 7).  Returns false -- nothing written -- when the boundary bin cannot be thinned below the
// candidate buffer (hundreds of equal distances): the caller then bisects as before.  All 256 threads call it.
// minimum / maximum over the 64 lanes (the same six DPP steps, identity for the lanes a step does not reach), wave-uniform
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    auto step = [](uint32_t x, uint32_t o) { return o < x ? o : x; };
    const int id = -1;
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x111, 0xF, 0xF, false)));
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x112, 0xF, 0xF, false)));
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x114, 0xF, 0xF, false)));
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x118, 0xF, 0xF, false)));
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x142, 0xA, 0xF, false)));
    v = step(v, static_cast<uint32_t>(__builtin_amdgcn_update_dpp(id, static_cast<int>(v), 0x143, 0xC, 0xF, false)));
    return __builtin_amdgcn_readlane(v, kWave - 1);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }

constexpr int kHistCand = 256;
struct alignas(16) TopkHistLds {
    uint32_t hist[256];
    uint64_t cand[kHistCand];
    uint32_t wmin[kNWave], wmax[kNWave], wcnt[kNWave];
    uint32_t bin, below, inbin, ncand;
};
__device__ __forceinline__ bool topk_hist_wg(const uint64_t (&key)[4], int k, uint64_t *fin, TopkHistLds &L) {
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    uint32_t hw[4];
    bool valid[4];
    uint32_t mn = 0xffffffffu, mx = 0u;
    int nv = 0;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        hw[s] = static_cast<uint32_t>(key[s] >> 32);
        valid[s] = key[s] != ~0ull;
        mn = valid[s] && hw[s] < mn ? hw[s] : mn;
        mx = valid[s] && hw[s] > mx ? hw[s] : mx;
        nv += __popcll(__ballot(valid[s]));
    }
    // the range of the valid distance words and their number, over the workgroup
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    L.hist[tid] = 0;
    if (lane == 0) {
        L.wmin[wave] = mn;
        L.wmax[wave] = mx;
        L.wcnt[wave] = static_cast<uint32_t>(nv);
    }
    if (tid < k) fin[tid] = ~0ull;
    if (tid == 0) L.ncand = 0;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < kNWave; w++) {
        mn = L.wmin[w] < mn ? L.wmin[w] : mn;
        mx = L.wmax[w] > mx ? L.wmax[w] : mx;
        total += L.wcnt[w];
    }
    if (total == 0) return true;  // (fin is all-ones)
    uint32_t K = static_cast<uint32_t>(k) < total ? static_cast<uint32_t>(k) : total;  // the K-th smallest valid key is the last one selected
    uint32_t base = mn;
    int sh = 32 - __builtin_clz((mx - mn) | 1u) - 8;  // (mx - mn) >> sh < 256
    sh = sh < 0 ? 0 : sh;
    bool in_range[4];
#pragma unroll
    for (int s = 0; s < 4; s++) in_range[s] = valid[s];
    for (;;) {
#pragma unroll
        for (int s = 0; s < 4; s++)
            if (in_range[s]) atomicAdd(&L.hist[(hw[s] - base) >> sh], 1u);
        __syncthreads();
        if (wave == 0) {
            const uint4 h = *reinterpret_cast<const uint4 *>(L.hist + 4 * lane);
            *reinterpret_cast<uint4 *>(L.hist + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);  // (for the next pass)
            const int mine = static_cast<int>(h.x + h.y + h.z + h.w);
            const int incl = wave_scan_incl(mine);
            const uint64_t reach = __ballot(incl >= static_cast<int>(K));  // (the last lane's total is the range's population >= K)
            const int Ln = __ffsll(static_cast<unsigned long long>(reach)) - 1;
            if (lane == Ln) {
                uint32_t c = static_cast<uint32_t>(incl - mine);  // keys in the bins of the lanes before
                uint32_t b = 4 * lane, inb = h.x;
                if (c + h.x < K) {
                    c += h.x;
                    b++;
                    inb = h.y;
                    if (c + h.y < K) {
                        c += h.y;
                        b++;
                        inb = h.z;
                        if (c + h.z < K) {
                            c += h.z;
                            b++;
                            inb = h.w;
                        }
                    }
                }
                L.bin = b;
                L.below = c;
                L.inbin = inb;
            }
        }
        __syncthreads();
        const uint32_t b = L.bin, below = L.below, inbin = L.inbin;
        // the next range: the boundary bin alone
#pragma unroll
        for (int s = 0; s < 4; s++) in_range[s] = in_range[s] && ((hw[s] - base) >> sh) == b;
        base += b << sh;
        K -= below;
        if (inbin <= 32u || sh == 0) break;
        sh = sh > 8 ? sh - 8 : 0;  // (L.bin / below / inbin are rewritten behind the next pass's first barrier: every thread has read them)
    }
    // selected: every valid key below the boundary bin (there are k_sel - K of them) and the keys in it (>= K, the K smallest count)
    int c = 0;
    uint64_t m[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const bool sel = valid[s] && (hw[s] < base || in_range[s]);
        m[s] = __ballot(sel);
        c += __popcll(m[s]);
    }
    uint32_t at = 0;
    if (lane == 0 && c > 0) at = atomicAdd(&L.ncand, static_cast<uint32_t>(c));
    at = __builtin_amdgcn_readfirstlane(at);
    const uint64_t below_me = (1ull << lane) - 1ull;
#pragma unroll
    for (int s = 0; s < 4; s++) {
        if ((m[s] >> lane) & 1ull) {
            const uint32_t i = at + static_cast<uint32_t>(__popcll(m[s] & below_me));
            if (i < static_cast<uint32_t>(kHistCand)) L.cand[i] = key[s];
        }
        at += static_cast<uint32_t>(__popcll(m[s]));
    }
    __syncthreads();
    const int nc = static_cast<int>(L.ncand);
    if (nc > kHistCand) return false;  // (uniform: the bisection path re-reads the keys)
    if (tid < nc) {
        const uint64_t v = L.cand[tid];
        int rank = 0, j = 0;
        for (; j + 4 <= nc; j += 4) {  // uniform addresses: LDS broadcasts, two keys per read
            const uint64_t o0 = L.cand[j], o1 = L.cand[j + 1], o2 = L.cand[j + 2], o3 = L.cand[j + 3];
            rank += (o0 < v ? 1 : 0) + (o1 < v ? 1 : 0) + (o2 < v ? 1 : 0) + (o3 < v ? 1 : 0);
        }
        for (; j < nc; j++) rank += L.cand[j] < v ? 1 : 0;
        if (rank < k) fin[rank] = v;
    }
    __syncthreads();
    return true;
}

// The k <= 64 smallest of n keys (key_of(i), i < n; distinct; all-ones = none), ascending, by the four waves of a
// workgroup: each wave picks the k smallest of its interleaved quarter (chunks of 4 x 256 keys, the best so far carried
// along), then all four merge their lists by rank.  Returns fin (k keys, all-ones padded) to wave 0, null to the others
// (behind the last barrier: they touch none of the buffers again).  lists: [kNWave][k], fin: [k], scratch: [kNWave][k] of LDS.
template <class KeyOf>
__device__ __forceinline__ const uint64_t *topk_small_wg(int64_t n, int k, uint64_t *lists, uint64_t *fin, uint64_t *scratch,
                                                         KeyOf key_of, bool hist = true /* (false: tools/micro/topk_select.hip times the bisection) */) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    constexpr int S = 4;
    if (hist && n <= S * kWG) {  // one chunk: by histograms (topk_hist_wg); a boundary of hundreds of equal distances falls through
        __shared__ TopkHistLds hist_lds;
        uint64_t key[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int64_t i = s * kWG + wave * kWave + lane;
            key[s] = i < n ? key_of(i) : ~0ull;
        }
        if (topk_hist_wg(key, k, fin, hist_lds)) return wave == 0 ? fin : nullptr;
    }
    uint64_t *myscr = scratch + wave * k;
    uint64_t *mylist = lists + wave * k;
    uint64_t carry = ~0ull;
    if (n <= 0 && lane < k) mylist[lane] = ~0ull;
    for (int64_t base = 0; base < n; base += S * kWG) {
        uint64_t key[S];
#pragma unroll
        for (int s = 0; s < S; s++) {
            const int64_t i = base + s * kWG + wave * kWave + lane;
            key[s] = i < n ? key_of(i) : ~0ull;
        }
        wave_topk_sorted<S>(key, carry, k, myscr, mylist, lane);
        carry = lane < k ? mylist[lane] : ~0ull;
    }
    // the k smallest of the four ascending lists: a key's rank in the union is its index plus, per other list, the number of
    // keys below it there (keys are distinct) -- three bisections in LDS side by side, every wave ranking its own list,
    // instead of a second bisection of the key space by wave 0 alone
    if (wave == 0 && lane < k) fin[lane] = ~0ull;
    __syncthreads();
    if (lane < k) {
        const uint64_t v = mylist[lane];
        if (v != ~0ull) {
            const uint64_t *o0 = lists + ((wave + 1) & (kNWave - 1)) * k, *o1 = lists + ((wave + 2) & (kNWave - 1)) * k,
                           *o2 = lists + ((wave + 3) & (kNWave - 1)) * k;
            int l0 = 0, h0 = k, l1 = 0, h1 = k, l2 = 0, h2 = k;
            for (int it = 0; (1 << it) <= k; it++) {  // floor(log2 k) + 1 steps settle the k + 1 possible counts
                const int m0 = (l0 + h0) >> 1, m1 = (l1 + h1) >> 1, m2 = (l2 + h2) >> 1;
                const uint64_t a0 = o0[m0 < k ? m0 : k - 1], a1 = o1[m1 < k ? m1 : k - 1], a2 = o2[m2 < k ? m2 : k - 1];
                if (l0 < h0) {
                    if (a0 < v) l0 = m0 + 1;
                    else h0 = m0;
                }
                if (l1 < h1) {
                    if (a1 < v) l1 = m1 + 1;
                    else h1 = m1;
                }
                if (l2 < h2) {
                    if (a2 < v) l2 = m2 + 1;
                    else h2 = m2;
                }
            }
            const int rank = lane + l0 + l1 + l2;
            if (rank < k) fin[rank] = v;
        }
    }
    __syncthreads();
    return wave == 0 ? fin : nullptr;
}

// ------------------------------------------------------------------------------------------------
// Row norms (ivf_flat.clj:171-177): one wave per row.
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(kWG) void row_norms_kernel(const float *rows, int64_t ld, int64_t n, float *out) {
    int lane = threadIdx.x & (kWave - 1);
    int64_t row = static_cast<int64_t>(blockIdx.x) * kNWave + (threadIdx.x >> 6);
    if (row >= n) return;
    float4 r[NCH];
    load_row<NCH>(r, rows + row * ld, static_cast<int>(ld / 4), lane, true);
    float s = wave_sum(lane_partial<NCH, false>(r, r));
    if (lane == 0) out[row] = __builtin_sqrtf(s);
}

// Merge the partial top-k lists of one query (contiguous keys) into its final ascending top-k.
// One wave per query.  out_ord / out_dist: [nq][k], padded with 0xffffffff / +inf.
struct MergeArgs {
    const uint64_t *partial;
    int64_t keys_per_query;
    int32_t nq, k;
    uint32_t *out_ord;
    float *out_dist;
};

// One workgroup of W waves per query: every wave folds a contiguous slice of the query's partial keys into
// its own ascending top-k (registers when k <= 64, LDS otherwise), then wave 0 folds the W lists.
// (A single wave scanning ~45k keys made the merge the longest kernel of a one-query IVF search.)
template <bool COH = false>
__device__ __forceinline__ void fold_keys(const uint64_t *in, int64_t n, int k, bool regk, uint64_t *list, int lane,
                                          uint64_t &mine, int &cnt) {
    uint64_t thr = ~0ull, cap = ~0ull;
    constexpr int U = 8;
    // the next block's loads are issued before this block is folded (one memory round trip per block otherwise)
    uint64_t nxt[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        int64_t i = u * kWave + lane;
        nxt[u] = i < n ? (COH ? coherent_load(in + i) : in[i]) : ~0ull;
    }
    for (int64_t base = 0; base < n; base += U * kWave) {
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; u++) key[u] = nxt[u];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + (U + u) * kWave + lane;
            nxt[u] = i < n ? (COH ? coherent_load(in + i) : in[i]) : ~0ull;
        }
        if (regk && base == 0) {  // start from a bound on the k-th smallest key instead of ~0 (kth_bound)
            uint64_t m = key[0];
#pragma unroll
            for (int u = 1; u < U; u++) m = key[u] < m ? key[u] : m;
            cap = kth_bound(m, k, lane);
            thr = cap;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t mask = __ballot(key[u] < thr);
            while (mask) {
                int b = __ffsll(static_cast<unsigned long long>(mask)) - 1;
                mask &= mask - 1;
                uint64_t kb = lane_bcast(key[u], b);
                if (kb < thr) {
                    if (regk) {
                        wave_insert_reg(mine, cnt, k, kb, lane);
                        const uint64_t kth = wave_kth_reg(mine, k);
                        thr = kth < cap ? kth : cap;
                    } else {
                        wave_insert(list, cnt, k, kb, lane);
                        thr = cnt == k ? list[k - 1] : ~0ull;
                    }
                }
            }
        }
    }
}


// Merge the partial top-k lists of query q (a.keys_per_query contiguous keys) with the W waves of this workgroup: every
// wave folds a slice into its own ascending top-k, wave 0 folds the W lists.  out_ord / out_dist: query q's k slots
// (0xffffffff / +inf padded).  The body of merge_topk_kernel, and the tail of a fused list scan (scan_kernel).
template <bool COH = false>
__device__ __forceinline__ void merge_topk_wg(const MergeArgs &a, int q, int W, unsigned char *smem, uint32_t *out_ord,
                                              float *out_dist) {
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);  // [W][k] (+ [k] for the final list when W > 1)
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const uint64_t *in = a.partial + static_cast<int64_t>(q) * a.keys_per_query;
    const bool regk = a.k <= kWave;
    const int64_t slice = (a.keys_per_query + W - 1) / W;
    const int64_t s0 = wave * slice;
    const int64_t sn = s0 + slice < a.keys_per_query ? slice : (a.keys_per_query > s0 ? a.keys_per_query - s0 : 0);
    uint64_t *list = lists + static_cast<size_t>(wave) * a.k;
    uint64_t mine = ~0ull;
    int cnt = 0;
    if (wave < W) {
        fold_keys<COH>(in + s0, sn, a.k, regk, list, lane, mine, cnt);
        if (regk) {
            if (lane < a.k) list[lane] = mine;
        } else {
            for (int i = cnt + lane; i < a.k; i += kWave) list[i] = ~0ull;
        }
    }
    __syncthreads();
    if (wave != 0) return;
    if (W > 1) {  // second level: W * k keys (sentinels included) -> final list behind the W slots
        uint64_t *fin = lists + static_cast<size_t>(W) * a.k;
        const int tot = W * a.k;
        uint64_t thr = ~0ull;
        mine = ~0ull;
        cnt = 0;
        for (int base = 0; base < tot; base += kWave) {
            const int i = base + lane;
            const uint64_t key = i < tot ? lists[i] : ~0ull;
            uint64_t mask = __ballot(key < thr);
            while (mask) {
                int b = __ffsll(static_cast<unsigned long long>(mask)) - 1;
                mask &= mask - 1;
                uint64_t kb = lane_bcast(key, b);
                if (kb < thr) {
                    if (regk) {
                        wave_insert_reg(mine, cnt, a.k, kb, lane);
                        thr = wave_kth_reg(mine, a.k);
                    } else {
                        wave_insert(fin, cnt, a.k, kb, lane);
                        thr = cnt == a.k ? fin[a.k - 1] : ~0ull;
                    }
                }
            }
        }
        if (regk) {
            if (lane < a.k) fin[lane] = mine;
        } else {
            for (int i = cnt + lane; i < a.k; i += kWave) fin[i] = ~0ull;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        list = fin;
    } else {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < a.k; i += kWave) {
        uint64_t key = list[i];
        bool ok = key != ~0ull;
        out_ord[i] = ok ? static_cast<uint32_t>(key) : 0xffffffffu;
        out_dist[i] = ok ? key_dist(key) : __uint_as_float(0x7f800000u);
    }
}

// ------------------------------------------------------------------------------------------------
// Scan: for each (query, contiguous row range) pair, stream the rows and either keep the per-wave
// top-k (MODE_TOPK), store the distances (MODE_STORE) or fold them into a running minimum
// (MODE_MINUPD).  IVF list scan (ivf_flat.clj:217-234), centroid routing (:261-269), k-means
// assignment (:79-90), k-means++ seeding (:43-49) and exact kNN (bench.clj:72-84) are all this
// kernel with different pair tables.
// ------------------------------------------------------------------------------------------------
struct Pair {
    int64_t row_begin, row_end;
    int32_t q;
    uint32_t ord_base;   // offset of this list's rows in the query's candidate stream over THIS handle's rows
    // the same offset in the candidate stream of the WHOLE index when this handle holds only some of the inverted
    // lists (hnswgpu_set_ivf_shard): shards order their results by it, so a merge over shards breaks ties exactly as
    // one unsharded search would (ivf_flat.clj:281-294 concatenates the probed partitions in probe order)
    uint32_t gord_base;
    uint32_t pad;
};

struct ScanArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld;
    int64_t nrows_all;  // implicit pairs: every query scans rows [0, nrows_all)
    const float *Q;
    int64_t qld;
    const float *q_norms;  // optional precomputed query norms (queries that are base rows)
    int32_t dim;
    int32_t metric;
    int32_t mode;
    const Pair *pairs;  // nullptr = implicit pairs
    // optional execution order of the pairs (a permutation that puts the pairs probing the same list next to each
    // other): with it, workgroups are dealt to the XCDs in runs of `run` neighbouring work items, so pairs that
    // stream the same rows run side by side on ONE L2 and the second reader finds the first one's lines there
    const int32_t *order;
    int32_t run;  // work items per run (a power of two, >= the mean number of pairs per list; set by the launcher)
    int32_t npairs;
    int32_t chunk_rows;
    int32_t nchunks;
    int32_t k;
    int32_t role;       // ROLE_* (profiling label only)
    uint64_t *partial;  // TOPK: [pair][chunk][wave][k] ([pair][chunk][k] with wg_merge)
    float *out;         // STORE: out[pair * out_stride + (row - row_begin)]; MINUPD: out[row]
    int64_t out_stride;
    // TOPK, k <= 64: the four per-wave lists of a workgroup are merged in LDS and ONE list is written -- a quarter of the
    // keys for whoever merges them
    int32_t wg_merge;
    // Fused tail of the IVF list scan (small batches: the launches around the scan cost as much as the scan): every
    // workgroup counts itself in done[q] after writing its partial list; the last one of a query merges that query's
    // partial lists, maps the winners to row ids (ivf_flat.clj:291-294) and writes the final results -- no merge,
    // decode or copy launch.  done[] is zero between calls (the last workgroup resets its counter).
    uint32_t *done;            // nullptr = no fused tail
    int32_t pairs_per_query;   // = nprobe; a query's workgroups: pairs_per_query * nchunks
    const int32_t *listids;
    int32_t *out_ids;          // [nq][k]
    float *out_dist;           // [nq][k]
    uint32_t *out_gord;        // optional [nq][k]: position in the candidate stream of the whole (sharded) index
};

// ROLE only names the caller in profiler output (rocprofv3 groups dispatches by kernel name); the
// code is identical for every role.
constexpr int ROLE_LIST_SCAN = 0, ROLE_ROUTE = 1, ROLE_ASSIGN = 2, ROLE_SEED = 3, ROLE_EXACT = 4;

template <int NCH, int RB, bool L2, int ROLE>
__global__ __launch_bounds__(kWG) void scan_kernel(ScanArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);  // [kNWave][k]
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int64_t bid = blockIdx.x;
    // chunk-major: every pair's chunk 0 first, then chunk 1, ...  Chunks past a short list's end are empty
    // workgroups; this order keeps them at the END of the dispatch instead of interleaved with working ones
    // (interleaved idle workgroups cost the tile kernel half its CU occupancy on k-means lists).
    int32_t pair, chunk;
    if (a.order) {
        // workgroup b runs on XCD b % 8: XCD x takes the runs x, x + 8, x + 16, ... of a.run work items each
        // (work items stay chunk-major: [chunk][pair in list order])
        const int64_t j = bid >> 3;
        const int64_t item = ((j / a.run) * 8 + (bid & 7)) * a.run + j % a.run;
        if (item >= static_cast<int64_t>(a.npairs) * a.nchunks) return;
        pair = a.order[item % a.npairs];
        chunk = static_cast<int32_t>(item / a.npairs);
    } else {
        pair = static_cast<int32_t>(bid % a.npairs);
        chunk = static_cast<int32_t>(bid / a.npairs);
    }
    int64_t rb0, rb1;
    int32_t qi;
    uint32_t ord_base;
    if (a.pairs) {
        Pair p = a.pairs[pair];
        rb0 = p.row_begin;
        rb1 = p.row_end;
        qi = p.q;
        ord_base = p.ord_base;
    } else {
        rb0 = 0;
        rb1 = a.nrows_all;
        qi = pair;
        ord_base = 0;
    }
    int64_t r0 = rb0 + static_cast<int64_t>(chunk) * a.chunk_rows;
    int64_t r1 = r0 + a.chunk_rows < rb1 ? r0 + a.chunk_rows : rb1;
    uint64_t *mylist = lists + wave * a.k;
    int cnt = 0;
    uint64_t thr = ~0ull;
    const bool regk = a.k <= kWave;  // top-k list in registers (one key per lane) instead of LDS
    uint64_t mine = ~0ull;
    if (r0 < r1) {
        float4 q[NCH];
        load_query<NCH>(q, a.Q + qi * a.qld, a.dim, lane);
        float qn = 0.0f;
        if (a.metric == METRIC_COS) qn = a.q_norms ? a.q_norms[qi] : query_norm<NCH>(q);
        const int nvec = static_cast<int>(a.ld / 4);
        for (int64_t base = r0 + wave * RB; base < r1; base += kNWave * RB) {
            float4 r[RB][NCH];
            // lane b fetches row b's precomputed norm with the rows, not after the reduction
            const float myrn = (a.metric == METRIC_COS && lane < RB && base + lane < r1) ? a.row_norms[base + lane] : 0.0f;
#pragma unroll
            for (int b = 0; b < RB; b++) load_row<NCH>(r[b], a.rows + (base + b) * a.ld, nvec, lane, base + b < r1);
            float s[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) s[b] = lane_partial<NCH, L2>(q, r[b]);
            const float tot = rows_sum_to_lane<RB>(s, lane);  // lane b: row b's sum (eight rows: one halving exchange)
#pragma unroll
            for (int b = 0; b < RB; b++) {
                int64_t row = base + b;
                if (row < r1) {  // wave-uniform
                    float rn = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), b));
                    float d = finish_dist(a.metric, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tot), b)), qn, rn);
                    if (a.mode == MODE_TOPK) {
                        uint64_t key = make_key(d, ord_base + static_cast<uint32_t>(row - rb0));
                        if (key < thr) {
                            if (regk) {
                                wave_insert_reg(mine, cnt, a.k, key, lane);
                                thr = wave_kth_reg(mine, a.k);  // ~0 until the list is full
                            } else {
                                wave_insert(mylist, cnt, a.k, key, lane);
                                thr = cnt == a.k ? mylist[a.k - 1] : ~0ull;
                            }
                        }
                    } else if (a.mode == MODE_STORE) {
                        if (lane == 0) a.out[pair * a.out_stride + (row - rb0)] = d;
                    } else {
                        if (lane == 0) {
                            float o = a.out[row];
                            if (d < o) a.out[row] = d;
                        }
                    }
                }
            }
        }
    }
    if (a.mode == MODE_TOPK) {
        if (a.wg_merge && regk) {
            // the workgroup's four lists -> one: wave 0 inserts the other waves' keys (ascending: stop at the first
            // key that no longer fits) into its own register list
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
                uint64_t *dst = a.partial + (static_cast<int64_t>(pair) * a.nchunks + chunk) * a.k;
                if (lane < a.k) {
                    if (a.done) coherent_store(dst + lane, mine);  // read by another workgroup of this launch
                    else dst[lane] = mine;
                }
            }
        } else {
            uint64_t *dst = a.partial + ((static_cast<int64_t>(pair) * a.nchunks + chunk) * kNWave + wave) * a.k;
            if (regk) {
                if (lane < a.k) {
                    if (a.done) coherent_store(dst + lane, mine);
                    else dst[lane] = mine;
                }
            } else {
                for (int i = lane; i < a.k; i += kWave) {
                    const uint64_t v = i < cnt ? mylist[i] : ~0ull;
                    if (a.done) coherent_store(dst + i, v);
                    else dst[i] = v;
                }
            }
        }
        if (a.done) {
            // ---- fused tail: am I the last workgroup of query qi?
            __shared__ int tail_last;
            wait_stores_acked();  // this workgroup's partial list has reached the point of coherence ...
            __syncthreads();
            if (threadIdx.x == 0) {
                const uint32_t total = static_cast<uint32_t>(a.pairs_per_query) * static_cast<uint32_t>(a.nchunks);
                // ... before it counts itself; the last one reads every list with coherent loads (kernels.hpp, top)
                const uint32_t prev = __hip_atomic_fetch_add(a.done + qi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tail_last = prev == total - 1 ? 1 : 0;
                if (tail_last) __hip_atomic_store(a.done + qi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
            if (!tail_last) return;
            MergeArgs m;
            m.partial = a.partial;
            m.keys_per_query = static_cast<int64_t>(a.pairs_per_query) * a.nchunks * ((a.wg_merge && regk) ? 1 : kNWave) * a.k;
            m.nq = 0;
            m.k = a.k;
            m.out_ord = nullptr;
            m.out_dist = nullptr;
            uint32_t *ord_s = reinterpret_cast<uint32_t *>(smem + sizeof(uint64_t) * (kNWave + 1) * a.k);  // [k]
            float *dist_s = reinterpret_cast<float *>(ord_s + a.k);                                         // [k]
            merge_topk_wg<true>(m, qi, kNWave, smem, ord_s, dist_s);  // waves 1..3 return from it after its barrier
            if (wave != 0) return;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const Pair *pp = a.pairs + static_cast<int64_t>(qi) * a.pairs_per_query;
            for (int i = lane; i < a.k; i += kWave) {
                const uint32_t o = ord_s[i];
                uint32_t go = 0xffffffffu;
                int32_t id = -1;
                if (o != 0xffffffffu) {
                    int p = 0;
                    while (p + 1 < a.pairs_per_query && pp[p + 1].ord_base <= o) p++;  // as ivf_decode_kernel
                    id = a.listids[pp[p].row_begin + (o - pp[p].ord_base)];
                    go = pp[p].gord_base + (o - pp[p].ord_base);
                }
                a.out_ids[static_cast<int64_t>(qi) * a.k + i] = id;
                a.out_dist[static_cast<int64_t>(qi) * a.k + i] = dist_s[i];
                if (a.out_gord) a.out_gord[static_cast<int64_t>(qi) * a.k + i] = go;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Gather-dot: distances from one query to base rows ids[0..m) -- the per-hop neighbour expansion
// of search-layer-ultra (ultra_fast.clj:185-204) as a stand-alone seam
// (simd-optimized/batch-cosine-distances, simd_optimized.clj:176-184).
// ------------------------------------------------------------------------------------------------
struct GatherArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld;
    int64_t n;
    const float *q;      // query y at q + y * qld
    int64_t qld;
    int32_t dim;
    int32_t metric;
    const int32_t *ids;  // nullptr = 0..m-1; otherwise query y's candidates are ids[y * m .. y * m + m)
    int32_t m;
    int32_t blocks_per_query;  // grid = nq * blocks_per_query
    float *out;          // optional: out[y * m + j]
    uint64_t *out_keys;  // optional: make_key(dist, j) (all-ones for an id outside [0, n)) -- the re-rank path
};

template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void gather_dist_kernel(GatherArgs a) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int64_t y = blockIdx.x / a.blocks_per_query;
    const int bx = static_cast<int>(blockIdx.x % a.blocks_per_query);
    int j0 = (bx * kNWave + wave) * RB;
    if (j0 >= a.m) return;
    float4 q[NCH];
    load_query<NCH>(q, a.q + y * a.qld, a.dim, lane);
    float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
    const int nvec = static_cast<int>(a.ld / 4);
    const int32_t *ids = a.ids ? a.ids + y * a.m : nullptr;
    float4 r[RB][NCH];
    int64_t rid[RB];
#pragma unroll
    for (int b = 0; b < RB; b++) {
        int j = j0 + b;
        int64_t id = -1;
        if (j < a.m) id = ids ? ids[j] : j;
        bool ok = id >= 0 && id < a.n;
        rid[b] = ok ? id : -1;
        load_row<NCH>(r[b], a.rows + (ok ? id : 0) * a.ld, nvec, lane, ok);
    }
#pragma unroll
    for (int b = 0; b < RB; b++) {
        float s = wave_sum(lane_partial<NCH, L2>(q, r[b]));
        if (j0 + b < a.m && lane == 0) {
            float d = __uint_as_float(0x7fc00000u);  // NaN for an out-of-range id
            if (rid[b] >= 0) d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[rid[b]] : 0.0f);
            if (a.out) a.out[y * a.m + j0 + b] = d;
            if (a.out_keys)
                a.out_keys[y * a.m + j0 + b] = rid[b] >= 0 ? make_key(d, static_cast<uint32_t>(j0 + b)) : ~0ull;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Compressed rows for the traversal's rejection test.
//
// search-layer-ultra admits a neighbour only if the result list is not full or its distance is BELOW the list's worst
// (ultra_fast.clj:195-198, strict <); with ef = 100 on the bench's 31k x 768 index that is 15.5 % of the 2,400
// distances a query evaluates (tools/prefilter_study.py) -- the other 84.5 % are computed, compared and dropped.  A
// rejected neighbour's distance is never stored or looked at again, so a LOWER BOUND that is already >= the worst
// decides it just as well, and a lower bound needs a quarter of the bytes and an eighth of the arithmetic: every row
// also exists as int8 codes c = round(v / s_v), s_v = max|v| / 127 (v' = s_v c) with its exact residual r_v = |v - v'|,
// the query is coded the same way once per search (q' = s_q a, r_q = |q - q'|), and a . c is an exact integer from
// v_dot4c_i32_i8.  Since |q . v - q' . v'| <= |q| r_v + r_q |v'| and | |q - v| - |q' - v'| | <= r_q + r_v:
//     cosine:  d(q, v) >= 1 - q'.v' / (|q||v|) - r_v / |v| - (r_q / |q|) (1 + r_v / |v|)
//     dot:     d(q, v) >= -q'.v' - |q| r_v - r_q (|v| + r_v)
//     L2:      d(q, v) >= |q' - v'| - r_q - r_v,      |q' - v'|^2 = s_q^2 a.a - 2 s_q s_v a.c + s_v^2 c.c
// Every residual is inflated by 1 % and an allowance two orders of magnitude above the f32 rounding of either side (the
// exact path's chain of <= 18 roundings per lane sum is < 1.2e-6 relative to |q||v|) is added, so "bound >= worst"
// implies "the f32 distance the exact path would compute is >= worst": the traversal, its results and its counters are
// unchanged bit for bit -- only rows that may be admitted are fetched in f32.  NaN / infinity anywhere makes the
// comparison false and the row takes the exact path.
//
// Layout: lane l of a wave owns the same elements of a row as in the f32 kernels (float4 number c * 64 + l, c < NCH);
// its NCH code words are stored side by side, 64 * NCH dwords per row, so a row is ONE coalesced load of NCH dwords
// per lane (768 B at dim 768 against 3 KB).  qmeta[row] = (s_v, E, Z, 1 / |v|): E = the row's share of the bound, Z = the
// metric's second per-row term (dot: 1.01 (|v| + r_v); L2: c.c; cosine: unused).
// ------------------------------------------------------------------------------------------------
// codes of this lane's elements of a row or query held as float4 r[NCH]: returns max|.| over the wave first
template <int NCH>
__device__ __forceinline__ float wave_absmax(const float4 (&r)[NCH], bool &bad) {
    float mx = 0.0f;
    bool b = false;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const float e[4] = {r[c].x, r[c].y, r[c].z, r[c].w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float m = __builtin_fabsf(e[j]);
            b = b || !(m <= 3.0e38f);  // NaN or infinity
            mx = m > mx ? m : mx;
        }
    }
    for (int off = 1; off < kWave; off <<= 1) {
        const float o = __shfl_xor(mx, off, kWave);
        mx = o > mx ? o : mx;
    }
    bad = __ballot(b) != 0;
    return mx;
}

// w = the int8 codes (4 per dword), res = this lane's share of |x - s code|^2, c2 = of code . code
template <int NCH>
__device__ __forceinline__ void encode_lane(const float4 (&r)[NCH], float mx, bool bad, uint32_t (&w)[NCH], float &res, int &c2) {
    const float sc = mx / 127.0f, inv = mx > 0.0f ? 127.0f / mx : 0.0f;
    res = 0.0f;
    c2 = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) {
        const float e[4] = {r[c].x, r[c].y, r[c].z, r[c].w};
        w[c] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float t = __builtin_rintf(e[j] * inv);
            t = t > 127.0f ? 127.0f : (t < -127.0f ? -127.0f : t);
            const int code = bad ? 0 : static_cast<int>(t);
            const float d = e[j] - sc * static_cast<float>(code);  // against the value the test reconstructs
            res = __builtin_fmaf(d, d, res);
            c2 += code * code;
            w[c] |= (static_cast<uint32_t>(code) & 0xffu) << (8 * j);
        }
    }
}

__device__ __forceinline__ int wave_sum_int(int x) {
    for (int off = 1; off < kWave; off <<= 1) x += __shfl_xor(x, off, kWave);
    return x;
}

// per-row bound terms (s_v, E, Z, 1 / |v|) from the row's residual, code norm and norm (both code layouts use it);
// .w = 1 / |v|: no second random read for the norm, no division in the bound
__device__ __forceinline__ float4 code_row_meta(int metric, float mx, float res, int c2, float nv, bool bad) {
    float E, Z = 0.0f;
    if (metric == METRIC_COS) {
        E = 1.01f * res / nv + 1.0e-4f;  // nv == 0: NaN or infinity -> exact path
    } else if (metric == METRIC_DOT) {
        E = 1.01f * res + 2.0e-5f * nv;  // times |q| at the point of use
        Z = 1.01f * (nv + res);          // times r_q
    } else {
        E = 1.01f * res + 4.0e-6f * nv;
        Z = static_cast<float>(c2);
    }
    if (bad || !(E >= 0.0f)) E = __uint_as_float(0x7fc00000u);  // NaN: the comparison fails, exact path
    return make_float4(mx / 127.0f, E, Z, 1.0f / nv);
}

template <int NCH>
__global__ __launch_bounds__(kWG) void quantize_rows_kernel(const float *rows, int64_t ld, int64_t n, int metric,
                                                            uint32_t *qrows, float4 *qmeta) {
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
    res = __builtin_sqrtf(wave_sum(res));
    c2 = wave_sum_int(c2);
    const float nv = __builtin_sqrtf(wave_sum(lane_partial<NCH, false>(r, r)));
    uint32_t *dst = qrows + (row * kWave + lane) * NCH;
#pragma unroll
    for (int c = 0; c < NCH; c++) dst[c] = w[c];
    if (lane == 0) qmeta[row] = code_row_meta(metric, mx, res, c2, nv, bad);
}

// The query's side of the test, once per search and wave: codes in the row layout + the scalars of the bounds.
struct QueryScal {
    float s;    // scale s_q
    float qn;   // |q|
    float rq;   // 1.01 r_q (NaN for a query with NaN / infinity: every neighbour takes the exact path)
    float eq;   // 1.01 r_q / |q|   (cosine)
    float a2;   // s_q^2 a.a = |q'|^2  (L2)
    float iqn;  // 1 / |q|   (cosine: the bound multiplies, the reciprocals' roundings are far inside its allowance)
    float pad[2];
};
template <int NCH>
struct QueryCode {
    uint32_t a[NCH];
    QueryScal sc;
};

template <int NCH>
__device__ __forceinline__ void encode_query(const float4 (&q)[NCH], QueryCode<NCH> &qc) {
    bool bad;
    const float mx = wave_absmax<NCH>(q, bad);
    float res;
    int c2;
    encode_lane<NCH>(q, mx, bad, qc.a, res, c2);
    qc.sc.s = mx / 127.0f;
    qc.sc.qn = __builtin_sqrtf(wave_sum(lane_partial<NCH, false>(q, q)));
    qc.sc.rq = bad ? __uint_as_float(0x7fc00000u) : 1.01f * __builtin_sqrtf(wave_sum(res));
    qc.sc.eq = qc.sc.rq / qc.sc.qn;
    qc.sc.a2 = qc.sc.s * qc.sc.s * static_cast<float>(wave_sum_int(c2));
    qc.sc.iqn = 1.0f / qc.sc.qn;
    qc.sc.pad[0] = qc.sc.pad[1] = 0.0f;
}

// a . c over this lane's elements (exact)
template <int NCH>
__device__ __forceinline__ int code_dot(const uint32_t (&a)[NCH], const uint32_t (&w)[NCH]) {
    int acc = 0;
#pragma unroll
    for (int c = 0; c < NCH; c++) acc = __builtin_amdgcn_sdot4(static_cast<int>(a[c]), static_cast<int>(w[c]), acc, false);
    return acc;
}

// Eight per-lane integers -> their eight wave totals, one per group of eight lanes: lane l ends up with the total of
// x[(l >> 3) & 7].  Halving exchange: v_permlane32_swap / v_permlane16_swap trade the half a lane does not keep for
// the half it does (4 + 2 swaps), one row_ror:8 DPP add, then three DPP adds inside the group of eight -- 17
// instructions for eight rows instead of eight six-step butterflies.  Integer sums: the order is irrelevant.
__device__ __forceinline__ int wave_sum8_int(const int (&x)[8], int lane) {
    unsigned y[4], z[2];
#pragma unroll
    for (int i = 0; i < 4; i++) {  // lanes 0-31 keep rows i, lanes 32-63 rows 4 + i
        auto p = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(x[i]), static_cast<unsigned>(x[4 + i]), false, false);
        y[i] = p[0] + p[1];
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {  // 16-lane rows 0 / 1 / 2 / 3 keep rows i, 2 + i, 4 + i, 6 + i
        auto p = __builtin_amdgcn_permlane16_swap(y[i], y[2 + i], false, false);
        z[i] = p[0] + p[1];
    }
    // 16-lane row R: z[0] = row 2R, z[1] = row 2R + 1
    const bool h3 = (lane & 8) != 0;
    const int keep = static_cast<int>(h3 ? z[1] : z[0]), send = static_cast<int>(h3 ? z[0] : z[1]);
    int v = keep + __builtin_amdgcn_update_dpp(0, send, 0x128, 0xF, 0xF, true);  // row_ror:8 = lane ^ 8
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);  // row_half_mirror
    return v;
}
// the row whose total wave_sum8_int leaves in lane l
__device__ __forceinline__ int wave_sum8_row(int lane) { return (lane >> 3) & 7; }

// the bounds of d(q, v) from the exact code dot product (see above): lb <= the f32 distance of the exact path <= ub;
// NaN when nothing can be said
__device__ __forceinline__ void code_bounds(int metric, int dot, const QueryScal &qc, float4 meta, float irn, float &lb, float &ub) {
    const float dh = static_cast<float>(dot) * (qc.s * meta.x);  // q' . v'
    if (metric == METRIC_L2) {
        const float v2 = meta.x * meta.x * meta.z;  // |v'|^2
        const float d2 = qc.a2 - 2.0f * dh + v2, dl = 2.0e-6f * (qc.a2 + v2), lo = d2 - dl;
        const float W = (qc.rq + 4.0e-6f * qc.qn) + meta.y;
        lb = __builtin_sqrtf(lo > 0.0f ? lo : 0.0f) - W;
        ub = __builtin_sqrtf(d2 + dl) * (1.0f + 1.0e-6f) + W;
        return;
    }
    if (metric == METRIC_DOT) {
        const float W = qc.qn * meta.y + qc.rq * meta.z;
        lb = -dh - W;
        ub = -dh + W;
        return;
    }
    const float c = 1.0f - dh * (qc.iqn * irn), W = meta.y + qc.eq * (1.0f + meta.y);  // |q| or |v| zero: NaN / infinity
    lb = c - W;
    ub = c + W;
}
__device__ __forceinline__ float code_lower_bound(int metric, int dot, const QueryScal &qc, float4 meta, float irn) {
    float lb, ub;
    code_bounds(metric, dot, qc, meta, irn, lb, ub);
    return lb;
}

// Diagnostic / test entry (hnswgpu_rejection_bounds): the bound of every listed row against one query, by the very
// functions the traversal uses.  One wave per eight rows.
template <int NCH>
__global__ __launch_bounds__(kWave) void code_bound_kernel(const float *Q, int dim, int metric, const uint32_t *qrows,
                                                           const float4 *qmeta, const int32_t *ids, int m, float *out,
                                                           float *out_ub) {
    const int lane = threadIdx.x;
    float4 q[NCH];
    load_query<NCH>(q, Q, dim, lane);
    QueryCode<NCH> qc;
    encode_query<NCH>(q, qc);
    const int j0 = blockIdx.x * 8;
    int acc[8];
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const int32_t rid = ids[j0 + b < m ? j0 + b : m - 1];
        uint32_t w[NCH];
        const uint32_t *rp = qrows + (static_cast<int64_t>(rid) * kWave + lane) * NCH;
#pragma unroll
        for (int c = 0; c < NCH; c++) w[c] = rp[c];
        acc[b] = code_dot<NCH>(qc.a, w);
    }
    const int tot = wave_sum8_int(acc, lane);
    const int j = j0 + wave_sum8_row(lane);
    if ((lane & 7) == 0 && j < m) {
        const int32_t rid = ids[j];
        const float4 mt = qmeta[rid];
        float lb, ub;
        code_bounds(metric, tot, qc.sc, mt, mt.w, lb, ub);
        out[j] = lb;
        if (out_ub) out_ub[j] = ub;  // (the IVF bounds pass derives its thresholds from upper bounds)
    }
}

// ------------------------------------------------------------------------------------------------
// HNSW traversal: one 256-thread workgroup per query, the whole layered search GPU-resident.
//
// Restates search-layer-ultra / search-knn (ultra_fast.clj:151-212, 346-374) with one sorted list:
//   * `nearest` (max-PQ) and `candidates` (min-PQ) become ONE ascending list in LDS; an entry
//     carries an "expanded" flag (bit 31 of the id word).  The next candidate is the first
//     unexpanded entry.  A candidate the reference would still hold after its eviction from
//     `nearest` can only be expanded if its distance EQUALS the current worst (:175-178 uses <=),
//     so evicted entries tied with the worst are kept as "ghosts" behind position ef.
//   * per expansion: <= M0 neighbour ids are read, visited-filtered against an LDS bitset
//     (HashSet visited, :156) with ballot/popcount compaction, their rows gathered (3 x 1 KiB wave
//     loads per 768-d row, RB rows in flight per wave), distances reduced with wave shuffles, then
//     all admitted neighbours are merged into the list in one parallel rank-and-scatter step that
//     is equivalent to the reference's sequential admit/evict loop (:195-204): an incoming
//     neighbour is admitted iff fewer than ef of {list, earlier neighbours} are <= it (strict <
//     against the worst), ties keep admission order.
//   * levels > 0 use ef = 1 (:373-374); the single survivor seeds the next level.
// In build mode (q_rows != nullptr) the query is a base row, and the best m entries of every level
// <= the node's level are emitted for the host-side linker.
// ------------------------------------------------------------------------------------------------
struct HnswArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld;
    int64_t n;
    int32_t dim;
    int32_t metric;
    const float *Q;
    int64_t qld;
    const int32_t *q_rows;    // build mode: query q is base row q_rows[q]
    const int32_t *q_levels;  // build mode: level of the node being inserted
    int32_t ref_start;        // build mode: the walk starts at min(level, entry's level) with the entry point itself
                              // (insert-single, ultra_fast.clj:247-248) instead of descending from the top layer
    int32_t nq;
    const int32_t *l0_adj;
    int32_t M0;
    const int64_t *up_off;
    const int32_t *up_adj;
    int32_t M;
    int32_t entry;
    int32_t max_level;
    int32_t ef;   // layer-0 breadth (ef-construction in build mode)
    int32_t k;    // results per query (M0 in build mode)
    int32_t cap;  // list capacity = ef + kGhost
    int32_t nwords;
    int32_t *out_ids;
    float *out_dist;
    int64_t *stats;
    int32_t *up_out_ids;  // build mode: [nq][up_stride] nearest node of each upper level <= q level
    float *up_out_dist;
    int32_t up_stride;
    // visited set in HBM (VG kernels; indexes too large for an LDS bitset): one slab of n generation
    // stamps per resident workgroup.  A node is visited in the current (query, layer) iff its stamp
    // equals that pair's generation number; nothing is ever cleared.
    uint32_t *vis;
    int64_t vis_stride;
    uint32_t gen_base;
    // Ghost overflow (more than cap - ef unexpanded candidates tied with the ef-th distance at once: the surplus ties
    // are not expanded, the reference would expand them).  First pass: such a query appends its index to again[]
    // (again_cnt = their number).  Second pass, same kernel with the largest list the LDS holds: q_index = again,
    // nq_dev = again_cnt -- work item i is query q_index[i], and only *nq_dev items exist.
    int32_t *again;
    int32_t *again_cnt;
    const int32_t *q_index;
    const int32_t *nq_dev;
    // L2 prefetch helpers (PF kernels: a handful of queries).  One CU gathers only ~21 GB/s from beyond its XCD's L2 but
    // 40-50 GB/s out of it, and while one query runs the chip is idle: pf_groups helper workgroups per query, placed on
    // the query's XCD (blockIdx % 8), read the neighbour rows of the best unexpanded candidates into that L2 ahead of
    // the traversal.  Pure hints through a mailbox per query (pf_mail: word [2] = done, 64-bit words [8..71] = a ring of
    // data-tagged entries (launch number << 8 | lap) << 32 | node): the traversal never waits for a helper -- not even
    // for its own stores --, so results and counters cannot depend on them.
    uint32_t *pf_mail;
    // ... and the helpers do more than warm the L2: they EVALUATE.  A distance is a property of (query, row), not of the
    // traversal's state, so one computed ahead of time by another workgroup -- with the same lane_partial + butterfly +
    // finish_dist, hence the same bits -- is as good as one computed on the spot.  For every hinted node a helper
    // publishes the distances of its slice of the node's neighbours: pf_res[query][ring slot][neighbour slot] =
    // (tag << 32) | distance bits, tag = the hint's own (launch number << 8 | lap).  When the traversal expands a node
    // whose words have arrived it reads them beside the adjacency row and gathers nothing for those neighbours.
    unsigned long long *pf_res;
    int32_t pf_groups;
    int32_t pf_hints;  // unexpanded entries looked at per expansion (the best pf_hints of the list)
    uint32_t pf_seq;
    // solo mode (solo_kernels.hpp: one query over several CUs): the node-keyed tables of what the helpers publish --
    // solo_rec[query][slot of a node][neighbour slot] = (tag << 32) | distance bits, solo_claim[query][slot] = tag of the node
    // whose evaluation somebody has asked for, tag = (launch number, node); 2^solo_log2s slots per query
    unsigned long long *solo_rec;
    uint32_t *solo_claim;
    int32_t solo_log2s;
    int32_t solo_chase;  // the helpers append the neighbours they find closer than the owner's window themselves
    // Host-polled completion (small synchronous calls whose queries and results live in mapped pinned host memory):
    // every workgroup counts itself in done_cnt when it has no work left; the last one copies *again_cnt to
    // host_again, resets the counter and stores flag_val to host_flag -- the host thread spins on that word.
    uint32_t *done_cnt;
    uint32_t *host_flag;
    int32_t *host_again;
    uint32_t flag_val;
    // int8 copy of the rows for the rejection test (quantize_rows_kernel); null = every neighbour is evaluated in f32
    const uint32_t *qrows;
    const float4 *qmeta;
    unsigned long long *rej_stats;  // profiling: [0] += f32 rows fetched, [1] += neighbours evaluated (null otherwise)
    unsigned long long *dbg;  // -DHG_HNSW_STAMPS diagnostic builds only: per-phase s_memrealtime totals
};

// -DHG_IVF_STAMPS diagnostic builds (tools/build_stamps.sh): absolute wall_clock64 stamps (100 MHz) of the phases of ONE
// IVF search's latency chain -- routing tail and finish kernel of query 0 -- into the buffer of
// hnswgpu_debug_set_tile_stamps: slots 16.. (tools/ivf_phase_stamps.py names them)
#ifdef HG_IVF_STAMPS
#define HG_IVF_STAMP(buf, slot, cond)                                      \
    do {                                                                   \
        if ((buf) && (cond)) (buf)[slot] = wall_clock64();                 \
    } while (0)
#else
#define HG_IVF_STAMP(buf, slot, cond) do { } while (0)
#endif

#ifdef HG_HNSW_STAMPS
#define HG_STAMP(slot)                                                  \
    do {                                                                \
        const unsigned long long t_now = wall_clock64(); \
        st_acc[slot] += t_now - st_prev;                                \
        st_prev = t_now;                                                \
    } while (0)
#else
#define HG_STAMP(slot) do { } while (0)
#endif

constexpr uint32_t kExpanded = 0x80000000u;

// NW = waves per query: 4 for latency (few queries), 1-2 for throughput (more queries resident per CU;
// with NW = 1 every barrier is wave-local).  VG = visited set in HBM (generation stamps) instead of
// the LDS bitset.  The arithmetic and the traversal are identical for every NW / VG.
// The grid is persistent: workgroup b serves queries b, b + gridDim.x, ...
constexpr int kPfMailWords = 160, kPfRing = 64;  // 32-bit words per query's mailbox; ring entries

// ---- L2 prefetch helpers of a handful of queries (HnswArgs::pf_mail) ------------------------------------------------
// Workgroup b sits on XCD b % 8; query t and its helpers share XCD t % 8:
//     b = (t % 8) + 8 * ((t / 8) * (1 + G) + role), role 0 = the traversal, 1..G = helpers
__device__ __forceinline__ void pf_place(const HnswArgs &a, int nq_eff, int &role, int &query, uint32_t *&mail) {
    const int u = blockIdx.x >> 3, team = 1 + a.pf_groups;
    role = u % team;
    query = (u / team) * 8 + (blockIdx.x & 7);
    mail = query < nq_eff ? a.pf_mail + static_cast<int64_t>(query) * kPfMailWords : nullptr;
}

// A helper: follow the query's ring until the traversal is done (or 20 ms have passed: never hang), reading ITS slice
// of every posted node's neighbour rows -- and throwing them away: they are in the XCD's L2 afterwards.
template <int NCH, int RB, bool L2, int NW>
__device__ __forceinline__ void pf_helper(const HnswArgs &a, uint32_t *mail, int pf_role, int pf_query) {
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    const int nvec = static_cast<int>(a.ld / 4);
    float4 hq[NCH];  // the query and its norm, exactly as the traversal holds them
    load_query<NCH>(hq, a.Q + static_cast<int64_t>(pf_query) * a.qld, a.dim, lane);
    const float hqn = a.metric == METRIC_COS ? query_norm<NCH>(hq) : 0.0f;
    const unsigned long long t_begin = wall_clock64();
    auto timed_out = [&]() { return wall_clock64() - t_begin > 2000000ull; };
    // the ring: the best unexpanded list entries, posted by wave 1 while wave 0 selects the next candidate.  (A second
    // ring with the closest fresh neighbour of every expansion, posted as soon as its distance was known, made the
    // search SLOWER -- 448 us against 426: dropped.)
    const unsigned long long *ring = reinterpret_cast<const unsigned long long *>(mail + 16);
    uint32_t e = 0;  // every helper follows every entry and fetches ITS slice of the node's neighbours
    for (;;) {
        const bool done = coherent_load(mail + 2) == a.pf_seq;
        const unsigned long long v = coherent_load(ring + (e % kPfRing));
        const uint32_t tag = static_cast<uint32_t>(v >> 32), node = static_cast<uint32_t>(v);
        // an entry of a LATER lap than the one I wait for: the traversal has lapped me, take up what is there
        // (an earlier lap: my entry has not been written yet)
        const uint32_t ahead = ((tag & 0xff) - ((e / kPfRing) & 0xff)) & 0xff;
        if ((tag >> 8) == a.pf_seq && ahead < 128) {
            e += ahead * kPfRing + 1;
            if (node < static_cast<uint32_t>(a.n)) {
                const int nb = a.l0_adj[static_cast<int64_t>(node) * a.M0 + (lane < a.M0 ? lane : a.M0 - 1)];
                // neighbours [lo, hi) are this helper's; its waves take them RB at a time
                const int per = (a.M0 + a.pf_groups - 1) / a.pf_groups, lo = (pf_role - 1) * per;
                const int hi = lo + per < a.M0 ? lo + per : a.M0;
                if (a.pf_res != nullptr) {
                    // distances of this helper's slice of the node's neighbours, RB rows per trip, by the entry's wave
                    if (static_cast<int>((e - 1) % NW) == wave) {
                        const uint32_t ee = e - 1;  // the entry in hand (e already points past it)
                        const unsigned long long tagw =
                            ((static_cast<unsigned long long>(a.pf_seq) << 8) | ((ee / kPfRing) & 0xff)) << 32;
                        unsigned long long *res = a.pf_res + (static_cast<int64_t>(pf_query) * kPfRing + (ee % kPfRing)) * kMaxDeg;
                        for (int j0 = lo; j0 < hi; j0 += RB) {
                            float4 r[RB][NCH];
                            // lane b < RB owns row b of the trip: its neighbour id (through LDS-free v_readlane in the
                            // unrolled loop below) and that row's norm
                            int32_t myid = -1;
#pragma unroll
                            for (int b = 0; b < RB; b++) {
                                const int j = j0 + b < hi ? j0 + b : hi - 1;
                                const int32_t t = __builtin_amdgcn_readlane(nb, j);
                                const bool ok = t >= 0 && t < a.n && j0 + b < hi;
                                if (lane == b) myid = ok ? t : -1;
                                load_row<NCH>(r[b], a.rows + static_cast<int64_t>(ok ? t : static_cast<int32_t>(node)) * a.ld, nvec, lane, true);
                            }
                            const float myrn = (a.metric == METRIC_COS && myid >= 0) ? a.row_norms[myid] : 0.0f;
                            float sums[RB];
#pragma unroll
                            for (int b = 0; b < RB; b++) sums[b] = lane_partial<NCH, L2>(hq, r[b]);
                            const float mine = rows_sum_to_lane<RB>(sums, lane);
                            if (lane < RB && myid >= 0) {
                                const float dv = finish_dist(a.metric, mine, hqn, myrn) + 0.0f;
                                coherent_store(res + (j0 + lane), tagw | __float_as_uint(dv));
                            }
                        }
                    }
                    continue;
                }
                for (int j0 = lo + wave * RB; j0 < hi; j0 += NW * RB) {
                    float4 r[RB][NCH];
#pragma unroll
                    for (int x = 0; x < RB; x++) {
                        const int j = j0 + x < hi ? j0 + x : hi - 1;
                        int32_t rid = __builtin_amdgcn_readlane(nb, j);
                        rid = (rid >= 0 && rid < a.n) ? rid : static_cast<int32_t>(node);
                        const float4 *rp = reinterpret_cast<const float4 *>(a.rows + static_cast<int64_t>(rid) * a.ld);
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) {
                            const int i = cc * kWave + lane;
                            r[x][cc] = rp[i < nvec ? i : nvec - 1];
                        }
                    }
#pragma unroll
                    for (int x = 0; x < RB; x++)
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) asm volatile("" ::"v"(r[x][cc].x));  // the loads must happen
                }
            }
            continue;  // look at the next entry right away
        }
        if (done || timed_out()) break;
        __builtin_amdgcn_s_sleep(8);
    }
}

// One wave of the traversal: post the best few unexpanded entries of `list` (from `start`) that were not posted a
// moment ago.  head / recent are that wave's posting state (recent: lane l holds the l-th recent node).  Data-tagged
// 8-byte entries: one agent-scope store each, nothing to wait for.
constexpr uint32_t kExpandedFlag = 0x80000000u;
__device__ __forceinline__ void pf_post(const HnswArgs &a, uint32_t *mail, const uint2 *list, int start, int len,
                                        uint32_t &head, uint32_t &recent, int lane, uint2 *hint_s) {
    int posted = 0;
    const int want = a.pf_hints;
    for (int base = start; base < len && posted < want; base += kWave) {
        const int i = base + lane;
        const uint32_t y = i < len ? list[i].y : kExpandedFlag;
        uint64_t m = __ballot(!(y & kExpandedFlag));
        for (; m && posted < want; m &= m - 1) {
            const int j = __ffsll(static_cast<unsigned long long>(m)) - 1;
            const uint32_t node = __builtin_amdgcn_readlane(y, j);
            posted++;
            if (__ballot(recent == node)) continue;  // hinted a moment ago
            if (lane == 0) {
                const unsigned long long tagv = (static_cast<unsigned long long>(a.pf_seq) << 8) | ((head / kPfRing) & 0xff);
                coherent_store(reinterpret_cast<unsigned long long *>(mail + 16) + (head % kPfRing), (tagv << 32) | node);
                hint_s[head % kPfRing] = make_uint2(node, head);  // the traversal's own copy of the ring: (node, entry)
            }
            if (lane == static_cast<int>(head % kWave)) recent = node;
            head++;
        }
    }
}

template <int NCH, int RB, bool L2, int NW, bool VG, bool PF = false>
__global__ __launch_bounds__(NW * kWave) void hnsw_search_kernel(HnswArgs a) {
    constexpr int kThreads = NW * kWave;
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS: [list: cap x (distance bits, node | expanded flag)] [cand_id | cand_d | cand_P: kMaxDeg each] [sc: 16] [part: NW x 64]
    //      [hint_s: kPfRing x 8 B] [bits: nwords] [posA: cap x 16 bit].  ONE list: the merge moves entries in place (they only
    //      ever move towards the tail), so a query costs 10 bytes of LDS per list slot instead of the 20 of a double buffer --
    //      at large ef the list is what bounds the queries a CU holds (ef 704: 12 instead of 7).
    uint2 *listA = reinterpret_cast<uint2 *>(smem);
    int32_t *cand_id = reinterpret_cast<int32_t *>(listA + a.cap);
    float *cand_d = reinterpret_cast<float *>(cand_id + kMaxDeg);
    int32_t *cand_P = reinterpret_cast<int32_t *>(cand_d + kMaxDeg);
    int32_t *sc = cand_P + kMaxDeg;  // scalars
    int32_t *part = sc + 16;         // [NW][64] per-wave partial counts of the merge
    uint2 *hint_s = reinterpret_cast<uint2 *>(part + NW * kWave);  // [kPfRing] (node, ring entry) of the hints posted (PF)
    uint32_t *bits = reinterpret_cast<uint32_t *>(hint_s + kPfRing);
    uint16_t *posA = reinterpret_cast<uint16_t *>(bits + a.nwords);  // [cap] merged position of every list entry (cap < 65536)
    uint32_t *stamps = VG ? a.vis + static_cast<int64_t>(blockIdx.x) * a.vis_stride : nullptr;
    // sc[0]=cursor sc[1]=ncand sc[2]=nadmit sc[3]=minP sc[4]=worst bits sc[5]=nghost sc[6]=ghost overflow (per query)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int nvec = static_cast<int>(a.ld / 4);
    uint32_t gen = a.gen_base;

    const int nq_eff = a.nq_dev ? (*a.nq_dev < a.nq ? *a.nq_dev : a.nq) : a.nq;
    int pf_role = 0, wi0 = blockIdx.x, wi_step = gridDim.x;
    uint32_t *mail = nullptr;
    if (PF) {
        pf_place(a, nq_eff, pf_role, wi0, mail);
        wi_step = 0x40000000;  // one query per traversal workgroup
        if (pf_role > 0 && wi0 < nq_eff) pf_helper<NCH, RB, L2, NW>(a, mail, pf_role, wi0);
        if (pf_role > 0) wi0 = nq_eff;  // helpers (and spare workgroups) take no query
    }
    for (int wi = wi0; wi < nq_eff; wi += wi_step) {
        const int qi = a.q_index ? a.q_index[wi] : wi;  // the query this work item serves
        uint2 *const curA = listA;
        uint32_t pf_head = 0, pf_recent = 0xffffffffu;  // wave 1: entries posted so far; lane l: the l-th recent node
        const float *qptr = a.q_rows ? a.rows + static_cast<int64_t>(a.q_rows[qi]) * a.ld : a.Q + qi * a.qld;
        float4 q[NCH];
        load_query<NCH>(q, qptr, a.dim, lane);
        float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
        QueryCode<NCH> qc;  // the query's side of the rejection test
        if (a.qrows != nullptr) encode_query<NCH>(q, qc);
        const int qlevel = a.q_levels ? a.q_levels[qi] : -1;

        int64_t n_eval = 0, n_hop = 0, n_exact = 0;
        int len = 0;
#ifdef HG_HNSW_STAMPS
        unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long st_prev = wall_clock64();
        const unsigned long long st_w0 = st_prev, st_c0 = clock64();
#endif
        __syncthreads();  // the previous query's result readers are done with the lists
        if (tid == 0) sc[6] = 0;
        if (PF && tid < kPfRing) hint_s[tid] = make_uint2(0xffffffffu, 0u);
        // seed: the entry point (ultra_fast.clj:358-359)
        {
            float4 r[NCH];
            load_row<NCH>(r, a.rows + static_cast<int64_t>(a.entry) * a.ld, nvec, lane, true);
            float s = wave_sum(lane_partial<NCH, L2>(q, r));
            float d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[a.entry] : 0.0f);
            if (tid == 0) curA[0] = make_uint2(__float_as_uint(d + 0.0f), static_cast<uint32_t>(a.entry));
            len = 1;
            n_eval = 1;
        }
        const int top_l = (a.ref_start && qlevel >= 0 && qlevel < a.max_level) ? qlevel : a.max_level;
        for (int level = top_l; level >= 0; level--) {
            int ef_l = level > 0 ? 1 : a.ef;
            // fresh visited set per layer (:156); entries carried from the level above are marked
            if (VG) {
                gen++;
            } else {
                for (int w = tid; w < a.nwords; w += kThreads) bits[w] = 0;
            }
            __syncthreads();
            if (len > ef_l) len = ef_l;
            // the reference re-evaluates its entry points at every layer (:162-167); the values are
            // reused here, but counted so that `evals` is the reference's number of distance calls
            if (level != top_l) n_eval += len;
            for (int i = tid; i < len; i += kThreads) {
                uint2 e = curA[i];
                e.y &= ~kExpanded;
                curA[i] = e;
                if (VG) atomicExch(&stamps[e.y], gen);
                else atomicOr(&bits[e.y >> 5], 1u << (e.y & 31));
            }
            __syncthreads();
            int cur_start = 0;
            const int deg = level == 0 ? a.M0 : a.M;
            HG_STAMP(0);  // seed / level set-up
            for (;;) {
                // ---- wave 0: next candidate (first unexpanded entry), its neighbour ids, visited filter,
                //      compaction.  One barrier for all of it.
                if (wave == 0) {
                    int found = -1;
                    for (int base = cur_start; base < len && found < 0; base += kWave) {
                        int i = base + lane;
                        bool un = i < len && !(curA[i].y & kExpanded);
                        uint64_t m = __ballot(un);
                        if (m) found = base + __ffsll(static_cast<unsigned long long>(m)) - 1;
                    }
                    int ncand = 0;
                    if (found >= 0) {
                        const uint32_t node = curA[found].y & ~kExpanded;
                        const int32_t *adj = level == 0 ? a.l0_adj + static_cast<int64_t>(node) * a.M0
                                                        : a.up_adj + (a.up_off[node] + (level - 1)) * a.M;
                        int nb = lane < deg ? adj[lane] : -1;
                        // has a helper published the distances of this node's neighbours?  (the newest hint of the node
                        // in the traversal's copy of the ring; its words carry the hint's own tag, read with one 64-bit load
                        // per neighbour slot beside the adjacency row: nothing here waits for a helper)
                        float pubd = 0.0f;
                        bool have = false;
                        if (PF && level == 0 && a.pf_res != nullptr) {
                            const uint2 h = hint_s[lane];
                            uint64_t hm = __ballot(h.x == node);
                            if (hm) {
                                uint32_t e_new = 0;
                                for (; hm; hm &= hm - 1) {
                                    const uint32_t e1 = __builtin_amdgcn_readlane(h.y, __ffsll(static_cast<unsigned long long>(hm)) - 1);
                                    e_new = e1 > e_new ? e1 : e_new;
                                }
                                const uint32_t tag = (a.pf_seq << 8) | ((e_new / kPfRing) & 0xff);
                                if (lane < deg) {
                                    const unsigned long long wv = coherent_load(
                                        a.pf_res + (static_cast<int64_t>(wi) * kPfRing + (e_new % kPfRing)) * kMaxDeg + lane);
                                    have = static_cast<uint32_t>(wv >> 32) == tag;
                                    pubd = __uint_as_float(static_cast<uint32_t>(wv));
                                }
                            }
                        }
                        bool fresh = false;
                        if (nb >= 0 && nb < a.n) {
                            if (VG) {
                                fresh = atomicExch(&stamps[nb], gen) != gen;
                            } else {
                                uint32_t bit = 1u << (nb & 31);
                                uint32_t old = atomicOr(&bits[nb >> 5], bit);
                                fresh = !(old & bit);
                            }
                        }
                        uint64_t m = __ballot(fresh);
                        int pos = __popcll(m & ((1ull << lane) - 1ull));
                        if (fresh) {
                            cand_id[pos] = nb;
                            if (PF) {
                                cand_P[pos] = have ? 1 : 0;     // (cand_P is the merge's from step 2 on; until then: published?)
                                if (have) cand_d[pos] = pubd;   // the gather below skips this candidate
                            }
                        }
                        ncand = __popcll(m);
                        if (lane == 0) curA[found].y = node | kExpanded;
                    }
                    if (lane == 0) {
                        sc[0] = found;
                        sc[1] = ncand;
                        sc[7] = 0;  // this hop's "needs the f32 row" mask (rejection test below)
                        sc[8] = 0;
                    }
                }
                if (PF && NW > 1 && wave == 1 && level == 0)
                    // meanwhile (wave 1 idles here): post the best few unexpanded candidates to the helpers.  It reads
                    // the list while wave 0 flags its pick -- a stale flag only costs a redundant hint.
                    pf_post(a, mail, curA, cur_start, len, pf_head, pf_recent, lane, hint_s);
                __syncthreads();
                HG_STAMP(1);  // select + adjacency + visited filter
                const int c = sc[0];
                if (c < 0) break;
                const int nc = sc[1];
                n_hop++;
                if (nc == 0) {
                    cur_start = c + 1;
                    __syncthreads();  // sc[0] / sc[1] are rewritten by wave 0 right away
                    continue;
                }
                n_eval += nc;
                const bool list_full = len >= ef_l;
                const float worst0 = list_full ? __uint_as_float(curA[ef_l - 1].x) : 0.0f;
                // ---- rejection test on the int8 rows (see quantize_rows_kernel): with a full list a neighbour whose
                //      LOWER BOUND is already >= the worst cannot be admitted (:195-198 is a strict <, the worst only
                //      shrinks within a hop) and its distance is never looked at again -- it gets +inf and no f32 fetch
                uint64_t needmask = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
                if (PF && a.pf_res != nullptr)  // distances a helper has published (cand_d holds them): nothing to gather
                    needmask &= ~__ballot(lane < nc && cand_P[lane] != 0);
                if (a.qrows != nullptr && list_full) {
                    // eight code rows per wave step (sixteen in flight cost 220 VGPRs at dim 768: a wave less per SIMD)
                    uint64_t wmask = 0;
                    const int own = wave_sum8_row(lane);  // the row of a step whose total this lane receives
                    for (int j0 = wave * 8; j0 < nc; j0 += NW * 8) {
                        // the lane that will hold row `own`'s total fetches that candidate's meta data up front
                        const int myj = j0 + own;
                        const bool ok = (lane & 7) == 0 && myj < nc;
                        const float4 mymeta = ok ? a.qmeta[cand_id[myj]] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                        const int lj = j0 + lane;
                        const int32_t ids8 = cand_id[(lane < 8 && lj < nc) ? lj : j0];  // past nc: a valid row, unused
                        uint32_t w[8][NCH];
#pragma unroll
                        for (int b = 0; b < 8; b++) {
                            const int32_t rid = __builtin_amdgcn_readlane(ids8, b);
                            const uint32_t *rp = a.qrows + (static_cast<int64_t>(rid) * kWave + lane) * NCH;
#pragma unroll
                            for (int cc = 0; cc < NCH; cc++) w[b][cc] = rp[cc];
                        }
                        int acc[8];
#pragma unroll
                        for (int b = 0; b < 8; b++) acc[b] = code_dot<NCH>(qc.a, w[b]);
                        const int tot = wave_sum8_int(acc, lane);
                        const float lb = code_lower_bound(a.metric, tot, qc.sc, mymeta, mymeta.w);
                        const bool need = ok && !(lb >= worst0);  // NaN: needs the exact distance
                        if (ok) cand_d[myj] = __uint_as_float(0x7f800000u);  // overwritten below if needed
                        // bit 8r of the ballot = row r of the step
                        const uint64_t m8 = ((__ballot(need) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
                        wmask |= m8 << j0;
                    }
                    if (NW > 1) {
                        if (lane == 0 && wmask) {
                            if (static_cast<uint32_t>(wmask)) atomicOr(reinterpret_cast<uint32_t *>(&sc[7]), static_cast<uint32_t>(wmask));
                            if (wmask >> 32) atomicOr(reinterpret_cast<uint32_t *>(&sc[8]), static_cast<uint32_t>(wmask >> 32));
                        }
                        __syncthreads();
                        needmask = static_cast<uint32_t>(sc[7]) | (static_cast<uint64_t>(static_cast<uint32_t>(sc[8])) << 32);
                    } else {
                        needmask = wmask;
                    }
                }
                // ---- gather rows + distances of the neighbours that need them: wave w takes the needed candidates of
                //      rank [w*RB + t*NW*RB, +RB) (all of them while the list is filling or without int8 rows)
                HG_STAMP(8);  // rejection test (int8 rows)
                const int nneed = __popcll(needmask);
                n_exact += nneed;
                const bool isset = (needmask >> lane) & 1ull;
                const int rank = __popcll(needmask & ((1ull << lane) - 1ull));
                for (int t0 = wave * RB; t0 < nneed; t0 += NW * RB) {
                    float4 r[RB][NCH];
                    // lane b takes the candidate of rank t0 + b; it fetches that candidate's id and precomputed norm up
                    // front, so that the norm's memory round trip overlaps the row gather instead of following the reduction
                    int myj = -1;
#pragma unroll
                    for (int b = 0; b < RB; b++) {
                        const uint64_t hit = __ballot(isset && rank == t0 + b);
                        const int jb = hit ? __ffsll(static_cast<unsigned long long>(hit)) - 1 : -1;
                        myj = lane == b ? jb : myj;
                    }
                    const int32_t myid = myj >= 0 ? cand_id[myj] : 0;
                    const float myrn = (a.metric == METRIC_COS && myj >= 0) ? a.row_norms[myid] : 0.0f;
#pragma unroll
                    for (int b = 0; b < RB; b++) {
                        const int32_t rid = __builtin_amdgcn_readlane(myid, b);
                        load_row<NCH>(r[b], a.rows + static_cast<int64_t>(rid) * a.ld, nvec, lane, t0 + b < nneed);
                    }
                    float s[RB];
#pragma unroll
                    for (int b = 0; b < RB; b++) s[b] = lane_partial<NCH, L2>(q, r[b]);
                    const float mine = rows_sum_to_lane<RB>(s, lane);  // lane b keeps candidate b's reduced sum
                    if (myj >= 0) cand_d[myj] = finish_dist(a.metric, mine, qn, myrn) + 0.0f;
                }
                __syncthreads();
                HG_STAMP(2);  // row gather + distances
                // ---- pre-filter: with a full list a candidate at or beyond the current worst can never be admitted
                //      (:195-198 is a strict <, and the worst only shrinks within a hop), and it changes no other
                //      entry's rank.  Late in a search that is most candidates: the merge loops below run over the
                //      survivors only, and a hop without survivors skips the merge and its three barriers.
                const float cdist = lane < nc ? cand_d[lane] : 0.0f;
                const int cbits = __float_as_int(cdist);
                const bool surv = lane < nc && (!list_full || cdist < worst0);
                const uint64_t smask = __ballot(surv);  // identical in every wave
                if (smask == 0) {
                    cur_start = c + 1;
                    continue;  // nobody touches sc[] between the barrier above and wave 0's next write
                }
                // ---- merge, step 1 (all waves): every list entry counts the candidates that go before it
                //      (its shift), every candidate counts the list entries that stay before it.  The
                //      candidate distances sit one per lane and are broadcast with v_readlane.
                int cntA = 0;  // lane j: # list entries (of this wave's share) with d <= d_j
                for (int base = 0; base < len; base += kThreads) {
                    const int i = base + tid;
                    const bool valid = i < len;
                    const float de = valid ? __uint_as_float(curA[i].x) : 0.0f;
                    int sh = 0;
                    for (uint64_t mm = smask; mm; mm &= mm - 1) {
                        const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
                        const float dj = __int_as_float(__builtin_amdgcn_readlane(cbits, j));
                        sh += (dj < de) ? 1 : 0;
                        const int below = __popcll(__ballot(valid && de <= dj));
                        cntA += (lane == j) ? below : 0;
                    }
                    if (valid) {
                        const int P = i + sh;
                        posA[i] = static_cast<uint16_t>(P < 65535 ? P : 65535);
                        if (P == ef_l - 1) sc[4] = static_cast<int32_t>(curA[i].x);
                    }
                }
                if (surv) part[wave * kWave + lane] = cntA;
                __syncthreads();
                HG_STAMP(3);  // merge step 1
                // ---- merge, step 2 (wave 0): rank, admission and final position of every candidate
                if (wave == 0) {
                    bool admitted = false;
                    int P = 0x7fffffff;
                    int before = 0, after_less = 0;
                    for (uint64_t mm = smask; mm; mm &= mm - 1) {
                        const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
                        const float o = __int_as_float(__builtin_amdgcn_readlane(cbits, j));
                        before += (j < lane && o <= cdist) ? 1 : 0;
                        after_less += (j > lane && o < cdist) ? 1 : 0;
                    }
                    if (surv) {
                        int below = 0;
#pragma unroll
                        for (int w = 0; w < NW; w++) below += part[w * kWave + lane];
                        const int r = below + before;
                        admitted = r < ef_l;
                        P = r + after_less;
                        if (admitted && P == ef_l - 1) sc[4] = cbits;
                    }
                    if (lane < nc) cand_P[lane] = admitted ? P : -1;
                    uint64_t am = __ballot(admitted);
                    int minP = admitted ? P : 0x7fffffff;
                    for (int off = 1; off < kWave; off <<= 1) {
                        int o = __shfl_xor(minP, off, kWave);
                        minP = o < minP ? o : minP;
                    }
                    if (lane == 0) {
                        sc[2] = __popcll(am);
                        sc[3] = minP;
                        sc[5] = 0;
                    }
                }
                __syncthreads();
                HG_STAMP(4);  // merge step 2
                const int nadm = sc[2];
                if (nadm == 0) {
                    cur_start = c + 1;
                    continue;
                }
                // ---- merge, step 3 (all waves): the entries move to their merged positions IN PLACE -- every entry moves
                //      towards the tail (by the number of admitted candidates that go before it), so the list is walked
                //      from its last block of kThreads entries down to the block of the first admitted position: a block is
                //      read, then (all waves have read their share) written; its writes land at or behind its own start,
                //      never on an entry that is still to be read.  Entries before the first admitted position stay where
                //      they are.  Entries pushed past ef whose distance ties the new worst stay as ghosts.
                const int total = len + nadm;
                const bool full = total > ef_l;
                const uint32_t wbits = static_cast<uint32_t>(sc[4]);
                const int minP = sc[3];
                int ghosts = 0;
                for (int base = ((len - 1) / kThreads) * kThreads; base >= 0 && base + kThreads > minP; base -= kThreads) {
                    const int i = base + tid;
                    const bool valid = i < len;
                    uint2 e = make_uint2(0u, 0u);
                    int P = 0;
                    if (valid) {
                        e = curA[i];
                        P = posA[i];
                    }
                    if (NW > 1) __syncthreads();  // (one wave: its LDS reads and writes are in program order)
                    bool gh = false;
                    if (valid) {
                        if (P < a.cap && P != i) curA[P] = e;
                        gh = full && P >= ef_l && P < a.cap && e.x == wbits;
                        if (full && P >= a.cap && e.x == wbits && !(e.y & kExpanded)) sc[6] = 1;  // a tie with no slot left
                    }
                    ghosts += __popcll(__ballot(gh));
                }
                {
                    // the admitted candidates into the gaps (every entry of the blocks above has been read: the loop's
                    // barriers; a candidate's position is at or behind minP, i.e. inside those blocks or behind the list)
                    bool gh = false;
                    if (tid < nc) {
                        const int P = cand_P[tid];
                        if (P >= 0) {
                            if (P < a.cap) curA[P] = make_uint2(static_cast<uint32_t>(cbits), static_cast<uint32_t>(cand_id[tid]));
                            gh = full && P >= ef_l && P < a.cap && static_cast<uint32_t>(cbits) == wbits;
                            if (full && P >= a.cap && static_cast<uint32_t>(cbits) == wbits) sc[6] = 1;
                        }
                    }
                    if (wave == 0) ghosts += __popcll(__ballot(gh));
                }
                if (lane == 0 && ghosts) atomicAdd(&sc[5], ghosts);
                __syncthreads();
                HG_STAMP(5);  // merge step 3
                const int newlen = full ? ef_l + sc[5] : total;
                len = newlen;
                cur_start = minP < c + 1 ? minP : c + 1;
            }
            // ---- level done
            if (a.q_rows && level > 0 && level <= qlevel && tid == 0) {
                a.up_out_ids[static_cast<int64_t>(qi) * a.up_stride + (level - 1)] =
                    len > 0 ? static_cast<int32_t>(curA[0].y & ~kExpanded) : -1;
                a.up_out_dist[static_cast<int64_t>(qi) * a.up_stride + (level - 1)] =
                    len > 0 ? __uint_as_float(curA[0].x) : 0.0f;
            }
        }
        // ---- results: ascending, take k (:362-370; the distances are reused, not recomputed)
        int real = len < a.ef ? len : a.ef;
        for (int i = tid; i < a.k; i += kThreads) {
            bool ok = i < real;
            a.out_ids[static_cast<int64_t>(qi) * a.k + i] = ok ? static_cast<int32_t>(curA[i].y & ~kExpanded) : -1;
            a.out_dist[static_cast<int64_t>(qi) * a.k + i] =
                ok ? __uint_as_float(curA[i].x) : __uint_as_float(0x7f800000u);
        }
#ifdef HG_HNSW_STAMPS
        HG_STAMP(6);
        st_acc[10] = clock64() - st_c0;
        st_acc[11] = wall_clock64() - st_w0;
        if (a.dbg && tid == 0 && qi == 0)
            for (int i = 0; i < 12; i++) a.dbg[i] = i == 7 ? static_cast<unsigned long long>(n_hop) : st_acc[i];
#endif
        if (a.again && tid == 0 && sc[6]) a.again[atomicAdd(a.again_cnt, 1)] = qi;  // sc[6]: ordered by the last merge's barrier
        if (a.stats && tid == 0) {
            a.stats[2 * static_cast<int64_t>(qi)] = n_eval;
            a.stats[2 * static_cast<int64_t>(qi) + 1] = n_hop;
        }
        if (a.rej_stats && tid == 0) {
            atomicAdd(a.rej_stats, static_cast<unsigned long long>(n_exact));
            atomicAdd(a.rej_stats + 1, static_cast<unsigned long long>(n_eval));
        }
        if (PF && tid == 0) coherent_store(mail + 2, a.pf_seq);  // the helpers may go
    }
    if (a.host_flag) {
        __threadfence_system();  // this thread's results (host memory) are visible system-wide ...
        __syncthreads();         // ... before the workgroup reports itself done
        if (tid == 0) {
            if (atomicAdd(a.done_cnt, 1u) == gridDim.x - 1) {  // the last workgroup: every result is out
                atomicExch(a.done_cnt, 0u);
                __threadfence();
                *a.host_again = a.again_cnt ? atomicAdd(a.again_cnt, 0) : 0;
                __threadfence_system();
                __hip_atomic_store(a.host_flag, a.flag_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

}  // namespace hg
