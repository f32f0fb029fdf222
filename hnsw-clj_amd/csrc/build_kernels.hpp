// build_kernels.hpp -- neighbour selection of the HNSW builder by the diversity heuristic of src/hnsw/graph.clj:162-198
// (get-neighbors-heuristic), used for a new node's own links and for an over-full neighbour list
// (prune-connections, graph.clj:208-232).  Included by hnsw.hip only.
#pragma once
#include "kernels.hpp"

namespace hg {

constexpr int kSelMaxCand = 4096;  // candidates of one selection task (ef_construction <= 4096; longer lists: the closest)

struct HeurArgs {
    const float *rows;
    const float *row_norms;
    int64_t ld;
    int32_t dim, metric;
    int32_t ntasks;
    const int64_t *off;       // [ntasks + 1] a task's candidates are cand_id / cand_d [off[t], off[t + 1])
    const int32_t *cand_id;   // ascending by distance; -1 entries at the end of a task's range are padding
    const float *cand_d;      // distance of the candidate to the task's base node (as the search / the stored edge has it)
    const int32_t *m;         // [ntasks] links wanted (2M at layer 0, M above), or null: m_all
    int32_t m_all;
    int32_t extend;           // extend-candidates? (graph.clj:191-195)
    int32_t keep_rows;        // the taken rows stay in registers (round 5; 0 = every candidate fetches them again: A/B)
    int32_t out_stride;       // >= max m
    int32_t *out_id;          // [ntasks][out_stride] in selection order, -1 padded
    float *out_d;             // [ntasks][out_stride]
    int32_t *out_cnt;         // [ntasks]
};

// One workgroup per task.  The candidates are walked in ascending (distance, id) order (graph.clj:165-171; the lists
// arrive sorted by distance, runs of equal distances are put in id order first); the closest is taken, every further
// one is taken unless it is closer to an already taken one than to the base node (:181-188: `(distance graph
// current-id res-id)` < current-dist), until m are taken (:177); with `extend` the discarded ones fill up in their order.
// The pair distances are the traversal's own arithmetic (lane_partial + butterfly + finish_dist on the stored norms: what
// oracle.c's pair_dist computes in its device order), each wave evaluating its share of the taken rows against the
// candidate's row, RB rows in flight -- one memory round trip per candidate.
template <int NCH, int RB, bool L2>
__global__ __launch_bounds__(kWG) void heuristic_select_kernel(HeurArgs a) {
    __shared__ int32_t cid[kSelMaxCand];
    __shared__ float cd[kSelMaxCand];
    __shared__ int32_t res[kMaxDeg];
    __shared__ uint8_t state[kSelMaxCand];
    __shared__ int32_t sflag[2];
    __shared__ int32_t sflag2[3];
    const int t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int64_t o0 = a.off[t];
    int C = static_cast<int>(a.off[t + 1] - o0);
    if (C > kSelMaxCand) C = kSelMaxCand;
    const int m = a.m ? a.m[t] : a.m_all;
    const int nvec = static_cast<int>(a.ld / 4);
    if (tid == 0) sflag[0] = C;
    __syncthreads();
    for (int i = tid; i < C; i += kWG) {
        const int32_t id = a.cand_id[o0 + i];
        cid[i] = id;
        cd[i] = a.cand_d[o0 + i];
        if (id < 0) atomicMin(&sflag[0], i);  // padding starts here
    }
    __syncthreads();
    C = sflag[0];
    // runs of equal distances -> ascending id (each member of a run finds its rank inside the run)
    int32_t my_id[kSelMaxCand / kWG], my_pos[kSelMaxCand / kWG];
#pragma unroll
    for (int u = 0; u < kSelMaxCand / kWG; u++) {
        const int i = tid + u * kWG;
        my_pos[u] = -1;
        if (i < C) {
            const float d = cd[i];
            const bool tie = (i > 0 && cd[i - 1] == d) || (i + 1 < C && cd[i + 1] == d);
            if (tie) {
                int lo = i, hi = i;
                while (lo > 0 && cd[lo - 1] == d) lo--;
                while (hi + 1 < C && cd[hi + 1] == d) hi++;
                const int32_t me = cid[i];
                int rank = 0;
                for (int j = lo; j <= hi; j++) rank += (cid[j] < me || (cid[j] == me && j < i)) ? 1 : 0;
                my_id[u] = me;
                my_pos[u] = lo + rank;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSelMaxCand / kWG; u++)
        if (my_pos[u] >= 0) cid[my_pos[u]] = my_id[u];
    __syncthreads();
    int nres = 0;
    if (tid == 0) {
        sflag[1] = 0;
        sflag2[0] = sflag2[1] = sflag2[2] = 0;
    }
    for (int i = tid; i < C; i += kWG) state[i] = 0;
    __syncthreads();
    // The rows already taken stay IN REGISTERS: wave w keeps taken rows [KR w, KR w + KR) from the moment each of them was the
    // candidate (its row is in every wave's registers then), so a candidate costs ONE row fetch -- the next candidate's, issued
    // while this one is decided -- and one barrier.  (Fetching the taken rows again for every candidate was up to 528 rows of a
    // 33-candidate task and ~1,600 of a new node's 200: the 1.25M x 1536 build moved ~20 TB that way.)  Same pairs, same
    // arithmetic, same order of decisions.  Tasks that want more links than 4 x KR, and rows beyond 6 KB, reload as before.
    constexpr int KR = 8;
    constexpr bool kKeep = NCH <= 6;
    if (kKeep && a.keep_rows && m <= kNWave * KR) {
        float4 r[KR][NCH];
#pragma unroll
        for (int t = 0; t < KR; t++)
#pragma unroll
            for (int c = 0; c < NCH; c++) r[t][c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        float myrn = 0.0f;  // lane t < KR: the norm of this wave's taken row t
        // (rows of up to 4 KB: the next candidate's row is fetched into registers while this one is decided; longer rows would
        // push the kernel past 256 registers -- one workgroup per CU -- so they are fetched when needed, out of the L2 the touches
        // below have filled, and the second workgroup on the CU covers the wait)
        constexpr bool kAhead = NCH <= 4;
        float4 q[NCH], qx[kAhead ? NCH : 1];
        float qn = 0.0f, qnx = 0.0f;
        if (kAhead && C > 0) {
            load_row<NCH>(q, a.rows + static_cast<int64_t>(cid[0]) * a.ld, nvec, lane, true);
            qn = a.metric == METRIC_COS ? a.row_norms[cid[0]] : 0.0f;
        }
        // (further ahead the rows are only TOUCHED -- wave w one dword per 128-byte line of candidate i + 2 + w's row, a register
        // each -- so that the fetch one candidate ahead finds them in the L2: the rows of a task are scattered over the base, and
        // at a workgroup or two per CU a fetch from memory is longer than a decision takes)
        uint32_t sink = 0, touched = 0;
        auto touch = [&](int i) -> uint32_t {
            const int32_t c = cid[i < C ? i : C - 1];
            const int line = lane < (nvec + 7) / 8 ? lane : 0;
            return reinterpret_cast<const uint32_t *>(a.rows + static_cast<int64_t>(c) * a.ld)[32 * line];
        };
        if (wave + 1 < C) touched = touch(1 + wave);
        for (int i = 0; i < C && nres < m; i++) {
            const float dc = cd[i];
            if constexpr (kAhead) {
                const bool more = i + 1 < C;
                const int32_t cx = cid[more ? i + 1 : i];
                load_row<NCH>(qx, a.rows + static_cast<int64_t>(cx) * a.ld, nvec, lane, more);
                qnx = (more && a.metric == METRIC_COS) ? a.row_norms[cx] : 0.0f;
            } else {
                load_row<NCH>(q, a.rows + static_cast<int64_t>(cid[i]) * a.ld, nvec, lane, true);
                qn = a.metric == METRIC_COS ? a.row_norms[cid[i]] : 0.0f;
            }
            sink ^= touched;  // (last iteration's: long since in)
            touched = (wave == (i & 3) && i + 5 < C) ? touch(i + 5) : 0u;
            bool closer = false;
            if (nres > 0) {
                const int have = nres - wave * KR;  // this wave's taken rows (<= 0: none yet)
                bool mine = false;
                if (have > 0) {
                    float sv[KR];
#pragma unroll
                    for (int t = 0; t < KR; t++) sv[t] = lane_partial<NCH, L2>(q, r[t]);
                    const float sm = rows_sum_to_lane<KR>(sv, lane);
                    if (lane < KR && lane < have) mine = finish_dist(a.metric, sm, qn, myrn) < dc;
                }
                // (three flags in turn: the one cleared here was last read before the PREVIOUS barrier and is written behind this one)
                if (__ballot(mine) && lane == 0) atomicOr(&sflag2[i % 3], 1);
                if (tid == 0) sflag2[(i + 1) % 3] = 0;
                __syncthreads();
                closer = sflag2[i % 3] != 0;
            }
            if (tid == 0) state[i] = closer ? 2 : 1;  // 1 = taken, 2 = discarded (read by thread 0 alone, below)
            if (!closer) {
                if (wave == nres / KR) {  // (uniform per wave)
                    const int slot = nres % KR;
#pragma unroll
                    for (int t = 0; t < KR; t++)
                        if (t == slot) {
#pragma unroll
                            for (int c = 0; c < NCH; c++) r[t][c] = q[c];
                        }
                    if (lane == slot) myrn = qn;
                }
                nres++;
            }
            if constexpr (kAhead) {
#pragma unroll
                for (int c = 0; c < NCH; c++) q[c] = qx[c];
                qn = qnx;
            }
        }
        if ((sink ^ touched) == 0x9e3779b9u && a.ntasks < 0) a.out_cnt[t] = -1;  // (never: keeps the touches alive)
        __syncthreads();
    } else {
    for (int i = 0; i < C && nres < m; i++) {
        const int32_t c = cid[i];
        const float dc = cd[i];
        bool closer = false;
        if (nres > 0) {
            float4 q[NCH];
            load_row<NCH>(q, a.rows + static_cast<int64_t>(c) * a.ld, nvec, lane, true);
            const float qn = a.metric == METRIC_COS ? a.row_norms[c] : 0.0f;
            bool mine = false;
            for (int r0 = wave * RB; r0 < nres; r0 += kNWave * RB) {
                float4 r[RB][NCH];
                float rn = 0.0f;
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const bool ok = r0 + b < nres;
                    const int32_t rid = ok ? res[r0 + b] : 0;
                    load_row<NCH>(r[b], a.rows + static_cast<int64_t>(rid) * a.ld, nvec, lane, ok);
                    if (lane == b && ok && a.metric == METRIC_COS) rn = a.row_norms[rid];
                }
                float s[RB];
#pragma unroll
                for (int b = 0; b < RB; b++) s[b] = lane_partial<NCH, L2>(q, r[b]);
                const float sm = rows_sum_to_lane<RB>(s, lane);
                if (lane < RB && r0 + lane < nres) mine = mine || (finish_dist(a.metric, sm, qn, rn) < dc);
            }
            if (__ballot(mine) && lane == 0) atomicOr(&sflag[1], 1);
            __syncthreads();
            closer = sflag[1] != 0;
            __syncthreads();  // everybody has read the flag before it is cleared
        }
        if (tid == 0) {
            state[i] = closer ? 2 : 1;  // 1 = taken, 2 = discarded
            if (!closer) res[nres] = c;
            sflag[1] = 0;
        }
        if (!closer) nres++;
        __syncthreads();
    }
    }
    // output: the taken ones in selection order; then, with extend, the discarded ones in their order (:191-195) -- wave 0,
    // 64 candidates per step (one thread walking the list was ~10 us of a 200-candidate task)
    if (wave == 0) {
        int n = 0;
        int32_t *oi = a.out_id + static_cast<int64_t>(t) * a.out_stride;
        float *od = a.out_d + static_cast<int64_t>(t) * a.out_stride;
        for (int pass = 1; pass <= (a.extend ? 2 : 1); pass++)
            for (int base = 0; base < C && n < m; base += kWave) {
                const int i = base + lane;
                const bool is = i < C && state[i] == pass;
                const uint64_t mk = __ballot(is);
                const int pos = n + __popcll(mk & ((1ull << lane) - 1ull));
                if (is && pos < m) {
                    oi[pos] = cid[i];
                    od[pos] = cd[i];
                }
                n += __popcll(mk);
            }
        n = n < m ? n : m;
        if (lane == 0) a.out_cnt[t] = n;
        for (int j = n + lane; j < a.out_stride; j += kWave) {
            oi[j] = -1;
            od[j] = 0.0f;
        }
    }
}

}  // namespace hg
