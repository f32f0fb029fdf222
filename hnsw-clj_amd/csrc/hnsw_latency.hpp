// hnsw_latency.hpp -- the traversal kernel for a HANDFUL of queries (single-query latency, the reference's T-threads
// protocol): the same search-layer-ultra / search-knn restatement as hnsw_search_kernel (kernels.hpp), same arithmetic,
// same list, same counters -- but software-pipelined ACROSS expansions, because one query on one CU is bound by the
// chain of dependent memory round trips (adjacency row -> neighbour rows -> merge), not by bandwidth:
//
//   * the NEXT candidate is known exactly as soon as the current candidate's neighbour distances are: it is the better
//     of (a) the first unexpanded entry of the list as it stands and (b) the closest neighbour that will be admitted
//     (ties: the list entry, it was admitted earlier).  Nothing about it depends on the merge itself.
//   * so right after the distances, wave 0 picks it, takes its adjacency row from a two-entry register cache (filled one
//     expansion ahead with the row of the runner-up), filters it against the visited set, and every wave ISSUES the row
//     gather of the next expansion -- and only then is the current expansion's merge (rank-and-scatter, three barriers)
//     done, under those loads.
//   * nothing is speculative about what is evaluated: every gathered row belongs to the expansion the reference
//     performs next (ultra_fast.clj:170-204), so distance evaluations and expansions still equal the oracle's counts.
//
// Per expansion the critical path shrinks from  adjacency + gather + merge  to  max(gather, merge).
// Included by hnsw.hip only.
#pragma once
#include "kernels.hpp"

namespace hg {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, and on gfx9 that
// waits for EVERY outstanding vector-memory operation (loads share vmcnt with stores): the row gather issued for the
// next expansion would be drained at the first barrier of the merge it is meant to hide under.  Between these
// barriers the workgroup's shared state (lists, candidate buffers, visited bitset, scalars) lives in LDS only.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

inline size_t hnsw_latency_lds_bytes(int cap, int nwords, int nw) {
    return sizeof(uint2) * 2 * cap + sizeof(int32_t) * 5 * kMaxDeg + sizeof(int32_t) * 32 + sizeof(int32_t) * nw * kWave +
           sizeof(int32_t) * cap + sizeof(uint32_t) * nwords;
}

template <int NCH, int RB, bool L2, int NW>
__global__ __launch_bounds__(NW * kWave) void hnsw_latency_kernel(HnswArgs a) {
    constexpr int kThreads = NW * kWave;
    extern __shared__ __align__(16) unsigned char smem[];
    uint2 *listA = reinterpret_cast<uint2 *>(smem);
    uint2 *listB = listA + a.cap;
    int32_t *cand_id = reinterpret_cast<int32_t *>(listB + a.cap);  // [2][kMaxDeg]: current / next expansion
    float *cand_d = reinterpret_cast<float *>(cand_id + 2 * kMaxDeg);  // [2][kMaxDeg]
    int32_t *cand_P = reinterpret_cast<int32_t *>(cand_d + 2 * kMaxDeg);
    int32_t *sc = cand_P + kMaxDeg;      // [32] scalars
    int32_t *part = sc + 32;             // [NW][64] per-wave partial counts of the merge
    int32_t *posA = part + NW * kWave;   // [cap] merged position of every list entry
    uint32_t *bits = reinterpret_cast<uint32_t *>(posA + a.cap);
    // sc[0]=list index of the candidate being expanded  sc[1]=its fresh neighbours  sc[2]=nadmit  sc[4]=worst bits
    // sc[5]=nghost  sc[6]=ghost overflow (per query)
    // sc[8]=next candidate found (0/1)  sc[9]=its fresh neighbours  sc[10]=lane of the next candidate among the current
    // neighbours (-1: it is a list entry)  sc[11]=its index in the list before the merge (-1: it is a neighbour)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int nvec = static_cast<int>(a.ld / 4);

    const int nq_eff = a.nq_dev ? (*a.nq_dev < a.nq ? *a.nq_dev : a.nq) : a.nq;
    for (int wi = blockIdx.x; wi < nq_eff; wi += gridDim.x) {
        const int qi = a.q_index ? a.q_index[wi] : wi;
        uint2 *curA = listA, *curB = listB;
        float4 q[NCH];
        load_query<NCH>(q, a.Q + qi * a.qld, a.dim, lane);
        const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;

        int64_t n_eval = 0, n_hop = 0;
        int len = 0;
#ifdef HG_HNSW_STAMPS
        unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long st_prev = wall_clock64();
        const unsigned long long st_w0 = st_prev, st_c0 = clock64();
#endif
        __syncthreads();  // the previous query's result readers are done with the lists
        if (tid == 0) sc[6] = 0;
        {  // seed: the entry point (ultra_fast.clj:358-359)
            float4 r[NCH];
            load_row<NCH>(r, a.rows + static_cast<int64_t>(a.entry) * a.ld, nvec, lane, true);
            const float s = wave_sum(lane_partial<NCH, L2>(q, r));
            const float d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[a.entry] : 0.0f);
            if (tid == 0) curA[0] = make_uint2(__float_as_uint(d + 0.0f), static_cast<uint32_t>(a.entry));
            len = 1;
            n_eval = 1;
        }
        for (int level = a.max_level; level >= 0; level--) {
            const int ef_l = level > 0 ? 1 : a.ef;
            for (int w = tid; w < a.nwords; w += kThreads) bits[w] = 0;  // fresh visited set per layer (:156)
            lds_barrier();
            if (len > ef_l) len = ef_l;
            if (level != a.max_level) n_eval += len;  // the reference re-evaluates its entry points (:162-167)
            for (int i = tid; i < len; i += kThreads) {
                uint2 e = curA[i];
                e.y &= ~kExpanded;
                curA[i] = e;
                atomicOr(&bits[e.y >> 5], 1u << (e.y & 31));
            }
            lds_barrier();
            const int deg = level == 0 ? a.M0 : a.M;
            int cur_start = 0;
            int b = 0;  // cand_id / cand_d buffer of the expansion in hand
            // wave 0's adjacency cache: two rows, lane l holds neighbour l (deg <= 64 = one lane each)
            int pf_node0 = -1, pf_node1 = -1, pf_nb0 = -1, pf_nb1 = -1;
            // rows in flight (issued one expansion ahead), and which candidates they belong to
            float4 r[RB][NCH];
#pragma unroll
            for (int x = 0; x < RB; x++)
#pragma unroll
                for (int cc = 0; cc < NCH; cc++) r[x][cc] = make_float4(0.f, 0.f, 0.f, 0.f);
            int32_t myid = 0;
            float myrn = 0.0f;

            // Wave 0: choose the next candidate, fetch / look up its adjacency, visited filter, compaction into buffer
            // `nb_buf`.  `have`: the distances of the expansion in hand (buffer b, `nc_cur` of them) take part.
            auto select_next = [&](int nb_buf, bool have, int nc_cur) {
                // (a) first unexpanded list entry
                int fo = -1;
                for (int base = cur_start; base < len && fo < 0; base += kWave) {
                    const int i = base + lane;
                    const bool un = i < len && !(curA[i].y & kExpanded);
                    const uint64_t m = __ballot(un);
                    if (m) fo = base + __ffsll(static_cast<unsigned long long>(m)) - 1;
                }
                const float d_fo = fo >= 0 ? __uint_as_float(curA[fo].x) : 0.0f;
                // (b) closest neighbour that will be admitted: strict < keeps the earliest among equals
                int bj = -1;
                float best = 0.0f;
                if (have && nc_cur > 0) {
                    const float cdist = lane < nc_cur ? cand_d[b * kMaxDeg + lane] : 0.0f;
                    const bool list_full = len >= ef_l;
                    const float worst0 = list_full ? __uint_as_float(curA[ef_l - 1].x) : 0.0f;
                    const bool surv = lane < nc_cur && (!list_full || cdist < worst0);
                    const int cb = __float_as_int(cdist);
                    for (uint64_t mm = __ballot(surv); mm; mm &= mm - 1) {
                        const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
                        const float dj = __int_as_float(__builtin_amdgcn_readlane(cb, j));
                        if (bj < 0 || dj < best) {
                            best = dj;
                            bj = j;
                        }
                    }
                }
                const bool use_new = bj >= 0 && (fo < 0 || best < d_fo);  // equal: the list entry was admitted first
                int ncand = 0;
                const bool any = use_new || fo >= 0;
                if (any) {
                    const uint32_t node = use_new ? static_cast<uint32_t>(cand_id[b * kMaxDeg + bj]) : (curA[fo].y & ~kExpanded);
                    int nbr;
                    if (static_cast<int>(node) == pf_node0) {
                        nbr = pf_nb0;
                        pf_node0 = -1;
                    } else if (static_cast<int>(node) == pf_node1) {
                        nbr = pf_nb1;
                        pf_node1 = -1;
                    } else {
                        const int32_t *adj = level == 0 ? a.l0_adj + static_cast<int64_t>(node) * a.M0
                                                        : a.up_adj + (a.up_off[node] + (level - 1)) * a.M;
                        nbr = adj[lane < deg ? lane : deg - 1];
#ifdef HG_HNSW_STAMPS
                        st_acc[8]++;  // adjacency row not in the register cache
#endif
                    }
                    if (lane >= deg) nbr = -1;
                    // the runner-up's adjacency row, one expansion ahead: the list entry that stays first unexpanded
                    int g = -1;
                    if (use_new) {
                        g = fo;
                    } else {
                        for (int base = fo + 1; base < len && g < 0; base += kWave) {
                            const int i = base + lane;
                            const bool un = i < len && !(curA[i].y & kExpanded);
                            const uint64_t m = __ballot(un);
                            if (m) g = base + __ffsll(static_cast<unsigned long long>(m)) - 1;
                        }
                    }
                    if (g >= 0) {
                        const int gn = static_cast<int>(curA[g].y & ~kExpanded);
                        if (gn != pf_node0 && gn != pf_node1) {
                            const int32_t *adj = level == 0 ? a.l0_adj + static_cast<int64_t>(gn) * a.M0
                                                            : a.up_adj + (a.up_off[gn] + (level - 1)) * a.M;
                            const int v = adj[lane < deg ? lane : deg - 1];  // not waited for before the next expansion
                            if (pf_node0 < 0) {
                                pf_node0 = gn;
                                pf_nb0 = v;
                            } else {
                                pf_node1 = gn;
                                pf_nb1 = v;
                            }
                        }
                    }
                    bool fresh = false;
                    if (nbr >= 0 && nbr < a.n) {
                        const uint32_t bit = 1u << (nbr & 31);
                        const uint32_t old = atomicOr(&bits[nbr >> 5], bit);
                        fresh = !(old & bit);
                    }
                    const uint64_t m = __ballot(fresh);
                    const int pos = __popcll(m & ((1ull << lane) - 1ull));
                    if (fresh) cand_id[nb_buf * kMaxDeg + pos] = nbr;
                    ncand = __popcll(m);
                    if (!use_new && lane == 0) curA[fo].y = node | kExpanded;
                }
                if (lane == 0) {
                    sc[8] = any ? 1 : 0;
                    sc[9] = ncand;
                    sc[10] = use_new ? bj : -1;
                    sc[11] = use_new ? -1 : fo;
                }
            };
            // Loads are issued UNCONDITIONALLY from clamped addresses and masked when they are consumed: a load under a
            // lane mask needs its destination zeroed first, and that write waits for every load still in flight (the
            // adjacency row just prefetched, the norms just requested) -- the very latencies this kernel overlaps.
            // Row slots beyond the candidate count are skipped by a scalar (wave-uniform) branch instead.
            const int wave_u = __builtin_amdgcn_readfirstlane(wave);
            auto issue_rows = [&](int buf, int ncnt) {  // first trip: candidates [wave * RB, + RB)
                const int ncnt_u = __builtin_amdgcn_readfirstlane(ncnt);
                const int myj = wave_u * RB + (lane < RB ? lane : 0);
                myid = cand_id[buf * kMaxDeg + (myj < ncnt_u ? myj : 0)];  // slot 0 holds a valid id whenever ncnt > 0
                if (ncnt_u > 0 && a.metric == METRIC_COS) myrn = a.row_norms[myid];
#pragma unroll
                for (int x = 0; x < RB; x++) {
                    if (wave_u * RB + x < ncnt_u) {
                        const int32_t rid = __builtin_amdgcn_readlane(myid, x);
                        const float4 *rp = reinterpret_cast<const float4 *>(a.rows + static_cast<int64_t>(rid) * a.ld);
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) {
                            const int i = cc * kWave + lane;
                            r[x][cc] = rp[i < nvec ? i : nvec - 1];
                        }
                    }
                }
            };
            auto reduce_rows = [&](int buf, int j0, int ncnt) {
                float s[RB];
#pragma unroll
                for (int x = 0; x < RB; x++) {
                    float4 t[NCH];
#pragma unroll
                    for (int cc = 0; cc < NCH; cc++)  // lanes past the row end re-read its last float4: zero them here
                        t[cc] = (cc * kWave + lane < nvec) ? r[x][cc] : make_float4(0.f, 0.f, 0.f, 0.f);
                    s[x] = lane_partial<NCH, L2>(q, t);
                }
#pragma unroll
                for (int x = 0; x < RB; x++) s[x] = wave_sum(s[x]);
                float mine = 0.0f;
#pragma unroll
                for (int x = 0; x < RB; x++) mine = lane == x ? s[x] : mine;
                const int myj = j0 + lane;
                if (lane < RB && myj < ncnt) cand_d[buf * kMaxDeg + myj] = finish_dist(a.metric, mine, qn, myrn) + 0.0f;
            };

            // ---- the level's first candidate
            if (wave == 0) select_next(0, false, 0);
            lds_barrier();
            bool go = sc[8] != 0;
            int c = sc[11], nc = sc[9];
            if (go) issue_rows(0, nc);
            cur_start = c + 1;
            HG_STAMP(0);  // level set-up, first candidate
            while (go) {
                n_hop++;
                n_eval += nc;
                // ---- distances of the expansion in hand: the first trip's rows were issued one expansion ago
                reduce_rows(b, wave * RB, nc);
                for (int j0 = NW * RB + wave_u * RB; j0 < nc; j0 += NW * RB) {  // deg > NW * RB only: further trips
                    const int myj = j0 + (lane < RB ? lane : 0);
                    myid = cand_id[b * kMaxDeg + (myj < nc ? myj : 0)];
                    if (a.metric == METRIC_COS) myrn = a.row_norms[myid];
#pragma unroll
                    for (int x = 0; x < RB; x++) {
                        const int32_t rid = __builtin_amdgcn_readlane(myid, x);  // a slot past nc: candidate 0 again, unused
                        const float4 *rp = reinterpret_cast<const float4 *>(a.rows + static_cast<int64_t>(rid) * a.ld);
#pragma unroll
                        for (int cc = 0; cc < NCH; cc++) {
                            const int i = cc * kWave + lane;
                            r[x][cc] = rp[i < nvec ? i : nvec - 1];
                        }
                    }
                    reduce_rows(b, j0, nc);
                }
                HG_STAMP(1);  // waiting for the rows in flight + reduction
                lds_barrier();
                HG_STAMP(2);  // barrier
                // ---- the next candidate (exact), its neighbours, and their rows on the way BEFORE this merge
                if (wave == 0) select_next(b ^ 1, true, nc);
                lds_barrier();
                HG_STAMP(3);  // choice of the next candidate, adjacency, visited filter
                const bool go_next = sc[8] != 0;
                const int nc_next = sc[9], sel_new = sc[10], sel_old = sc[11];
                if (go_next) issue_rows(b ^ 1, nc_next);
                HG_STAMP(4);  // issue of the next expansion's row loads
                // ---- merge of the expansion in hand (as hnsw_search_kernel: rank-and-scatter, :195-204)
                int p_next = sel_old;  // list index of the next candidate after this merge
                bool merged = false;
                if (nc > 0) {
                    const float cdist = lane < nc ? cand_d[b * kMaxDeg + lane] : 0.0f;
                    const int cbits = __float_as_int(cdist);
                    const bool list_full = len >= ef_l;
                    const float worst0 = list_full ? __uint_as_float(curA[ef_l - 1].x) : 0.0f;
                    const bool surv = lane < nc && (!list_full || cdist < worst0);
                    const uint64_t smask = __ballot(surv);  // identical in every wave
                    if (smask != 0) {
                        int cntA = 0;
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
                                posA[i] = P;
                                if (P == ef_l - 1) sc[4] = static_cast<int32_t>(curA[i].x);
                            }
                        }
                        if (surv) part[wave * kWave + lane] = cntA;
                        lds_barrier();
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
                                const int rr = below + before;
                                admitted = rr < ef_l;
                                P = rr + after_less;
                                if (admitted && P == ef_l - 1) sc[4] = cbits;
                            }
                            if (lane < nc) cand_P[lane] = admitted ? P : -1;
                            const uint64_t am = __ballot(admitted);
                            if (lane == 0) {
                                sc[2] = __popcll(am);
                                sc[5] = 0;
                            }
                        }
                        lds_barrier();
                        const int nadm = sc[2];
                        if (nadm > 0) {
                            const int total = len + nadm;
                            const bool full = total > ef_l;
                            const uint32_t wbits = static_cast<uint32_t>(sc[4]);
                            int ghosts = 0;
                            for (int base = 0; base < len; base += kThreads) {
                                const int i = base + tid;
                                bool gh = false;
                                if (i < len) {
                                    const uint2 e = curA[i];
                                    const int P = posA[i];
                                    if (P < a.cap) curB[P] = e;
                                    gh = full && P >= ef_l && P < a.cap && e.x == wbits;
                                    if (full && P >= a.cap && e.x == wbits && !(e.y & kExpanded)) sc[6] = 1;
                                }
                                ghosts += __popcll(__ballot(gh));
                            }
                            {
                                bool gh = false;
                                if (tid < nc) {
                                    const int P = cand_P[tid];
                                    if (P >= 0) {
                                        // the neighbour chosen as next candidate enters the list already expanded
                                        const uint32_t fl = tid == sel_new ? kExpanded : 0u;
                                        if (P < a.cap)
                                            curB[P] = make_uint2(static_cast<uint32_t>(cbits),
                                                                 static_cast<uint32_t>(cand_id[b * kMaxDeg + tid]) | fl);
                                        gh = full && P >= ef_l && P < a.cap && static_cast<uint32_t>(cbits) == wbits;
                                        if (full && P >= a.cap && static_cast<uint32_t>(cbits) == wbits) sc[6] = 1;
                                    }
                                }
                                if (wave == 0) ghosts += __popcll(__ballot(gh));
                            }
                            if (lane == 0 && ghosts) atomicAdd(&sc[5], ghosts);
                            lds_barrier();
                            const int newlen = full ? ef_l + sc[5] : total;
                            {
                                uint2 *t = curA;
                                curA = curB;
                                curB = t;
                            }
                            len = newlen;
                            merged = true;
                            if (go_next) p_next = sel_new >= 0 ? cand_P[sel_new] : posA[sel_old];
                        }
                    }
                }
                (void)merged;
                HG_STAMP(5);  // merge
                // every entry before the next candidate's position is expanded (it is the best unexpanded one)
                cur_start = p_next + 1;
                b ^= 1;
                nc = nc_next;
                go = go_next;
            }
            // ---- level done
        }
        // ---- results: ascending, take k (:362-370; the distances are reused, not recomputed)
        const int real = len < a.ef ? len : a.ef;
        for (int i = tid; i < a.k; i += kThreads) {
            const bool ok = i < real;
            a.out_ids[static_cast<int64_t>(qi) * a.k + i] = ok ? static_cast<int32_t>(curA[i].y & ~kExpanded) : -1;
            a.out_dist[static_cast<int64_t>(qi) * a.k + i] = ok ? __uint_as_float(curA[i].x) : __uint_as_float(0x7f800000u);
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
