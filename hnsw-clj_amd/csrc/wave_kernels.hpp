// wave_kernels.hpp -- the HNSW traversal of LARGE launches: one wave per query, the candidate list as a main list in LDS plus an
// admission buffer in the wave's registers.
//
// search-layer-ultra (ultra_fast.clj:151-212) on the single-workgroup kernel (kernels.hpp: hnsw_search_kernel) merges every
// expansion's admitted neighbours into ONE sorted list by rank and scatter: every list entry is compared with every admitted
// neighbour, every expansion.  With one wave per query (the shape of every launch that fills the chip) that is the traversal's
// largest compute phase -- at the headline's ef 640 the merge took 2.9 of the 8.2 us an expansion takes per wave, the same again
// went into selecting the next candidate from LDS, and the launch runs no faster with the int8 rejection test than without
// (24 % fewer bytes, same time): it is bound by instructions, not by memory.  The list here is the one solo_kernels.hpp's
// sequencer keeps (round 5): `nearest` = main[0, pm) + buffer[0, pb), an admitted neighbour enters the buffer by a ballot and a
// lane shift (several at once: one pass decides them exactly as the reference's loop would), what leaves `nearest` is a pointer
// move, the next candidate comes out of two register windows, and every 63 admissions the buffer is merged into the main list
// by per-lane binary searches.  Ids, distance bits and both counters are hnsw_search_kernel's (the parity suite runs both).
#pragma once
#include "kernels.hpp"

namespace hg {

__device__ __forceinline__ uint32_t wl_key(uint32_t b) {  // orderable key of distance bits: scalar compares are integer compares
    return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u);
}

// The list of ONE wave's traversal of one layer.  All scalars are wave-uniform.
struct WaveList {
    uint2 *main;    // LDS [cap] (distance bits, node | expanded flag), ascending
    uint2 *img;     // LDS [64]: the buffer's image while it is merged
    int lane, cap, ef_l;
    int lm, pm;                 // main entries; those of them in `nearest`
    float bd;                   // buffer, lane k: distance (+inf behind the entries) ...
    uint32_t bi;                // ... and node | expanded flag
    int nb, pb;                 // buffer entries; those of them in `nearest`
    uint64_t bun;               // bit k: buffer entry k is unexpanded
    float fd;                   // front window of the main list, lane l: entry fbase + l
    uint32_t fi;
    int fbase;
    uint64_t fun;               // bit l: entry fbase + l exists and is unexpanded
    float td;                   // tail window: distance of main entry tbase + l (the entries around pm)
    int tbase;
    float worst;                // of `nearest`, while it holds ef entries
    uint32_t worst_k;
    bool overflow;              // ties with the worst may have been cut off: the query is repeated with a larger list

    __device__ __forceinline__ bool full() const { return pm + pb >= ef_l; }
    __device__ __forceinline__ void load_front(int from) {
        fbase = from;
        const int i = fbase + lane;
        uint2 e = make_uint2(0u, kExpanded);
        if (i < lm) e = main[i];
        fd = __uint_as_float(e.x);
        fi = e.y;
        fun = __builtin_amdgcn_ballot_w64(i < lm && !(e.y & kExpanded));
    }
    __device__ __forceinline__ void load_tail() {
        tbase = pm > kWave ? pm - kWave : 0;
        const int i = tbase + lane;
        td = i < lm ? __uint_as_float(main[i].x) : 0.0f;
    }
    __device__ __forceinline__ void top_worst() {  // the later of main[pm - 1] and buffer[pb - 1]: the worst of `nearest`
        const int im = pm - 1 - tbase, ib = pb - 1;
        const uint32_t wmb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(td), im > 0 ? im : 0));
        const uint32_t wbb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), ib > 0 ? ib : 0));
        const uint32_t km = pm > 0 ? wl_key(wmb) : 0u, kb2 = pb > 0 ? wl_key(wbb) : 0u;
        worst_k = kb2 >= km ? kb2 : km;
        worst = __uint_as_float(kb2 >= km ? wbb : wmb);
    }
    __device__ __forceinline__ void begin_level(int entries, int ef) {
        ef_l = ef;
        lm = entries;
        pm = entries;
        bd = __uint_as_float(0x7f800000u);
        bi = kExpanded;
        nb = pb = 0;
        bun = 0;
        load_front(0);
        load_tail();
        top_worst();
    }
    // Merge the buffer into the main list, in place: buffer entry k goes to k + (main entries <= it), main entry i to i + (buffer
    // entries < it) -- the main entries are the older ones.  Blocks from the tail down to the first position that changes; an
    // entry moves towards the tail by at most 63, into blocks already read.
    __device__ __forceinline__ void compact() {
        if (nb == 0) return;
        if (lane < nb) img[lane] = make_uint2(__float_as_uint(bd), bi);
        const int first_un = fun ? fbase + __ffsll(static_cast<unsigned long long>(fun)) - 1 : fbase + kWave;
        int lo = 0, hi = lm;  // upper bound of bd in the main list
        for (int span = lm; span > 0; span >>= 1) {
            const int mid = (lo + hi) >> 1;
            const float v = __uint_as_float(main[mid < lm ? mid : lm - 1].x);
            const bool act = lo < hi;
            const bool go = act && v <= bd;
            lo = go ? mid + 1 : lo;
            hi = (act && !go) ? mid : hi;
        }
        const int Pk = lane + lo;
        const int minP = __builtin_amdgcn_readlane(Pk, 0);
        const int total = lm + nb;
        const bool isfull = full();
        for (int base = ((lm - 1) / kWave) * kWave; base >= 0 && base + kWave > minP; base -= kWave) {
            const int i = base + lane;
            const bool valid_i = i < lm;
            uint2 e = make_uint2(0u, 0u);
            if (valid_i) e = main[i];
            const float de = __uint_as_float(e.x);
            int l2 = 0, h2 = nb;  // lower bound of de in the buffer
#pragma unroll
            for (int it = 0; it < 7; it++) {
                const int mid = (l2 + h2) >> 1;
                const float v = __uint_as_float(img[mid < nb ? mid : nb - 1].x);
                const bool act = l2 < h2;
                const bool go = act && v < de;
                l2 = go ? mid + 1 : l2;
                h2 = (act && !go) ? mid : h2;
            }
            const int Pe = i + l2;
            if (valid_i && Pe < cap && Pe != i) main[Pe] = e;
        }
        if (lane < nb && Pk < cap) main[Pk] = make_uint2(__float_as_uint(bd), bi);
        if (isfull) {
            // behind `nearest` only what ties its worst can still be expanded (:175-178): that run stays (hnsw_search_kernel's
            // ghosts), as far as the list has room
            const uint32_t wbits = main[ef_l - 1].x;
            int phys = (total < cap ? total : cap) - ef_l;
            const bool more = phys > kWave || total > cap;
            phys = phys > kWave ? kWave : phys;
            const bool tie = lane < phys && main[ef_l + (lane < phys ? lane : 0)].x == wbits;
            const uint64_t nt = ~__builtin_amdgcn_ballot_w64(tie);
            const int run = nt ? __ffsll(static_cast<unsigned long long>(nt)) - 1 : kWave;
            if (run == phys && more) overflow = true;
            lm = ef_l + run;
            pm = ef_l;
        } else {
            lm = total;
            pm = total;
        }
        nb = 0;
        pb = 0;
        bun = 0;
        bi = kExpanded;
        bd = __uint_as_float(0x7f800000u);
        load_front(first_un < minP ? first_un : minP);
        load_tail();
    }
    // The next candidate (:170-178): the smaller of the first unexpanded entries of the two sequences (a tie: the main list's,
    // it is the older one); false when none is left that is <= the worst of `nearest`.
    __device__ __forceinline__ bool pop(uint32_t &node) {
        while (fun == 0 && fbase + kWave < lm) load_front(fbase + kWave);
        if ((fun | bun) == 0) return false;
        const int lf = fun ? __ffsll(static_cast<unsigned long long>(fun)) - 1 : 0;
        const int kb = bun ? __ffsll(static_cast<unsigned long long>(bun)) - 1 : 0;
        const uint32_t dmk = fun ? wl_key(static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(fd), lf))) : 0xffffffffu;
        const uint32_t dbk = bun ? wl_key(static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), kb))) : 0xffffffffu;
        const uint32_t nm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(fi), lf));
        const uint32_t nbf = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bi), kb));
        const bool take_main = fun != 0 && dmk <= dbk;
        if (full() && (take_main ? dmk : dbk) > worst_k) return false;
        node = take_main ? nm : nbf;
        if (take_main) {
            if (lane == lf) {
                fi |= kExpanded;
                main[fbase + lf].y = fi;
            }
            fun &= fun - 1;
        } else {
            if (lane == kb) bi |= kExpanded;
            bun &= bun - 1;
        }
        return true;
    }
    // Admission (:195-204) of the fresh neighbours in `smask` (lane j: distance `dist`, node `id`; adjacency order; already
    // known to be < the worst the expansion found, or `nearest` not full).
    __device__ __forceinline__ void admit(uint64_t smask, float dist, uint32_t id) {
        if (smask == 0) return;
        if (nb + __popcll(smask) > kWave - 1) {  // (63 admissions at most between two merges)
            compact();
            top_worst();
        }
        if (smask & (smask - 1)) {
            // two or more: one pass in adjacency order decides every admission exactly as the sequential loop would -- a survivor
            // is admitted iff fewer than ef of {`nearest` as the expansion found it, the survivors before it} are <= it -- and
            // collects the merge counts; the admitted ones enter the buffer together (a scatter through LDS), and what they push
            // out of `nearest` (:203-204) is the nev largest of its two tails (a merge-path split, all lanes at once)
            int before = 0, arank = 0, cball = 0, shb = 0;
            uint64_t am = 0;
            const bool tvalid = tbase + lane < pm;
#pragma unroll 1
            for (uint64_t mm = smask; mm; mm &= mm - 1) {
                const int sv = __ffsll(static_cast<unsigned long long>(mm)) - 1;
                const float ds = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dist), sv));
                const int cb = __popcll(__builtin_amdgcn_ballot_w64(lane < pb && bd <= ds));
                const int cw = __popcll(__builtin_amdgcn_ballot_w64(tvalid && td <= ds));
                const int bs = __builtin_amdgcn_readlane(before, sv);
                // (smaller than every main entry the tail window shows, and the window does not start at 0: at most ef - 64
                // main + buffer entries are <= it, and fewer than 64 survivors precede it)
                const int tm = (cw == 0 && tbase > 0) ? 0 : tbase + cw;
                const bool adm = tm + cb + bs < ef_l;
                before += (lane > sv && ds <= dist) ? 1 : 0;
                if (adm) {
                    am |= 1ull << sv;
                    shb += (ds < bd) ? 1 : 0;
                    arank += (ds < dist || (ds == dist && sv < lane)) ? 1 : 0;
                    cball = lane == sv ? cb : cball;
                }
            }
            const int nadm = __popcll(am);
            if (nadm == 0) return;
            const bool isadm = (am >> lane) & 1ull;
            if (lane < nb) img[lane + shb] = make_uint2(__float_as_uint(bd), bi);
            if (isadm) img[cball + arank] = make_uint2(__float_as_uint(dist), id);
            nb += nadm;
            {
                const uint2 e = img[lane];
                bd = lane < nb ? __uint_as_float(e.x) : __uint_as_float(0x7f800000u);
                bi = lane < nb ? e.y : kExpanded;
            }
            bun = __builtin_amdgcn_ballot_w64(lane < nb && !(bi & kExpanded));
            const int pbn = pb + nadm;
            const int nev = pm + pbn > ef_l ? pm + pbn - ef_l : 0;
            if (nev) {
                // lane e: e entries leave the main list's tail, nev - e the buffer's.  Right iff what stays is before what
                // leaves: main entries are the older ones (a tie: the buffer entry leaves)
                const int e = lane, eb = nev - lane;
                const bool feas = e <= nev && e <= pm && eb <= pbn;
                const int i_mk = pm - e - 1, i_me = pm - e, i_bk = pbn - eb - 1, i_be = pbn - eb;
                const float m_keep = __uint_as_float(main[i_mk > 0 ? i_mk : 0].x);
                const float m_ev = __uint_as_float(main[(feas && e > 0) ? i_me : 0].x);
                const float b_keep = __uint_as_float(img[(feas && i_bk > 0) ? i_bk : 0].x);
                const float b_ev = __uint_as_float(img[(feas && eb > 0) ? i_be : 0].x);
                const bool ca = i_mk < 0 || eb == 0 || m_keep <= b_ev;
                const bool cb2 = i_bk < 0 || e == 0 || b_keep < m_ev;
                const uint64_t okm = __builtin_amdgcn_ballot_w64(feas && ca && cb2);
                int em = okm ? __ffsll(static_cast<unsigned long long>(okm)) - 1 : -1;
                if (em < 0) {  // (cannot happen for comparable distances; NaNs: one at a time, the sequential rule)
                    em = 0;
                    int pmm = pm, pbb = pbn;
                    for (int t = 0; t < nev; t++) {
                        const uint32_t wmb = pmm > 0 ? main[pmm - 1].x : 0u;
                        const uint32_t wbb = pbb > 0 ? img[pbb - 1].x : 0u;
                        if (pmm > 0 && (pbb == 0 || wl_key(wmb) > wl_key(wbb))) {
                            pmm--;
                            em++;
                        } else {
                            pbb--;
                        }
                    }
                }
                pm -= em;
                pb = pbn - (nev - em);
            } else {
                pb = pbn;
            }
            top_worst();
            return;
        }
        // ONE survivor: straight into its place
        const int j = __ffsll(static_cast<unsigned long long>(smask)) - 1;
        const uint32_t djb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(dist), j));
        const uint32_t idj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id), j));
        if (full() && wl_key(djb) >= worst_k) return;  // (:195-198, a strict <)
        const float dj = __uint_as_float(djb);
        const int r0 = __popcll(__builtin_amdgcn_ballot_w64(bd <= dj));  // behind the buffer entries <= it (lanes >= nb hold +inf)
        const int r = r0 < nb ? r0 : nb;                                 // (an infinite distance: behind everything)
        const float sd = __uint_as_float(wave_shr1(__float_as_uint(bd)));
        const uint32_t si = wave_shr1(bi);
        bd = lane > r ? sd : (lane == r ? dj : bd);
        bi = lane > r ? si : (lane == r ? idj : bi);
        const uint64_t lowm = (1ull << r) - 1ull;
        bun = (bun & lowm) | ((bun & ~lowm) << 1) | (1ull << r);
        nb++;
        pb++;
        {  // (:203-204) if `nearest` now holds ef + 1, its worst leaves it: the later of the two tails (a tie: the buffer's)
            const int over = pm + pb > ef_l ? 1 : 0;
            const int im = pm - 1 - tbase;
            const uint32_t wmb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(td), im > 0 ? im : 0));
            const uint32_t wbb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), pb - 1));
            const int evm = (over && pm > 0 && wl_key(wmb) > wl_key(wbb)) ? 1 : 0;
            pm -= evm;
            pb -= over - evm;
        }
        top_worst();
    }
    __device__ __forceinline__ int end_level() {
        compact();
        return lm;
    }
};

// One wave per query (workgroups of one wave; persistent: workgroup b serves queries b, b + gridDim.x, ...).  VG = visited set in
// HBM stamps.  Build launches too (q_rows: the query is a base row, the best entry of every upper level <= the node's level is
// emitted for the linker, all ef candidates of layer 0 are the result); the repeat pass stays with hnsw_search_kernel.
template <int NCH, int RB, bool L2, bool VG>
__global__ __launch_bounds__(kWave) void hnsw_wave_kernel(HnswArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    // LDS: [main: cap x 8] [img: 64 x 8] [cand_id | cand_d: kMaxDeg each] [bits: nwords]
    uint2 *const curA = reinterpret_cast<uint2 *>(smem);
    uint2 *const img = curA + a.cap;
    int32_t *const cand_id = reinterpret_cast<int32_t *>(img + kWave);
    float *const cand_d = reinterpret_cast<float *>(cand_id + kMaxDeg);
    uint32_t *const bits = reinterpret_cast<uint32_t *>(cand_d + kMaxDeg);
    uint32_t *stamps = VG ? a.vis + static_cast<int64_t>(blockIdx.x) * a.vis_stride : nullptr;
    const int lane = threadIdx.x;
    const int nvec = static_cast<int>(a.ld / 4);
    uint32_t gen = a.gen_base;
    for (int qi = blockIdx.x; qi < a.nq; qi += gridDim.x) {
        const float *qptr = a.q_rows ? a.rows + static_cast<int64_t>(a.q_rows[qi]) * a.ld : a.Q + static_cast<int64_t>(qi) * a.qld;
        float4 q[NCH];
        load_query<NCH>(q, qptr, a.dim, lane);
        const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
        QueryCode<NCH> qc;  // the query's side of the rejection test
        if (a.qrows != nullptr) encode_query<NCH>(q, qc);
        const int qlevel = a.q_levels ? a.q_levels[qi] : -1;
        int64_t n_eval = 0, n_hop = 0, n_exact = 0;
        int len = 0;
        bool over = false;
        {  // seed: the entry point (ultra_fast.clj:358-359)
            float4 r[NCH];
            load_row<NCH>(r, a.rows + static_cast<int64_t>(a.entry) * a.ld, nvec, lane, true);
            const float s = wave_sum(lane_partial<NCH, L2>(q, r));
            const float d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[a.entry] : 0.0f);
            if (lane == 0) curA[0] = make_uint2(__float_as_uint(d + 0.0f), static_cast<uint32_t>(a.entry));
            len = 1;
            n_eval = 1;
        }
        const int top_l = (a.ref_start && qlevel >= 0 && qlevel < a.max_level) ? qlevel : a.max_level;
        for (int level = top_l; level >= 0; level--) {
            const int ef_l = level > 0 ? 1 : a.ef;
            // fresh visited set per layer (:156); entries carried from the level above are marked
            if (VG) {
                gen++;
            } else {
                for (int w = lane; w < a.nwords; w += kWave) bits[w] = 0;
            }
            if (len > ef_l) len = ef_l;
            // the reference re-evaluates its entry points at every layer (:162-167); the values are reused here, but counted so
            // that `evals` is the reference's number of distance calls
            if (level != top_l) n_eval += len;
            for (int i = lane; i < len; i += kWave) {
                uint2 e = curA[i];
                e.y &= ~kExpanded;
                curA[i] = e;
                if (VG) atomicExch(&stamps[e.y], gen);
                else atomicOr(&bits[e.y >> 5], 1u << (e.y & 31));
            }
            const int deg = level == 0 ? a.M0 : a.M;
            WaveList L;
            L.main = curA;
            L.img = img;
            L.lane = lane;
            L.cap = a.cap;
            L.overflow = false;
            L.begin_level(len, ef_l);
            for (;;) {
                uint32_t node;
                if (!L.pop(node)) break;
                const int32_t *adj = level == 0 ? a.l0_adj + static_cast<int64_t>(node) * a.M0
                                                : a.up_adj + (a.up_off[node] + (level - 1)) * a.M;
                const int nbr = lane < deg ? adj[lane] : -1;
                bool fresh = false;
                if (nbr >= 0 && nbr < a.n) {
                    if (VG) {
                        fresh = atomicExch(&stamps[nbr], gen) != gen;
                    } else {
                        const uint32_t bit = 1u << (nbr & 31);
                        const uint32_t old = atomicOr(&bits[nbr >> 5], bit);
                        fresh = !(old & bit);
                    }
                }
                const uint64_t fm = __builtin_amdgcn_ballot_w64(fresh);
                n_hop++;
                if (fm == 0) continue;
                const int nc = __popcll(fm);
                if (fresh) cand_id[__popcll(fm & ((1ull << lane) - 1ull))] = nbr;  // adjacency order
                n_eval += nc;
                const bool list_full = L.full();
                const float worst0 = list_full ? L.worst : 0.0f;
                // ---- rejection test on the int8 rows (kernels.hpp: quantize_rows_kernel): with a full list a neighbour whose
                //      LOWER BOUND is already >= the worst cannot be admitted (:195-198) -- it gets +inf and no f32 fetch
                uint64_t needmask = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
                if (a.qrows != nullptr && list_full) {
                    uint64_t wmask = 0;
                    const int own = wave_sum8_row(lane);  // the row of a step whose total this lane receives
                    for (int j0 = 0; j0 < nc; j0 += 8) {
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
                        const uint64_t m8 = ((__builtin_amdgcn_ballot_w64(need) & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
                        wmask |= m8 << j0;
                    }
                    needmask = wmask;
                }
                // ---- gather rows + distances of the neighbours that need them, RB rows per trip (lane b takes the candidate of
                //      rank t0 + b and fetches its norm up front)
                const int nneed = __popcll(needmask);
                n_exact += nneed;
                const bool isset = (needmask >> lane) & 1ull;
                const int rank = __popcll(needmask & ((1ull << lane) - 1ull));
                for (int t0 = 0; t0 < nneed; t0 += RB) {
                    float4 r[RB][NCH];
                    int myj = -1;
#pragma unroll
                    for (int b = 0; b < RB; b++) {
                        const uint64_t hit = __builtin_amdgcn_ballot_w64(isset && rank == t0 + b);
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
                // ---- admission: lane j = the j-th fresh neighbour
                const float cdist = lane < nc ? cand_d[lane] : 0.0f;
                const uint32_t cid = lane < nc ? static_cast<uint32_t>(cand_id[lane]) : 0u;
                L.admit(__builtin_amdgcn_ballot_w64(lane < nc && (!list_full || cdist < worst0)), cdist, cid);
            }
            len = L.end_level();
            over = over || L.overflow;
            if (a.q_rows && level > 0 && level <= qlevel && lane == 0) {  // build: the nearest node of this layer, for the linker
                a.up_out_ids[static_cast<int64_t>(qi) * a.up_stride + (level - 1)] = len > 0 ? static_cast<int32_t>(curA[0].y & ~kExpanded) : -1;
                a.up_out_dist[static_cast<int64_t>(qi) * a.up_stride + (level - 1)] = len > 0 ? __uint_as_float(curA[0].x) : 0.0f;
            }
        }
        // ---- results: ascending, take k (:362-370; the distances are reused, not recomputed)
        const int real = len < a.ef ? len : a.ef;
        for (int i = lane; i < a.k; i += kWave) {
            const bool ok = i < real;
            a.out_ids[static_cast<int64_t>(qi) * a.k + i] = ok ? static_cast<int32_t>(curA[i].y & ~kExpanded) : -1;
            a.out_dist[static_cast<int64_t>(qi) * a.k + i] = ok ? __uint_as_float(curA[i].x) : __uint_as_float(0x7f800000u);
        }
        if (lane == 0) {
            if (a.again && over) a.again[atomicAdd(a.again_cnt, 1)] = qi;
            if (a.stats) {
                a.stats[2 * static_cast<int64_t>(qi)] = n_eval;
                a.stats[2 * static_cast<int64_t>(qi) + 1] = n_hop;
            }
            if (a.rej_stats) {
                atomicAdd(a.rej_stats, static_cast<unsigned long long>(n_exact));
                atomicAdd(a.rej_stats + 1, static_cast<unsigned long long>(n_eval));
            }
        }
    }
    if (a.host_flag) {
        __threadfence_system();  // this thread's results (host memory) are visible system-wide ...
        if (lane == 0) {
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
