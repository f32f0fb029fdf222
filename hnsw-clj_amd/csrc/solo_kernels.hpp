// solo_kernels.hpp -- ONE HNSW query over several CUs (small launches: a handful of queries on an otherwise idle chip).
//
// search-layer-ultra (ultra_fast.clj:151-212) is a sequential loop: pop the nearest unexpanded candidate, evaluate its
// unvisited neighbours, admit them one by one.  A workgroup that does that alone pays two dependent memory round trips per
// expansion (adjacency row, then the neighbours' rows) through one CU's path to memory: 2.8 us per expansion, 1.9 ms per
// query at ef 640 ("One query, one CU").  But a distance is a property of (query, row), not of the
// traversal's state -- whoever computes it, with lane_partial + the butterfly + finish_dist, gets the same bits.  So the
// loop is split by what is sequential and what is not:
//
//   * the OWNER workgroup keeps the reference's order.  Its wave 0 (the sequencer) holds the candidate list, the visited
//     set and the admission rule (:170-204) in LDS and does nothing else: per expansion it looks the popped node up in an
//     LDS cache of (node -> neighbour ids, neighbour distances), filters the visited ones and merges -- no memory access at
//     all when the cache has everything.  What the cache does not have it fetches (one more look at the published words)
//     or computes itself: results never depend on anybody else's progress.
//   * its waves 1-2 (the fetchers) look AHEAD: the best T unexpanded list entries will be popped soon, so they claim each
//     of them for evaluation (claim word, ring entry), poll the words the helpers publish for it and copy ids + distances
//     into the LDS cache -- all off the sequencer's critical path.  They also publish tau, the distance of the T-th
//     unexpanded entry.
//   * G HELPER workgroups per query (other CUs, same XCD first) follow the ring; for every entry each of them evaluates ITS
//     slice of the node's neighbours and publishes (tag | distance bits) words in a node-keyed table.  And they CHASE: a
//     neighbour found closer than tau will be popped before anything the owner has asked for, so the helper that found it
//     claims it and appends it to the ring itself -- the descent does not wait for the round trip through the owner.
//
// Every published word carries its own tag (launch number, node): a word is either the right one or ignored; nobody ever
// waits for a helper (helpers leave on the owner's done word or after 20 ms), so ids, distance bits and both traversal
// counters are the single-workgroup kernel's by construction -- the whole parity suite runs through this path.
#pragma once
#include "kernels.hpp"

namespace hg {

// LDS accesses that another wave of the workgroup must see in program order: volatile, and EXPLICITLY in the LDS address space
// (a volatile access through a generic pointer is a flat instruction with a wait behind it: the compiler does not infer the
// address space of volatile accesses)
#define HG_LDS __attribute__((address_space(3)))
template <class T>
__device__ __forceinline__ volatile HG_LDS T *ldsv(T *p) {
    return (volatile HG_LDS T *)p;
}
template <class T>
__device__ __forceinline__ HG_LDS T *ldsp(T *p) {
    return (HG_LDS T *)p;
}

constexpr int kSoloRing = 256;                      // ring entries (64-bit) per query
constexpr int kSoloMailWords = 16 + 2 * kSoloRing;  // 32-bit words per query: [0..1] ring head (launch << 32 | count), [2] done, [4..5] tau (launch << 32 | float bits), [16..] ring
constexpr int kSoloMaxQueries = 128;                 // queries per launch served this way (hnsw.hip: kPfMaxQueries)
constexpr int kSoloSlots = 64;                      // LDS cache slots of the owner (one ballot finds a node)
constexpr int kSoloFetchers = 2;                    // owner waves 1..kSoloFetchers; each manages kSoloSlots / kSoloFetchers slots
constexpr uint32_t kSoloFree = 0xffffffffu, kSoloBusy = 0xfffffffeu;

__device__ __forceinline__ uint32_t solo_tag(uint32_t seq, uint32_t node) { return ((seq & 0x3fffu) << 18) | node; }  // node < 2^18
// slot of a node in the per-query tables: the node itself when the table has a slot per row, a multiplicative hash otherwise
__device__ __forceinline__ uint32_t solo_slot(uint32_t node, int log2s, int64_t n) {
    return (static_cast<int64_t>(1) << log2s) >= n ? node : (node * 2654435761u) >> (32 - log2s);
}
__device__ __forceinline__ uint32_t solo_half(uint32_t node) { return ((node * 2654435761u) >> 9) & 1u; }  // which fetcher looks after a node

// Workgroup b sits on XCD b % 8 (observed; for speed only): query t and its helpers share XCD t % 8:
//     b = (t % 8) + 8 * ((t / 8) * (1 + G) + role), role 0 = the owner, 1..G = helpers
__device__ __forceinline__ void solo_place(const HnswArgs &a, int &role, int &query) {
    const int u = blockIdx.x >> 3, team = 1 + a.pf_groups;
    role = u % team;
    query = (u / team) * 8 + (blockIdx.x & 7);
}

// ---- a helper workgroup -------------------------------------------------------------------------------------------------
// Follows the query's ring until the owner is done (or 20 ms have passed: never hang).  Entry e is taken by wave e % 4 of
// EVERY helper; helper `role` evaluates neighbours [lo, hi) of the entry's node, RB rows per trip.
template <int NCH, int RB, bool L2>
__device__ __forceinline__ void solo_helper(const HnswArgs &a, uint32_t *mail, int role, int query) {
    constexpr int NW = 4;
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nvec = static_cast<int>(a.ld / 4);
    float4 hq[NCH];  // the query and its norm, exactly as the owner holds them
    load_query<NCH>(hq, a.Q + static_cast<int64_t>(query) * a.qld, a.dim, lane);
    const float hqn = a.metric == METRIC_COS ? query_norm<NCH>(hq) : 0.0f;
    const unsigned long long t_begin = wall_clock64();
    auto timed_out = [&]() { return wall_clock64() - t_begin > 2000000ull; };
    const unsigned long long *ring = reinterpret_cast<const unsigned long long *>(mail + 16);
    unsigned long long *ringw = reinterpret_cast<unsigned long long *>(mail + 16);
    unsigned long long *head = reinterpret_cast<unsigned long long *>(mail);
    const int64_t S = static_cast<int64_t>(1) << a.solo_log2s;
    unsigned long long *rec = a.solo_rec + static_cast<int64_t>(query) * S * a.M0;
    uint32_t *claim = a.solo_claim + static_cast<int64_t>(query) * S;
    const unsigned long long seq_hi = static_cast<unsigned long long>(a.pf_seq) << 32;
    if (a.solo_chase) {
        // the ring count of THIS launch starts at (launch, 0): launch numbers only grow, so the maximum is a reset that
        // any number of producers may apply in any order.  Returned and waited for before this wave's first append.
        unsigned long long old = 0;
        if (lane == 0) old = atomicMax(head, seq_hi);
        asm volatile("" ::"v"(old));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    unsigned long long n_chase = 0, n_entries = 0;
    const int per = (a.M0 + a.pf_groups - 1) / a.pf_groups, lo = (role - 1) * per;
    const int hi = lo + per < a.M0 ? lo + per : a.M0;
    uint32_t e = 0;
    for (;;) {
        const bool done = coherent_load(mail + 2) == a.pf_seq;
        const unsigned long long v = coherent_load(ring + (e % kSoloRing));
        const unsigned long long tw = coherent_load(reinterpret_cast<const unsigned long long *>(mail + 4));
        const float tau = __uint_as_float(static_cast<uint32_t>(tw >> 32) == a.pf_seq ? static_cast<uint32_t>(tw) : 0xff800000u);  // none yet: -inf
        const uint32_t tag = static_cast<uint32_t>(v >> 32), node = static_cast<uint32_t>(v);
        // an entry of a LATER lap than the one I wait for: the producers have lapped me, take up what is there
        // (an earlier lap: my entry has not been written yet)
        const uint32_t ahead = ((tag & 0xff) - ((e / kSoloRing) & 0xff)) & 0xff;
        if (done) break;  // (whatever the ring still holds: nobody will read what it leads to)
        if ((tag >> 8) == a.pf_seq && ahead < 128) {
            const uint32_t ee = e + ahead * kSoloRing;  // the entry in hand
            e = ee + 1;
            if (node < static_cast<uint32_t>(a.n) && static_cast<int>(ee % NW) == wave && lo < hi) {
                const int nb = a.l0_adj[static_cast<int64_t>(node) * a.M0 + (lane < a.M0 ? lane : a.M0 - 1)];
                const unsigned long long tagw = static_cast<unsigned long long>(solo_tag(a.pf_seq, node)) << 32;
                n_entries++;
                unsigned long long *res = rec + static_cast<int64_t>(solo_slot(node, a.solo_log2s, a.n)) * a.M0;
                for (int j0 = lo; j0 < hi; j0 += RB) {
                    float4 r[RB][NCH];
                    int32_t myid = -1;  // lane b < RB owns row b of the trip: its neighbour id and that row's norm
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
                    const bool own = lane < RB && myid >= 0;
                    const float dv = finish_dist(a.metric, mine, hqn, myrn) + 0.0f;
                    if (own) coherent_store(res + (j0 + lane), tagw | __float_as_uint(dv));
                    if (a.solo_chase) {
                        // closer than the T-th entry the owner has in its window: it will be popped before that one --
                        // claim it (one exchange per node and launch wins) and append it to the ring at once
                        bool want = own && dv < tau;
                        const uint32_t mytag = solo_tag(a.pf_seq, static_cast<uint32_t>(myid));
                        // (a maximum, not an exchange: the first claim of a slot in this launch wins -- launch numbers grow, the
                        // tables are zeroed where the 14 bits start over -- and a helper leaves a slot that already carries this
                        // launch's number alone, be it this node's or one that shares the slot: two such nodes would otherwise
                        // take the slot from each other, and append each other, for ever)
                        if (want) want = (atomicMax(claim + solo_slot(static_cast<uint32_t>(myid), a.solo_log2s, a.n), mytag) >> 18) != (mytag >> 18);
                        const uint64_t pm = __builtin_amdgcn_ballot_w64(want);
                        n_chase += __popcll(pm);
                        if (pm) {
                            const int leader = __ffsll(static_cast<unsigned long long>(pm)) - 1;
                            unsigned long long base = 0;
                            if (lane == leader) base = atomicAdd(head, static_cast<unsigned long long>(__popcll(pm)));
                            const uint32_t blo = __builtin_amdgcn_readlane(static_cast<uint32_t>(base), leader);
                            const uint32_t bhi = __builtin_amdgcn_readlane(static_cast<uint32_t>(base >> 32), leader);
                            if (want && bhi == a.pf_seq) {
                                const uint32_t pe = blo + __popcll(pm & ((1ull << lane) - 1ull));
                                const unsigned long long tagv = (static_cast<unsigned long long>(a.pf_seq) << 8) | ((pe / kSoloRing) & 0xff);
                                coherent_store(ringw + (pe % kSoloRing), (tagv << 32) | static_cast<uint32_t>(myid));
                            }
                        }
                    }
                }
            }
            continue;  // look at the next entry right away
        }
        if (timed_out()) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (a.dbg && lane == 0) {  // diagnostics: entries this wave evaluated its slice of, nodes it appended to the ring itself
        if (role == 1) atomicAdd(a.dbg + 59, n_entries);
        atomicAdd(a.dbg + 60, n_chase);
    }
}

// LDS of the owner workgroup
struct SoloLds {
    uint2 *list;                  // [cap] (distance bits, node | expanded flag), ascending: the MAIN list
    unsigned long long *c_valid;  // [kSoloSlots] neighbour slots of the cached node whose distance is in c_d
    unsigned long long *g_need;   // [1] gather hand-over: the neighbour slots the assistant wave computes
    int32_t *g_ids;               // [64] ... their node ids (lane = neighbour slot)
    float *g_d;                   // [64] ... and the distances it returns
    uint2 *win;                   // [kSoloFetchers][64] scratch of the fetchers' window compaction
    uint2 *bmir;                  // [64] mirror of the sequencer's admission buffer (the fetchers and the compaction read it)
    uint32_t *bits;               // [nwords] visited set
    int32_t *sc;                  // [32] scalars: [0] first unexpanded index of the main list [1] its length [2] fin [3] buffer entries [6] ghost overflow [8] gather requests [9] gather answers [13] level done
    uint32_t *c_node;             // [kSoloSlots] node a slot holds (kSoloFree / kSoloBusy)
    uint32_t *c_done;             // [kSoloSlots] node the sequencer has consumed from the slot (written by the sequencer only)
    uint32_t *c_state;            // [kSoloSlots] 1 = c_id holds the node's adjacency row
    int32_t *c_id;                // [kSoloSlots][kMaxDeg]
    float *c_d;                   // [kSoloSlots][kMaxDeg]
};

__host__ __device__ inline size_t solo_lds_bytes(int cap, int nwords) {
    return sizeof(uint2) * cap + 8 * kSoloSlots + 8 + 8 * 64 + sizeof(uint2) * (kSoloFetchers + 1) * 64 + 4 * static_cast<size_t>(nwords) + 4 * 32 +
           4 * 3 * kSoloSlots + 2 * 4 * static_cast<size_t>(kSoloSlots) * kMaxDeg + 16;
}

__device__ __forceinline__ SoloLds solo_carve(unsigned char *smem, int cap, int nwords) {
    SoloLds s;
    s.list = reinterpret_cast<uint2 *>(smem);
    s.c_valid = reinterpret_cast<unsigned long long *>(s.list + cap);
    s.g_need = s.c_valid + kSoloSlots;
    s.g_ids = reinterpret_cast<int32_t *>(s.g_need + 1);
    s.g_d = reinterpret_cast<float *>(s.g_ids + 64);
    s.win = reinterpret_cast<uint2 *>(s.g_d + 64);
    s.bmir = s.win + kSoloFetchers * 64;
    s.bits = reinterpret_cast<uint32_t *>(s.bmir + 64);
    s.sc = reinterpret_cast<int32_t *>(s.bits + nwords);
    s.c_node = reinterpret_cast<uint32_t *>(s.sc + 32);
    s.c_done = s.c_node + kSoloSlots;
    s.c_state = s.c_done + kSoloSlots;
    s.c_id = reinterpret_cast<int32_t *>(s.c_state + kSoloSlots);
    s.c_d = reinterpret_cast<float *>(s.c_id + kSoloSlots * kMaxDeg);
    return s;
}

// ---- a fetcher wave of the owner (level 0 only) ---------------------------------------------------------------------------
// Both look at the front of the main list and at the admission buffer's mirror; a node is looked after by the fetcher its hash
// names.  Lane l < 32 manages cache slot f * 32 + l (node, valid mask and state also in its registers: one writer per slot).
__device__ __forceinline__ void solo_fetcher(const HnswArgs &a, const SoloLds &L, uint32_t *mail, int query, int f,
                                             unsigned long long *cnt_push, unsigned long long *cnt_steal) {
    constexpr int kMine = kSoloSlots / kSoloFetchers;
    const int lane = threadIdx.x & (kWave - 1);
    volatile HG_LDS int32_t *sc = ldsv(L.sc);
    volatile HG_LDS uint32_t *c_node = ldsv(L.c_node), *c_done = ldsv(L.c_done), *c_state = ldsv(L.c_state);
    volatile HG_LDS unsigned long long *c_valid = ldsv(L.c_valid);
    volatile HG_LDS unsigned long long *list = ldsv(reinterpret_cast<unsigned long long *>(L.list));  // whole entries
    volatile HG_LDS unsigned long long *bmir = ldsv(reinterpret_cast<unsigned long long *>(L.bmir));
    volatile HG_LDS unsigned long long *win = ldsv(reinterpret_cast<unsigned long long *>(L.win + f * 64));
    HG_LDS int32_t *c_id = ldsp(L.c_id);
    HG_LDS float *c_d = ldsp(L.c_d);
    const int slot = f * kMine + lane;  // meaningful for lane < kMine
    const bool mgr = lane < kMine;
    uint32_t s_node = kSoloFree, s_state = 0;
    unsigned long long s_valid = 0;
    const int deg = a.M0;
    const unsigned long long degmask = deg >= 64 ? ~0ull : ((1ull << deg) - 1ull);
    const int64_t S = static_cast<int64_t>(1) << a.solo_log2s;
    const unsigned long long *rec = a.solo_rec + static_cast<int64_t>(query) * S * a.M0;
    uint32_t *claim = a.solo_claim + static_cast<int64_t>(query) * S;
    unsigned long long *head = reinterpret_cast<unsigned long long *>(mail);
    unsigned long long *ringw = reinterpret_cast<unsigned long long *>(mail + 16);
    const int T = a.pf_hints;
    uint32_t tau_sent = 0x7fc00000u;  // (a NaN: nothing sent yet)
    unsigned long long n_push = 0, n_steal = 0;
    {
        unsigned long long old = 0;
        if (lane == 0) old = atomicMax(head, static_cast<unsigned long long>(a.pf_seq) << 32);
        asm volatile("" ::"v"(old));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    while (sc[2] == 0) {
        bool busy = false;
        // ---- the window: the first T unexpanded entries of the sequencer's two sequences (the front of the main list, the
        //      admission buffer's mirror), merged.  The sequencer may be in the middle of an update: an entry can show twice
        //      or not at all -- these are hints.
        uint2 wm = make_uint2(0u, kSoloFree), wb = make_uint2(0u, kSoloFree);  // lane t: the t-th unexpanded entry of either
        int cm = 0, cb = 0;
        {
            const int cs = sc[0], ln = sc[1];
            for (int blk = 0; blk < 2 && cm < T; blk++) {
                const int i = cs + blk * 64 + lane;
                uint2 en = make_uint2(0u, kExpanded);
                if (i < ln) {
                    const unsigned long long ev = list[i];
                    en.x = static_cast<uint32_t>(ev);
                    en.y = static_cast<uint32_t>(ev >> 32);
                }
                const bool un = !(en.y & kExpanded);
                const uint64_t um = __builtin_amdgcn_ballot_w64(un);
                const int pos = cm + __popcll(um & ((1ull << lane) - 1ull));
                if (un && pos < 64) win[pos] = (static_cast<unsigned long long>(en.y) << 32) | en.x;
                cm += __popcll(um);
            }
            cm = cm < T ? cm : T;
            if (lane < cm) {
                const unsigned long long wv = win[lane];
                wm = make_uint2(static_cast<uint32_t>(wv), static_cast<uint32_t>(wv >> 32));
            }
            const int lb = sc[3];
            uint2 en = make_uint2(0u, kExpanded);
            if (lane < lb) {
                const unsigned long long ev = bmir[lane];
                en.x = static_cast<uint32_t>(ev);
                en.y = static_cast<uint32_t>(ev >> 32);
            }
            const bool un = !(en.y & kExpanded);
            const uint64_t um = __builtin_amdgcn_ballot_w64(un);
            const int pos = __popcll(um & ((1ull << lane) - 1ull));
            if (un) win[pos] = (static_cast<unsigned long long>(en.y) << 32) | en.x;
            cb = __popcll(um);
            cb = cb < T ? cb : T;
            if (lane < cb) {
                const unsigned long long wv = win[lane];
                wb = make_uint2(static_cast<uint32_t>(wv), static_cast<uint32_t>(wv >> 32));
            }
        }
        // merged position of either entry: the main list's entries are the older ones
        int rm = lane, rb = lane;
        {
            const float fm_ = __uint_as_float(wm.x), fb_ = __uint_as_float(wb.x);
            const int tmax = cm > cb ? cm : cb;
            for (int t = 0; t < tmax; t++) {
                const float bt = __int_as_float(__builtin_amdgcn_readlane(static_cast<int>(wb.x), t));
                const float mt = __int_as_float(__builtin_amdgcn_readlane(static_cast<int>(wm.x), t));
                rm += (t < cb && bt < fm_) ? 1 : 0;
                rb += (t < cm && mt <= fb_) ? 1 : 0;
            }
        }
        int wcnt = cm + cb;
        wcnt = wcnt < T ? wcnt : T;
        if (lane < cm && rm < 64) win[rm] = (static_cast<unsigned long long>(wm.y) << 32) | wm.x;
        if (lane < cb && rb < 64) win[rb] = (static_cast<unsigned long long>(wb.y) << 32) | wb.x;
        uint2 w = make_uint2(0u, kSoloFree);
        if (lane < wcnt) {
            const unsigned long long wv = win[lane];
            w.x = static_cast<uint32_t>(wv);
            w.y = static_cast<uint32_t>(wv >> 32);
        }
        // tau: what a helper compares a fresh distance with before it chases the node: the last entry of the window
        if (f == 0) {
            uint32_t tl = 0xff800000u;  // nothing unexpanded: -inf
            if (wcnt > 0) tl = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.x), wcnt - 1));
            if (tl != tau_sent) {
                if (lane == 0)
                    coherent_store(reinterpret_cast<unsigned long long *>(mail + 4), (static_cast<unsigned long long>(a.pf_seq) << 32) | tl);
                tau_sent = tl;
            }
        }
        // ---- slots the sequencer has consumed are free again
        if (mgr && s_node < kSoloBusy && c_done[slot] == s_node) {
            s_node = kSoloFree;
            c_node[slot] = kSoloFree;
        }
        // ---- which window nodes are mine, and which of them has no slot yet
        const bool mine = lane < wcnt && w.y < static_cast<uint32_t>(a.n) && solo_half(w.y) == static_cast<uint32_t>(f);
        bool seen = false, isnew = false;
        uint64_t unc = 0;
        for (uint64_t mm = __builtin_amdgcn_ballot_w64(mine); mm; mm &= mm - 1) {
            const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
            const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.y), j));
            const uint64_t hit = __builtin_amdgcn_ballot_w64(mgr && s_node == c);
            if (hit) seen = seen || (lane == __ffsll(static_cast<unsigned long long>(hit)) - 1);
            else unc |= 1ull << j;
        }
        for (uint64_t mm = unc; mm; mm &= mm - 1) {
            const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
            const uint32_t c = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(w.y), j));
            uint64_t fm = __builtin_amdgcn_ballot_w64(mgr && s_node == kSoloFree);
            bool steal = false;
            if (!fm) {  // no free slot: take one whose node has left the window (the sequencer re-checks the node after its reads)
                fm = __builtin_amdgcn_ballot_w64(mgr && !seen && !isnew);
                steal = true;
            }
            if (!fm) break;
            const int Ls = __ffsll(static_cast<unsigned long long>(fm)) - 1;
            if (lane == Ls) {
                c_node[slot] = kSoloBusy;
                c_state[slot] = 0;
                c_valid[slot] = 0;
                c_node[slot] = c;
                s_node = c;
                s_state = 0;
                s_valid = 0;
                isnew = true;
            }
            n_steal += steal ? 1 : 0;
            busy = true;
        }
        // ---- claim the new nodes; the ones nobody has claimed before go to the ring
        {
            bool push = false;
            if (isnew) {
                const uint32_t mytag = solo_tag(a.pf_seq, s_node);
                push = atomicMax(claim + solo_slot(s_node, a.solo_log2s, a.n), mytag) != mytag;  // (the owner asks once per slot it fills)
            }
            const uint64_t pm = __builtin_amdgcn_ballot_w64(push);
            if (pm) {
                const int leader = __ffsll(static_cast<unsigned long long>(pm)) - 1;
                unsigned long long base = 0;
                if (lane == leader) base = atomicAdd(head, static_cast<unsigned long long>(__popcll(pm)));
                const uint32_t blo = __builtin_amdgcn_readlane(static_cast<uint32_t>(base), leader);
                const uint32_t bhi = __builtin_amdgcn_readlane(static_cast<uint32_t>(base >> 32), leader);
                if (push && bhi == a.pf_seq) {
                    const uint32_t pe = blo + __popcll(pm & ((1ull << lane) - 1ull));
                    const unsigned long long tagv = (static_cast<unsigned long long>(a.pf_seq) << 8) | ((pe / kSoloRing) & 0xff);
                    coherent_store(ringw + (pe % kSoloRing), (tagv << 32) | s_node);
                }
                n_push += __popcll(pm);
            }
        }
        // ---- poll the incomplete slots, four at a time: lane j = neighbour slot j of the node
        const bool pend = mgr && s_node < kSoloBusy && (s_valid & degmask) != degmask;
        for (uint64_t pm = __builtin_amdgcn_ballot_w64(pend); pm;) {
            int Lq[4];
            uint32_t cq[4], stq[4];
            unsigned long long wq[4];
            int32_t idq[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                Lq[u] = pm ? __ffsll(static_cast<unsigned long long>(pm)) - 1 : -1;
                pm &= pm - 1;  // (0 & anything stays 0)
                const int Lc = Lq[u] < 0 ? 0 : Lq[u];
                cq[u] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(s_node), Lc));
                stq[u] = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(s_state), Lc));
                wq[u] = 0;
                idq[u] = -1;
                if (Lq[u] >= 0 && lane < deg) {
                    wq[u] = coherent_load(rec + static_cast<int64_t>(solo_slot(cq[u], a.solo_log2s, a.n)) * a.M0 + lane);
                    if (!stq[u]) idq[u] = a.l0_adj[static_cast<int64_t>(cq[u]) * a.M0 + lane];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (Lq[u] < 0) continue;
                const int sl = f * kMine + Lq[u];
                if (!stq[u]) {
                    c_id[sl * kMaxDeg + lane] = idq[u];  // (lanes >= deg: -1)
                } else {
                    idq[u] = c_id[sl * kMaxDeg + lane];
                }
                const bool ok = lane < deg && static_cast<uint32_t>(wq[u] >> 32) == solo_tag(a.pf_seq, cq[u]);
                const bool pad = lane < deg && (idq[u] < 0 || idq[u] >= a.n);
                if (ok) c_d[sl * kMaxDeg + lane] = __uint_as_float(static_cast<uint32_t>(wq[u]));
                const uint64_t vm = __builtin_amdgcn_ballot_w64(ok || pad);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the data before the words that announce it
                if (lane == Lq[u]) {
                    if (vm & ~s_valid) busy = true;
                    s_valid |= vm;
                    c_valid[slot] = s_valid;
                    if (!s_state) {
                        c_state[slot] = 1;
                        s_state = 1;
                    }
                }
            }
        }
        if (!__builtin_amdgcn_ballot_w64(busy)) __builtin_amdgcn_s_sleep(4);
    }
    if (cnt_push && lane == 0) {
        atomicAdd(cnt_push, n_push);
        atomicAdd(cnt_steal, n_steal);
    }
}

#ifdef HG_SOLO_STAMPS
#define SOLO_STAMP(slot)                            \
    do {                                            \
        const unsigned long long t_now = clock64(); \
        st_acc[slot] += t_now - st_prev;            \
        st_prev = t_now;                            \
    } while (0)
#else
#define SOLO_STAMP(slot) do { } while (0)
#endif

// Distances of the neighbours in `need` (lane j = neighbour slot j, id in nb_id): the rows gathered RB per trip -- lane b takes
// the candidate of rank t0 + b and fetches its norm up front -- lane_partial + butterfly + finish_dist, into `dist` of lane j.
template <int NCH, int RB, bool L2>
__device__ __forceinline__ void solo_gather(const HnswArgs &a, const float4 (&q)[NCH], float qn, uint64_t need, int32_t nb_id,
                                            float &dist, int lane, int nvec) {
    const int nneed = __popcll(need);
    const bool isset = (need >> lane) & 1ull;
    const int rank = __popcll(need & ((1ull << lane) - 1ull));
    for (int t0 = 0; t0 < nneed; t0 += RB) {
        float4 r[RB][NCH];
        int myj = -1;
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const uint64_t hit = __builtin_amdgcn_ballot_w64(isset && rank == t0 + b);
            const int jb = hit ? __ffsll(static_cast<unsigned long long>(hit)) - 1 : -1;
            myj = lane == b ? jb : myj;
        }
        const int32_t myid = __shfl(nb_id, myj >= 0 ? myj : 0, kWave);
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
        const float dv = finish_dist(a.metric, mine, qn, myrn) + 0.0f;
#pragma unroll
        for (int b = 0; b < RB; b++) {
            const int jb = __builtin_amdgcn_readlane(myj, b);
            const float vb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dv), b));
            if (lane == jb) dist = vb;
        }
    }
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
// The sequencer's list is TWO sorted sequences: the main list in LDS and an admission buffer of up to 64 entries in its
// registers (lane k = the k-th smallest; mirrored to LDS for the fetchers).  The reference's `nearest` (ultra_fast.clj:158)
// is main[0, pm) + buffer[0, pb), pm + pb <= ef; its `candidates` are the unexpanded entries of both.  An admitted neighbour
// (:195-198) enters the buffer by one ballot and one lane shift; the entry it pushes out of `nearest` (:203-204) is the
// later of main[pm - 1] and buffer[pb - 1] -- a pointer moves, nothing else.  Evicted entries stay where they are: the
// next candidate is the smaller of the first unexpanded entries of either sequence, and it is expanded iff it is still
// <= the worst of `nearest` (:175-178) -- the reference's own loop, instead of the single-workgroup kernel's positional
// merge of every expansion (one wave moved ~6 blocks of a 640-entry list per admission: 4 us per expansion).  Every 64
// admissions, and when the layer is done, the buffer is merged into the main list in place (per-lane binary searches).
// RB: rows in flight per trip of the sequencer's own gathers; RBH: of a helper wave.
template <int NCH, int RB, int RBH, bool L2>
__global__ __launch_bounds__(kWG) void hnsw_solo_kernel(HnswArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (uniform, and the compiler knows it: the branches on it are scalar)
    const int nvec = static_cast<int>(a.ld / 4);
    int role, query;
    solo_place(a, role, query);
    uint32_t *mail = query < a.nq ? a.pf_mail + static_cast<int64_t>(query) * kSoloMailWords : nullptr;
    if (role > 0 && query < a.nq) solo_helper<NCH, RBH, L2>(a, mail, role, query);
    if (role == 0 && query < a.nq) {
        const SoloLds Ls = solo_carve(smem, a.cap, a.nwords);
        uint2 *const curA = Ls.list;
        uint32_t *const bits = Ls.bits;
        volatile HG_LDS int32_t *sc = ldsv(Ls.sc);
        volatile HG_LDS uint2 *const bm = ldsv(Ls.bmir);
        volatile HG_LDS uint32_t *const vnode = ldsv(Ls.c_node), *const vstate = ldsv(Ls.c_state), *const vdone = ldsv(Ls.c_done);
        volatile HG_LDS unsigned long long *const vvalid = ldsv(Ls.c_valid);
        volatile HG_LDS int32_t *const vcid = ldsv(Ls.c_id);
        volatile HG_LDS float *const vcd = ldsv(Ls.c_d);
        const int qi = query;
        float4 q[NCH];
        load_query<NCH>(q, a.Q + static_cast<int64_t>(qi) * a.qld, a.dim, lane);
        const float qn = a.metric == METRIC_COS ? query_norm<NCH>(q) : 0.0f;
        int64_t n_eval = 0, n_hop = 0, n_exact = 0;
        unsigned long long h_cached = 0, h_full = 0, h_poll = 0, h_gather = 0, h_l0 = 0, h_compact = 0;
        int lm = 0;  // entries of the main list
        if (tid < 32) sc[tid] = 0;
        if (tid < kSoloSlots) {
            Ls.c_node[tid] = kSoloFree;
            Ls.c_done[tid] = kSoloFree;
            Ls.c_state[tid] = 0;
            Ls.c_valid[tid] = 0;
        }
        // seed: the entry point (ultra_fast.clj:358-359)
        if (wave == 0) {
            float4 r[NCH];
            load_row<NCH>(r, a.rows + static_cast<int64_t>(a.entry) * a.ld, nvec, lane, true);
            const float s = wave_sum(lane_partial<NCH, L2>(q, r));
            const float d = finish_dist(a.metric, s, qn, a.metric == METRIC_COS ? a.row_norms[a.entry] : 0.0f);
            if (lane == 0) curA[0] = make_uint2(__float_as_uint(d + 0.0f), static_cast<uint32_t>(a.entry));
            lm = 1;
            n_eval = 1;
        }
        const int64_t S = static_cast<int64_t>(1) << a.solo_log2s;
        const unsigned long long *rec = a.solo_rec + static_cast<int64_t>(query) * S * a.M0;
        int g_seq = 0;  // gather requests so far: handed over by the sequencer / taken by the assistant wave (sc[8] starts at 0)
        for (int level = a.max_level; level >= 0; level--) {
            const int ef_l = level > 0 ? 1 : a.ef;
            for (int w = tid; w < a.nwords; w += kWG) bits[w] = 0;  // fresh visited set per layer (:156)
            __syncthreads();
            if (wave == 0) {
                // ================= the sequencer: search-layer-ultra on this layer, one wave, no barrier =================
                if (lm > ef_l) lm = ef_l;
                // the reference re-evaluates its entry points at every layer (:162-167); the values are reused here, but
                // counted so that `evals` is the reference's number of distance calls
                if (level != a.max_level) n_eval += lm;
                for (int i = lane; i < lm; i += kWave) {
                    uint2 e = curA[i];
                    e.y &= ~kExpanded;
                    curA[i] = e;
                    atomicOr(&bits[e.y >> 5], 1u << (e.y & 31));
                }
                const int deg = level == 0 ? a.M0 : a.M;
                // ---- the two sequences
                float bd = __uint_as_float(0x7f800000u);  // buffer, lane k: distance (+inf behind the entries) ...
                uint32_t bi = kExpanded;    // ... and node | expanded flag
                int nb = 0, pb = 0;         // buffer entries; those of them in `nearest`
                uint64_t bun = 0;           // bit k: buffer entry k is unexpanded
                int pm = lm;                // main entries in `nearest`
                float fd = 0.0f;            // front window of the main list, lane l: entry fbase + l
                uint32_t fi = kExpanded;
                int fbase = 0;
                uint64_t fun = 0;           // bit l: entry fbase + l exists and is unexpanded
                float td = 0.0f;            // tail window: distance of main entry tbase + l (the entries around pm)
                int tbase = 0;
                float worst = 0.0f;         // of `nearest`, while it holds ef entries
                bool dirty = false;         // the buffer differs from its mirror
                bool dirty_sc = false;      // ... only its entry count does
                auto load_front = [&](int from) {
                    fbase = from;
                    const int i = fbase + lane;
                    uint2 e = make_uint2(0u, kExpanded);
                    if (i < lm) e = curA[i];
                    fd = __uint_as_float(e.x);
                    fi = e.y;
                    fun = __builtin_amdgcn_ballot_w64(i < lm && !(e.y & kExpanded));
                };
                auto load_tail = [&]() {
                    tbase = pm > kWave ? pm - kWave : 0;
                    const int i = tbase + lane;
                    td = i < lm ? __uint_as_float(curA[i].x) : 0.0f;
                };
                // (scalar decisions compare ORDERABLE KEYS of the distance bits with integer instructions: a float compare of two
                // uniform values is a vector instruction whose result the scalar unit then waits ~25 cycles for)
                auto fkey = [](uint32_t b) { return b ^ (static_cast<uint32_t>(static_cast<int32_t>(b) >> 31) | 0x80000000u); };
                uint32_t worst_k = 0;
                auto top_worst = [&]() {  // the later of main[pm - 1] and buffer[pb - 1]: the worst of `nearest`
                    const int im = pm - 1 - tbase, ib = pb - 1;
                    const uint32_t wmb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(td), im > 0 ? im : 0));
                    const uint32_t wbb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), ib > 0 ? ib : 0));
                    const uint32_t km = pm > 0 ? fkey(wmb) : 0u, kb2 = pb > 0 ? fkey(wbb) : 0u;
                    worst_k = kb2 >= km ? kb2 : km;
                    worst = __uint_as_float(kb2 >= km ? wbb : wmb);
                };
                // Merge the buffer into the main list, in place: buffer entry k goes to k + (main entries <= it), main entry i
                // to i + (buffer entries < it) -- the main entries are the older ones.  Blocks from the tail down to the first
                // position that changes; an entry moves towards the tail by at most 64, into blocks already read.
                auto compact = [&]() {
                    if (nb == 0) return;
                    h_compact++;
                    if (lane < nb) {
                        bm[lane].x = __float_as_uint(bd);
                        bm[lane].y = bi;
                    }
                    // (the first unexpanded main entry, in the coordinates before the merge: nothing in front of it or of minP moves)
                    const int first_un = fun ? fbase + __ffsll(static_cast<unsigned long long>(fun)) - 1 : fbase + kWave;
                    int lo = 0, hi = lm;  // upper bound of bd in the main list
                    for (int span = lm; span > 0; span >>= 1) {
                        const int mid = (lo + hi) >> 1;
                        const float v = __uint_as_float(curA[mid < lm ? mid : lm - 1].x);
                        const bool act = lo < hi;
                        const bool go = act && v <= bd;
                        lo = go ? mid + 1 : lo;
                        hi = (act && !go) ? mid : hi;
                    }
                    const int Pk = lane + lo;
                    const int minP = __builtin_amdgcn_readlane(Pk, 0);
                    const int total = lm + nb, top = pm + pb;
                    const bool full = top >= ef_l;
                    for (int base = ((lm - 1) / kWave) * kWave; base >= 0 && base + kWave > minP; base -= kWave) {
                        const int i = base + lane;
                        const bool valid_i = i < lm;
                        uint2 e = make_uint2(0u, 0u);
                        if (valid_i) e = curA[i];
                        const float de = __uint_as_float(e.x);
                        int l2 = 0, h2 = nb;  // lower bound of de in the buffer
#pragma unroll
                        for (int it = 0; it < 7; it++) {
                            const int mid = (l2 + h2) >> 1;
                            const float v = __uint_as_float(bm[mid < nb ? mid : nb - 1].x);
                            const bool act = l2 < h2;
                            const bool go = act && v < de;
                            l2 = go ? mid + 1 : l2;
                            h2 = (act && !go) ? mid : h2;
                        }
                        const int Pe = i + l2;
                        if (valid_i && Pe < a.cap && Pe != i) curA[Pe] = e;
                    }
                    if (lane < nb && Pk < a.cap) curA[Pk] = make_uint2(__float_as_uint(bd), bi);
                    if (full) {
                        // behind `nearest` only what ties its worst can still be expanded (:175-178): that run stays (the
                        // single-workgroup kernel's ghosts), as far as the list has room
                        const uint32_t wbits = curA[ef_l - 1].x;
                        int phys = (total < a.cap ? total : a.cap) - ef_l;
                        const bool more = phys > kWave || total > a.cap;
                        phys = phys > kWave ? kWave : phys;
                        const bool tie = lane < phys && curA[ef_l + (lane < phys ? lane : 0)].x == wbits;
                        const uint64_t nt = ~__builtin_amdgcn_ballot_w64(tie);
                        const int run = nt ? __ffsll(static_cast<unsigned long long>(nt)) - 1 : kWave;
                        if (run == phys && more && lane == 0) sc[6] = 1;  // ties may have been cut off: the query is repeated
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
                    dirty = true;
                    load_front(first_un < minP ? first_un : minP);
                    load_tail();
                };
                load_front(0);
                load_tail();
                top_worst();
                if (lane == 0) {
                    sc[0] = 0;
                    sc[1] = lm;
                    sc[3] = 0;
                }
#ifdef HG_SOLO_STAMPS
                unsigned long long st_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                unsigned long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                unsigned long long st_prev = clock64();
#endif
                for (;;) {
                    SOLO_STAMP(9);
                    // ---- next candidate: the smaller of the first unexpanded entries of the two sequences (a tie: the main
                    //      list's, it is the older one)
                    while (fun == 0 && fbase + kWave < lm) load_front(fbase + kWave);
                    if ((fun | bun) == 0) break;
                    const int lf = fun ? __ffsll(static_cast<unsigned long long>(fun)) - 1 : 0;
                    const int kb = bun ? __ffsll(static_cast<unsigned long long>(bun)) - 1 : 0;
                    const uint32_t dmk = fun ? fkey(static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(fd), lf))) : 0xffffffffu;
                    const uint32_t dbk = bun ? fkey(static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), kb))) : 0xffffffffu;
                    const uint32_t nm = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(fi), lf));
                    const uint32_t nbf = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(bi), kb));
                    const bool take_main = fun != 0 && dmk <= dbk;
                    if (pm + pb >= ef_l && (take_main ? dmk : dbk) > worst_k) break;  // (:175-178; nothing behind it can qualify either)
                    const uint32_t node = take_main ? nm : nbf;
                    if (take_main) {
                        if (lane == lf) {
                            fi |= kExpanded;
                            curA[fbase + lf].y = fi;
                        }
                        fun &= fun - 1;
                        if (lane == 0) sc[0] = fun ? fbase + __ffsll(static_cast<unsigned long long>(fun)) - 1 : fbase + kWave;
                    } else {
                        if (lane == kb) {
                            bi |= kExpanded;
                            bm[kb].y = bi;
                        }
                        bun &= bun - 1;
                    }
                    SOLO_STAMP(0);
                    // ---- its neighbours and their distances: the LDS cache first
                    int32_t nb_id = -1;
                    float dist = 0.0f;
                    unsigned long long valid = 0;
                    bool have_ids = false;
                    if (level == 0) {
                        h_l0++;
                        const uint32_t cn = lane < kSoloSlots ? vnode[lane] : kSoloFree;
                        const uint64_t hm = __builtin_amdgcn_ballot_w64(cn == node);
                        if (hm) {
                            const int sl = __ffsll(static_cast<unsigned long long>(hm)) - 1;
                            // (in this order, volatile: the valid mask before the data it announces, the node again after
                            // everything -- still this node's slot then: the reads were this node's)
                            const uint32_t st = vstate[sl];
                            const unsigned long long vm = vvalid[sl];
                            const int32_t cid = vcid[sl * kMaxDeg + lane];
                            const float cd = vcd[sl * kMaxDeg + lane];
                            const uint32_t n2 = vnode[sl];
                            if (n2 == node && st) {
                                have_ids = true;
                                nb_id = lane < deg ? cid : -1;
                                dist = cd;
                                valid = vm;
                                if (lane == 0) vdone[sl] = node;
                                h_cached++;
                            }
                        }
                    }
                    unsigned long long pubw = 0;
                    if (!have_ids) {
                        const int32_t *adj = level == 0 ? a.l0_adj + static_cast<int64_t>(node) * a.M0
                                                        : a.up_adj + (a.up_off[node] + (level - 1)) * a.M;
                        if (level == 0 && lane < deg)  // the words the helpers may have published, beside the adjacency row
                            pubw = coherent_load(rec + static_cast<int64_t>(solo_slot(node, a.solo_log2s, a.n)) * a.M0 + lane);
                        nb_id = lane < deg ? adj[lane] : -1;
                        if (level == 0) {
                            const bool ok = lane < deg && static_cast<uint32_t>(pubw >> 32) == solo_tag(a.pf_seq, node);
                            if (ok) dist = __uint_as_float(static_cast<uint32_t>(pubw));
                            valid = __builtin_amdgcn_ballot_w64(ok);
                        }
                    }
                    SOLO_STAMP(1);
                    bool fresh = false;
                    if (nb_id >= 0 && nb_id < a.n) {
                        const uint32_t bit = 1u << (nb_id & 31);
                        const uint32_t old = atomicOr(&bits[nb_id >> 5], bit);
                        fresh = !(old & bit);
                    }
                    const uint64_t fm = __builtin_amdgcn_ballot_w64(fresh);
                    n_hop++;
                    if (fm == 0) continue;
                    n_eval += __popcll(fm);
                    uint64_t need = fm & ~valid;
                    SOLO_STAMP(2);
                    if (level == 0 && need == 0) h_full++;
                    if (level == 0 && need && have_ids) {
                        // the cache was behind: one direct look at the published words of the neighbours still missing
                        if ((need >> lane) & 1ull)
                            pubw = coherent_load(rec + static_cast<int64_t>(solo_slot(node, a.solo_log2s, a.n)) * a.M0 + lane);
                        const bool ok = ((need >> lane) & 1ull) && static_cast<uint32_t>(pubw >> 32) == solo_tag(a.pf_seq, node);
                        if (ok) dist = __uint_as_float(static_cast<uint32_t>(pubw));
                        need &= ~__builtin_amdgcn_ballot_w64(ok);
                        if (need == 0) h_poll++;
                    }
                    SOLO_STAMP(3);
                    if (need) {
                        // ---- what nobody has published: gather the rows and compute, RB rows per trip (lane b takes the
                        //      candidate of rank t0 + b and fetches its norm up front)
                        if (level == 0) h_gather++;
                        const int nneed = __popcll(need);
                        n_exact += nneed;
                        // more than one trip: the assistant wave takes every second one (hand-over through LDS: ids and mask
                        // before the request word, its distances before the answer word)
                        uint64_t mine_m = need, theirs = 0;
                        if (nneed > RB) {
                            const int rank = __popcll(need & ((1ull << lane) - 1ull));
                            mine_m = __builtin_amdgcn_ballot_w64(((need >> lane) & 1ull) && ((rank / RB) & 1) == 0);
                            theirs = need & ~mine_m;
                            ldsv(Ls.g_ids)[lane] = nb_id;
                            g_seq++;
                            if (lane == 0) {
                                ldsv(Ls.g_need)[0] = theirs;
                                sc[8] = g_seq;
                            }
                        }
                        solo_gather<NCH, RB, L2>(a, q, qn, mine_m, nb_id, dist, lane, nvec);
                        if (theirs) {
                            while (sc[9] != g_seq) __builtin_amdgcn_s_sleep(1);
                            const float gd = ldsv(Ls.g_d)[lane];
                            if ((theirs >> lane) & 1ull) dist = gd;
                        }
                    }
                    SOLO_STAMP(4);
                    // ---- admission (:195-204), the reference's own loop: the fresh neighbours in adjacency order, each
                    //      against the worst of `nearest` as the ones before it have left it
                    const bool isfull = pm + pb >= ef_l;
                    const uint64_t smask = __builtin_amdgcn_ballot_w64(fresh && (!isfull || dist < worst));  // (the worst only shrinks)
#ifdef HG_SOLO_STAMPS
                    if (level == 0 && smask) {
                        dg[isfull ? 0 : 1]++;
                        dg[isfull ? 2 : 3] += __popcll(smask);
                    }
#endif
                    // (63 admissions at most between two merges: the buffer has 64 lanes, and the tail window of 64 main entries
                    // covers every eviction)
                    SOLO_STAMP(5);
                    if (smask && nb + __popcll(smask) > kWave - 1) {
                        compact();
                        top_worst();
                    }
                    SOLO_STAMP(6);
                    if (smask & (smask - 1)) {
                        // ---- two or more survivors: all of them at once.  One pass in adjacency order decides every admission
                        //      exactly as the sequential loop would -- a survivor is admitted iff fewer than ef of {`nearest` as the
                        //      expansion found it, the survivors before it} are <= it (the entries those have pushed out of
                        //      `nearest` meanwhile were larger than it anyway) -- and collects the merge counts; the admitted ones
                        //      then enter the buffer together (a scatter through its LDS mirror), and what they push out of `nearest`
                        //      (:203-204) is the nev largest of its two tails, found by all lanes at once (a merge-path split).
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
                            // (smaller than every main entry the tail window shows, and the window does not start at 0: at most
                            // ef - 64 main + buffer entries are <= it, and fewer than 64 survivors precede it)
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
                        if (nadm) {
                            const bool isadm = (am >> lane) & 1ull;
                            if (lane < nb) {
                                bm[lane + shb].x = __float_as_uint(bd);
                                bm[lane + shb].y = bi;
                            }
                            if (isadm) {
                                bm[cball + arank].x = __float_as_uint(dist);
                                bm[cball + arank].y = static_cast<uint32_t>(nb_id);
                            }
                            nb += nadm;
                            {
                                const uint32_t ex = bm[lane].x, ey = bm[lane].y;
                                bd = lane < nb ? __uint_as_float(ex) : __uint_as_float(0x7f800000u);
                                bi = lane < nb ? ey : kExpanded;
                            }
                            bun = __builtin_amdgcn_ballot_w64(lane < nb && !(bi & kExpanded));
                            const int pbn = pb + nadm;
                            const int nev = pm + pbn > ef_l ? pm + pbn - ef_l : 0;
                            if (nev) {
                                // lane e: e entries leave the main list's tail, nev - e the buffer's.  Right iff what stays is
                                // before what leaves: main entries are the older ones (a tie: the buffer entry leaves)
                                const int e = lane, eb = nev - lane;
                                const bool feas = e <= nev && e <= pm && eb <= pbn;
                                const int i_mk = pm - e - 1, i_me = pm - e, i_bk = pbn - eb - 1, i_be = pbn - eb;
                                const float m_keep = __uint_as_float(curA[i_mk > 0 ? i_mk : 0].x);
                                const float m_ev = __uint_as_float(curA[(feas && e > 0) ? i_me : 0].x);
                                const float b_keep = __uint_as_float(bm[(feas && i_bk > 0) ? i_bk : 0].x);
                                const float b_ev = __uint_as_float(bm[(feas && eb > 0) ? i_be : 0].x);
                                const bool ca = i_mk < 0 || eb == 0 || m_keep <= b_ev;
                                const bool cb2 = i_bk < 0 || e == 0 || b_keep < m_ev;
                                const uint64_t okm = __builtin_amdgcn_ballot_w64(feas && ca && cb2);
                                int em = okm ? __ffsll(static_cast<unsigned long long>(okm)) - 1 : -1;
                                if (em < 0) {  // (cannot happen for comparable distances; NaNs: one at a time, the sequential rule)
                                    em = 0;
                                    int pmm = pm, pbb = pbn;
                                    for (int t = 0; t < nev; t++) {
                                        const uint32_t wmb = pmm > 0 ? curA[pmm - 1].x : 0u;
                                        const uint32_t wbb = pbb > 0 ? bm[pbb - 1].x : 0u;
                                        if (pmm > 0 && (pbb == 0 || fkey(wmb) > fkey(wbb))) {
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
                            dirty_sc = true;  // (the mirror IS the buffer now)
                        }
                    }
#pragma unroll 1
                    for (uint64_t mm = (smask & (smask - 1)) == 0 ? smask : 0; mm; mm &= mm - 1) {  // ONE survivor: straight into its place
                        const int j = __ffsll(static_cast<unsigned long long>(mm)) - 1;
                        const uint32_t djb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(dist), j));
                        const uint32_t idj = static_cast<uint32_t>(__builtin_amdgcn_readlane(nb_id, j));
                        if (pm + pb >= ef_l && fkey(djb) >= worst_k) continue;  // (:195-198, a strict <)
                        // behind the buffer entries <= it (ties: admission order; lanes >= nb hold +inf)
                        const float dj = __uint_as_float(djb);
                        const int r0 = __popcll(__builtin_amdgcn_ballot_w64(bd <= dj));
                        const int r = r0 < nb ? r0 : nb;  // (an infinite distance: behind everything)
                        const float sd = __uint_as_float(wave_shr1(__float_as_uint(bd)));
                        const uint32_t si = wave_shr1(bi);
                        bd = lane > r ? sd : (lane == r ? dj : bd);
                        bi = lane > r ? si : (lane == r ? idj : bi);
                        const uint64_t lowm = (1ull << r) - 1ull;
                        bun = (bun & lowm) | ((bun & ~lowm) << 1) | (1ull << r);
                        nb++;
                        pb++;
                        // (:203-204) if `nearest` now holds ef + 1, its worst leaves it: the later of the two tails (a tie: the
                        // buffer's, it is the younger).  Straight-line: integer selects on the keys, no branch.
                        {
                            const int over = pm + pb > ef_l ? 1 : 0;
                            const int im = pm - 1 - tbase;
                            const uint32_t wmb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(td), im > 0 ? im : 0));
                            const uint32_t wbb = static_cast<uint32_t>(__builtin_amdgcn_readlane(__float_as_int(bd), pb - 1));
                            const int evm = (over && pm > 0 && fkey(wmb) > fkey(wbb)) ? 1 : 0;
                            pm -= evm;
                            pb -= over - evm;
                        }
                        top_worst();
                        dirty = true;
                    }
                    SOLO_STAMP(7);
                    if (dirty) Ls.bmir[lane] = make_uint2(__float_as_uint(bd), bi);  // the fetchers see the buffer through its mirror
                    if (dirty || dirty_sc) {
                        if (lane == 0) {  // (plain stores: hints, and the LDS takes a wave's stores in order)
                            Ls.sc[3] = nb;
                            Ls.sc[1] = lm;
                            Ls.sc[0] = fun ? fbase + __ffsll(static_cast<unsigned long long>(fun)) - 1 : fbase + kWave;
                        }
                        asm volatile("" ::: "memory");
                        dirty = false;
                        dirty_sc = false;
                    }
                    SOLO_STAMP(8);
                }
                compact();
                if (lane == 0) {
                    sc[13] = level + 1;             // the assistant may go ...
                    if (level == 0) sc[2] = 1;      // ... and the fetchers
                }
#ifdef HG_SOLO_STAMPS
                if (level == 0 && a.dbg && lane == 0) {
                    for (int i = 0; i < 10; i++) atomicAdd(a.dbg + 40 + i, st_acc[i]);
                    for (int i = 0; i < 8; i++) atomicAdd(a.dbg + 50 + i, dg[i]);
                }
#endif
            } else if (level == 0 && wave <= kSoloFetchers) {
                solo_fetcher(a, Ls, mail, query, wave - 1, a.dbg ? a.dbg + 38 : nullptr, a.dbg ? a.dbg + 39 : nullptr);
            } else if (wave == kSoloFetchers + 1) {
                // ================= the assistant: its share of the sequencer's own gathers =================
                for (;;) {
                    int sq;
                    while ((sq = sc[8]) == g_seq && sc[13] != level + 1) __builtin_amdgcn_s_sleep(1);
                    if (sq == g_seq) break;  // the level is done (the sequencer waits for every answer before it says so)
                    g_seq = sq;
                    const uint64_t theirs = ldsv(Ls.g_need)[0];
                    const int32_t ids = ldsv(Ls.g_ids)[lane];
                    float dist = 0.0f;
                    solo_gather<NCH, RB, L2>(a, q, qn, theirs, ids, dist, lane, nvec);
                    ldsv(Ls.g_d)[lane] = dist;
                    if (lane == 0) sc[9] = sq;
                }
            }
            __syncthreads();
        }
        // ---- results: ascending, take k (:362-370; the distances are reused, not recomputed)
        if (wave == 0) {
            if (lane == 0) coherent_store(mail + 2, a.pf_seq);  // the helpers may go
            const int real = lm < a.ef ? lm : a.ef;
            for (int i = lane; i < a.k; i += kWave) {
                const bool ok = i < real;
                a.out_ids[static_cast<int64_t>(qi) * a.k + i] = ok ? static_cast<int32_t>(curA[i].y & ~kExpanded) : -1;
                a.out_dist[static_cast<int64_t>(qi) * a.k + i] = ok ? __uint_as_float(curA[i].x) : __uint_as_float(0x7f800000u);
            }
            if (lane == 0) {
                if (a.again && sc[6]) a.again[atomicAdd(a.again_cnt, 1)] = qi;
                if (a.stats) {
                    a.stats[2 * static_cast<int64_t>(qi)] = n_eval;
                    a.stats[2 * static_cast<int64_t>(qi) + 1] = n_hop;
                }
                if (a.rej_stats) {
                    atomicAdd(a.rej_stats, static_cast<unsigned long long>(n_exact));
                    atomicAdd(a.rej_stats + 1, static_cast<unsigned long long>(n_eval));
                }
                if (a.dbg) {  // diagnostics (hnswgpu_debug_set_tile_stamps): how the level-0 expansions were served
                    atomicAdd(a.dbg + 32, h_l0);
                    atomicAdd(a.dbg + 33, h_cached);
                    atomicAdd(a.dbg + 34, h_full);
                    atomicAdd(a.dbg + 35, h_poll);
                    atomicAdd(a.dbg + 36, h_gather);
                    atomicAdd(a.dbg + 37, static_cast<unsigned long long>(n_exact));
                    atomicAdd(a.dbg + 58, h_compact);
                }
            }
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
