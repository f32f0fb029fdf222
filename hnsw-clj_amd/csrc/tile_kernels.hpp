// tile_kernels.hpp -- kernels of the tiled (MFMA) scan path; contract and argument structs: tile_args.hpp.
// Included by engine.hip only (the kernels are not templates).
#pragma once
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

// Eight waves per workgroup (two per SIMD).  Every wave owns 32 rows of the 256-row tile and stages ITS rows
// global -> registers (two K-steps ahead) -> a wave-private LDS slab, so the K loop has no workgroup barrier at
// all: the only shared LDS data, the resident query group, is read-only.  With two independent instruction
// streams per SIMD, one wave's LDS / memory waits hide under the other wave's MFMA chain.
__global__ __launch_bounds__(kTileThreads) void tile_scan_kernel(TileArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int ldq = tile_ldq(a.dim);
    const int tq = tile_tq(a.dim);                               // queries resident in LDS
    float *Bs = reinterpret_cast<float *>(smem);                 // [tq][ldq]     resident query group
    float *As = Bs + tq * ldq;                                   // [8][32][36]   wave-private row slabs
    float *qn_s = As + kTileRows * kTileLdA;                     // [32]
    int64_t *ob_s = reinterpret_cast<int64_t *>(qn_s + kTileQ);  // [32] output bases (-1 = empty slot)
    float *rn_s = reinterpret_cast<float *>(ob_s + kTileQ);      // [256] row norms of the current tile
    int32_t *qi_s = reinterpret_cast<int32_t *>(rn_s + kTileRows);  // [32] query index of every slot (-1 = empty)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    // Persistent mode (a.work_ctr, the IVF work list): a workgroup per CU takes work items off per-XCD counters until
    // the list is empty -- no launch gap between items, no idle CU while another still has several items queued, and
    // the list can be cut into fewer, longer items (one query-group fill per 4 tiles instead of per 2).  Each XCD
    // first drains ITS contiguous eighth of the list (the groups that scan the same rows run side by side on one L2),
    // then helps the others.
    int32_t *s_item_p = qi_s + kTileQ;  // [1] the work item in hand (dynamic LDS: the static budget is spoken for)
    for (int iter = 0;; iter++) {
    int g, chunk;
    if (a.work_ctr) {
        __syncthreads();  // every wave is done with the previous item's LDS (query group, slot tables)
        if (tid == 0) {
            const int nitems = *a.nitems, per_xcd = (nitems + 7) >> 3;
            int got = -1;
            for (int t = 0; t < 8 && got < 0; t++) {
                const int x = (static_cast<int>(blockIdx.x) + t) & 7;
                const int i = atomicAdd(a.work_ctr + x, 1);
                if (i < per_xcd && x * per_xcd + i < nitems) got = x * per_xcd + i;
            }
            *s_item_p = got;
        }
        __syncthreads();
        const int item = *s_item_p;
        if (item < 0) break;
        g = a.wi_group[item];
        chunk = a.wi_chunk[item];
    } else if (iter > 0) {
        break;
    } else if (a.members) {
        // The work list is ordered (list, chunk, group): the groups that scan the SAME rows are neighbours.
        // Workgroups are dealt to the 8 XCDs round-robin, so neighbours in blockIdx would land on eight different
        // L2s and every group would pull its rows over the fabric again.  Give each XCD one contiguous eighth of
        // the work list instead (workgroup b -> XCD b % 8, slot b / 8): the second and third group of a chunk then
        // run beside the first on the same L2 and find its lines there.  Idle workgroups still sit at the end.
        const int nitems = *a.nitems;
        int item = blockIdx.x;
        if (!(a.dbg & 1)) {
            const int per_xcd = (nitems + 7) >> 3;
            const int slot = blockIdx.x >> 3;
            if (slot >= per_xcd) break;
            item = (blockIdx.x & 7) * per_xcd + slot;
        }
        if (item >= nitems) break;
        g = a.wi_group[item];
        chunk = a.wi_chunk[item];
    } else {
        g = blockIdx.x / a.nchunks;
        chunk = blockIdx.x % a.nchunks;
    }

    int64_t rb0, rb1;
    int cnt;
    if (a.members) {
        if (g >= *a.ngroups) continue;
        int s = a.grp_seg[g];
        rb0 = a.seg_off[s];
        rb1 = a.seg_off[s + 1];
        cnt = a.grp_mem_cnt[g];
    } else {
        rb0 = 0;
        rb1 = a.nrows_all;
        cnt = a.nq - g * tq < tq ? a.nq - g * tq : tq;
    }
    // split THIS segment evenly into round(tiles / tiles_per_chunk) chunks of whole tiles (no tiny tail chunk
    // that would reload the query group for a handful of rows); a.nchunks is the bound for the longest segment
    const int64_t tiles = (rb1 - rb0 + kTileRows - 1) / kTileRows;
    const int64_t nch = tile_nchunks(rb1 - rb0, a.chunk_rows, a.nchunks);
    if (chunk >= nch) continue;
    const int64_t per = (tiles + nch - 1) / nch * kTileRows;
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * per;
    const int64_t r1 = r0 + per < rb1 ? r0 + per : rb1;
    if (r0 >= r1 || cnt <= 0) continue;
    unsigned long long t_start = 0, c_start = 0;
    if (a.dbg_buf) {
        t_start = __builtin_amdgcn_s_memrealtime();
        c_start = __builtin_amdgcn_s_memtime();
    }

    // ---- resident query group -> LDS (zero rows for empty slots).  Two steps, so that no load in the copy loop
    //      depends on another load: (1) 32 threads resolve their slot's query index / output base / norm into LDS;
    //      (2) every thread copies its float4s with all of its loads in flight together.  (A loop of
    //      load -> wait -> store, with the member lookup inside it, cost 12-24 dependent memory round trips per
    //      workgroup before the first MFMA.)
    if (tid < kTileQ) {
        int qi = -1;
        int64_t ob = -1;
        if (tid < cnt) {
            if (a.members) {
                GroupMember m = a.members[a.grp_mem_begin[g] + tid];
                qi = m.q;
                ob = m.out_base;
            } else {
                qi = g * tq + tid;
                ob = static_cast<int64_t>(qi) * a.out_stride;
            }
        }
        qi_s[tid] = qi;
        ob_s[tid] = ob;
        qn_s[tid] = (qi >= 0 && a.q_norms) ? a.q_norms[qi] : 0.0f;
    }
    __syncthreads();
    const int nvec = static_cast<int>(a.ld / 4);
    const int nk = (a.dim + kTileK - 1) / kTileK;
    const int kps = tile_phase_steps(a.dim);   // K-steps of one residency phase
    const int nphase = (nk + kps - 1) / kps;   // 1 for dim <= 896
    // columns [phase * kps * 32, + kps * 32) of every resident query -> Bs (zero beyond the row end)
    auto fill_queries = [&](int phase) {
        constexpr int kQU = 14;  // tq x ldq stays under 100 KiB -> at most 14 float4 per thread: one batch
        const int per_row = kps * kTileK / 4, total = tq * per_row, c0 = phase * per_row;
        for (int f0 = tid; f0 < total; f0 += kTileThreads * kQU) {
            float4 v[kQU];
            int off[kQU];    // LDS offset of the float4 (-1 = past the end of the group)
            bool live[kQU];  // false = zero fill (empty slot, or a column past the row end)
#pragma unroll
            for (int u = 0; u < kQU; u++) {  // unconditional loads from clamped addresses; masked at the store
                int f = f0 + u * kTileThreads;
                const bool in = f < total;
                f = in ? f : total - 1;
                const int slot = f / per_row, c4 = f - slot * per_row;
                const int qi = qi_s[slot];
                const int qc = qi >= 0 ? qi : 0, cg = c0 + c4, cc = cg < nvec ? cg : nvec - 1;
                v[u] = reinterpret_cast<const float4 *>(a.Qp + static_cast<int64_t>(qc) * a.ld)[cc];
                off[u] = in ? slot * ldq + 4 * c4 : -1;
                live[u] = qi >= 0 && cg < nvec;
            }
#pragma unroll
            for (int u = 0; u < kQU; u++)
                if (off[u] >= 0) *reinterpret_cast<float4 *>(Bs + off[u]) = live[u] ? v[u] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    if (nphase == 1) {
        fill_queries(0);
        __syncthreads();  // single phase: the only workgroup barrier after the slot table -- the query group is complete
    }

    const int half = lane >> 5, li = lane & 31;
    float *Aw = As + wave * 32 * kTileLdA;  // this wave's slab: 32 rows x 32 floats (+4 pad)
    float *rnw = rn_s + wave * 32;
    const int64_t ob = li < tq ? ob_s[li] : -1;
    const float qn = li < tq ? qn_s[li] : 0.0f;
    uint64_t best = ~0ull;  // argmin mode: best (distance, row) of query column li over this wave's rows
    uint64_t best16 = ~0ull;  // ... of column lane & 15 in the narrow (<= 16 queries) layout
    for (int64_t tb = r0; tb < r1; tb += kTileRows) {
        const int64_t t0 = tb + wave * 32;  // this wave's 32 rows of the 256-row tile
        // single phase: no barrier below, a wave without rows in this tile (ragged chunk end) is done; several
        // phases: every wave takes part in the refills' barriers and multiplies clamped rows it never stores
        if (nphase == 1 && t0 >= r1) break;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        // staging registers as named members (an array passed by reference to the helpers ended up in scratch)
        struct Stage {
            float4 v0, v1, v2, v3;
        };
        Stage stA, stB;
        // Unconditional loads from clamped addresses: rows past the segment end feed outputs that are never stored;
        // columns past the row end re-read the row's last float4 and meet the ZERO padding of the query group in
        // LDS (finite x 0 adds exactly nothing to the f32 chain); K-steps past the end re-read the last one into
        // registers nobody uses.  No branch and no select around a load, so the compiler can count vmcnt instead
        // of draining it -- with branches in the loop every stage_store waited for the loads issued just before it.
        auto load1 = [&](int ks, int u) -> float4 {
            int f = lane + kWave * u;
            int row = f >> 3, c4 = ks * (kTileK / 4) + (f & 7);
            int64_t gr = t0 + row;
            gr = gr < r1 ? gr : r1 - 1;
            c4 = c4 < nvec ? c4 : nvec - 1;
            return reinterpret_cast<const float4 *>(a.rows + gr * a.ld)[c4];
        };
        auto stage_load = [&](Stage &st, int ks) {  // ks: K-step of the whole row
            ks = ks < nk ? ks : nk - 1;
            st.v0 = load1(ks, 0);
            st.v1 = load1(ks, 1);
            st.v2 = load1(ks, 2);
            st.v3 = load1(ks, 3);
        };
        auto store1 = [&](const float4 &v, int u) {
            int f = lane + kWave * u;
            *reinterpret_cast<float4 *>(Aw + (f >> 3) * kTileLdA + (f & 7) * 4) = v;
        };
        auto stage_store = [&](const Stage &st) {
            store1(st.v0, 0);
            store1(st.v1, 1);
            store1(st.v2, 2);
            store1(st.v3, 3);
        };
        // LDS -> registers for one K-step: A from this wave's slab, B from the resident query group
        auto lds_read = [&](float4 (&av)[kTileK / 8], float4 (&bv)[kTileK / 8], int ks) {  // ks: step within the phase
            ks = ks < kps ? ks : kps - 1;
            const float *Ab = Aw + li * kTileLdA + 4 * half;
            const float *Bb = Bs + (li & (tq - 1)) * ldq + ks * kTileK + 4 * half;  // columns >= tq repeat, unused
#pragma unroll
            for (int t = 0; t < kTileK / 8; t++) {
                av[t] = *reinterpret_cast<const float4 *>(Ab + 8 * t);
                bv[t] = *reinterpret_cast<const float4 *>(Bb + 8 * t);
            }
        };
        auto mfma4 = [&](const float4 &x, const float4 &y) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, y.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, y.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, y.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, y.w, acc, 0, 0, 0);
        };
        // One K-step (a macro, so that every register set is a named variable: the lambda form of this step put
        // the staging registers in scratch memory), software-pipelined inside the wave: registers `cur` hold K-step ks (read from LDS during the
        // previous step), `st` holds the global rows of K-step ks+1.  After the first 4 MFMAs the slab is
        // overwritten with ks+1 and read back into `nxt` (LDS ops of one wave execute in issue order, and the reads
        // of ks completed before this step began), `st` is refilled with K-step ks+3, and the remaining 12 MFMAs
        // cover all three latencies.  Reading and multiplying the same step back to back instead left the two
        // waves of a SIMD in lockstep (both waiting on LDS, then both queueing for the matrix core).
#ifndef HG_TILE_ABLATE  // diagnostic builds (tools/tile_variant.sh) drop parts of the step; 0 in the product
#define HG_TILE_ABLATE 0
#endif
#define HG_TILE_STEP(cav, cbv, nav, nbv, st, ks_)                  \
    do {                                                          \
        if (!(HG_TILE_ABLATE & 1)) mfma4(cav[0], cbv[0]);         \
        __builtin_amdgcn_sched_barrier(0);                        \
        if (!(HG_TILE_ABLATE & 4)) stage_store(st);               \
        if (!(HG_TILE_ABLATE & 2)) lds_read(nav, nbv, (ks_) + 1); \
        if (!(HG_TILE_ABLATE & 8)) stage_load(st, kb + (ks_) + 3); \
        __builtin_amdgcn_sched_barrier(0);                        \
        if (!(HG_TILE_ABLATE & 1)) {                              \
            mfma4(cav[1], cbv[1]);                                \
            mfma4(cav[2], cbv[2]);                                \
            mfma4(cav[3], cbv[3]);                                \
        }                                                         \
        __builtin_amdgcn_sched_barrier(0);                        \
    } while (0)
        // this wave's row norms -> its LDS slot now; they are needed only after the K loop
        if (lane < 32) rnw[lane] = (a.metric == METRIC_COS && t0 + lane < r1) ? a.row_norms[t0 + lane] : 0.0f;
        if (cnt <= 16) {
            // ---- narrow groups (<= 16 queries): v_mfma_f32_16x16x4_f32, two 16-row blocks x 16 columns per wave, half
            //      the matrix-core time of the 32-column tile whose other columns would be empty.  Lane l = (i = l & 15,
            //      s = l >> 4) carries k-slot s of A[i][.] / B[.][i]; the slots are fed k = 8t + {0, 4, 1, 5} and then
            //      8t + {2, 6, 3, 7}: the same f32 chain over k as the wide tile (hardware-checked bit for bit,
            //      tools/micro/mfma16_order.hip), so one restatement (oracle/oracle.c "MFMA order") covers both.
            const int i16 = lane & 15, ksl = lane >> 4, kq = 4 * (ksl & 1);
            const bool khi = (ksl >> 1) != 0;
            f32x4v accA, accB;  // rows 0..15 and 16..31 of this wave's 32
#pragma unroll
            for (int i = 0; i < 4; i++) accA[i] = accB[i] = 0.0f;
            // one K-step's operands after the k-slot selection: [t][u] = element for MFMA u of block t
            float na0[4][2], na1[4][2], nb[4][2], ca0[4][2], ca1[4][2], cb[4][2];
            auto lds_read16 = [&](float (&a0)[4][2], float (&a1)[4][2], float (&b)[4][2], int ks) {
                ks = ks < kps ? ks : kps - 1;
                const float *A0 = Aw + i16 * kTileLdA + kq, *A1 = Aw + (16 + i16) * kTileLdA + kq;
                const float *B0 = Bs + i16 * ldq + ks * kTileK + kq;
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float4 x = *reinterpret_cast<const float4 *>(A0 + 8 * t);
                    const float4 y = *reinterpret_cast<const float4 *>(A1 + 8 * t);
                    const float4 z = *reinterpret_cast<const float4 *>(B0 + 8 * t);
                    a0[t][0] = khi ? x.y : x.x;
                    a0[t][1] = khi ? x.w : x.z;
                    a1[t][0] = khi ? y.y : y.x;
                    a1[t][1] = khi ? y.w : y.z;
                    b[t][0] = khi ? z.y : z.x;
                    b[t][1] = khi ? z.w : z.z;
                }
            };
            auto mfma16 = [&](const float (&a0)[4][2], const float (&a1)[4][2], const float (&b)[4][2], int t) {
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][0], b[t][0], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][0], b[t][0], accB, 0, 0, 0);
                accA = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][1], b[t][1], accA, 0, 0, 0);
                accB = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][1], b[t][1], accB, 0, 0, 0);
            };
#define HG_TILE_STEP16(ca0_, ca1_, cb_, na0_, na1_, nb_, st, ks_)       \
    do {                                                               \
        mfma16(ca0_, ca1_, cb_, 0);                                    \
        __builtin_amdgcn_sched_barrier(0);                             \
        stage_store(st);                                               \
        lds_read16(na0_, na1_, nb_, (ks_) + 1);                        \
        stage_load(st, kb + (ks_) + 3);                                \
        __builtin_amdgcn_sched_barrier(0);                             \
        mfma16(ca0_, ca1_, cb_, 1);                                    \
        mfma16(ca0_, ca1_, cb_, 2);                                    \
        mfma16(ca0_, ca1_, cb_, 3);                                    \
        __builtin_amdgcn_sched_barrier(0);                             \
    } while (0)
            for (int phase = 0; phase < nphase; phase++) {
                const int kb = phase * kps;
                const int kn = nk - kb < kps ? nk - kb : kps;
                if (nphase > 1) {
                    __syncthreads();
                    fill_queries(phase);
                    __syncthreads();
                }
                stage_load(stA, kb);
                stage_load(stB, kb + 1);
                stage_store(stA);
                stage_load(stA, kb + 2);
                lds_read16(ca0, ca1, cb, 0);
                int ks = 0;
                for (; ks + 1 < kn; ks += 2) {
                    HG_TILE_STEP16(ca0, ca1, cb, na0, na1, nb, stB, ks);
                    HG_TILE_STEP16(na0, na1, nb, ca0, ca1, cb, stA, ks + 1);
                }
                if (ks < kn) HG_TILE_STEP16(ca0, ca1, cb, na0, na1, nb, stB, ks);
            }
#undef HG_TILE_STEP16
            // epilogue: lane (i16, ksl) holds D[4 ksl + r][i16] in accA[r] and D[16 + 4 ksl + r][i16] in accB[r]
            const int64_t ob16 = ob_s[i16];
            const float qn16 = qn_s[i16];
            if (a.out_key) {
                if (ob16 >= 0) {
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const int i = (r < 4 ? 0 : 16) + 4 * ksl + (r & 3);
                        const int64_t gr = t0 + i;
                        const float dv = finish_dist(a.metric, r < 4 ? accA[r & 3] : accB[r & 3], qn16, rnw[i]);
                        const uint64_t key = gr < r1 ? make_key(dv, static_cast<uint32_t>(gr - rb0)) : ~0ull;
                        best16 = key < best16 ? key : best16;
                    }
                }
            } else {
                float *T = Aw;  // [16 columns][kTileLdA]: T[c][i] = distance of column c to row i
                float4 v;
                int i = 4 * ksl;
                v.x = finish_dist(a.metric, accA[0], qn16, rnw[i + 0]) + 0.0f;
                v.y = finish_dist(a.metric, accA[1], qn16, rnw[i + 1]) + 0.0f;
                v.z = finish_dist(a.metric, accA[2], qn16, rnw[i + 2]) + 0.0f;
                v.w = finish_dist(a.metric, accA[3], qn16, rnw[i + 3]) + 0.0f;
                *reinterpret_cast<float4 *>(T + i16 * kTileLdA + i) = v;
                i = 16 + 4 * ksl;
                v.x = finish_dist(a.metric, accB[0], qn16, rnw[i + 0]) + 0.0f;
                v.y = finish_dist(a.metric, accB[1], qn16, rnw[i + 1]) + 0.0f;
                v.z = finish_dist(a.metric, accB[2], qn16, rnw[i + 2]) + 0.0f;
                v.w = finish_dist(a.metric, accB[3], qn16, rnw[i + 3]) + 0.0f;
                *reinterpret_cast<float4 *>(T + i16 * kTileLdA + i) = v;
                const int64_t gr = t0 + li;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int c = 2 * j + half;
                    const float dv = T[c * kTileLdA + li];
                    const int64_t obc = ob_s[c];
                    if (obc >= 0 && gr < r1) a.out[obc + (gr - rb0)] = dv;
                }
            }
            continue;
        }
        float4 av0[kTileK / 8], bv0[kTileK / 8], av1[kTileK / 8], bv1[kTileK / 8];
        for (int phase = 0; phase < nphase; phase++) {
            const int kb = phase * kps;                          // first K-step of the phase
            const int kn = nk - kb < kps ? nk - kb : kps;        // its K-steps
            if (nphase > 1) {
                __syncthreads();  // every wave has multiplied the previous phase (or tile): Bs may change
                fill_queries(phase);
                __syncthreads();
            }
            stage_load(stA, kb);
            stage_load(stB, kb + 1);
            stage_store(stA);
            stage_load(stA, kb + 2);
            lds_read(av0, bv0, 0);
            // entering step ks: cur regs = ks, the `st` passed in = ks+1, the other st = ks+2
            int ks = 0;
            for (; ks + 1 < kn; ks += 2) {
                HG_TILE_STEP(av0, bv0, av1, bv1, stB, ks);
                HG_TILE_STEP(av1, bv1, av0, bv0, stA, ks + 1);
            }
            if (ks < kn) HG_TILE_STEP(av0, bv0, av1, bv1, stB, ks);
        }
#undef HG_TILE_STEP
        // ---- epilogue: D[row i][query col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        if (HG_TILE_ABLATE & 16) asm volatile("" ::"v"(acc[0]), "v"(acc[15]));  // keep the chain alive
        if (ob >= 0 && !(HG_TILE_ABLATE & 16)) {
            if (a.out_key) {  // fused argmin: first minimum wins (strict <, ivf_flat.clj:86-89) = smallest (distance, row)
#pragma unroll
                for (int reg = 0; reg < 16; reg++) {
                    int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    int64_t gr = t0 + i;
                    float dv = finish_dist(a.metric, acc[reg], qn, rnw[i]);
                    uint64_t key = gr < r1 ? make_key(dv, static_cast<uint32_t>(gr - rb0)) : ~0ull;
                    best = key < best ? key : best;
                }
            }
        }
        if (!a.out_key && !(HG_TILE_ABLATE & 16)) {
            // Dense output: query column c's distances to this wave's 32 rows are 32 CONSECUTIVE floats of c's
            // candidate array.  In the MFMA layout a lane holds one column, so a store instruction would touch 64
            // different arrays (64 four-byte segments); transposed through the wave's LDS slab (its K loop is
            // over), lanes 0..31 / 32..63 write two 128-byte runs per instruction.
            float *T = Aw;  // [32 columns][kTileLdA]: T[c][i] = distance of column c to row i
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const int i = 8 * g4 + 4 * half;
                float4 v;
                v.x = finish_dist(a.metric, acc[4 * g4 + 0], qn, rnw[i + 0]) + 0.0f;
                v.y = finish_dist(a.metric, acc[4 * g4 + 1], qn, rnw[i + 1]) + 0.0f;
                v.z = finish_dist(a.metric, acc[4 * g4 + 2], qn, rnw[i + 2]) + 0.0f;
                v.w = finish_dist(a.metric, acc[4 * g4 + 3], qn, rnw[i + 3]) + 0.0f;
                *reinterpret_cast<float4 *>(T + li * kTileLdA + i) = v;
            }
            const int64_t gr = t0 + li;
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int c = 2 * j + half;
                const float dv = T[c * kTileLdA + li];
                const int64_t obc = ob_s[c];
                if (obc >= 0 && gr < r1) a.out[obc + (gr - rb0)] = dv;
            }
        }
    }
    if (a.out_key && cnt <= 16) {  // narrow layout: lanes i16 + 16 s, s = 0..3, hold four row quarters of column i16
        uint64_t other = __shfl_xor(best16, 16, kWave);
        best16 = other < best16 ? other : best16;
        other = __shfl_xor(best16, 32, kWave);
        best16 = other < best16 ? other : best16;
        if (lane < 16 && ob_s[lane] >= 0 && best16 != ~0ull)
            atomicMin(a.out_key + (static_cast<int64_t>(g) * tq + lane), best16);
    } else if (a.out_key && ob >= 0) {  // lanes li and li + 32 hold the two row-halves of query column li
        const uint64_t other = __shfl_xor(best, 32, kWave);
        best = other < best ? other : best;
        if (half == 0 && best != ~0ull) atomicMin(a.out_key + (static_cast<int64_t>(g) * tq + li), best);
    }
    if (a.dbg_buf) {  // diagnostics only: when and where this workgroup ran
        __syncthreads();
        if (tid == 0) {
            unsigned long long *o = a.dbg_buf + 4ull * blockIdx.x;
            o[0] = t_start;
            o[1] = __builtin_amdgcn_s_memrealtime();
            // HW_REG_HW_ID (id 4, all 32 bits: cu_id 11:8, sh_id 12, se_id 15:13) | HW_REG_XCC_ID (id 20) << 32
            //   | shader-clock cycles of this workgroup / 16 << 36 (against the 100 MHz stamps: the clock it ran at)
            o[2] = static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(0xF804)) |
                   ((static_cast<unsigned long long>(__builtin_amdgcn_s_getreg(0xF814)) & 0xf) << 32) |
                   (((__builtin_amdgcn_s_memtime() - c_start) >> 4) << 36);
            o[3] = static_cast<unsigned long long>((r1 - r0 + kTileRows - 1) / kTileRows) | (static_cast<unsigned long long>(cnt) << 32);
        }
    }
    }  // work items
}

// W = a.wpq waves per query.  W = 1: four queries per 256-thread workgroup, one wave each.  W > 1: one query per
// workgroup of W waves; wave w folds the 512-candidate blocks w, w + W, ... and wave 0 merges the W partial lists.
// (One wave per query left a batch of 1024 queries with four waves per CU, each walking 31k candidates through
// ~60 dependent load round trips: 143 us per launch, twice per batched IVF search.)  Keys are a total order on
// (distance, position), so the result does not depend on W.
// The body of select_topk_kernel for query q served by W waves of this workgroup (W = 1: this wave alone).  Also the
// tail of ivf_route_kernel (ivf.hip), whose last workgroup per query picks the probed lists in the same launch.
template <bool COH = false>
__device__ __forceinline__ void select_topk_wg(const SelectArgs &a, int q, int W, unsigned char *smem,
                                               const uint64_t **final_keys = nullptr) {
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    uint64_t *lists = reinterpret_cast<uint64_t *>(smem);
    uint64_t *list = lists + static_cast<size_t>(wave) * a.k;
    const float *in = a.dist + static_cast<int64_t>(q) * a.stride;
    const int64_t n = a.q_cnt ? a.q_cnt[q] : a.cnt_all;
    int cnt = 0;
    uint64_t thr = ~0ull;
    const bool regk = a.k <= kWave;
    uint64_t mine = ~0ull;
    constexpr int U = 8;  // 8 independent 64-wide loads in flight per iteration (one latency per 512 candidates)
    const int64_t first = W == 1 ? 0 : static_cast<int64_t>(wave) * U * kWave;
    uint64_t cap = ~0ull;  // bound on the k-th smallest key, known before anything is inserted (kth_bound)
    // the next block's loads are issued before this block is folded: one memory round trip per block would
    // otherwise sit between every two folds (15 blocks per wave at 31k candidates and 4 waves)
    const int64_t step = static_cast<int64_t>(W) * U * kWave;
    // element u of a lane's block: candidate base + idx_of(u).  Scalar form: u * 64 + lane (one 4-byte load per lane and
    // instruction: 256 B per wave instruction); vector form (a.vec4: rows 16-B aligned): two float4 per lane,
    // (u / 4) * 256 + lane * 4 + u % 4 -- a quarter of the load instructions for the same 512 candidates.  Keys carry
    // their own position, so the order in which a block is walked does not matter.
    const bool vec = a.vec4 != 0 && !COH;
    auto idx_of = [&](int u) -> int64_t { return vec ? (u >> 2) * (4 * kWave) + lane * 4 + (u & 3) : u * kWave + lane; };
    auto load_block = [&](float (&dst)[U], int64_t base) {
        if (vec) {
#pragma unroll
            for (int g4 = 0; g4 < U / 4; g4++) {
                const int64_t i = base + g4 * (4 * kWave) + lane * 4;
                // a float4 may straddle the end of the query's candidates (never the end of the buffer: DevBuf slack)
                const float4 t = i < n ? *reinterpret_cast<const float4 *>(in + i) : make_float4(0.f, 0.f, 0.f, 0.f);
                dst[4 * g4 + 0] = t.x;
                dst[4 * g4 + 1] = t.y;
                dst[4 * g4 + 2] = t.z;
                dst[4 * g4 + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int64_t i = base + u * kWave + lane;
                dst[u] = i < n ? (COH ? coherent_load(in + i) : in[i]) : __uint_as_float(0x7fc00000u);
            }
        }
    };
    float vn[U];
    load_block(vn, first);
    for (int64_t base = first; base < n; base += step) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = vn[u];
        load_block(vn, base + step);
        uint64_t key[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int64_t i = base + idx_of(u);
            key[u] = i < n ? make_key(v[u], static_cast<uint32_t>(i)) : ~0ull;
        }
        if (regk && base == first) {
            uint64_t m = key[0];
#pragma unroll
            for (int u = 1; u < U; u++) m = key[u] < m ? key[u] : m;
            cap = kth_bound(m, a.k, lane);
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
                        wave_insert_reg(mine, cnt, a.k, kb, lane);
                        const uint64_t kth = wave_kth_reg(mine, a.k);
                        thr = kth < cap ? kth : cap;
                    } else {
                        wave_insert(list, cnt, a.k, kb, lane);
                        thr = cnt == a.k ? list[a.k - 1] : ~0ull;
                    }
                }
            }
        }
    }
    if (regk) {
        if (lane < a.k) list[lane] = mine;
    } else {
        for (int i = cnt + lane; i < a.k; i += kWave) list[i] = ~0ull;
    }
    if (W > 1) {
        __syncthreads();
        if (wave != 0) return;
        // second level: W * k keys (sentinels included) -> final list behind the W slots
        uint64_t *fin = lists + static_cast<size_t>(W) * a.k;
        const int tot = W * a.k;
        thr = ~0ull;
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
        list = fin;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (final_keys) *final_keys = list;  // (the wave that finishes the job: k ascending keys in LDS, all-ones padded)
    if (!a.out_ord) return;
    for (int i = lane; i < a.k; i += kWave) {
        const uint64_t key = list[i];
        const bool ok = key != ~0ull;
        a.out_ord[static_cast<int64_t>(q) * a.k + i] = ok ? static_cast<uint32_t>(key) : 0xffffffffu;
        a.out_dist[static_cast<int64_t>(q) * a.k + i] = ok ? key_dist(key) : __uint_as_float(0x7f800000u);
    }
}

__global__ __launch_bounds__(1024) void select_topk_kernel(SelectArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int W = a.wpq;
    const int q = W == 1 ? blockIdx.x * kNWave + (threadIdx.x >> 6) : blockIdx.x;
    if (q >= a.nq) return;  // W > 1: uniform over the workgroup
    select_topk_wg(a, q, W, smem);
}

}  // namespace hg
