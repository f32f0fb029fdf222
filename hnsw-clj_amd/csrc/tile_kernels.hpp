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
    const int tq = tile_tq(a.dim);                               // queries resident in LDS (32 / 16 / 8)
    float *Bs = reinterpret_cast<float *>(smem);                 // [tq][ldq]     resident query group
    float *As = Bs + tq * ldq;                                   // [8][32][36]   wave-private row slabs
    float *qn_s = As + kTileRows * kTileLdA;                     // [32]
    int64_t *ob_s = reinterpret_cast<int64_t *>(qn_s + kTileQ);  // [32] output bases (-1 = empty slot)
    float *rn_s = reinterpret_cast<float *>(ob_s + kTileQ);      // [256] row norms of the current tile
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    int g, chunk;
    if (a.members) {
        if (static_cast<int>(blockIdx.x) >= *a.nitems) return;
        g = a.wi_group[blockIdx.x];
        chunk = a.wi_chunk[blockIdx.x];
    } else {
        g = blockIdx.x / a.nchunks;
        chunk = blockIdx.x % a.nchunks;
    }

    int64_t rb0, rb1;
    int cnt;
    if (a.members) {
        if (g >= *a.ngroups) return;
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
    if (chunk >= nch) return;
    const int64_t per = (tiles + nch - 1) / nch * kTileRows;
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * per;
    const int64_t r1 = r0 + per < rb1 ? r0 + per : rb1;
    if (r0 >= r1 || cnt <= 0) return;
    unsigned long long t_start = 0;
    if (a.dbg_buf) t_start = __builtin_amdgcn_s_memrealtime();

    // ---- resident query group -> LDS (zero rows for empty slots)
    if (tid < kTileQ) {
        if (tid < cnt) {
            int qi;
            int64_t ob;
            if (a.members) {
                GroupMember m = a.members[a.grp_mem_begin[g] + tid];
                qi = m.q;
                ob = m.out_base;
            } else {
                qi = g * tq + tid;
                ob = static_cast<int64_t>(qi) * a.out_stride;
            }
            ob_s[tid] = ob;
            qn_s[tid] = a.q_norms ? a.q_norms[qi] : 0.0f;
        } else {
            ob_s[tid] = -1;
            qn_s[tid] = 0.0f;
        }
    }
    const int nvec = static_cast<int>(a.ld / 4);
    const int nk = (a.dim + kTileK - 1) / kTileK;
    for (int f = tid; f < tq * (nk * kTileK / 4); f += kTileThreads) {
        int slot = f / (nk * kTileK / 4), c4 = f % (nk * kTileK / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < cnt && c4 < nvec) {
            int qi = a.members ? a.members[a.grp_mem_begin[g] + slot].q : g * tq + slot;
            v = reinterpret_cast<const float4 *>(a.Qp + static_cast<int64_t>(qi) * a.ld)[c4];
        }
        *reinterpret_cast<float4 *>(Bs + slot * ldq + 4 * c4) = v;
    }
    __syncthreads();  // the only workgroup barrier: the query group is complete

    const int half = lane >> 5, li = lane & 31;
    float *Aw = As + wave * 32 * kTileLdA;  // this wave's slab: 32 rows x 32 floats (+4 pad)
    float *rnw = rn_s + wave * 32;
    const int64_t ob = li < tq ? ob_s[li] : -1;
    const float qn = li < tq ? qn_s[li] : 0.0f;
    for (int64_t t0 = r0 + wave * 32; t0 < r1; t0 += kTileRows) {  // this wave's 32 rows of every 256-row tile
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        float4 stA[4], stB[4];
        // unconditional loads from clamped addresses (rows past the segment end feed outputs that are never
        // stored; columns past the row end are zeroed by a select): no branches, counted vmcnt possible
        auto stage_load = [&](float4 (&st)[4], int ks) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int f = lane + kWave * u;
                int row = f >> 3, c4 = ks * (kTileK / 4) + (f & 7);
                int64_t gr = t0 + row;
                gr = gr < r1 ? gr : r1 - 1;
                int c4c = c4 < nvec ? c4 : nvec - 1;
                float4 v = reinterpret_cast<const float4 *>(a.rows + gr * a.ld)[c4c];
                st[u] = c4 < nvec ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stage_store = [&](const float4 (&st)[4]) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int f = lane + kWave * u;
                *reinterpret_cast<float4 *>(Aw + (f >> 3) * kTileLdA + (f & 7) * 4) = st[u];
            }
        };
        auto compute = [&](int ks) {
            const float *Ab = Aw + li * kTileLdA + 4 * half;
            const float *Bb = Bs + (li & (tq - 1)) * ldq + ks * kTileK + 4 * half;  // columns >= tq repeat, unused
            float4 av[kTileK / 8], bv[kTileK / 8];
#pragma unroll
            for (int t = 0; t < kTileK / 8; t++) {
                av[t] = *reinterpret_cast<const float4 *>(Ab + 8 * t);
                bv[t] = *reinterpret_cast<const float4 *>(Bb + 8 * t);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the 8 LDS reads ahead of the MFMA chain
            if (a.dbg & 1) {
                asm volatile("" ::"v"(av[0].x), "v"(bv[0].x), "v"(av[3].w), "v"(bv[3].w));
                return;
            }
#pragma unroll
            for (int t = 0; t < kTileK / 8; t++) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].x, bv[t].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].y, bv[t].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].z, bv[t].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t].w, bv[t].w, acc, 0, 0, 0);
            }
        };
        // this wave's row norms -> its LDS slot now; they are needed only after the K loop
        if (lane < 32) rnw[lane] = (a.metric == METRIC_COS && t0 + lane < r1) ? a.row_norms[t0 + lane] : 0.0f;
        stage_load(stA, 0);
        stage_store(stA);
        if (nk > 1) stage_load(stA, 1);
        if (nk > 2) stage_load(stB, 2);
        // wave-private slab: LDS operations of one wave execute in issue order, so the reads of a K-step see the
        // stores before them and the stores of the next K-step land after these reads
        for (int ks = 0; ks < nk; ks += 2) {
            compute(ks);  // stA holds K-step ks+1
            if (ks + 1 < nk) stage_store(stA);
            if (ks + 3 < nk && !(a.dbg & 2)) stage_load(stA, ks + 3);
            if (ks + 1 >= nk) break;
            compute(ks + 1);  // stB holds K-step ks+2
            if (ks + 2 < nk) stage_store(stB);
            if (ks + 4 < nk && !(a.dbg & 2)) stage_load(stB, ks + 4);
        }
        // ---- epilogue: D[row i][query col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        if (ob >= 0 && !(a.dbg & 4)) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                int64_t gr = t0 + i;
                float dv = finish_dist(a.metric, acc[reg], qn, rnw[i]) + 0.0f;
                if (gr < r1) a.out[ob + (gr - rb0)] = dv;
            }
        }
    }
    if (a.dbg_buf) {  // diagnostics only: when and where this workgroup ran
        __syncthreads();
        if (tid == 0) {
            unsigned long long *o = a.dbg_buf + 4ull * blockIdx.x;
            o[0] = t_start;
            o[1] = __builtin_amdgcn_s_memrealtime();
            o[2] = __builtin_amdgcn_s_getreg(0x1804);  // HW_REG_HW_ID
            o[3] = static_cast<unsigned long long>((r1 - r0 + kTileRows - 1) / kTileRows) | (static_cast<unsigned long long>(cnt) << 32);
        }
    }
}

__global__ __launch_bounds__(kWG) void select_topk_kernel(SelectArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * kNWave + wave;
    if (q >= a.nq) return;
    uint64_t *list = reinterpret_cast<uint64_t *>(smem) + static_cast<size_t>(wave) * a.k;
    const float *in = a.dist + static_cast<int64_t>(q) * a.stride;
    const int64_t n = a.q_cnt ? a.q_cnt[q] : a.cnt_all;
    int cnt = 0;
    uint64_t thr = ~0ull;
    const bool regk = a.k <= kWave;
    uint64_t mine = ~0ull;
    constexpr int U = 8;  // 8 independent 64-wide loads in flight per iteration (one latency per 512 candidates)
    for (int64_t base = 0; base < n; base += U * kWave) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + u * kWave + lane;
            v[u] = i < n ? in[i] : __uint_as_float(0x7fc00000u);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            int64_t i = base + u * kWave + lane;
            uint64_t key = i < n ? make_key(v[u], static_cast<uint32_t>(i)) : ~0ull;
            uint64_t mask = __ballot(key < thr);
            while (mask) {
                int b = __ffsll(static_cast<unsigned long long>(mask)) - 1;
                mask &= mask - 1;
                uint64_t kb = __shfl(key, b, kWave);
                if (kb < thr) {
                    if (regk) {
                        wave_insert_reg(mine, cnt, a.k, kb, lane);
                        thr = wave_kth_reg(mine, a.k);
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
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int i = lane; i < a.k; i += kWave) {
        bool ok = i < cnt;
        uint64_t key = ok ? list[i] : ~0ull;
        a.out_ord[static_cast<int64_t>(q) * a.k + i] = ok ? static_cast<uint32_t>(key) : 0xffffffffu;
        a.out_dist[static_cast<int64_t>(q) * a.k + i] = ok ? key_dist(key) : __uint_as_float(0x7f800000u);
    }
}

}  // namespace hg
