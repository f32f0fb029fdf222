// tile_kernels.hpp -- kernels of the tiled (MFMA) scan path; contract and argument structs: tile_args.hpp.
// Included by engine.hip only (the kernels are not templates).
#pragma once
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

__global__ __launch_bounds__(kWG) void tile_scan_kernel(TileArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int ldq = tile_ldq(a.dim);
    float *Bs = reinterpret_cast<float *>(smem);                 // [32][ldq]   resident query group
    float *As = Bs + kTileQ * ldq;                               // [2][128][36] streamed row tiles
    float *qn_s = As + 2 * kTileRows * kTileLdA;                 // [32]
    int64_t *ob_s = reinterpret_cast<int64_t *>(qn_s + kTileQ);  // [32] output bases (-1 = empty slot)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int g = blockIdx.x / a.nchunks;
    const int chunk = blockIdx.x % a.nchunks;

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
        cnt = a.nq - g * kTileQ < kTileQ ? a.nq - g * kTileQ : kTileQ;
    }
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * a.chunk_rows;
    const int64_t r1 = r0 + a.chunk_rows < rb1 ? r0 + a.chunk_rows : rb1;
    if (r0 >= r1 || cnt <= 0) return;

    // ---- resident query group -> LDS (zero rows for empty slots)
    if (tid < kTileQ) {
        int64_t ob = -1;
        float qn = 0.0f;
        if (tid < cnt) {
            int qi;
            if (a.members) {
                GroupMember m = a.members[a.grp_mem_begin[g] + tid];
                qi = m.q;
                ob = m.out_base;
            } else {
                qi = g * kTileQ + tid;
                ob = static_cast<int64_t>(qi) * a.out_stride;
            }
            qn = a.q_norms ? a.q_norms[qi] : 0.0f;
            ob_s[tid] = ob;
            qn_s[tid] = qn;
        } else {
            ob_s[tid] = -1;
            qn_s[tid] = 0.0f;
        }
    }
    const int nvec = static_cast<int>(a.ld / 4);
    const int nk = (a.dim + kTileK - 1) / kTileK;
    for (int f = tid; f < kTileQ * (nk * kTileK / 4); f += kWG) {
        int slot = f / (nk * kTileK / 4), c4 = f % (nk * kTileK / 4);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot < cnt && c4 < nvec) {
            int qi = a.members ? a.members[a.grp_mem_begin[g] + slot].q : g * kTileQ + slot;
            v = reinterpret_cast<const float4 *>(a.Qp + static_cast<int64_t>(qi) * a.ld)[c4];
        }
        *reinterpret_cast<float4 *>(Bs + slot * ldq + 4 * c4) = v;
    }
    __syncthreads();

    const int half = lane >> 5, li = lane & 31;
    for (int64_t t0 = r0; t0 < r1; t0 += kTileRows) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.0f;
        // stage K-step 0
        float4 st[4];
        auto stage_load = [&](int ks) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int f = tid + kWG * u;
                int row = f >> 3, c4 = ks * (kTileK / 4) + (f & 7);
                int64_t gr = t0 + row;
                st[u] = (gr < r1 && c4 < nvec) ? reinterpret_cast<const float4 *>(a.rows + gr * a.ld)[c4]
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stage_store = [&](int buf) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                int f = tid + kWG * u;
                int row = f >> 3, c = (f & 7) * 4;
                *reinterpret_cast<float4 *>(As + (buf * kTileRows + row) * kTileLdA + c) = st[u];
            }
        };
        stage_load(0);
        stage_store(0);
        __syncthreads();
        for (int ks = 0; ks < nk; ks++) {
            const int buf = ks & 1;
            if (ks + 1 < nk) stage_load(ks + 1);
            const float *Ab = As + (buf * kTileRows + wave * 32 + li) * kTileLdA + 4 * half;
            const float *Bb = Bs + li * ldq + ks * kTileK + 4 * half;
#pragma unroll
            for (int t = 0; t < kTileK / 8; t++) {
                float4 av = *reinterpret_cast<const float4 *>(Ab + 8 * t);
                float4 bv = *reinterpret_cast<const float4 *>(Bb + 8 * t);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
            }
            if (ks + 1 < nk) stage_store(buf ^ 1);
            __syncthreads();
        }
        // ---- epilogue: D[row i][query col]: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
        const int64_t ob = ob_s[li];
        const float qn = qn_s[li];
        if (ob >= 0) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                int64_t gr = t0 + wave * 32 + i;
                if (gr < r1) {
                    float rn = a.metric == METRIC_COS ? a.row_norms[gr] : 0.0f;
                    a.out[ob + (gr - rb0)] = finish_dist(a.metric, acc[reg], qn, rn) + 0.0f;
                }
            }
        }
        __syncthreads();
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
    for (int64_t base = 0; base < n; base += kWave) {
        int64_t i = base + lane;
        uint64_t key = i < n ? make_key(in[i], static_cast<uint32_t>(i)) : ~0ull;
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
