// l2_kernels.hpp -- the batched scan for the Euclidean metric (euclidean-distance-ultra, ultra_fast.clj:43-51).
//
// sqrt(sum (q - v)^2) is not a bilinear form, so the MFMA tile kernel cannot compute it in the reference's arithmetic
// (|q|^2 + |v|^2 - 2 q.v loses the exact zeros the reference's tests pin), and until now every L2 batch ran as one GEMV
// per (query, row range) pair: the rows streamed from L2 / HBM once PER QUERY (batch 1024 IVF: 9.4 ms against 0.9 ms
// for cosine).  Here a workgroup keeps a group of up to 32 queries resident in LDS -- the same groups, work list and
// outputs as tile_scan_kernel (TileArgs) -- and every wave holds RB rows in registers and walks the group's queries
// over them: a row is fetched once per GROUP, the work per (row, query) pair is the GEMV kernel's own
// lane_partial + wave_sum + sqrt, so the distances are bit-identical to scan_kernel's (oracle/oracle.c "device order").
// VALU-bound (63 % of the VALU issue rate by SQ_INSTS_VALU at batch 1024); pairing rows for v_pk_add_f32 /
// v_pk_fma_f32 changed neither the counted instructions nor the time and was dropped.
//
// Round 2: the same kernel serves cosine / dot (template flag L2M = false) for the MID-SIZE batches of the IVF search --
// 0.5 to 2 (query, list) pairs per list, where the GEMV scan streams a list once per pair (the second reader at best
// finds the first one's lines in L2) and the MFMA tile kernel would change the summation order: here a list is
// fetched once per group and every distance keeps the GEMV order, bit for bit.
#pragma once
#include "kernels.hpp"
#include "tile_args.hpp"

namespace hg {

__host__ inline size_t l2_group_lds_bytes(int64_t ld) {
    return sizeof(float) * kTileQ * static_cast<size_t>(ld) + sizeof(int64_t) * kTileQ + sizeof(int32_t) * kTileQ +
           sizeof(float) * kTileQ;
}

template <int NCH, int RB, bool L2M = true>
__global__ __launch_bounds__(kTileThreads) void l2_group_kernel(TileArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int nvec = static_cast<int>(a.ld / 4);
    float4 *Bs = reinterpret_cast<float4 *>(smem);                             // [32][nvec] resident query group
    int64_t *ob_s = reinterpret_cast<int64_t *>(Bs + kTileQ * nvec);           // [32] output bases (-1 = empty slot)
    int32_t *qi_s = reinterpret_cast<int32_t *>(ob_s + kTileQ);                // [32] query index (-1 = empty)
    float *qn_s = reinterpret_cast<float *>(qi_s + kTileQ);                    // [32] query norms (cosine)
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int tq = kTileQ;
    // ---- work item -> (group, chunk): identical to tile_scan_kernel (XCD-contiguous slices of the work list)
    int g, chunk;
    if (a.members) {
        const int nitems = *a.nitems;
        const int per_xcd = (nitems + 7) >> 3;
        const int slot = blockIdx.x >> 3;
        if (slot >= per_xcd) return;
        const int item = (blockIdx.x & 7) * per_xcd + slot;
        if (item >= nitems) return;
        g = a.wi_group[item];
        chunk = a.wi_chunk[item];
    } else {
        g = blockIdx.x / a.nchunks;
        chunk = blockIdx.x % a.nchunks;
    }
    int64_t rb0, rb1;
    int cnt;
    if (a.members) {
        if (g >= *a.ngroups) return;
        const int s = a.grp_seg[g];
        rb0 = a.seg_off[s];
        rb1 = a.seg_off[s + 1];
        cnt = a.grp_mem_cnt[g];
    } else {
        rb0 = 0;
        rb1 = a.nrows_all;
        cnt = a.nq - g * tq < tq ? a.nq - g * tq : tq;
    }
    const int64_t tiles = (rb1 - rb0 + kTileRows - 1) / kTileRows;
    const int64_t nch = tile_nchunks(rb1 - rb0, a.chunk_rows, a.nchunks);
    if (chunk >= nch) return;
    const int64_t per = (tiles + nch - 1) / nch * kTileRows;
    const int64_t r0 = rb0 + static_cast<int64_t>(chunk) * per;
    const int64_t r1 = r0 + per < rb1 ? r0 + per : rb1;
    if (r0 >= r1 || cnt <= 0) return;

    // ---- slot table, then the query group with all of a thread's loads in flight (see tile_scan_kernel)
    if (tid < kTileQ) {
        int qi = -1;
        int64_t ob = -1;
        if (tid < cnt) {
            if (a.members) {
                const GroupMember m = a.members[a.grp_mem_begin[g] + tid];
                qi = m.q;
                ob = m.out_base;
            } else {
                qi = g * tq + tid;
                ob = static_cast<int64_t>(qi) * a.out_stride;
            }
        }
        qi_s[tid] = qi;
        ob_s[tid] = ob;
        qn_s[tid] = (!L2M && a.metric == METRIC_COS && qi >= 0 && a.q_norms) ? a.q_norms[qi] : 0.0f;
    }
    __syncthreads();
    {
        constexpr int kQU = 16;  // 32 x 256 float4 at most = 16 per thread
        const int total = cnt * nvec;
        for (int f0 = tid; f0 < total; f0 += kTileThreads * kQU) {
            float4 v[kQU];
            int fc[kQU];
#pragma unroll
            for (int u = 0; u < kQU; u++) {  // clamped, unconditional: all loads of a thread in flight together
                const int f = f0 + u * kTileThreads;
                fc[u] = f < total ? f : total - 1;
                const int slot = fc[u] / nvec, c4 = fc[u] - slot * nvec;
                v[u] = reinterpret_cast<const float4 *>(a.Qp + static_cast<int64_t>(qi_s[slot]) * a.ld)[c4];
            }
            // ... and unconditional stores: an index past the end re-writes element total - 1 with its own value
            // (a store under `f < total` let the compiler sink each load into its branch: load, wait, store, 16 times)
#pragma unroll
            for (int u = 0; u < kQU; u++) Bs[fc[u]] = v[u];  // slot * nvec + c4 == f
        }
    }
    __syncthreads();

    uint64_t best = ~0ull;  // argmin mode: lane q keeps the best (distance, row) of query slot q over this wave's rows
    for (int64_t base = r0 + wave * RB; base < r1; base += kTileWaves * RB) {
        float4 r[RB][NCH];
        // lane b fetches row b's precomputed norm with the rows (cosine), as scan_kernel does
        const float myrn = (!L2M && a.metric == METRIC_COS && lane < RB && base + lane < r1) ? a.row_norms[base + lane] : 0.0f;
#pragma unroll
        for (int b = 0; b < RB; b++) load_row<NCH>(r[b], a.rows + (base + b) * a.ld, nvec, lane, base + b < r1);
        for (int q = 0; q < cnt; q++) {
            const float qn = L2M ? 0.0f : qn_s[q];
            float4 qv[NCH];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const int i = c * kWave + lane;
                qv[c] = i < nvec ? Bs[q * nvec + i] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float s[RB];
#pragma unroll
            for (int b = 0; b < RB; b++) s[b] = lane_partial<NCH, L2M>(qv, r[b]);
            const float mine = rows_sum_to_lane<RB>(s, lane);  // lane b: row b's sum (eight rows: one halving exchange)
            if (a.out_key) {  // first minimum wins (strict <, ivf_flat.clj:86-89) = smallest (distance, row)
                uint64_t kmin = ~0ull;
#pragma unroll
                for (int b = 0; b < RB; b++) {
                    const int64_t row = base + b;
                    const float sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), b));
                    const float rn = L2M ? 0.0f : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myrn), b));
                    const float dv = L2M ? __builtin_sqrtf(sb) : finish_dist(a.metric, sb, qn, rn);
                    const uint64_t key = row < r1 ? make_key(dv, static_cast<uint32_t>(row - rb0)) : ~0ull;
                    kmin = key < kmin ? key : kmin;
                }
                if (lane == q) best = kmin < best ? kmin : best;
            } else {
                // lane b takes row b's distance: RB consecutive floats of the query's array
                // lane b holds row b's norm already: the distance is finished where it is stored
                if (lane < RB && base + lane < r1)
                    a.out[ob_s[q] + (base + lane - rb0)] = L2M ? __builtin_sqrtf(mine) : finish_dist(a.metric, mine, qn, myrn);
            }
        }
    }
    if (a.out_key && lane < cnt && best != ~0ull) atomicMin(a.out_key + (static_cast<int64_t>(g) * tq + lane), best);
}

}  // namespace hg
