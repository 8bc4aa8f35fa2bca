// Which rows a batch's selection can read the score of (src/layers.py:226-237 scores exactly `unique_nodes` = the batch and its
// neighbours): one byte map per batch, built per epoch from the batches' PLANS (pcg_plan_batches / pcg_plan_epochs) - a plan
// slot holds every row's record (CSR start, degree, centre) and the rows' degree-tier queues, so nothing here walks indptr.
//
//   mark_sweep_kernel   one workgroup per (batch, id range of 2^shift nodes): the batch's LONG rows (> 512 neighbours: the q16
//                       and q4 queues - most of a power-law batch's neighbours) restricted to the range - two binary searches per row, the rows are sorted by id - are swept
//                       into a bitmap of the range in LDS (ds_or: no global scatter, no global atomic), then the bitmap is
//                       expanded into the byte map with coalesced 16-byte stores.  EVERY byte of every map is written here
//                       (one or zero): no zeroing pass.
//   mark_short_kernel   afterwards: the centres, the train positives, and the rows of <= 512 neighbours (one byte store per
//                       neighbour; rows of <= 64 four to a wave - a 16-lane group each -, longer ones a wave each).
//
// Why: at 10 M nodes / 200 M edges a batch of 4096 degree-biased picks has ~7 M neighbours, most of them in a few hundred hub
// rows; marking them with scattered one-byte stores (mark_long_kernel, score.hip: every store its own partially written line)
// took 0.88 ms per 20-batch epoch + 0.46 ms for the other rows + the zeroing pass - 67 us per step, 17 % of it.
#include "choose.h"

namespace pcg {

struct MarkPlanArgs {
    const int32_t *nodes;
    int32_t n_total, B, B_tail, n_rel, n_slots, tail_slot;
    const int32_t *indices[PCG_MAX_REL];
    Workspace w_full, w_tail;          // slot 0's plan part carved for B / for the shorter last batch (pointers only)
    int64_t plan_stride;
    unsigned char *maps;
    int64_t map_stride, n_nodes;
    const int32_t *train_pos;
    int32_t n_pos;
    int32_t shift, n_ranges;           // a range = 2^shift node ids
};

constexpr int SWEEP_THREADS = 1024;
constexpr int SWEEP_WAVES = SWEEP_THREADS / PCG_WAVE;
constexpr int SWEEP_PIECE = 4 * PCG_WAVE;      // ids a wave takes at a time: four loads of 64 in flight
constexpr int SWEEP_ROWS = 512;                // rows a workgroup locates at a time (one per thread of its first eight waves)

// first positions in nbr[0, d) whose ids are >= x0 / >= x1: two binary searches side by side (their probes are independent
// loads: one memory round trip per level for both; every probe unconditional - index clamped)
__device__ __forceinline__ void lower_bound_ids2(const int32_t *__restrict__ nbr, int d, int64_t x0, int64_t x1, int &p0, int &p1) {
    int lo0 = 0, hi0 = d, lo1 = 0, hi1 = d;
    while (lo0 < hi0 || lo1 < hi1) {
        const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
        const int64_t v0 = nbr[m0 < d ? m0 : d - 1], v1 = nbr[m1 < d ? m1 : d - 1];
        if (lo0 < hi0) {
            if (v0 < x0) lo0 = m0 + 1;
            else hi0 = m0;
        }
        if (lo1 < hi1) {
            if (v1 < x1) lo1 = m1 + 1;
            else hi1 = m1;
        }
    }
    p0 = lo0;
    p1 = lo1;
}

__global__ void __launch_bounds__(SWEEP_THREADS) mark_sweep_kernel(const MarkPlanArgs a) {
    extern __shared__ __align__(16) uint32_t bm[];                       // 2^shift bits
    __shared__ long long s_beg[SWEEP_ROWS];
    __shared__ int s_len[SWEEP_ROWS], s_rel[SWEEP_ROWS], s_pre[SWEEP_ROWS];
    __shared__ int s_wave[SWEEP_WAVES];
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    const int s = (int)blockIdx.x / a.n_ranges, q = (int)blockIdx.x - s * a.n_ranges;
    Workspace w = s == a.tail_slot ? a.w_tail : a.w_full;
    shift_plan(w, (int64_t)s * a.plan_stride);
    const int Bs = s == a.tail_slot ? a.B_tail : a.B;
    const int64_t lo_id = (int64_t)q << a.shift, hi_id = lo_id + (1ll << a.shift);
    const int words = 1 << (a.shift - 5);
    for (int i = tid; i < words; i += SWEEP_THREADS) bm[i] = 0u;
    // the rows swept here: the two longest degree tiers (> 512 neighbours: the q16 and q4 queues)
    const int n16 = (int)w.counters[C_N16], n_long = n16 + (int)w.counters[C_N4];
    for (int j0 = 0; j0 < n_long; j0 += SWEEP_ROWS) {
        // this block of <= 512 rows: a thread finds its row's stretch inside the range (the ids of a row ascend)
        const int j = j0 + tid;
        int len = 0;
        if (tid < SWEEP_ROWS && j < n_long) {
            const int row = j < n16 ? w.q16[j] : w.q4[j - n16];
            const RowRec p = w.recs[row];
            const int r = row / Bs;
            const int32_t *nbr = nullptr;
            for (int rr = 0; rr < PCG_MAX_REL; ++rr)                      // (a kernel-argument array: indexed by a constant)
                if (rr == r) nbr = a.indices[rr];
            nbr += p.start;
            int p0, p1;
            lower_bound_ids2(nbr, p.d, lo_id, hi_id, p0, p1);
            s_beg[tid] = p.start + p0;
            s_rel[tid] = r;
            len = p1 - p0;
        }
        if (tid < SWEEP_ROWS) s_len[tid] = len;
        // inclusive prefix of the rows' piece counts
        const int pieces = (len + SWEEP_PIECE - 1) / SWEEP_PIECE;
        const int inc = wave_incl_scan(pieces, lane);
        if (lane == PCG_WAVE - 1) s_wave[wave] = inc;
        __syncthreads();
        int base = 0, total = 0;
        for (int x = 0; x < SWEEP_ROWS / PCG_WAVE; ++x) {
            const int t = s_wave[x];
            if (x < wave) base += t;
            total += t;
        }
        if (tid < SWEEP_ROWS) s_pre[tid] = base + inc;
        __syncthreads();
        // the block's pieces dealt out over the waves; a wave finds its piece's row by bisection in the prefix
        for (int pc = wave; pc < total; pc += SWEEP_WAVES) {
            int lo = 0, hi = SWEEP_ROWS - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_pre[mid] > pc) hi = mid;
                else lo = mid + 1;
            }
            const int rw = lo;
            const int c = pc - (rw > 0 ? s_pre[rw - 1] : 0);
            const int r = s_rel[rw], n = s_len[rw];
            const int64_t jb = s_beg[rw];
            const int32_t *nbr = nullptr;
            for (int rr = 0; rr < PCG_MAX_REL; ++rr)
                if (rr == r) nbr = a.indices[rr];
            int32_t idv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {                                  // (unconditional loads: clamped)
                const int i = c * SWEEP_PIECE + u * PCG_WAVE + lane;
                idv[u] = nbr[jb + (i < n ? i : n - 1)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = c * SWEEP_PIECE + u * PCG_WAVE + lane;
                const int64_t bit = (int64_t)idv[u] - lo_id;
                if (i < n && bit >= 0 && bit < (1ll << a.shift)) atomicOr(&bm[bit >> 5], 1u << (bit & 31));
            }
        }
        __syncthreads();
    }
    __syncthreads();
    // the range's bytes: one per node, 16 per store; beyond the table (the map's padding) zero
    unsigned char *__restrict__ map = a.maps + (int64_t)s * a.map_stride;
    for (int i = tid; i < 2 * words; i += SWEEP_THREADS) {               // half a word (16 nodes) per thread and turn
        const int64_t at = lo_id + 16ll * i;
        if (at >= a.map_stride) break;
        const uint32_t h = (bm[i >> 1] >> (16 * (i & 1))) & 0xFFFFu;
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t x = (h >> (4 * k)) & 0xFu;
            o[k] = (x & 1u) | ((x & 2u) << 7) | ((x & 4u) << 14) | ((x & 8u) << 21);
        }
        *reinterpret_cast<uint4 *>(map + at) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// the rows of <= 512 neighbours, the centres, the train positives: one byte store each.  Items of a batch:
//   [groups of four rows of the qa queue | groups of four rows of the q0 queue | rows of q1]
__global__ void __launch_bounds__(256) mark_short_kernel(const MarkPlanArgs a) {
    const int lane = lane_id();
    {
        const int64_t nthreads = (int64_t)gridDim.x * gridDim.y * blockDim.x;
        const int64_t tid = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        for (int64_t i = tid; i < (int64_t)a.n_pos * a.n_slots; i += nthreads) {
            const int slot = (int)(i / a.n_pos);
            const int32_t v = a.train_pos[i - (int64_t)slot * a.n_pos];
            if ((uint32_t)v < (uint64_t)a.n_nodes) a.maps[(int64_t)slot * a.map_stride + v] = 1;
        }
        for (int64_t i = tid; i < a.n_total; i += nthreads) {            // the centres' own scores are read too
            const int32_t v = a.nodes[i];
            if ((uint32_t)v < (uint64_t)a.n_nodes) a.maps[(i / a.B) * a.map_stride + v] = 1;
        }
    }
    // blockIdx.y = the batch: its queue lengths are read once per workgroup
    const int s = (int)blockIdx.y;
    Workspace w = s == a.tail_slot ? a.w_tail : a.w_full;
    shift_plan(w, (int64_t)s * a.plan_stride);
    const int Bs = s == a.tail_slot ? a.B_tail : a.B;
    const int na = (int)w.counters[C_NA], n0 = (int)w.counters[C_N0], n1 = (int)w.counters[C_N1];
    const int ga = (na + 3) >> 2, g0 = (n0 + 3) >> 2;
    unsigned char *__restrict__ map = a.maps + (int64_t)s * a.map_stride;
    const int slot_waves = (int)gridDim.x * (int)(blockDim.x >> 6);
    for (int it = (int)blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6); it < ga + g0 + n1; it += slot_waves) {
        int j = it;
        if (j < ga + g0) {
            // four rows of <= 64 neighbours, a 16-lane group each
            const bool in_a = j < ga;
            const int grp = lane >> 4, sub = lane & 15;
            const int qi = 4 * (in_a ? j : j - ga) + grp, qn = in_a ? na : n0;
            const int32_t *qq = in_a ? w.qa : w.q0;
            const int row = qq[qi < qn ? qi : qn - 1];
            const RowRec p = w.recs[row];
            const int r = row / Bs;
            const int32_t *nbr = nullptr;
            for (int rr = 0; rr < PCG_MAX_REL; ++rr)
                if (rr == r) nbr = a.indices[rr];
            nbr += p.start;
            const int d = qi < qn ? p.d : 0;
            int32_t idv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = u * 16 + sub;
                idv[u] = nbr[i < d ? i : (d > 0 ? d - 1 : 0)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (u * 16 + sub < d && (uint32_t)idv[u] < (uint64_t)a.n_nodes) map[idv[u]] = 1;
            continue;
        }
        j -= ga + g0;
        const int row = w.q1[j];
        const RowRec p = w.recs[row];
        const int r = row / Bs;
        const int32_t *nbr = nullptr;
        for (int rr = 0; rr < PCG_MAX_REL; ++rr)
            if (rr == r) nbr = a.indices[rr];
        nbr += p.start;
        const int d = p.d;
        constexpr int CU = 4;
        for (int j0 = 0; j0 < d; j0 += CU * PCG_WAVE) {
            int32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int i = j0 + u * PCG_WAVE + lane;
                idv[u] = nbr[i < d ? i : d - 1];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u)
                if (j0 + u * PCG_WAVE + lane < d && (uint32_t)idv[u] < (uint64_t)a.n_nodes) map[idv[u]] = 1;
        }
    }
}

}  // namespace pcg

extern "C" {

/* pcg_mark_touched from the batches' plans: see include/pcgnn.h */
int pcg_mark_touched_planned(const pcg_graph_desc *g, const int32_t *nodes, int32_t n_total, int32_t B, const void *plans,
                             int64_t plan_stride, int64_t list_capacity, uint8_t *maps, int64_t map_stride, void *stream) {
    if (!g || !nodes || n_total < 0 || B < 1 || !plans || !maps || g->n_rel < 1 || g->n_rel > PCG_MAX_REL) return PCG_E_ARG;
    if (map_stride < pcg::touched_bytes(g->n_nodes) || (map_stride & 15) != 0 || (reinterpret_cast<uintptr_t>(maps) & 15u) != 0)
        return PCG_E_ARG;
    if (list_capacity < 1 || list_capacity >= (1ll << 31) || plan_stride < 0 || (plan_stride & 255) != 0) return PCG_E_ARG;
    if (n_total == 0) return PCG_OK;
    pcg::MarkPlanArgs a;
    a.nodes = nodes;
    a.n_total = n_total;
    a.B = B;
    a.n_slots = (n_total + B - 1) / B;
    a.B_tail = n_total - (a.n_slots - 1) * B;
    a.tail_slot = a.B_tail == B ? -1 : a.n_slots - 1;
    a.n_rel = g->n_rel;
    for (int r = 0; r < PCG_MAX_REL; ++r) {
        a.indices[r] = r < g->n_rel ? g->indices[r] : nullptr;
        if (r < g->n_rel && !a.indices[r]) return PCG_E_ARG;
    }
    unsigned char *pl = static_cast<unsigned char *>(const_cast<void *>(plans));
    pcg::carve(g, B, list_capacity, pl, pl, &a.w_full);               // (only the plan part's pointers are used)
    pcg::carve(g, a.B_tail, list_capacity, pl, pl, &a.w_tail);
    if (a.n_slots > 1 && plan_stride < pcg::carve(g, B, list_capacity, nullptr, nullptr, nullptr).plan_bytes) return PCG_E_ARG;
    a.plan_stride = plan_stride;
    a.maps = maps;
    a.map_stride = map_stride;
    a.n_nodes = g->n_nodes;
    a.train_pos = g->train_pos;
    a.n_pos = g->train_pos ? g->n_pos : 0;
    // the id ranges: 2^shift nodes each, shift in [16, 19] (8 .. 64 KB of LDS: two workgroups per CU): the widest that still gives
    // every CU a workgroup - every workgroup locates every long row of its batch in its range (two binary searches per row: random
    // loads, the pass's bottleneck: 1540 workgroups of 2^17 ids took 462 us per epoch at 10 M nodes), so fewer, wider ranges win
    int shift = 19;
    while (shift > 16 && (int64_t)a.n_slots * ((map_stride + (1ll << shift) - 1) >> shift) < 256) --shift;
    a.shift = shift;
    a.n_ranges = (int)((map_stride + (1ll << shift) - 1) >> shift);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t smem = (size_t)1 << (shift - 3);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pcg::mark_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                64 * 1024) != hipSuccess)
            return PCG_E_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(pcg::mark_sweep_kernel, dim3(a.n_slots * a.n_ranges), dim3(pcg::SWEEP_THREADS), smem, st, a);
    PCG_LAUNCH_CHECK();
    // per batch: as many workgroups as its rows could need (a wave per row of > 64 neighbours, per four shorter ones), the whole
    // launch a few thousand
    int per_slot = (g->n_rel * B + 7) / 8;
    const int most = (4096 + a.n_slots - 1) / a.n_slots;
    per_slot = per_slot > most ? most : (per_slot < 1 ? 1 : per_slot);
    hipLaunchKernelGGL(pcg::mark_short_kernel, dim3(per_slot, a.n_slots), dim3(256), 0, st, a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
