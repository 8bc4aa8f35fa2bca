// choose + aggregate for gfx950: the PC-GNN hot path.  This file: the plan and the entry points.
//
//   plan     per (relation r, centre b) row the kept-count bound cap = (deg > k+1 ? k : deg) + m (+1), exclusive
//            offsets into the selection list, five degree-tier queues and the 128-entry chunk table of the gather, a
//            32-byte record per row.  One workgroup (plan_kernel), two passes of many (plan_count / plan_write), or the
//            same two passes riding along the score pass and the train-pos sort (front_a / front_b: pcg_step_front).
//   select   select.hip: ONE persistent launch chooses every row's neighbours (+ minority picks) into its list region
//   gather   gather.hip: chip-wide balanced segmented mean over the lists
//
// Reference lines replaced: src/layers.py:217-219, 246-262, 587-624, 633-738;
// src/graphsage.py:62-96, 200-232 (keep-all + add_self + sqrt normalisation).
#include <limits.h>
#include <stddef.h>

#include "choose.h"

namespace pcg {

__device__ __forceinline__ RowRec row_plan(const ChooseArgs &a, int row, int64_t node_off = 0) {
    RowRec p;
    const int r = row / a.B;
    const int64_t b = node_off + (row - r * a.B);
    p.node = a.nodes[b];
    p.start = a.g.indptr[r][p.node];
    p.d = (int)(a.g.indptr[r][p.node + 1] - p.start);
    p.k = (int)ceil((double)p.d * a.thr[r]);             // layers.py:260
    p.m = 0;
    if (a.train_flag && a.labels[b] == 1) {              // layers.py:675
        p.m = (int)((double)p.k * a.rho[r]);             // layers.py:681
        if (p.m > a.g.n_pos) p.m = a.g.n_pos;
        if (p.m < 0) p.m = 0;
    }
    p.lbeg = 0;
    p.chunk0 = 0;
    return p;
}

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
// four of the five tier counts of a thread / tile (each <= 4096) share one 64-bit word, 16 bits each: one scan;
// the fifth (rows of <= 16 neighbours) is scanned on its own
// tier of a row: 0 = four-per-wave group (<= 16 neighbours and nothing to do after the selection), 1 = one key per lane
// (<= 64; also the short rows that go on to minority picks / the self union: they get a wave of their own instead of
// queueing behind each other inside a group), 2 = one wave (<= 512), 3 / 4 = workgroup rows (<= 4096 / longer)
__device__ __forceinline__ int row_tier(int d, bool tail) {
    return (d <= TA_CAP && !tail) ? 0 : d <= TB_CAP ? 1 : d <= T1_CAP ? 2 : d <= T4_CAP ? 3 : 4;
}
__device__ __forceinline__ long long tier_word(int tier) { return tier == 0 ? 0ll : 1ll << (16 * (tier - 1)); }
struct TierCounts {
    int na, n0, n1, n4, n16;
};
__device__ __forceinline__ TierCounts tier_unpack(long long w, int na) {
    TierCounts t;
    t.na = na;
    t.n0 = (int)(w & 0xFFFF);
    t.n1 = (int)((w >> 16) & 0xFFFF);
    t.n4 = (int)((w >> 32) & 0xFFFF);
    t.n16 = (int)((w >> 48) & 0xFFFF);
    return t;
}
__device__ __forceinline__ void tier_add(TierCounts &a, const TierCounts &b) {
    a.na += b.na; a.n0 += b.n0; a.n1 += b.n1; a.n4 += b.n4; a.n16 += b.n16;
}
__device__ __forceinline__ void tier_push(const Workspace &w, int tier, int row, TierCounts &o) {
    if (tier == 0) w.qa[o.na++] = row;
    else if (tier == 1) w.q0[o.n0++] = row;
    else if (tier == 2) w.q1[o.n1++] = row;
    else if (tier == 3) w.q4[o.n4++] = row;
    else w.q16[o.n16++] = row;
}
__device__ __forceinline__ void tier_finish(const Workspace &w, const TierCounts &t, bool overflow) {
    w.counters[C_NA] = overflow ? 0 : t.na;
    w.counters[C_N0] = overflow ? 0 : t.n0;
    w.counters[C_N1] = overflow ? 0 : t.n1;
    w.counters[C_N4] = overflow ? 0 : t.n4;
    w.counters[C_N16] = overflow ? 0 : t.n16;
    for (int i = 0; i < 8; ++i) w.heads[16 * i] = 0;      // the select kernel's work-queue heads
    w.heads[13] = w.heads[14] = 0;                        // ... and the long-row kernel's (departures, cursor)
}
// exclusive scan of one value per thread over the block; returns the block total through `total`
template <typename T>
__device__ __forceinline__ T block_excl_scan(T v, T *lds /* >= waves */, T &total) {
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    T inc = v;
    for (int o = 1; o < PCG_WAVE; o <<= 1) {
        const T t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == PCG_WAVE - 1) lds[wave] = inc;
    __syncthreads();
    T pre = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        const T x = lds[w];
        if (w < wave) pre += x;
        tot += x;
    }
    __syncthreads();
    total = tot;
    return pre + inc - v;
}

// The plan's four per-row quantities (list entries, gather chunks, tier membership) scanned over the block with ONE pair of
// barriers: 32-bit DPP scans inside every wave (the tier of a row is a one in one of five 8-bit counters - a wave has at most
// 64 rows), the waves' totals through LDS as one int4 each, the list offset widened to 64 bits across waves.
struct PlanScan {
    long long cap;
    int chunk;
    TierCounts t;
};
__device__ __forceinline__ void plan_scan_add(PlanScan &s, const int4 x) {
    s.cap += (long long)(uint32_t)x.x;
    s.chunk += x.y;
    s.t.na += x.z & 0xFF; s.t.n0 += (x.z >> 8) & 0xFF; s.t.n1 += (x.z >> 16) & 0xFF; s.t.n4 += (x.z >> 24) & 0xFF;
    s.t.n16 += x.w;
}
// tier < 0: the thread has no row.  pre = sums over the block's threads before this one, tot = over the whole block
__device__ __forceinline__ void block_excl_scan_plan(int cap, int nch, int tier, int4 *lds /* >= waves */, PlanScan &pre,
                                                     PlanScan &tot) {
    const int lane = lane_id(), wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int w1 = (tier >= 0 && tier < 4) ? 1 << (8 * tier) : 0, w2 = tier == 4 ? 1 : 0;
    const int4 inc = make_int4(wave_incl_scan(cap, lane), wave_incl_scan(nch, lane), wave_incl_scan(w1, lane),
                               wave_incl_scan(w2, lane));
    if (lane == PCG_WAVE - 1) lds[wave] = inc;
    __syncthreads();
    pre = {0, 0, {0, 0, 0, 0, 0}};
    tot = {0, 0, {0, 0, 0, 0, 0}};
    for (int w = 0; w < nw; ++w) {
        const int4 x = lds[w];
        if (w < wave) plan_scan_add(pre, x);
        plan_scan_add(tot, x);
    }
    __syncthreads();
    plan_scan_add(pre, make_int4(inc.x - cap, inc.y - nch, inc.z - w1, inc.w - w2));
}

// chunk descriptors of one row: {row, first list entry, entries in use, chunks of the row}.  The plan writes the
// capacity share; the select kernel overwrites .z with what the row actually uses (so nothing has to fill the tail).
__device__ __forceinline__ void write_chunk_desc(const Workspace &w, int row, int o_chunk, long long o_cap, int cap) {
    const int nch = (cap + CHUNK - 1) / CHUNK;
    for (int j = 0; j < nch; ++j)
        w.chunk_desc[o_chunk + j] = make_int4(row, (int)o_cap + j * CHUNK, cap - j * CHUNK < CHUNK ? cap - j * CHUNK : CHUNK, nch);
}

// the whole plan of one batch by ONE workgroup of PLAN_THREADS threads (tiles of PLAN_THREADS * PLAN_PER rows)
__device__ __forceinline__ void plan_block_body(const ChooseArgs &a) {
    __shared__ int lds[PLAN_THREADS / PCG_WAVE];
    __shared__ long long lds64[PLAN_THREADS / PCG_WAVE];
    // per-relation constants in LDS: a per-lane relation index then costs one ds_read instead of a
    // waterfall loop over the kernel-argument arrays (which also serialised every load behind it)
    __shared__ const int64_t *t_indptr[PCG_MAX_REL];
    __shared__ double t_thr[PCG_MAX_REL], t_rho[PCG_MAX_REL];
    if (threadIdx.x < PCG_MAX_REL) {
        t_indptr[threadIdx.x] = a.g.indptr[threadIdx.x < a.g.n_rel ? threadIdx.x : 0];
        t_thr[threadIdx.x] = a.thr[threadIdx.x];
        t_rho[threadIdx.x] = a.rho[threadIdx.x];
    }
    __syncthreads();
    // branch-free optional inputs
    const int32_t *lab_ptr = (a.train_flag && a.labels) ? a.labels : a.nodes;
    const int lab_on = (a.train_flag && a.labels) ? 1 : 0;
    const int rows = a.g.n_rel * a.B;
#define PLAN_STAMP(slot) do { if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)rows * 8 + (slot)] = wall_clock64(); } while (0)
    PLAN_STAMP(0);
    long long run_cap = 0;
    int run_chunk = 0;
    TierCounts run = {0, 0, 0, 0, 0};
    bool overflow = false;
    for (int tile0 = 0; tile0 < rows; tile0 += PLAN_THREADS * PLAN_PER) {
        const int r0 = tile0 + threadIdx.x * PLAN_PER;
        RowRec rec[PLAN_PER];
        int cap[PLAN_PER];
        long long cap_sum = 0;
        int chunk_sum = 0, na_sum = 0;
        long long tiers = 0;
        // the dependent loads of the 4 rows are issued level by level, not row by row
        int nodev[PLAN_PER], labv[PLAN_PER], rel[PLAN_PER], bidx[PLAN_PER];
        long long s0v[PLAN_PER], s1v[PLAN_PER];
#pragma unroll
        for (int i = 0; i < PLAN_PER; ++i) {
            const int row = r0 + i < rows ? r0 + i : rows - 1;
            rel[i] = row / a.B;
            bidx[i] = row - rel[i] * a.B;
            nodev[i] = a.nodes[bidx[i]];
            labv[i] = lab_ptr[bidx[i]];
        }
#pragma unroll
        for (int i = 0; i < PLAN_PER; ++i) {
            const int64_t *ip = t_indptr[rel[i]];
            s0v[i] = ip[nodev[i]];
            s1v[i] = ip[nodev[i] + 1];
        }
#pragma unroll
        for (int i = 0; i < PLAN_PER; ++i) {
            const int row = r0 + i;
            cap[i] = 0;
            if (row < rows) {
                RowRec p;
                p.node = nodev[i];
                p.start = s0v[i];
                p.d = (int)(s1v[i] - s0v[i]);
                p.k = (int)ceil((double)p.d * t_thr[rel[i]]);             // layers.py:260
                p.m = 0;
                if (lab_on && labv[i] == 1) {                             // layers.py:675
                    p.m = (int)((double)p.k * t_rho[rel[i]]);             // layers.py:681
                    if (p.m > a.g.n_pos) p.m = a.g.n_pos;
                    if (p.m < 0) p.m = 0;
                }
                p.lbeg = p.chunk0 = 0;
                rec[i] = p;
                cap[i] = rec_cap(p, a.add_self);
                cap_sum += cap[i];
                chunk_sum += (cap[i] + CHUNK - 1) / CHUNK;
                const int tier = row_tier(p.d, p.m > 0 || a.add_self);
                tiers += tier_word(tier);
                na_sum += tier == 0;
            }
        }
        PLAN_STAMP(1);
        // four scans: four of the tier counts (each <= 4 per thread, <= 4096 per tile) share one word
        long long t_cap, t_tiers;
        int t_chunk, t_na;
        long long o_cap = run_cap + block_excl_scan<long long>(cap_sum, lds64, t_cap);
        int o_chunk = run_chunk + block_excl_scan(chunk_sum, lds, t_chunk);
        const int o_na = block_excl_scan(na_sum, lds, t_na);
        TierCounts o = tier_unpack(block_excl_scan<long long>(tiers, lds64, t_tiers), o_na);
        tier_add(o, run);
        run_cap += t_cap; run_chunk += t_chunk;
        tier_add(run, tier_unpack(t_tiers, t_na));
        overflow = overflow || run_cap > a.w.list_capacity || (long long)run_chunk > a.w.chunk_cap;
        PLAN_STAMP(2);
#pragma unroll
        for (int i = 0; i < PLAN_PER; ++i) {
            const int row = r0 + i;
            if (row >= rows) continue;
            const int nch = (cap[i] + CHUNK - 1) / CHUNK;
            a.w.row_begin[row] = o_cap;
            a.w.chunk_begin[row] = o_chunk;
            rec[i].lbeg = (int)o_cap;
            rec[i].chunk0 = o_chunk;
            a.w.recs[row] = rec[i];
            if (!overflow) {
                write_chunk_desc(a.w, row, o_chunk, o_cap, cap[i]);
                tier_push(a.w, row_tier(rec[i].d, rec[i].m > 0 || a.add_self), row, o);
            }
            o_cap += cap[i];
            o_chunk += nch;
        }
    }
    PLAN_STAMP(3);
    if (threadIdx.x == 0) {
        a.w.row_begin[rows] = run_cap;
        a.w.chunk_begin[rows] = run_chunk;
        tier_finish(a.w, run, overflow);
        a.w.counters[C_NCHUNK] = overflow ? 0 : run_chunk;
        if (overflow && a.status) atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
    }
#undef PLAN_STAMP
}

__global__ void __launch_bounds__(PLAN_THREADS) plan_kernel(const ChooseArgs a) { plan_block_body(a); }

// Large batches (rows > PLAN_THREADS * PLAN_PER): the same plan in two launches of many workgroups.
//   plan_count : every block works out the records of its rows and their totals
//   plan_write : every block adds up the totals of the blocks before it (a few dozen values), then scans its
//                own rows and writes offsets / queues / chunk table exactly as the single-block kernel does
struct PlanTotals {
    long long cap;
    int chunk, n0, n1, n4, n16, na;
    int pad[2];
};
static_assert(sizeof(PlanTotals) <= 64, "the workspace carve reserves 64 bytes per plan block");

// pass 1, workgroup `block` of THREADS threads, one row per thread
// (w: the workspace the plan goes to - a.w, or a plan slot behind it; node_off: where the batch starts in a.nodes / a.labels)
template <int THREADS>
__device__ __forceinline__ void plan_count_body(const ChooseArgs &a, const Workspace &w, PlanTotals *totals, int block,
                                                int64_t node_off = 0) {
    __shared__ int4 lds4[THREADS / PCG_WAVE];
    const int rows = a.g.n_rel * a.B;
    const int row = block * THREADS + (int)threadIdx.x;
    int cap = 0, tier = -1;
    if (row < rows) {
        const RowRec p = row_plan(a, row, node_off);
        w.recs[row] = p;
        cap = rec_cap(p, a.add_self);
        tier = row_tier(p.d, p.m > 0 || a.add_self);
    }
    PlanScan pre, tot;
    block_excl_scan_plan(cap, (cap + CHUNK - 1) / CHUNK, tier, lds4, pre, tot);
    if (threadIdx.x == 0) {
        PlanTotals t;
        t.cap = tot.cap;
        t.chunk = tot.chunk;
        t.n0 = tot.t.n0; t.n1 = tot.t.n1; t.n4 = tot.t.n4; t.n16 = tot.t.n16; t.na = tot.t.na;
        t.pad[0] = t.pad[1] = 0;
        totals[block] = t;
    }
}

// pass 2, workgroup `block` of THREADS threads (one row per thread); pass 1 ran n_count_blocks workgroups of
// COUNT_THREADS rows each (THREADS is a multiple of COUNT_THREADS)
template <int THREADS, int COUNT_THREADS>
__device__ __forceinline__ void plan_write_body(const ChooseArgs &a, const Workspace &w, const PlanTotals *totals, int block,
                                                int n_count_blocks) {
    __shared__ int4 lds4[THREADS / PCG_WAVE];
    const int rows = a.g.n_rel * a.B;
    long long run_cap = 0, all_cap = 0;
    int run_chunk = 0, all_chunk = 0;
    TierCounts run = {0, 0, 0, 0, 0}, all = {0, 0, 0, 0, 0};
    // the totals of the count blocks before this block's rows, and of all of them: wave 0 reads them a lane each (one load
    // latency; a loop of uniform loads is one scalar-load round trip per count block - 48 of them at B = 4096) and adds them up
    // across lanes; everybody picks the 14 sums up from LDS
    const int before = block * (THREADS / COUNT_THREADS);
    __shared__ long long s_sum[2][8];
    // this thread's row record is requested first (unconditionally, index clamped): its latency hides behind the totals
    const int rows_ = a.g.n_rel * a.B;
    const int row_ = block * THREADS + (int)threadIdx.x;
    const RowRec rec_early = w.recs[row_ < rows_ ? row_ : rows_ - 1];
    if (threadIdx.x < PCG_WAVE) {
        long long v[2][7] = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}};
        for (int bk = (int)threadIdx.x; bk < n_count_blocks; bk += PCG_WAVE) {
            const PlanTotals t = totals[bk];
            const long long q[7] = {t.cap, t.chunk, t.na, t.n0, t.n1, t.n4, t.n16};
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                v[1][j] += q[j];
                v[0][j] += bk < before ? q[j] : 0;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                long long x = v[h][j];
                for (int o = 1; o < PCG_WAVE; o <<= 1) x += __shfl_xor(x, o);
                if (threadIdx.x == 0) s_sum[h][j] = x;
            }
    }
    __syncthreads();
    run_cap = s_sum[0][0]; run_chunk = (int)s_sum[0][1];
    run = {(int)s_sum[0][2], (int)s_sum[0][3], (int)s_sum[0][4], (int)s_sum[0][5], (int)s_sum[0][6]};
    all_cap = s_sum[1][0]; all_chunk = (int)s_sum[1][1];
    all = {(int)s_sum[1][2], (int)s_sum[1][3], (int)s_sum[1][4], (int)s_sum[1][5], (int)s_sum[1][6]};
    const bool overflow = all_cap > w.list_capacity || (long long)all_chunk > w.chunk_cap;
    const int row = block * THREADS + (int)threadIdx.x;
    const RowRec rec = rec_early;
    const int cap = row < rows ? rec_cap(rec, a.add_self) : 0;
    const int nch = (cap + CHUNK - 1) / CHUNK;
    const int tier = row < rows ? row_tier(rec.d, rec.m > 0 || a.add_self) : -1;
    PlanScan pre, tot;
    block_excl_scan_plan(cap, nch, tier, lds4, pre, tot);
    const long long o_cap = run_cap + pre.cap;
    const int o_chunk = run_chunk + pre.chunk;
    TierCounts o = pre.t;
    tier_add(o, run);
    if (row < rows) {
        w.row_begin[row] = o_cap;
        w.chunk_begin[row] = o_chunk;
        w.recs[row].lbeg = (int)o_cap;
        w.recs[row].chunk0 = o_chunk;
        if (!overflow) {
            write_chunk_desc(w, row, o_chunk, o_cap, cap);
            tier_push(w, tier, row, o);
        }
    }
    if (block == 0 && threadIdx.x == 0) {
        w.row_begin[rows] = all_cap;
        w.chunk_begin[rows] = all_chunk;
        tier_finish(w, all, overflow);
        w.counters[C_NCHUNK] = overflow ? 0 : all_chunk;
        if (overflow && a.status) atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
    }
}

__global__ void __launch_bounds__(PLAN_THREADS) plan_count(const ChooseArgs a, PlanTotals *totals) {
    plan_count_body<PLAN_THREADS>(a, a.w, totals, (int)blockIdx.x);
}
__global__ void __launch_bounds__(PLAN_THREADS) plan_write(const ChooseArgs a, const PlanTotals *totals) {
    plan_write_body<PLAN_THREADS, PLAN_THREADS>(a, a.w, totals, (int)blockIdx.x, (int)gridDim.x);
}

// The plans of ALL batches of an epoch in ONE launch (they depend on the picked ids, their labels and the CSR degrees only - not on
// any parameter - so they do not belong on a step's critical path): batch s = nodes[s * B_full, min((s + 1) * B_full, n_total))
// is planned into plan slot s (slot 0's plan part + s * stride bytes) by its own nb_full workgroups, one row per thread.
// Workgroup j of a batch needs the sums (list entries, chunks, tier counts) over the rows of workgroups 0 .. j - 1.  Every
// workgroup works out its own rows' records and totals first and PUBLISHES the totals (write-through stores, then this launch's
// tag); then it reads its predecessors' totals - lane t of wave 0 waits for workgroup t's tag - and adds them up: one pass of the
// dependent loads (centre, then its two row offsets: ~2 us each, the picks and the offsets come from HBM) and one hand-off,
// whatever the batch size (a workgroup recounting its predecessors' rows itself took 17 us per epoch at batch 1024 and 32-39 us
// at 4096; a count launch + a write launch 7 + 11 us).  Nobody waits on a workgroup that itself waits before publishing, and a
// batch's workgroups have consecutive ids; the wait is bounded (PCG_ST_SYNC_TIMEOUT).  The tag is one more than the launches
// that have planned into this slot so far (heads[H_PLAN_SEQ], moved on by the batch's last workgroup once it has read every
// predecessor).  The batch's last workgroup then knows the batch's totals: it writes the counters, the select kernel's queue
// heads and the overflow verdict.  Rows are guarded one by one against the list / chunk capacities (a prefix is all a workgroup
// knows), so nothing is written out of bounds when a batch overflows; its counters are zeroed then and nothing is selected.
// `full` is carved for B_full rows per relation, `tail` for the last, shorter batch (its layout differs; same slot pitch).
// bump: a device counter incremented once per epoch (the sampler's epoch number: the picks were made before this launch).
// (the arguments are used in place - no per-slot copy, no pointer to them: the relation arrays inside are indexed per lane, and
//  a copy, or an argument whose address is taken, lives in scratch)
constexpr int H_PLAN_SEQ = 12;           // word of Workspace::heads
constexpr int PLAN_SPIN_MAX = 1 << 21;
__device__ __forceinline__ void plan_slot(const ChooseArgs &a, int s, int block, int64_t stride, int64_t node_off) {
    __shared__ int4 lds4[PLAN_THREADS / PCG_WAVE];
    __shared__ long long s_part[PCG_WAVE][8];
    __shared__ long long s_run[8];
    const int rows = a.g.n_rel * a.B;
    const int nb = (rows + PLAN_THREADS - 1) / PLAN_THREADS;
    if (block >= nb) return;
    Workspace w = a.w;                       // the slot's plan part (pointers only)
    shift_plan(w, (int64_t)s * stride);
    const int tid = (int)threadIdx.x, lane = lane_id(), wave = tid >> 6;
    PlanTotals *totals = reinterpret_cast<PlanTotals *>(w.plan_totals);
    const int tag = (int)w.heads[H_PLAN_SEQ] + 1;
#define SLOT_STAMP(k) do { if (a.stamps && tid == 0 && block == nb - 1 && s == 0) a.stamps[(size_t)rows * 8 + (k)] = wall_clock64(); } while (0)
    SLOT_STAMP(0);
    // ---- 1. this workgroup's rows: records, totals; the totals are published at once
    const int row = block * PLAN_THREADS + tid;
    RowRec rec = row_plan(a, row < rows ? row : rows - 1, node_off);
    const int cap = row < rows ? rec_cap(rec, a.add_self) : 0;
    const int nch = (cap + CHUNK - 1) / CHUNK;
    const int tier = row < rows ? row_tier(rec.d, rec.m > 0 || a.add_self) : -1;
    PlanScan pre, tot;
    block_excl_scan_plan(cap, nch, tier, lds4, pre, tot);
    if (tid == 0 && block + 1 < nb) {                                       // (the last workgroup has no successor)
        int *dst = reinterpret_cast<int *>(&totals[block]);
        const int v[8] = {(int)(tot.cap & 0xFFFFFFFFll), (int)(tot.cap >> 32), tot.chunk, tot.t.n0, tot.t.n1, tot.t.n4, tot.t.n16, tot.t.na};
        static_assert(offsetof(PlanTotals, chunk) == 8 && offsetof(PlanTotals, na) == 28 && offsetof(PlanTotals, pad) == 32, "PlanTotals layout");
#pragma unroll
        for (int j = 0; j < 8; ++j) __hip_atomic_store(dst + j, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(dst + 8, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // pad[0]: "published in this launch"
    }
    SLOT_STAMP(1);
    // ---- 2. the predecessors' totals
    long long run_cap = 0;
    int run_chunk = 0;
    TierCounts run = {0, 0, 0, 0, 0};
    if (block > 0) {                         // (workgroup-uniform)
        if (wave == 0) {
            for (int t0 = 0; t0 < block; t0 += PCG_WAVE) {
                const int t = t0 + lane;
                long long q[7] = {0, 0, 0, 0, 0, 0, 0};
                if (t < block) {
                    const int *src = reinterpret_cast<const int *>(&totals[t]);
                    for (int spins = 0;; ++spins) {
                        if (__hip_atomic_load(src + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) break;
                        if (spins >= PLAN_SPIN_MAX) {                        // (cannot happen short of a lost workgroup: reported, not hung)
                            if (a.status) atomicOr(a.status, (uint32_t)PCG_ST_SYNC_TIMEOUT);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    int v[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __hip_atomic_load(src + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q[0] = ((long long)v[1] << 32) | (unsigned int)v[0];
                    q[1] = v[2]; q[2] = v[7]; q[3] = v[3]; q[4] = v[4]; q[5] = v[5]; q[6] = v[6];      // chunk, na, n0, n1, n4, n16
                }
#pragma unroll
                for (int j = 0; j < 7; ++j) s_part[lane][j] = (t0 == 0 ? 0 : s_part[lane][j]) + q[j];
            }
        }
        __syncthreads();
        if (tid < 7) {
            long long sum = 0;
            for (int l = 0; l < PCG_WAVE; ++l) sum += s_part[l][tid];
            s_run[tid] = sum;
        }
        __syncthreads();
        run_cap = s_run[0];
        run_chunk = (int)s_run[1];
        run = {(int)s_run[2], (int)s_run[3], (int)s_run[4], (int)s_run[5], (int)s_run[6]};
    }
    SLOT_STAMP(2);
    // ---- 3. offsets, queues, chunk table
    const long long o_cap = run_cap + pre.cap;
    const int o_chunk = run_chunk + pre.chunk;
    TierCounts o = pre.t;
    tier_add(o, run);
    if (row < rows) {
        w.row_begin[row] = o_cap;
        w.chunk_begin[row] = o_chunk;
        rec.lbeg = (int)(o_cap < (long long)INT_MAX ? o_cap : (long long)INT_MAX);
        rec.chunk0 = o_chunk;
        w.recs[row] = rec;
        if (o_cap + cap <= w.list_capacity && (long long)o_chunk + nch <= w.chunk_cap) {
            write_chunk_desc(w, row, o_chunk, o_cap, cap);
            tier_push(w, tier, row, o);
        }
    }
    if (block == nb - 1 && tid == 0) {       // the batch's last workgroup: the totals
        const long long all_cap = run_cap + tot.cap;
        const int all_chunk = run_chunk + tot.chunk;
        TierCounts all = tot.t;
        tier_add(all, run);
        const bool overflow = all_cap > w.list_capacity || (long long)all_chunk > w.chunk_cap;
        w.row_begin[rows] = all_cap;
        w.chunk_begin[rows] = all_chunk;
        tier_finish(w, all, overflow);
        w.counters[C_NCHUNK] = overflow ? 0 : all_chunk;
        w.heads[H_PLAN_SEQ] = (uint32_t)tag;                                 // (every predecessor has published - and read the old value long ago)
        if (overflow && a.status) atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
    }
    SLOT_STAMP(3);
#undef SLOT_STAMP
}
// Several epochs' batches in one launch (pcg_plan_epochs): slot s = batch s % slots_per_epoch of epoch s / slots_per_epoch, whose
// picks start at nodes[epoch * epoch_nodes]; an epoch's last batch (tail_b, or -1) may be the shorter one.  bump += bump_by.
__global__ void __launch_bounds__(PLAN_THREADS) plan_batches_kernel(const ChooseArgs full, const ChooseArgs tail, int n_slots,
                                                                    int tail_b, int64_t stride, int nb_full,
                                                                    unsigned long long *__restrict__ bump, int slots_per_epoch,
                                                                    int64_t epoch_nodes, int bump_by) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && bump) bump[0] += (unsigned long long)bump_by;
    const int s = (int)blockIdx.x / nb_full, block = (int)blockIdx.x - s * nb_full;
    if (s >= n_slots) return;
    const int e = s / slots_per_epoch, b = s - e * slots_per_epoch;
    const int64_t node_off = (int64_t)e * epoch_nodes + (int64_t)b * full.B;
    if (b == tail_b) plan_slot(tail, s, block, stride, node_off);
    else plan_slot(full, s, block, stride, node_off);
}

// The front of a training step in two launches instead of four: the plan's two passes ride along the score pass
// and the train-pos sort (both have idle CUs at dataset scale, and the plan needs neither's result):
//   front_a: [plan pass 1 workgroups | score_table workgroups]          (256 threads)
//   front_b: [plan pass 2 workgroups | rank-sort workgroups]            (1024 threads)
//            + [Adam blocks]: the previous training step's deferred parameter update (everything but the label classifier,
//              whose 2F + 2 parameters - the ones this pass reads - that step's dense kernel has updated itself)
__global__ void __launch_bounds__(FRONT_COUNT_THREADS) front_a_kernel(const ChooseArgs a, PlanTotals *totals, int n_plan_blocks,
                                                                      int n_key_blocks, uint64_t *__restrict__ raw_keys,
                                                                      const float *__restrict__ W, const float *__restrict__ bias,
                                                                      int64_t row_begin, int64_t row_end, float *__restrict__ s0,
                                                                      const DeferredAdam ad, int n_adam_blocks,
                                                                      const int32_t *__restrict__ row_ids, int64_t pos_row_base,
                                                                      const unsigned char *__restrict__ touched) {
    __shared__ float part[4][PCG_WAVE];
    __shared__ unsigned short sel[4 * MARK_GROUP];       // score_marked_body
    const int b = (int)blockIdx.x;
    if (a.sort_done && b == 0 && threadIdx.x == 0) a.sort_done[0] = 0u;
    if (b < n_plan_blocks)
        plan_count_body<FRONT_COUNT_THREADS>(a, a.w, totals, b);
    else if (b < n_plan_blocks + n_key_blocks)      // the train positives' sort keys, from their feature rows
        pos_key_body(a.g.X, a.g.feat_dim, a.g.feat_stride, W, bias, a.g.train_pos, a.g.n_pos, raw_keys, b - n_plan_blocks,
                     n_key_blocks, pos_row_base);
    else if (b < n_plan_blocks + n_key_blocks + n_adam_blocks) {
        if (ad.pending[0] == 1u)                    // (one word, the same for every thread)
            adam_reduce_body(ad.theta, ad.m, ad.v, ad.slabs, (int)ad.pending[1], ad.n_params, 0, ad.p_end, ad.step_counter, ad.h,
                             nullptr, 1, b - n_plan_blocks - n_key_blocks, part);
    } else if (touched)       // only the rows the batch's selection can read (the whole table: row_begin == 0, no row_ids)
        score_marked_body(a.g.X, a.g.feat_dim, a.g.feat_stride, W, bias, row_end, s0, touched,
                          b - n_plan_blocks - n_key_blocks - n_adam_blocks,
                          (int)gridDim.x - n_plan_blocks - n_key_blocks - n_adam_blocks, sel);
    else
        score_table_body(a.g.X, a.g.feat_dim, a.g.feat_stride, W, bias, row_begin, row_end, s0,
                         b - n_plan_blocks - n_key_blocks - n_adam_blocks,
                         (int)gridDim.x - n_plan_blocks - n_key_blocks - n_adam_blocks, row_ids);
}

// The front of a PARTITIONED rank's training step (pc-gnn_amd/dist.py), one launch behind the gradient all-reduce:
//   [Adam on EVERY parameter from the all-reduced gradient, if one is waiting (ad.pending[0] == 1; ad.slabs = that gradient, one
//    "slab") || the train positives' unsorted keys from their replicated rows || the score of every row the rank holds, stored by
//    node id]
// The score and key workgroups need the label classifier AFTER that update, which other workgroups of this launch are storing:
// each works it out for itself - 2F + 2 parameters - from a snapshot of the classifier's parameters and Adam state taken after
// the previous update (pcg_gather_lists_dist refreshes it every step) and the gradient, with the same statement of the
// arithmetic (adam_update), into LDS.  Nothing in the launch waits for anything inside it.
struct FrontDist {
    pcg_graph_desc g;
    int64_t row_begin, row_end;
    float *s0;
    const int32_t *row_ids;
    int64_t pos_row_base;
    uint64_t *raw_keys;
    int32_t n_key_blocks, n_adam_blocks;
    DeferredAdam ad;
    const float *snap;         // [3][nc]: the classifier's parameters, m, v as of the last applied update
    int32_t nc;                // 2F + 2
    int64_t off_clf;
    uint32_t *zero_word;
};
constexpr int FRONT_DIST_NC = 2 * 512 + 2;
__global__ void __launch_bounds__(FRONT_COUNT_THREADS) front_dist_kernel(const FrontDist f) {
    __shared__ float part[4][PCG_WAVE];
    __shared__ float wl[FRONT_DIST_NC + 2];
    const int b = (int)blockIdx.x;
    if (f.zero_word && b == 0 && threadIdx.x == 0) f.zero_word[0] = 0u;
    const bool waiting = f.ad.pending[0] == 1u;                     // (one word, the same for every thread)
    if (b < f.n_adam_blocks) {
        if (waiting)
            adam_reduce_body(f.ad.theta, f.ad.m, f.ad.v, f.ad.slabs, 1, f.ad.n_params, 0, f.ad.p_end, f.ad.step_counter, f.ad.h, nullptr,
                             1, b, part);
        return;
    }
    const float t = (float)f.ad.step_counter[0];
    for (int i = (int)threadIdx.x; i < f.nc; i += FRONT_COUNT_THREADS) {
        float p = f.snap[i];
        if (waiting) {
            float mi, vi;
            p = adam_update(p, f.snap[f.nc + i], f.snap[2 * f.nc + i], f.ad.slabs[f.off_clf + i], t, f.ad.h, mi, vi);
        }
        wl[i] = p;
    }
    __syncthreads();
    const float *W = wl, *bias = wl + (f.nc - 2);
    if (b < f.n_adam_blocks + f.n_key_blocks)
        pos_key_body(f.g.X, f.g.feat_dim, f.g.feat_stride, W, bias, f.g.train_pos, f.g.n_pos, f.raw_keys, b - f.n_adam_blocks,
                     f.n_key_blocks, f.pos_row_base);
    else
        score_table_body(f.g.X, f.g.feat_dim, f.g.feat_stride, W, bias, f.row_begin, f.row_end, f.s0,
                         b - f.n_adam_blocks - f.n_key_blocks, (int)gridDim.x - f.n_adam_blocks - f.n_key_blocks, f.row_ids);
}

__global__ void __launch_bounds__(PLAN_THREADS) front_b_kernel(const ChooseArgs a, const PlanTotals *totals, int n_write_blocks,
                                                               int n_count_blocks, uint64_t *__restrict__ keys, int cap,
                                                               const uint64_t *__restrict__ raw_keys, uint32_t *pending,
                                                               float *__restrict__ center_out, int64_t center_id_offset) {
    __shared__ __align__(16) uint64_t sh[RANK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    if (pending && blockIdx.x == 0 && threadIdx.x == 0) pending[0] = 0u;   // front_a (the launch before) has applied the deferred update
    if ((int)blockIdx.x < n_write_blocks) {
        // (partitioned path) the centres' own scores, looked up by global id: center_out[b] = s0[nodes[b] + offset]
        const int b = (int)blockIdx.x * PLAN_THREADS + (int)threadIdx.x;
        if (center_out && b < a.B) center_out[b] = a.s0[(int64_t)a.nodes[b] + center_id_offset];
        plan_write_body<PLAN_THREADS, FRONT_COUNT_THREADS>(a, a.w, totals, (int)blockIdx.x, n_count_blocks);
    } else
        rank_sort_body(a.s0, a.g.train_pos, a.g.n_pos, cap, keys, (int)blockIdx.x - n_write_blocks, sh, part, raw_keys);
}

// planned: the plan is already in the workspace (pcg_step_front)
static int launch_select(const ChooseArgs &a, hipStream_t st, bool planned = false) {
    const pcg_graph_desc &g = a.g;
    const int rows = g.n_rel * a.B;
    if (planned) {
    } else if (rows <= PLAN_THREADS * PLAN_PER) {
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(PLAN_THREADS), 0, st, a);
        PCG_LAUNCH_CHECK();
    } else {
        const int nb = (rows + PLAN_THREADS - 1) / PLAN_THREADS;       // one row per thread: spread over many CUs
        PlanTotals *tot = reinterpret_cast<PlanTotals *>(a.w.plan_totals);
        hipLaunchKernelGGL(plan_count, dim3(nb), dim3(PLAN_THREADS), 0, st, a, tot);
        PCG_LAUNCH_CHECK();
        hipLaunchKernelGGL(plan_write, dim3(nb), dim3(PLAN_THREADS), 0, st, a, tot);
        PCG_LAUNCH_CHECK();
    }
    return launch_select_rows(a, st);
}

static unsigned long long *g_stamps = nullptr;

}  // namespace pcg

extern "C" {

/* diagnostic: per-row phase timestamps of the select kernels ([rows][8] uint64, wall_clock64 ticks
 * = 10 ns); pass NULL to switch off.  Not part of the product path. */
void pcg_debug_set_stamps(void *ptr) { pcg::g_stamps = static_cast<unsigned long long *>(ptr); }

int64_t pcg_choose_workspace_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity) {
    if (!g || B < 0 || list_capacity < 0 || list_capacity >= (1ll << 31)) return PCG_E_ARG;
    return pcg::carve1(g, B, list_capacity, nullptr, nullptr);
}

int64_t pcg_choose_plan_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity) {
    if (!g || B < 0 || list_capacity < 0 || list_capacity >= (1ll << 31)) return PCG_E_ARG;
    return pcg::carve(g, B, list_capacity, nullptr, nullptr, nullptr).plan_bytes;
}

int64_t pcg_choose_data_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity) {
    if (!g || B < 0 || list_capacity < 0 || list_capacity >= (1ll << 31)) return PCG_E_ARG;
    return pcg::carve(g, B, list_capacity, nullptr, nullptr, nullptr).data_bytes;
}

int64_t pcg_choose_workspace_offset(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, int32_t which) {
    if (!g || B < 0 || list_capacity < 0) return PCG_E_ARG;
    pcg::Workspace w;
    unsigned char *base = reinterpret_cast<unsigned char *>(4096);   // fake base, only differences are used
    pcg::carve1(g, B, list_capacity, base, &w);
    switch (which) {
        case 0: return reinterpret_cast<unsigned char *>(w.row_begin) - base;
        case 1: return reinterpret_cast<unsigned char *>(w.len) - base;
        case 2: return reinterpret_cast<unsigned char *>(w.list) - base;
        case 3: return reinterpret_cast<unsigned char *>(w.chunk_begin) - base;
        case 4: return reinterpret_cast<unsigned char *>(w.chunk_desc) - base;
        case 5: return reinterpret_cast<unsigned char *>(w.counters) - base;
        case 6: return reinterpret_cast<unsigned char *>(w.partial) - base;
        case 7: return reinterpret_cast<unsigned char *>(w.recs) - base;
        default: return PCG_E_ARG;
    }
}

int64_t pcg_sel_capacity_row(int64_t deg, double threshold, double rho, int32_t positive_train, int32_t n_pos,
                             int32_t add_self) {
    const int64_t k = (int64_t)ceil((double)deg * threshold);
    int64_t cap = (deg > k + 1) ? k : deg;
    if (positive_train) {
        int64_t m = (int64_t)((double)k * rho);
        if (m > n_pos) m = n_pos;
        if (m > 0) cap += m;
    }
    return cap + (add_self ? 1 : 0);
}

static int choose_args(pcg::ChooseArgs &a, const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                       const float *s0, const float *center_s0, const uint64_t *pos_keys, const double *thresholds,
                       const double *rho, int32_t train_flag, int32_t add_self, int32_t *cnt, void *workspace,
                       int64_t list_capacity, uint32_t *status, bool plan_only = false, const void *plan = nullptr) {
    if (!nodes || !thresholds || !workspace || !status) return PCG_E_ARG;
    if (!plan_only && !s0) return PCG_E_ARG;      // (the plan reads neither the scores nor the sorted keys)
    if (list_capacity < 1 || list_capacity >= (1ll << 31)) return PCG_E_ARG;
    if (train_flag && !rho) return PCG_E_ARG;
    if (g->n_rel < 1 || g->n_rel > PCG_MAX_REL) return PCG_E_ARG;
    if (train_flag && (!labels || (!plan_only && g->n_pos > 0 && (!pos_keys || !g->train_pos)))) return PCG_E_ARG;
    for (int r = 0; r < g->n_rel; ++r)
        if (!g->indptr[r] || !g->indices[r]) return PCG_E_ARG;
    a.g = *g;
    a.nodes = nodes;
    a.labels = labels;
    a.B = B;
    a.s0 = s0;
    a.center_s0 = center_s0;
    a.center_off = 0;
    a.pos_keys = pos_keys;
    for (int r = 0; r < PCG_MAX_REL; ++r) a.thr[r] = r < g->n_rel ? thresholds[r] : 0.0;
    for (int r = 0; r < PCG_MAX_REL; ++r) a.rho[r] = (r < g->n_rel && rho) ? rho[r] : 0.0;
    a.train_flag = train_flag;
    a.add_self = add_self;
    a.cnt = cnt;
    a.status = status;
    a.stamps = pcg::g_stamps;
    a.raw_keys = nullptr;
    a.sort_out = nullptr;
    a.sort_done = a.rank_acc = a.group_ticket = nullptr;
    a.n_sort = a.sort_cap = a.sort_slices = a.sort_slice_len = 0;
    a.pending_clear = nullptr;
    a.clf.clf_next = nullptr;
    a.n_wg_units = 0;
    pcg::carve1(g, B, list_capacity, static_cast<unsigned char *>(workspace), &a.w,
                static_cast<unsigned char *>(const_cast<void *>(plan)));
    return PCG_OK;
}

// what a *_planned call may add: where the plan lives (null: inside `workspace`), and the in-kernel train-pos sort
struct PlannedExtra {
    const void *plan = nullptr;
    uint32_t *sync_words = nullptr;     // non-null: pos_keys' scratch half holds the UNSORTED keys (pcg_step_scores_train); the
                                        // select kernel sorts them itself ([3] = arrival counter, zero on entry) and clears [1]
    int64_t center_off = 0;
};

static int choose_select(bool planned, const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const float *s0, const float *center_s0, const uint64_t *pos_keys, const double *thresholds,
                         const double *rho, int32_t train_flag, int32_t add_self, int32_t *cnt, void *workspace,
                         int64_t list_capacity, uint32_t *status, void *stream, const PlannedExtra &x = PlannedExtra()) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;  // empty trailing batch (model_handler.py:134 produces one): nothing to do
    if (!cnt) return PCG_E_ARG;
    if (x.plan && !planned) return PCG_E_ARG;
    pcg::ChooseArgs a;
    const int rc = choose_args(a, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                               workspace, list_capacity, status, false, x.plan);
    if (rc != PCG_OK) return rc;
    a.center_off = x.center_off;
    if (x.sync_words) {
        a.pending_clear = x.sync_words + 1;
        if (train_flag && g->n_pos > 0 && g->n_pos <= pcg::RANK_MAX) {
            const int64_t cap = pcg_pos_sort_capacity(g->n_pos) / 2;
            a.sort_out = const_cast<uint64_t *>(pos_keys);
            a.raw_keys = pos_keys + cap;
            a.sort_cap = (int32_t)cap;
            a.n_sort = (g->n_pos + PCG_WAVE - 1) / PCG_WAVE;
            a.sort_done = x.sync_words + 3;
            a.rank_acc = x.sync_words + 4;
            a.group_ticket = x.sync_words + 4 + pcg::RANK_MAX;
        }
    }
    return pcg::launch_select(a, static_cast<hipStream_t>(stream), planned);
}

int pcg_choose_select(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B, const float *s0,
                      const float *center_s0, const uint64_t *pos_keys, const double *thresholds, const double *rho,
                      int32_t train_flag, int32_t add_self, int32_t *cnt, void *workspace, int64_t list_capacity,
                      uint32_t *status, void *stream) {
    return choose_select(false, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                         workspace, list_capacity, status, stream);
}

int pcg_choose_select_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                              const float *s0, const float *center_s0, const uint64_t *pos_keys, const double *thresholds,
                              const double *rho, int32_t train_flag, int32_t add_self, int32_t *cnt, void *workspace,
                              const void *plan, int64_t list_capacity, uint32_t *status, uint32_t *sync_words,
                              int64_t center_id_offset, void *stream) {
    if (center_id_offset < 0 || (center_id_offset > 0 && center_s0)) return PCG_E_ARG;
    PlannedExtra x;
    x.sync_words = sync_words;
    x.center_off = center_id_offset;
    x.plan = plan;
    return choose_select(true, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                         workspace, list_capacity, status, stream, x);
}

/* The plans of all batches of an epoch (or of one batch: n_total <= B): ONE launch per epoch, off every step's critical path. */
int pcg_plan_batches(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t n_total, int32_t B,
                     const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *plans,
                     int64_t plan_stride, int64_t list_capacity, uint32_t *status, uint64_t *bump_counter, void *stream) {
    return pcg_plan_epochs(g, nodes, labels, n_total, 1, B, thresholds, rho, train_flag, add_self, plans, plan_stride, list_capacity,
                           status, bump_counter, stream);
}

/* The plans of every batch of n_epochs epochs of n_total picks each (epoch e's picks at nodes + e * n_total): ONE launch. */
int pcg_plan_epochs(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t n_total, int32_t n_epochs, int32_t B,
                    const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *plans,
                    int64_t plan_stride, int64_t list_capacity, uint32_t *status, uint64_t *bump_counter, void *stream) {
    if (!g || n_total < 0 || n_epochs < 1 || B < 1 || !plans || plan_stride < 0 || (plan_stride & 255) != 0) return PCG_E_ARG;
    if (n_total == 0) return PCG_OK;
    const int slots_per_epoch = (n_total + B - 1) / B;
    if ((int64_t)slots_per_epoch * n_epochs > (1 << 20)) return PCG_E_ARG;
    const int n_slots = slots_per_epoch * n_epochs;
    const int B_tail = n_total - (slots_per_epoch - 1) * B;
    if (n_slots > 1 && plan_stride < pcg::carve(g, B, list_capacity, nullptr, nullptr, nullptr).plan_bytes) return PCG_E_ARG;
    // (the data part is not touched by the plan: any non-null base will do for the argument check)
    pcg::ChooseArgs full, tail;
    int rc = choose_args(full, g, nodes, labels, B, nullptr, nullptr, nullptr, thresholds, rho, train_flag, add_self, nullptr, plans,
                         list_capacity, status, true, plans);
    if (rc != PCG_OK) return rc;
    rc = choose_args(tail, g, nodes, labels, B_tail, nullptr, nullptr, nullptr, thresholds, rho, train_flag, add_self, nullptr, plans,
                     list_capacity, status, true, plans);
    if (rc != PCG_OK) return rc;
    const int nb_full = (g->n_rel * B + pcg::PLAN_THREADS - 1) / pcg::PLAN_THREADS;
    const int tail_b = B_tail == B ? -1 : slots_per_epoch - 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pcg::plan_batches_kernel, dim3(n_slots * nb_full), dim3(pcg::PLAN_THREADS), 0, st, full, tail, n_slots, tail_b,
                       plan_stride, nb_full, reinterpret_cast<unsigned long long *>(bump_counter), slots_per_epoch, (int64_t)n_total,
                       (int)n_epochs);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

/* first half: class-0 logits of rows [row_begin, row_end) -> s0_out[row]  ||  plan pass 1  (|| a deferred Adam update) */
// no_plan: the launch is [train-pos keys || deferred Adam || score pass] only (the batch's plan was made elsewhere: pcg_plan_batches)
static int front_a(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end,
                   float *s0_out, const int32_t *row_ids, uint64_t *pos_keys, const int32_t *nodes, const int32_t *labels, int32_t B,
                   const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *workspace,
                   int64_t list_capacity, uint32_t *status, const pcg::DeferredAdam *ad, void *stream, bool no_plan = false,
                   uint32_t *zero_word = nullptr, int64_t pos_row_base = -1, const uint8_t *touched = nullptr) {
    if (touched && (row_ids || row_begin != 0 || g->feat_stride > 512)) return PCG_E_ARG;
    if (!g || !g->X || !W || !b || !s0_out || B < 0) return PCG_E_ARG;
    if (g->feat_dim < 1 || g->feat_stride < g->feat_dim || g->feat_stride % 4 != 0) return PCG_E_ARG;
    if ((reinterpret_cast<uintptr_t>(g->X) & 15u) != 0) return PCG_E_ARG;
    if (row_begin < 0 || row_end > g->n_nodes || row_begin > row_end) return PCG_E_ARG;
    if (B == 0 && row_ids && !no_plan) return PCG_E_ARG;
    if (B == 0 && !no_plan) return pcg_score_table(g, W, b, row_begin, row_end, s0_out, stream);   // (then _b gathers its keys itself)
    pcg::ChooseArgs a;
    if (no_plan) {
        a = pcg::ChooseArgs();
        a.g = *g;
        a.B = 0;
    } else {
        const int rc = choose_args(a, g, nodes, labels, B, nullptr, nullptr, nullptr, thresholds, rho, train_flag, add_self, nullptr,
                                   workspace, list_capacity, status, true);
        if (rc != PCG_OK) return rc;
    }
    a.sort_done = zero_word;                 // (front_a_kernel zeroes it: the select kernel's arrival counter)
    const int rows = g->n_rel * B;
    pcg::PlanTotals *tot = no_plan ? nullptr : reinterpret_cast<pcg::PlanTotals *>(a.w.plan_totals);
    const int n_count = (rows + pcg::FRONT_COUNT_THREADS - 1) / pcg::FRONT_COUNT_THREADS;
    const int n_score = (int)pcg::score_table_blocks(row_end - row_begin, g->feat_stride);
    // the train positives' unsorted keys go to the scratch half of pos_keys (rank-sort sizes only)
    const bool raw = pos_keys && train_flag && g->n_pos > 0 && g->n_pos <= pcg::RANK_MAX && (g->train_pos || pos_row_base >= 0);
    const int rows_per_block = 4 * (PCG_WAVE / pcg::lanes_per_row(g->feat_stride));
    int n_key = raw ? (g->n_pos + rows_per_block - 1) / rows_per_block : 0;
    if (n_key > 256) n_key = 256;
    uint64_t *raw_keys = raw ? pos_keys + pcg_pos_sort_capacity(g->n_pos) / 2 : nullptr;
    pcg::DeferredAdam none = {};
    const int n_adam = ad ? (int)((ad->p_end + PCG_WAVE - 1) / PCG_WAVE) : 0;
    hipLaunchKernelGGL(pcg::front_a_kernel, dim3(n_count + n_key + n_adam + n_score), dim3(pcg::FRONT_COUNT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, tot, n_count, n_key, raw_keys, W, b, row_begin, row_end, s0_out,
                       ad ? *ad : none, n_adam, row_ids, pos_row_base, touched);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_step_front_a(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end,
                     float *s0_out, const int32_t *row_ids, uint64_t *pos_keys, const int32_t *nodes, const int32_t *labels,
                     int32_t B, const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self,
                     void *workspace, int64_t list_capacity, uint32_t *status, void *stream) {
    if (row_ids && pos_keys) return PCG_E_ARG;       // (the keys-from-feature-rows group indexes X by node id)
    return front_a(g, W, b, row_begin, row_end, s0_out, row_ids, pos_keys, nodes, labels, B, thresholds, rho, train_flag, add_self,
                   workspace, list_capacity, status, nullptr, stream);
}

/* second half: train-pos sort by s0 (if train_flag and n_pos > 0)  ||  plan pass 2 */
static int front_b(const pcg_graph_desc *g, const float *s0, uint64_t *pos_keys, int32_t raw_keys_ready,
                   const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                   int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status,
                   uint32_t *pending, float *center_out, int64_t center_id_offset, void *stream) {
    if (!g || !s0 || B < 0 || (center_out && center_id_offset < 0)) return PCG_E_ARG;
    const bool sort = train_flag && g->n_pos > 0;
    if (sort && (!pos_keys || !g->train_pos)) return PCG_E_ARG;
    if (B == 0) return sort ? pcg_pos_sort(g, s0, pos_keys, stream) : PCG_OK;
    pcg::ChooseArgs a;
    const int rc = choose_args(a, g, nodes, labels, B, s0, nullptr, pos_keys, thresholds, rho, train_flag, add_self, nullptr,
                               workspace, list_capacity, status);
    if (rc != PCG_OK) return rc;
    const int rows = g->n_rel * B;
    pcg::PlanTotals *tot = reinterpret_cast<pcg::PlanTotals *>(a.w.plan_totals);
    const int n_count = (rows + pcg::FRONT_COUNT_THREADS - 1) / pcg::FRONT_COUNT_THREADS;
    const int n_write = (rows + pcg::PLAN_THREADS - 1) / pcg::PLAN_THREADS;
    const bool rank = sort && g->n_pos <= pcg::RANK_MAX;
    const int n_sort = rank ? (g->n_pos + PCG_WAVE - 1) / PCG_WAVE : 0;
    const int64_t cap = sort ? pcg_pos_sort_capacity(g->n_pos) / 2 : 0;
    const uint64_t *raw_keys = (rank && raw_keys_ready) ? pos_keys + cap : nullptr;
    hipLaunchKernelGGL(pcg::front_b_kernel, dim3(n_write + n_sort), dim3(pcg::PLAN_THREADS), 0, static_cast<hipStream_t>(stream), a,
                       tot, n_write, n_count, pos_keys, (int)cap, raw_keys, pending, center_out, center_id_offset);
    PCG_LAUNCH_CHECK();
    if (sort && !rank) return pcg_pos_sort(g, s0, pos_keys, stream);    // many positives: the bucket sort's own launches
    return PCG_OK;
}

int pcg_step_front_b(const pcg_graph_desc *g, const float *s0, uint64_t *pos_keys, int32_t raw_keys_ready,
                     const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                     int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status,
                     float *center_s0_out, int64_t center_id_offset, void *stream) {
    return front_b(g, s0, pos_keys, raw_keys_ready, nodes, labels, B, thresholds, rho, train_flag, add_self, workspace,
                   list_capacity, status, nullptr, center_s0_out, center_id_offset, stream);
}

/* pcg_step_front of a TRAINING step, with the previous step's deferred Adam update riding along the score pass */
int pcg_step_front_train(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, float *s0, uint64_t *pos_keys,
                         const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                         int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status, const float *slabs,
                         const int32_t *step_counter, uint32_t *sync_words, double lr, double beta1, double beta2, double eps,
                         double weight_decay, void *stream) {
    if (!g || !theta || !m || !v || !slabs || !step_counter || !sync_words || B < 1) return PCG_E_ARG;
    const int64_t n_params = pcg_dense_n_params(g->feat_dim, emb, g->n_rel);
    const int64_t o_clf = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 3, 0), o_b = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 4, 0);
    if (n_params < 0 || o_clf < 0) return PCG_E_ARG;
    pcg::DeferredAdam ad;
    ad.theta = theta; ad.m = m; ad.v = v;
    ad.slabs = slabs;
    ad.n_params = n_params;
    ad.p_end = o_clf;
    ad.step_counter = step_counter;
    ad.pending = sync_words + 1;
    ad.h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    const int rc = front_a(g, theta + o_clf, theta + o_b, 0, g->n_nodes, s0, nullptr, pos_keys, nodes, labels, B, thresholds, rho, 1, add_self,
                           workspace, list_capacity, status, &ad, stream);
    if (rc != PCG_OK) return rc;
    return front_b(g, s0, pos_keys, 1, nodes, labels, B, thresholds, rho, 1, add_self, workspace, list_capacity, status,
                   sync_words + 1, nullptr, 0, stream);
}

/* The front of a training step whose plan exists already (pcg_plan_batches): ONE launch
 *   [train-pos keys from their feature rows || the previous step's deferred Adam update || score pass over the table]
 * and no sort: pcg_choose_gather_planned(..., sync_words) sorts the keys inside the select kernel.  More than RANK_MAX train
 * positives: the bucket sort's own launches follow here and the keys are sorted on return. */
int pcg_step_scores_train(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, float *s0, uint64_t *pos_keys,
                          const float *slabs, const int32_t *step_counter, uint32_t *sync_words, double lr, double beta1,
                          double beta2, double eps, double weight_decay, const uint8_t *touched, void *stream) {
    if (!g || !theta || !m || !v || !slabs || !step_counter || !sync_words) return PCG_E_ARG;
    if (g->n_pos > 0 && (!pos_keys || !g->train_pos)) return PCG_E_ARG;
    const int64_t n_params = pcg_dense_n_params(g->feat_dim, emb, g->n_rel);
    const int64_t o_clf = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 3, 0), o_b = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 4, 0);
    if (n_params < 0 || o_clf < 0) return PCG_E_ARG;
    pcg::DeferredAdam ad;
    ad.theta = theta; ad.m = m; ad.v = v;
    ad.slabs = slabs;
    ad.n_params = n_params;
    ad.p_end = o_clf;
    ad.step_counter = step_counter;
    ad.pending = sync_words + 1;
    ad.h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    const int rc = front_a(g, theta + o_clf, theta + o_b, 0, g->n_nodes, s0, nullptr, pos_keys, nullptr, nullptr, 0, nullptr, nullptr, 1, 0,
                           nullptr, 1, nullptr, &ad, stream, true, sync_words + 3, -1, touched);
    if (rc != PCG_OK) return rc;
    if (g->n_pos > pcg::RANK_MAX) return pcg_pos_sort(g, s0, pos_keys, stream);
    return PCG_OK;
}

/* scores (+ unsorted train-pos keys) on their own, no plan, no Adam: the front of a step of the PARTITIONED path, whose table
 * rows are [owned | train-pos | halo] and whose scores are indexed by global node id (row_ids), and of any caller that plans its
 * batches with pcg_plan_batches and updates its parameters itself. */
int pcg_step_scores(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end, float *s0_out,
                    const int32_t *row_ids, uint64_t *pos_keys, int64_t pos_row_base, uint32_t *sync_words, const uint8_t *touched,
                    void *stream) {
    if (!g || !sync_words) return PCG_E_ARG;
    if (pos_keys && row_ids && pos_row_base < 0) return PCG_E_ARG;     // (with row_ids a node id is not a table row)
    if (pos_row_base >= 0 && pos_row_base + g->n_pos > g->n_nodes) return PCG_E_ARG;
    const int rc = front_a(g, W, b, row_begin, row_end, s0_out, row_ids, pos_keys, nullptr, nullptr, 0, nullptr, nullptr, 1, 0, nullptr, 1,
                           nullptr, nullptr, stream, true, sync_words + 3, pos_row_base, touched);
    if (rc != PCG_OK) return rc;
    // (too many train positives for the select kernel's own sort: the bucket sort's launches, by node id - not for row_ids tables)
    if (pos_keys && !row_ids && pos_row_base < 0 && g->n_pos > pcg::RANK_MAX) return pcg_pos_sort(g, s0_out, pos_keys, stream);
    return PCG_OK;
}

/* pcg_step_scores for a partitioned rank's TRAINING step, with the optimizer riding in it (one launch, behind the all-reduce):
 * if sync_words[1] == 1 ("a gradient is waiting": pcg_wgrad(flag_set) / pcg_grad_reduce) torch.optim.Adam's update from `grad`
 * (the all-reduced gradient, n_params floats) is applied to EVERY parameter by some workgroups, while the others score rows
 * [row_begin, row_end) -> s0_out[row_ids[row]] and form the train positives' unsorted keys (rows pos_row_base + i) with the label
 * classifier AFTER that update, which each works out for itself from clf_snap [3 * (2 feat_dim + 2)] - the classifier's
 * parameters, m, v as of the last applied update (pcg_gather_lists_dist refreshes it every step; the caller after a flush) -
 * and the gradient.  The flag is cleared by the step's select launch (pcg_choose_select_planned(sync_words)); the launch zeroes
 * sync_words[3].  n_pos > 16384: pos_keys must be NULL (the caller sorts by its own launches). */
int pcg_step_scores_dist(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, const float *grad,
                         const float *clf_snap, int64_t row_begin, int64_t row_end, float *s0_out, const int32_t *row_ids,
                         uint64_t *pos_keys, int64_t pos_row_base, const int32_t *step_counter, uint32_t *sync_words, double lr,
                         double beta1, double beta2, double eps, double weight_decay, void *stream) {
    if (!g || !g->X || !theta || !m || !v || !grad || !clf_snap || !s0_out || !step_counter || !sync_words) return PCG_E_ARG;
    if (g->feat_dim < 1 || g->feat_dim > 512 || g->feat_stride < g->feat_dim || g->feat_stride % 4 != 0) return PCG_E_ARG;
    if ((reinterpret_cast<uintptr_t>(g->X) & 15u) != 0) return PCG_E_ARG;
    if (row_begin < 0 || row_end > g->n_nodes || row_begin > row_end) return PCG_E_ARG;
    if (pos_keys && (pos_row_base < 0 || pos_row_base + g->n_pos > g->n_nodes || g->n_pos > pcg::RANK_MAX)) return PCG_E_ARG;
    const int64_t n_params = pcg_dense_n_params(g->feat_dim, emb, g->n_rel);
    const int64_t o_clf = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 3, 0);
    if (n_params < 0 || o_clf < 0) return PCG_E_ARG;
    pcg::FrontDist f;
    f.g = *g;
    f.row_begin = row_begin;
    f.row_end = row_end;
    f.s0 = s0_out;
    f.row_ids = row_ids;
    f.pos_row_base = pos_row_base;
    const bool raw = pos_keys && g->n_pos > 0;
    f.raw_keys = raw ? pos_keys + pcg_pos_sort_capacity(g->n_pos) / 2 : nullptr;
    const int rows_per_block = 4 * (PCG_WAVE / pcg::lanes_per_row(g->feat_stride));
    int n_key = raw ? (g->n_pos + rows_per_block - 1) / rows_per_block : 0;
    f.n_key_blocks = n_key > 256 ? 256 : n_key;
    f.n_adam_blocks = (int)((n_params + PCG_WAVE - 1) / PCG_WAVE);
    f.ad.theta = theta; f.ad.m = m; f.ad.v = v;
    f.ad.slabs = grad;
    f.ad.n_params = n_params;
    f.ad.p_end = n_params;
    f.ad.step_counter = step_counter;
    f.ad.pending = sync_words + 1;
    f.ad.h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    f.snap = clf_snap;
    f.nc = 2 * g->feat_dim + 2;
    f.off_clf = o_clf;
    f.zero_word = sync_words + 3;
    const int n_score = (int)pcg::score_table_blocks(row_end - row_begin, g->feat_stride);
    hipLaunchKernelGGL(pcg::front_dist_kernel, dim3(f.n_adam_blocks + f.n_key_blocks + n_score), dim3(pcg::FRONT_COUNT_THREADS), 0,
                       static_cast<hipStream_t>(stream), f);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int32_t pcg_pos_sort_in_select(int32_t n_pos) { return n_pos > 0 && n_pos <= pcg::RANK_MAX ? 1 : 0; }

/* uint32 words of the `sync_words` buffer: [0] dense ticket, [1] update pending, [2] its slab count, [3] the in-kernel sort's group
 * counter, then the sort's rank accumulators (one per train positive, up to RANK_MAX) and group tickets */
int32_t pcg_sync_words_count(void) { return 4 + pcg::RANK_MAX + pcg::RANK_MAX / PCG_WAVE + 4; }     // (+ 4: the dense kernel's "staged" counter)

int pcg_step_front(const pcg_graph_desc *g, const float *W, const float *b, float *s0, uint64_t *pos_keys,
                   const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                   int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status,
                   void *stream) {
    if (!g) return PCG_E_ARG;
    const int rc = pcg_step_front_a(g, W, b, 0, g->n_nodes, s0, nullptr, pos_keys, nodes, labels, B, thresholds, rho, train_flag, add_self,
                                    workspace, list_capacity, status, stream);
    if (rc != PCG_OK) return rc;
    return pcg_step_front_b(g, s0, pos_keys, B > 0 ? 1 : 0, nodes, labels, B, thresholds, rho, train_flag, add_self, workspace,
                            list_capacity, status, nullptr, 0, stream);
}

int pcg_choose_aggregate(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const float *s0, const float *center_s0, const uint64_t *pos_keys,
                         const double *thresholds, const double *rho, int32_t train_flag, int32_t norm,
                         int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt, void *workspace,
                         int64_t list_capacity, uint32_t *status, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!g->X || !agg) return PCG_E_ARG;
    const int rc = pcg_choose_select(g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag,
                                     add_self, cnt, workspace, list_capacity, status, stream);
    if (rc != PCG_OK) return rc;
    return pcg_aggregate_lists(g->X, g->feat_dim, g->feat_stride, g->n_nodes, g->n_rel * B, cnt, g, B, workspace, list_capacity,
                               norm, agg, agg_stride, status, stream);
}

int pcg_choose_gather_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B, const float *s0,
                              const float *center_s0, const uint64_t *pos_keys, const double *thresholds, const double *rho,
                              int32_t train_flag, int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt, void *workspace,
                              const void *plan, int64_t list_capacity, uint32_t *status, uint32_t *sync_words, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!g->X || !agg) return PCG_E_ARG;
    PlannedExtra x;
    x.plan = plan;
    x.sync_words = sync_words;
    const int rc = choose_select(true, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                                 workspace, list_capacity, status, stream, x);
    if (rc != PCG_OK) return rc;
    return pcg_gather_lists_planned(g->X, g->feat_dim, g->feat_stride, g->n_nodes, g->n_rel * B, cnt, g, B, workspace, plan, list_capacity,
                                    agg, agg_stride, status, stream);
}

/* select + gather of a TRAINING step whose label classifier is stepped on its own (ClfStep / SideJob in choose.h): two launches,
 *   select_rows         [+ one workgroup: the classifier's forward / loss / Adam for THIS batch: clf_next <- the updated classifier]
 *   gather_train_kernel [+ the previous step's deferred Adam update of every other parameter || (score_next) the NEXT step's
 *                          score pass and unsorted train-pos keys with clf_next]
 * followed by pcg_train_dense(adam_clf = 2).  s0 / pos_keys: read by the select launch (this step's scores; unsorted keys in the
 * scratch half), rewritten by the gather launch for the next step. */
int pcg_choose_gather_train(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B, float *s0,
                            uint64_t *pos_keys, const double *thresholds, const double *rho, int32_t add_self, float *agg,
                            int32_t agg_stride, int32_t *cnt, void *workspace, const void *plan, int64_t list_capacity,
                            uint32_t *status, uint32_t *sync_words, float *theta, float *m, float *v, int32_t emb, float *clf_next,
                            const float *slabs, const int32_t *step_counter, float lambda_1, float inv_count, double lr, double beta1,
                            double beta2, double eps, double weight_decay, int32_t score_next, const uint8_t *next_touched,
                            const float *acts, int32_t act_ld, float *wg_scratch, int32_t keys_sorted, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!g->X || !agg || !s0 || !sync_words || !theta || !m || !v || !clf_next || !slabs || !step_counter || !labels) return PCG_E_ARG;
    if (acts && (act_ld < 16 || act_ld % 16 != 0 || (reinterpret_cast<uintptr_t>(acts) & 15u) != 0 || emb % 16 != 0)) return PCG_E_ARG;
    if (acts && pcg::wgrad_kparts(act_ld / 16) > 1 && !wg_scratch) return PCG_E_ARG;
    if (g->feat_stride > 512) return PCG_E_UNSUPPORTED;
    const int64_t n_params = pcg_dense_n_params(g->feat_dim, emb, g->n_rel);
    const int64_t o_clf = pcg_dense_param_offset(g->feat_dim, emb, g->n_rel, 3, 0);
    if (n_params < 0 || o_clf < 0) return PCG_E_ARG;
    pcg::ChooseArgs a;
    int rc = choose_args(a, g, nodes, labels, B, s0, nullptr, pos_keys, thresholds, rho, 1, add_self, cnt, workspace, list_capacity,
                         status, false, plan);
    if (rc != PCG_OK) return rc;
    if (!cnt) return PCG_E_ARG;
    const bool rank = g->n_pos > 0 && g->n_pos <= pcg::RANK_MAX;
    const int64_t cap = g->n_pos > 0 ? pcg_pos_sort_capacity(g->n_pos) / 2 : 0;
    // keys_sorted: pos_keys' first half holds THIS step's keys sorted already (the previous step's pcg_train_dense(sort_keys), or
    // pcg_pos_sort behind pcg_step_scores): no in-kernel sort, no row waits, a hub row's window search runs beside its key pass
    if (rank && !keys_sorted) {                        // the select kernel sorts the unsorted keys itself
        a.sort_out = pos_keys;
        a.raw_keys = pos_keys + cap;
        a.sort_cap = (int32_t)cap;
        a.n_sort = (g->n_pos + PCG_WAVE - 1) / PCG_WAVE;
        a.sort_done = sync_words + 3;
        a.rank_acc = sync_words + 4;
        a.group_ticket = sync_words + 4 + pcg::RANK_MAX;
    }
    const pcg::AdamHyper h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    a.clf.clf_next = clf_next;
    a.clf.theta_clf = theta + o_clf;
    a.clf.m = m + o_clf;
    a.clf.v = v + o_clf;
    a.clf.step_counter = step_counter;
    a.clf.scale = inv_count * lambda_1;
    a.clf.h = h;
    // (a slice of <= 1024 rows per workgroup, at most 8 of them; their gradients meet in the first slabs' classifier entries -
    //  free until this step's dense launch writes them -, the arrival ticket is the dense kernel's unused one)
    a.clf.n_wg = (B + 1023) / 1024 > 8 ? 8 : (B + 1023) / 1024;
    a.clf.part = const_cast<float *>(slabs) + o_clf;
    a.clf.part_stride = n_params;
    a.clf.ticket = sync_words;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // the deferred update's weight-gradient workgroups (acts): in the gather launch (where they cost ~3 us of its ~10) - or, an
    // option that measured slower, as the first units of the select launch
    pcg::WgradArgs wga = {};
    int wg_tiles = 0;
    if (acts) {
        wga.acts = acts; wga.ld = act_ld;
        wga.F = g->feat_dim; wga.E = emb; wga.R = g->n_rel;
        wga.theta = theta; wga.m = m; wga.v = v;
        wga.step_counter = step_counter;
        wga.h = h;
        wga.pending = sync_words + 1;
        wga.n_kblocks = act_ld / 16;             // (the batch size the engine's buffers were made for: the expected one)
        wga.kparts = pcg::wgrad_kparts(act_ld / 16);
        wga.tickets = reinterpret_cast<uint32_t *>(wg_scratch);
        wga.partials = wg_scratch ? wg_scratch + (pcg::wgrad_tiles(g->feat_dim, emb, g->n_rel, 1) + 63) / 64 * 64 : nullptr;
        wga.grad_out = nullptr;
        wga.flag_set = nullptr;
        wga.apply = 1;
        wga.with_clf = 0;
        wg_tiles = pcg::wgrad_tiles(g->feat_dim, emb, g->n_rel, 0);
        // (A/B knob PCG_WGRAD_IN_SELECT=1.  Measured, YelpChi-like batch 1024: select_rows 16.2 -> 19.9 us, gather launch 10.5 -> 8.2:
        //  43.8 vs 42.9 us per step - the workgroups that start on their rows ~7 us late end the launch later than the gather's
        //  riders cost.  Off; profiles/r04/x_wgrad_in_select_yelp_bench.log)
        static int in_select = -1;
        if (in_select < 0) {
            const char *e = getenv("PCG_WGRAD_IN_SELECT");
            in_select = e ? atoi(e) : 0;
        }
        if (in_select && wga.kparts == 1 && wg_tiles % 2 == 0 && wg_tiles / 2 <= 96 && g->feat_stride <= 256) {
            a.wg = wga;
            a.n_wg_units = wg_tiles / 2;
        }
    }
    rc = pcg::launch_select(a, st, true);
    if (rc != PCG_OK) return rc;
    pcg::SideJob sd;
    sd.ad.theta = theta; sd.ad.m = m; sd.ad.v = v;
    sd.ad.slabs = slabs;
    sd.ad.n_params = n_params;
    sd.ad.p_end = o_clf;
    sd.ad.step_counter = step_counter;
    sd.ad.pending = sync_words + 1;
    sd.ad.h = h;
    sd.n_adam_blocks = acts ? 0 : (int)((o_clf + PCG_WAVE - 1) / PCG_WAVE);
    // acts: the previous step's pcg_train_dense(adam_clf = 3) left activations, not slabs - the weight gradients are GEMMs over
    // its batch, each output tile's workgroup applying Adam to its own parameters (wgrad.h) - here unless the select launch did it
    sd.wg = wga;
    sd.n_wgrad_blocks = (acts && a.n_wg_units == 0) ? wg_tiles * wga.kparts : 0;
    {
        static int prio = -1, off = -1;          // A/B knobs (timing experiments only: PCG_WGRAD_OFF=1 skips the update)
        if (prio < 0) {
            const char *e = getenv("PCG_WGRAD_PRIO"), *o = getenv("PCG_WGRAD_OFF");
            prio = e ? atoi(e) : 0;
            off = o ? atoi(o) : 0;
        }
        sd.wg_prio = prio;
        if (off) sd.n_wgrad_blocks = 0;
    }
    sd.W = score_next ? clf_next : nullptr;
    sd.bias = clf_next + 2 * g->feat_dim;
    sd.s0 = s0;
    sd.touched = next_touched;
    // (the one-launch bucket sort - 16384 < n_pos <= 131072 - works on raw keys formed here too)
    const bool one_launch = pcg_pos_sort_one_launch(g->n_pos) != 0;
    sd.raw_keys = (score_next && (rank || one_launch) && g->train_pos) ? pos_keys + cap : nullptr;
    const int rows_per_block = 4 * (PCG_WAVE / pcg::lanes_per_row(g->feat_stride));
    int n_key = sd.raw_keys ? (g->n_pos + rows_per_block - 1) / rows_per_block : 0;
    sd.n_key_blocks = n_key > 256 ? 256 : n_key;
    sd.n_score_blocks = score_next ? (int)pcg::score_table_blocks(g->n_nodes, g->feat_stride) : 0;
    sd.zero_word = sync_words + 3;
    rc = pcg::launch_gather_train(g->X, g->feat_dim, g->feat_stride, g->n_nodes, cnt, g, B, a.w, agg, agg_stride, status, sd, st);
    if (rc != PCG_OK) return rc;
    if (score_next && one_launch) return pcg::launch_bk_onepass(pos_keys + cap, g->n_pos, pos_keys, (int)cap, status, st);
    if (score_next && g->n_pos > pcg::RANK_MAX) return pcg_pos_sort(g, s0, pos_keys, stream);
    return PCG_OK;
}

int pcg_choose_aggregate_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                                 const float *s0, const float *center_s0, const uint64_t *pos_keys,
                                 const double *thresholds, const double *rho, int32_t train_flag, int32_t norm,
                                 int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt, void *workspace,
                                 int64_t list_capacity, uint32_t *status, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!g->X || !agg) return PCG_E_ARG;
    const int rc = pcg_choose_select_planned(g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag,
                                             add_self, cnt, workspace, nullptr, list_capacity, status, nullptr, 0, stream);
    if (rc != PCG_OK) return rc;
    return pcg_aggregate_lists(g->X, g->feat_dim, g->feat_stride, g->n_nodes, g->n_rel * B, cnt, g, B, workspace, list_capacity,
                               norm, agg, agg_stride, status, stream);
}

}  // extern "C"
