// choose + aggregate for gfx950: the PC-GNN hot path in four kinds of launches.
//
//   plan     per (relation r, centre b) row the kept-count bound cap = (deg > k+1 ? k : deg) + m (+1), exclusive
//            offsets into the selection list, four degree-tier queues and the 128-entry chunk table of the gather.
//            One workgroup (plan_kernel), two passes of many (plan_count / plan_write), or the same two passes riding
//            along the score pass and the train-pos sort (front_a / front_b: pcg_step_front).
//   select   ONE persistent launch (select_rows): every workgroup takes hub rows (deg > 4096) and mid rows
//            (513..4096) whole, then rows of <= 512 one per wave - longest first, statically assigned.  Per row:
//              1. neighbour ids + distance keys |s0[c] - s0[j]| -> LDS (8 gathers in flight per lane)
//              2. exact k-th smallest key by a bracketed counting search (two pivots by interpolation in value space,
//                 every 4th round a bit-space midpoint; the first bracket's survivors compacted; <= 64 survivors
//                 finished on registers); ties by row position via ballot prefix counts
//              3. kept ids compacted (ascending) in LDS and written to the row's list
//              4. minority over-sampling for positive centres: 64-ary window search in the per-step sorted train-pos
//                 keys; de-duplicated against (3) by LDS binary search; a duplicate leaves a hole (-1) so slots -
//                 and sums - keep a fixed order
//   gather   chip-wide balanced: one wave per 128-entry chunk of a row's list, feature rows gathered 64/lpr per
//            wave-instruction (128-B rows: 8 rows = 1 KiB), 8 in flight, f32 segmented sum; single-chunk rows are
//            finished here
//   combine  rows longer than one chunk: partial sums added in chunk order (bitwise reproducible), divided by |set|
//            (or its sqrt)
//
// Reference lines replaced: src/layers.py:217-219, 246-262, 587-624, 633-738;
// src/graphsage.py:62-96, 200-232 (keep-all + add_self + sqrt normalisation).
#include <limits.h>

#include "common.h"

namespace pcg {

constexpr int T0_CAP = 128;      // short rows: one wave per row, taken last
constexpr int T1_CAP = 512;      // row length handled by a single wave (keys + ids: 4 KB of LDS)
constexpr int T4_CAP = 4096;     // ... by a whole 8-wave workgroup with keys, ids and first-pass survivors in LDS (48 KB)
constexpr int HUB_LDS_CAP = 6144;  // hub row whose keys + first-pass survivors fit the 48 KB (ids re-read from the CSR)
constexpr int SEL_NW = 8;        // waves per select workgroup
constexpr int SEL_BLOCKS = 768;  // persistent select workgroups: 3 per CU
constexpr int SEL_LDS_WORDS = 3 * T4_CAP;
constexpr int CHUNK = 128;       // list entries per gather work item
constexpr int UNROLL = 8;        // row-gather instructions in flight per wave
constexpr int KEY_UNROLL = 8;    // neighbour-score gathers in flight per lane
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_PER = 4;      // rows per plan thread per tile
constexpr int GATHER_BLOCKS = 2048;

// counters (uint32) at the head of the workspace
enum { C_N1 = 0, C_N4 = 1, C_N16 = 2, C_NCHUNK = 5, C_N0 = 8 };

struct RowRec {            // 32 bytes, written by plan, read by select (one 32-B load instead of a 3-deep chain)
    int64_t start;         // offset of the row in indices[r]
    int32_t node, d, k, m;
    int32_t keep_all;
    int32_t pad;
};

struct Workspace {
    uint32_t *counters;    // [64]
    int64_t *row_begin;    // [rows + 1] start of every row's region in list
    int32_t *chunk_begin;  // [rows + 1]
    int32_t *len;          // [rows]     entries (holes included) actually written
    int32_t *q0, *q1, *q4, *q16;  // [rows] each: the rows of every degree tier
    struct RowRec *recs;   // [rows] what plan worked out per row
    unsigned char *plan_totals;   // [blocks of the two-pass plan] PlanTotals
    int4 *chunk_desc;      // [chunk_cap] per gather chunk: {row, first list entry, entries of the row's region in it, chunks of the row}
    float *partial;        // [chunk_cap, feat_stride]
    int32_t *list;         // [list_capacity]  chosen ids; -1 = hole
    uint32_t *scratch;     // [SEL_BLOCKS * 2 * max_degree] when rows beyond the LDS tiers exist (max_degree > HUB_LDS_CAP)
    int64_t list_capacity, chunk_cap;
};

static int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

static int64_t carve(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, unsigned char *base, Workspace *w) {
    const int64_t rows = (int64_t)g->n_rel * B;
    const int64_t chunk_cap = list_capacity / CHUNK + rows + 1;
    int64_t off = 0;
    auto take = [&](int64_t bytes) {
        const int64_t o = off;
        off += align256(bytes);
        return base ? base + o : nullptr;
    };
    unsigned char *p;
    p = take(256);                                 if (w) w->counters = reinterpret_cast<uint32_t *>(p);
    p = take(8 * (rows + 1));                      if (w) w->row_begin = reinterpret_cast<int64_t *>(p);
    p = take(4 * (rows + 1));                      if (w) w->chunk_begin = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->len = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q0 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q1 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q4 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q16 = reinterpret_cast<int32_t *>(p);
    p = take(32 * rows);                           if (w) w->recs = reinterpret_cast<RowRec *>(p);
    p = take(64 * (rows / 256 + 2));               if (w) w->plan_totals = p;   // PlanTotals (<= 64 B) per 256 rows
    p = take(16 * chunk_cap);                      if (w) w->chunk_desc = reinterpret_cast<int4 *>(p);
    p = take(4 * chunk_cap * g->feat_stride);      if (w) w->partial = reinterpret_cast<float *>(p);
    p = take(4 * list_capacity);                   if (w) w->list = reinterpret_cast<int32_t *>(p);
    p = take(g->max_degree > HUB_LDS_CAP ? (int64_t)SEL_BLOCKS * 2 * g->max_degree * 4 : 0);
    if (w) {
        w->scratch = reinterpret_cast<uint32_t *>(p);
        w->list_capacity = list_capacity;
        w->chunk_cap = chunk_cap;
    }
    return off;
}

struct ChooseArgs {
    pcg_graph_desc g;
    const int32_t *nodes;
    const int32_t *labels;
    int32_t B;
    const float *s0;
    const float *center_s0;
    const uint64_t *pos_keys;
    double thr[PCG_MAX_REL];
    double rho[PCG_MAX_REL];
    int32_t train_flag, add_self;
    int32_t *cnt;          // [rows] |chosen set|
    uint32_t *status;
    unsigned long long *stamps;   // diagnostic only (pcg_debug_set_stamps): [rows][8] wall-clock ticks per phase, else null
    Workspace w;
};

#define PCG_STAMP(slot)                                                               \
    do {                                                                              \
        if (a.stamps && tid == 0) a.stamps[(size_t)row * 8 + (slot)] = wall_clock64(); \
    } while (0)

__device__ __forceinline__ RowRec row_plan(const ChooseArgs &a, int row) {
    RowRec p;
    const int r = row / a.B, b = row - r * a.B;
    p.node = a.nodes[b];
    p.start = a.g.indptr[r][p.node];
    p.d = (int)(a.g.indptr[r][p.node + 1] - p.start);
    p.k = (int)ceil((double)p.d * a.thr[r]);             // layers.py:260
    p.keep_all = !(p.d > p.k + 1);                       // layers.py:662
    p.m = 0;
    if (a.train_flag && a.labels[b] == 1) {              // layers.py:675
        p.m = (int)((double)p.k * a.rho[r]);             // layers.py:681
        if (p.m > a.g.n_pos) p.m = a.g.n_pos;
        if (p.m < 0) p.m = 0;
    }
    p.pad = 0;
    return p;
}

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
// the four tier counts of a thread / tile (each <= 4096) share one 64-bit word, 16 bits each: one scan
__device__ __forceinline__ long long tier_word(int d) {
    return d <= T0_CAP ? 1ll : d <= T1_CAP ? (1ll << 16) : d <= T4_CAP ? (1ll << 32) : (1ll << 48);
}
struct TierCounts {
    int n0, n1, n4, n16;
};
__device__ __forceinline__ TierCounts tier_unpack(long long w) {
    TierCounts t;
    t.n0 = (int)(w & 0xFFFF);
    t.n1 = (int)((w >> 16) & 0xFFFF);
    t.n4 = (int)((w >> 32) & 0xFFFF);
    t.n16 = (int)((w >> 48) & 0xFFFF);
    return t;
}
__device__ __forceinline__ void tier_push(const Workspace &w, int d, int row, TierCounts &o) {
    if (d <= T0_CAP) w.q0[o.n0++] = row;
    else if (d <= T1_CAP) w.q1[o.n1++] = row;
    else if (d <= T4_CAP) w.q4[o.n4++] = row;
    else w.q16[o.n16++] = row;
}
__device__ __forceinline__ void tier_finish(const Workspace &w, const TierCounts &t, bool overflow) {
    w.counters[C_N0] = overflow ? 0 : t.n0;
    w.counters[C_N1] = overflow ? 0 : t.n1;
    w.counters[C_N4] = overflow ? 0 : t.n4;
    w.counters[C_N16] = overflow ? 0 : t.n16;
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


__global__ void __launch_bounds__(PLAN_THREADS) plan_kernel(const ChooseArgs a) {
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
    TierCounts run = {0, 0, 0, 0};
    bool overflow = false;
    for (int tile0 = 0; tile0 < rows; tile0 += PLAN_THREADS * PLAN_PER) {
        const int r0 = tile0 + threadIdx.x * PLAN_PER;
        RowRec rec[PLAN_PER];
        int cap[PLAN_PER];
        long long cap_sum = 0;
        int chunk_sum = 0;
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
                p.keep_all = !(p.d > p.k + 1);                            // layers.py:662
                p.m = 0;
                if (lab_on && labv[i] == 1) {                             // layers.py:675
                    p.m = (int)((double)p.k * t_rho[rel[i]]);             // layers.py:681
                    if (p.m > a.g.n_pos) p.m = a.g.n_pos;
                    if (p.m < 0) p.m = 0;
                }
                p.pad = 0;
                rec[i] = p;
                cap[i] = (p.keep_all ? p.d : p.k) + p.m + (a.add_self ? 1 : 0);
                cap_sum += cap[i];
                chunk_sum += (cap[i] + CHUNK - 1) / CHUNK;
                tiers += tier_word(p.d);
            }
        }
        PLAN_STAMP(1);
        // three scans: the tier counts (each <= 4 per thread, <= 4096 per tile) share one word
        long long t_cap, t_tiers;
        int t_chunk;
        long long o_cap = run_cap + block_excl_scan<long long>(cap_sum, lds64, t_cap);
        int o_chunk = run_chunk + block_excl_scan(chunk_sum, lds, t_chunk);
        TierCounts o = tier_unpack(block_excl_scan<long long>(tiers, lds64, t_tiers));
        o.n0 += run.n0; o.n1 += run.n1; o.n4 += run.n4; o.n16 += run.n16;
        const TierCounts tt = tier_unpack(t_tiers);
        run_cap += t_cap; run_chunk += t_chunk;
        run.n0 += tt.n0; run.n1 += tt.n1; run.n4 += tt.n4; run.n16 += tt.n16;
        overflow = overflow || run_cap > a.w.list_capacity || (long long)run_chunk > a.w.chunk_cap;
        PLAN_STAMP(2);
#pragma unroll
        for (int i = 0; i < PLAN_PER; ++i) {
            const int row = r0 + i;
            if (row >= rows) continue;
            const int nch = (cap[i] + CHUNK - 1) / CHUNK;
            a.w.row_begin[row] = o_cap;
            a.w.chunk_begin[row] = o_chunk;
            a.w.recs[row] = rec[i];
            if (!overflow) {
                for (int j = 0; j < nch; ++j)
                    a.w.chunk_desc[o_chunk + j] = make_int4(row, (int)o_cap + j * CHUNK, cap[i] - j * CHUNK < CHUNK ? cap[i] - j * CHUNK : CHUNK, nch);
                tier_push(a.w, rec[i].d, row, o);
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
}

// Large batches (rows > PLAN_THREADS * PLAN_PER): the same plan in two launches of many workgroups.
//   plan_count : every block works out the records of its PLAN_THREADS * PLAN_PER rows and their totals
//   plan_write : every block adds up the totals of the blocks before it (a few dozen values), then scans its
//                own rows and writes offsets / queues / chunk table exactly as the single-block kernel does
struct PlanTotals {
    long long cap;
    int chunk, n0, n1, n4, n16;
    int pad[3];
};
static_assert(sizeof(PlanTotals) <= 64, "the workspace carve reserves 64 bytes per plan block");

__device__ __forceinline__ int row_cap(const RowRec &p, int add_self) {
    return (p.keep_all ? p.d : p.k) + p.m + (add_self ? 1 : 0);
}

// pass 1, workgroup `block` of THREADS threads, one row per thread
template <int THREADS>
__device__ __forceinline__ void plan_count_body(const ChooseArgs &a, PlanTotals *totals, int block) {
    __shared__ int lds[THREADS / PCG_WAVE];
    __shared__ long long lds64[THREADS / PCG_WAVE];
    const int rows = a.g.n_rel * a.B;
    const int row = block * THREADS + (int)threadIdx.x;
    long long cap_sum = 0, tiers = 0;
    int chunk_sum = 0;
    if (row < rows) {
        const RowRec p = row_plan(a, row);
        a.w.recs[row] = p;
        const int cap = row_cap(p, a.add_self);
        cap_sum = cap;
        chunk_sum = (cap + CHUNK - 1) / CHUNK;
        tiers = tier_word(p.d);
    }
    long long t_cap, t_tiers;
    int t_chunk;
    block_excl_scan<long long>(cap_sum, lds64, t_cap);
    block_excl_scan(chunk_sum, lds, t_chunk);
    block_excl_scan<long long>(tiers, lds64, t_tiers);
    if (threadIdx.x == 0) {
        const TierCounts tt = tier_unpack(t_tiers);
        PlanTotals t;
        t.cap = t_cap;
        t.chunk = t_chunk;
        t.n0 = tt.n0; t.n1 = tt.n1; t.n4 = tt.n4; t.n16 = tt.n16;
        t.pad[0] = t.pad[1] = t.pad[2] = 0;
        totals[block] = t;
    }
}

// pass 2, workgroup `block` of THREADS threads (one row per thread); pass 1 ran n_count_blocks workgroups of
// COUNT_THREADS rows each (THREADS is a multiple of COUNT_THREADS)
template <int THREADS, int COUNT_THREADS>
__device__ __forceinline__ void plan_write_body(const ChooseArgs &a, const PlanTotals *totals, int block, int n_count_blocks) {
    __shared__ int lds[THREADS / PCG_WAVE];
    __shared__ long long lds64[THREADS / PCG_WAVE];
    const int rows = a.g.n_rel * a.B;
    long long run_cap = 0, all_cap = 0;
    int run_chunk = 0, all_chunk = 0;
    TierCounts run = {0, 0, 0, 0}, all = {0, 0, 0, 0};
    const int before = block * (THREADS / COUNT_THREADS);
    for (int bk = 0; bk < n_count_blocks; ++bk) {       // a few dozen uniform loads
        const PlanTotals t = totals[bk];
        if (bk < before) {
            run_cap += t.cap; run_chunk += t.chunk;
            run.n0 += t.n0; run.n1 += t.n1; run.n4 += t.n4; run.n16 += t.n16;
        }
        all_cap += t.cap; all_chunk += t.chunk;
        all.n0 += t.n0; all.n1 += t.n1; all.n4 += t.n4; all.n16 += t.n16;
    }
    const bool overflow = all_cap > a.w.list_capacity || (long long)all_chunk > a.w.chunk_cap;
    const int row = block * THREADS + (int)threadIdx.x;
    RowRec rec;
    rec.d = 0;
    int cap = 0;
    if (row < rows) {
        rec = a.w.recs[row];
        cap = row_cap(rec, a.add_self);
    }
    const int nch = (cap + CHUNK - 1) / CHUNK;
    long long t_cap, t_tiers;
    int t_chunk;
    const long long o_cap = run_cap + block_excl_scan<long long>((long long)cap, lds64, t_cap);
    const int o_chunk = run_chunk + block_excl_scan(nch, lds, t_chunk);
    TierCounts o = tier_unpack(block_excl_scan<long long>(row < rows ? tier_word(rec.d) : 0ll, lds64, t_tiers));
    o.n0 += run.n0; o.n1 += run.n1; o.n4 += run.n4; o.n16 += run.n16;
    if (row < rows) {
        a.w.row_begin[row] = o_cap;
        a.w.chunk_begin[row] = o_chunk;
        if (!overflow) {
            for (int j = 0; j < nch; ++j)
                a.w.chunk_desc[o_chunk + j] = make_int4(row, (int)o_cap + j * CHUNK, cap - j * CHUNK < CHUNK ? cap - j * CHUNK : CHUNK, nch);
            tier_push(a.w, rec.d, row, o);
        }
    }
    if (block == 0 && threadIdx.x == 0) {
        a.w.row_begin[rows] = all_cap;
        a.w.chunk_begin[rows] = all_chunk;
        tier_finish(a.w, all, overflow);
        a.w.counters[C_NCHUNK] = overflow ? 0 : all_chunk;
        if (overflow && a.status) atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
    }
}

__global__ void __launch_bounds__(PLAN_THREADS) plan_count(const ChooseArgs a, PlanTotals *totals) {
    plan_count_body<PLAN_THREADS>(a, totals, (int)blockIdx.x);
}
__global__ void __launch_bounds__(PLAN_THREADS) plan_write(const ChooseArgs a, const PlanTotals *totals) {
    plan_write_body<PLAN_THREADS, PLAN_THREADS>(a, totals, (int)blockIdx.x, (int)gridDim.x);
}

// The front of a training step in two launches instead of four: the plan's two passes ride along the score pass
// and the train-pos sort (both have idle CUs at dataset scale, and the plan needs neither's result):
//   front_a: [plan pass 1 workgroups | score_table workgroups]          (256 threads)
//   front_b: [plan pass 2 workgroups | rank-sort workgroups]            (1024 threads)
constexpr int FRONT_COUNT_THREADS = 256;

__global__ void __launch_bounds__(FRONT_COUNT_THREADS) front_a_kernel(const ChooseArgs a, PlanTotals *totals, int n_plan_blocks,
                                                                      int n_key_blocks, uint64_t *__restrict__ raw_keys,
                                                                      const float *__restrict__ W, const float *__restrict__ bias,
                                                                      int64_t row_begin, int64_t row_end, float *__restrict__ s0) {
    const int b = (int)blockIdx.x;
    if (b < n_plan_blocks)
        plan_count_body<FRONT_COUNT_THREADS>(a, totals, b);
    else if (b < n_plan_blocks + n_key_blocks)      // the train positives' sort keys, from their feature rows
        pos_key_body(a.g.X, a.g.feat_dim, a.g.feat_stride, W, bias, a.g.train_pos, a.g.n_pos, raw_keys, b - n_plan_blocks,
                     n_key_blocks);
    else
        score_table_body(a.g.X, a.g.feat_dim, a.g.feat_stride, W, bias, row_begin, row_end, s0,
                         b - n_plan_blocks - n_key_blocks, (int)gridDim.x - n_plan_blocks - n_key_blocks);
}

__global__ void __launch_bounds__(PLAN_THREADS) front_b_kernel(const ChooseArgs a, const PlanTotals *totals, int n_write_blocks,
                                                               int n_count_blocks, uint64_t *__restrict__ keys, int cap,
                                                               const uint64_t *__restrict__ raw_keys) {
    __shared__ uint64_t sh[RANK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    if ((int)blockIdx.x < n_write_blocks)
        plan_write_body<PLAN_THREADS, FRONT_COUNT_THREADS>(a, totals, (int)blockIdx.x, n_count_blocks);
    else
        rank_sort_body(a.s0, a.g.train_pos, a.g.n_pos, cap, keys, (int)blockIdx.x - n_write_blocks, sh, part, raw_keys);
}

// ---------------------------------------------------------------------------------------------
// select
// ---------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void grp_sync() {
    if constexpr (NW > 1) __syncthreads();
}

// exclusive prefix of a wave-uniform value over the group's waves, and the total
template <int NW>
__device__ __forceinline__ void grp_scan(int v, int wave, int lane, int *red, int &prefix, int &total) {
    if constexpr (NW == 1) {
        prefix = 0;
        total = v;
    } else {
        if (lane == 0) red[wave] = v;
        __syncthreads();
        int p = 0, t = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int x = red[w];
            if (w < wave) p += x;
            t += x;
        }
        __syncthreads();
        prefix = p;
        total = t;
    }
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    for (int o = 1; o < PCG_WAVE; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
        v = t < v ? t : v;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    for (int o = 1; o < PCG_WAVE; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_xor((int)v, o);
        v = t > v ? t : v;
    }
    return v;
}

__device__ __forceinline__ bool sorted_contains(const uint32_t *list, int n, uint32_t x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (list[mid] < x) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && list[lo] == x;
}

__device__ __forceinline__ float pos_score(const uint64_t *pk, int i) { return from_orderable((uint32_t)(pk[i] >> 32)); }
__device__ __forceinline__ uint32_t pos_dkey(const uint64_t *pk, int i, float c) { return dist_key(c, pos_score(pk, i)); }

// First x in [lo, hi] with pred(x) false, pred being true on a prefix of [lo, hi).
// 64 probes per step (one memory latency each) instead of one.
template <class Pred>
__device__ __forceinline__ int wave_partition_point(int lo, int hi, int lane, Pred pred) {
    for (;;) {
        const int n = hi - lo;
        if (n <= 0) return lo;
        if (n <= PCG_WAVE) {
            const int idx = lo + lane;
            return lo + wave_count(idx < hi && pred(idx));
        }
        const int step = (n + PCG_WAVE - 1) >> 6;
        int q = lo + (lane + 1) * step - 1;
        if (q > hi - 1) q = hi - 1;
        const int c = wave_count(pred(q));
        if (c == PCG_WAVE) return hi;
        int qc = lo + (c + 1) * step - 1;   // first probe that answered false
        if (qc > hi - 1) qc = hi - 1;
        if (c > 0) {
            int ql = lo + c * step - 1;
            if (ql > hi - 1) ql = hi - 1;
            lo = ql + 1;
        }
        hi = qc;
    }
}

// first index in [i0, end) whose distance key != kstar (or end); all lanes take part
__device__ __forceinline__ int run_end_fwd(const uint64_t *pk, float c, uint32_t kstar, int i0, int end, int lane) {
    for (int i = i0; i < end; i += PCG_WAVE) {
        const int j = i + lane;
        const bool same = j < end && pos_dkey(pk, j, c) == kstar;
        const uint64_t bad = ~__ballot(same);
        if (bad) {
            const int f = i + (__ffsll((unsigned long long)bad) - 1);
            return f < end ? f : end;
        }
    }
    return end;
}
// smallest x in [low, i0+1] such that every index in [x, i0] has key == kstar (i0+1 if none)
__device__ __forceinline__ int run_begin_bwd(const uint64_t *pk, float c, uint32_t kstar, int i0, int low, int lane) {
    for (int i = i0; i >= low; i -= PCG_WAVE) {
        const int j = i - lane;
        const bool same = j >= low && pos_dkey(pk, j, c) == kstar;
        const uint64_t bad = ~__ballot(same);
        if (bad) {
            const int f = i - (__ffsll((unsigned long long)bad) - 1);  // first non-matching going down
            return (f >= low ? f : low - 1) + 1;
        }
    }
    return low;
}

// rank of this lane's value among the valid lanes' values: lt = #smaller, eq = #equal (itself included)
__device__ __forceinline__ void wave_rank(uint32_t v, bool valid, int &lt, int &eq) {
    const uint64_t vm = __ballot(valid);
    lt = 0;
    eq = 0;
    for (int j = 0; j < PCG_WAVE; ++j) {
        if (!((vm >> j) & 1ull)) continue;                                   // wave-uniform
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)v, j);    // j is wave-uniform: v_readlane
        lt += o < v;
        eq += o == v;
    }
}
// r-th smallest (0-based) value among the valid lanes, given their (lt, eq)
__device__ __forceinline__ uint32_t wave_order_stat(uint32_t v, bool valid, int lt, int eq, int r) {
    const uint64_t hit = __ballot(valid && lt <= r && r < lt + eq);
    const int src = __ffsll((unsigned long long)hit) - 1;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src < 0 ? 0 : src);
}

// One counting pass over cur[0..n): c1 = #keys in [lo, p1), c2 = #keys in [lo, p2) (p1 <= p2 <= hi+1);
// mid_wave = this wave's share of the keys in [p1, p2).  cnts: 2 LDS ints (NW > 1).
template <int NW>
__device__ __forceinline__ void count_pass(const uint32_t *cur, int n, uint32_t lo, uint32_t hi, uint32_t p1,
                                           uint32_t p2, int wave, int lane, int *cnts, int &c1, int &c2, int &mid_wave) {
    constexpr int NT = NW * PCG_WAVE;
    if constexpr (NW > 1) {
        if (threadIdx.x < 2) cnts[threadIdx.x] = 0;
        __syncthreads();
    }
    int a1 = 0, a2 = 0;
    for (int base = wave * PCG_WAVE; base < n; base += NT) {   // wave-uniform trip count
        const int i = base + lane;
        const uint32_t key = i < n ? cur[i] : 0xFFFFFFFFu;
        const bool in = i < n && key >= lo && key <= hi;
        a1 += wave_count(in && key < p1);
        a2 += wave_count(in && key < p2);
    }
    mid_wave = a2 - a1;
    if constexpr (NW == 1) {
        c1 = a1;
        c2 = a2;
    } else {
        if (lane == 0) {
            atomicAdd(&cnts[0], a1);
            atomicAdd(&cnts[1], a2);
        }
        __syncthreads();
        c1 = cnts[0];
        c2 = cnts[1];
        __syncthreads();
    }
}

// Copies the keys of cur[0..n) that lie in [lo, hi] to dst (wave by wave, in order); `mine` = how many of them this
// wave's iterations hold (count_pass's mid_wave for the same bracket).  One scan instead of an atomic per iteration.
template <int NW>
__device__ __forceinline__ void compact_range(const uint32_t *cur, int n, uint32_t lo, uint32_t hi, uint32_t *dst,
                                              int mine, int wave, int lane, int *red) {
    constexpr int NT = NW * PCG_WAVE;
    int run, tot;
    grp_scan<NW>(mine, wave, lane, red, run, tot);
    for (int base = wave * PCG_WAVE; base < n; base += NT) {
        const int i = base + lane;
        const uint32_t key = i < n ? cur[i] : 0xFFFFFFFFu;
        const bool b = i < n && key >= lo && key <= hi;
        const uint64_t mm = __ballot(b);
        if (b) dst[run + __popcll(mm & lanemask_lt())] = key;
        run += __popcll(mm);
    }
    grp_sync<NW>();
}

// ---------------------------------------------------------------------------
// One (relation, centre) row: choose; write the row's region of the list.
// ---------------------------------------------------------------------------
// keys: >= deg uint32 distance keys (one wave: later the kept ids, compacted in place); ids: >= deg, the row's neighbour
// ids; ckeys: >= deg, survivors of the first bracketing pass, later the kept ids of a workgroup row (null for one wave);
// cand: 64; red: 2*NW+2 ints (NW > 1).  LDS or, for over-long hub rows, global scratch - separate instantiations, so that
// the LDS ones compile to ds_* instructions.
// CSR_IDS: the neighbour ids are not staged (ids unused) but read again from the CSR row when the kept ones are
// compacted - for hub rows, where the LDS is better spent on keys and survivors and the re-read is coalesced and L2-hot.
template <int NW, bool CSR_IDS = false>
__device__ __forceinline__ void select_row(const ChooseArgs &a, int row, uint32_t *keys, uint32_t *ids, uint32_t *ckeys,
                                           uint32_t *cand, int *red) {
    const int lane = lane_id();
    const int wave = (NW > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int tid = wave * PCG_WAVE + lane;
    constexpr int NT = NW * PCG_WAVE;

    if (a.stamps && tid == 0) a.stamps[(size_t)row * 8] = wall_clock64() | ((unsigned long long)blockIdx.x << 54);   // + who ran it
    const RowRec p = a.w.recs[row];
    const int d = p.d, k = p.k, m = p.m, node = p.node;
    const bool keep_all = p.keep_all != 0;
    const int r = row / a.B;
    const int32_t *__restrict__ nbr = a.g.indices[r] + p.start;
    // the centre's class-0 logit (not part of the plan record: the plan can then run before / beside the score pass)
    const float c = a.center_s0 ? a.center_s0[row - r * a.B] : a.s0[node];
    int32_t *__restrict__ out = a.w.list + a.w.row_begin[row];

    // ---- 1. neighbour ids and distance keys -> LDS ----------------------------
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (int base = tid; base < d; base += NT * KEY_UNROLL) {
        uint32_t id[KEY_UNROLL];
        float sc[KEY_UNROLL];
#pragma unroll
        for (int u = 0; u < KEY_UNROLL; ++u) {
            const int i = base + u * NT;
            id[u] = i < d ? (uint32_t)nbr[i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < KEY_UNROLL; ++u) sc[u] = keep_all ? 0.f : a.s0[id[u]];
#pragma unroll
        for (int u = 0; u < KEY_UNROLL; ++u) {
            const int i = base + u * NT;
            if (i < d) {
                if constexpr (!CSR_IDS) ids[i] = id[u];
                if (!keep_all) {
                    const uint32_t key = dist_key(c, sc[u]);
                    keys[i] = key;
                    kmin = key < kmin ? key : kmin;
                    kmax = key > kmax ? key : kmax;
                }
            }
        }
    }
    grp_sync<NW>();
    PCG_STAMP(1);

    // ---- 2. k-th smallest key: bracketed counting search ------------------------
    // invariant: the k-th smallest key lies in [lo, hi]; below = #keys < lo; ncand = #keys in [lo, hi]
    uint32_t kstar = 0xFFFFFFFFu;
    int need = 0;          // how many of the keys == kstar are kept (in position order)
    int n_equal = 0;       // how many keys == kstar there are
    if (!keep_all) {
        kmin = wave_min_u32(kmin);
        kmax = wave_max_u32(kmax);
        if constexpr (NW > 1) {
            if (lane == 0) {
                red[wave] = (int)kmin;
                red[NW + wave] = (int)kmax;
            }
            __syncthreads();
            for (int w = 0; w < NW; ++w) {
                const uint32_t x = (uint32_t)red[w], y = (uint32_t)red[NW + w];
                kmin = x < kmin ? x : kmin;
                kmax = y > kmax ? y : kmax;
            }
            __syncthreads();
        }
        uint32_t lo = kmin, hi = kmax;
        int below = 0, ncand = d, round = 0;
        const uint32_t *cur = keys;      // where the candidates live: the row's keys, or the compacted survivors
        int n_cur = d;
        while (ncand > PCG_WAVE && lo < hi) {
            // two pivots p1 <= p2 in (lo, hi] bracket the wanted rank: interpolation in value space, the
            // bracket sized to hold ~48 candidates if they were uniform (never narrower than 16 % of the
            // interval); every 4th round is a bit-space midpoint, which bounds the rounds for any input
            const uint32_t span = hi - lo;
            uint32_t p1 = lo + (span >> 1) + (span & 1u), p2 = p1;           // bit-space midpoint, in (lo, hi]
            if ((round & 3) != 3) {
                const float lv = __uint_as_float(lo), hv = __uint_as_float(hi);
                const float inv = 1.f / (float)ncand;
                const float f = ((float)(k - below) - 0.5f) * inv;
                const float w = fmaxf(0.08f, 24.f * inv);
                const float g1 = lv + (f - w) * (hv - lv), g2 = lv + (f + w) * (hv - lv);
                p1 = (g1 > lv && g1 <= hv) ? __float_as_uint(g1) : lo + 1;
                p2 = (g2 > lv && g2 <= hv) ? __float_as_uint(g2) : hi + 1;    // hi + 1: no upper cut (keys < 2^31: no wrap)
                if (p2 < p1) p2 = p1;
            }
            const bool first = cur == keys && ckeys != nullptr;
            int c1, c2, mid_wave;
            count_pass<NW>(cur, n_cur, lo, hi, p1, p2, wave, lane, red, c1, c2, mid_wave);
            if (below + c1 >= k) {              // k-th < p1
                hi = p1 - 1;
                ncand = c1;
            } else if (below + c2 >= k) {       // p1 <= k-th < p2
                lo = p1;
                hi = p2 - 1;
                below += c1;
                ncand = c2 - c1;
                if (first) {                    // compact the survivors: later passes are short
                    compact_range<NW>(cur, n_cur, lo, hi, ckeys, mid_wave, wave, lane, red);
                    cur = ckeys;
                    n_cur = ncand;
                }
            } else {                            // k-th >= p2
                lo = p2;
                below += c2;
                ncand -= c2;
            }
            ++round;
        }
        if (a.stamps && tid == 0) a.stamps[(size_t)row * 8 + 7] = (unsigned long long)round | ((unsigned long long)ncand << 32);
        if (lo == hi) {
            kstar = lo;
            need = k - below;
            n_equal = ncand;
        } else {
            // <= 64 survivors: one per lane, ranked in-wave
            constexpr bool coop = NW > 1;
            uint32_t *cd = cand;
            if (coop) {
                if (threadIdx.x == 0) red[2] = 0;
                __syncthreads();
            }
            int seen = 0;
            for (int base = coop ? wave * PCG_WAVE : 0; base < n_cur; base += coop ? NT : PCG_WAVE) {
                const int i = base + lane;
                const uint32_t key = i < n_cur ? cur[i] : 0u;
                const bool isc = i < n_cur && key >= lo && key <= hi;
                const uint64_t bm = __ballot(isc);
                const int cn = __popcll(bm);
                int at = seen;
                if (coop) {
                    int o = 0;
                    if (lane == 0 && cn) o = atomicAdd(&red[2], cn);
                    at = __builtin_amdgcn_readfirstlane(o);
                }
                if (isc) cd[at + __popcll(bm & lanemask_lt())] = key;
                seen += cn;
            }
            if (coop) __syncthreads();
            const bool have = lane < ncand;
            const uint32_t ck = have ? cd[lane] : 0xFFFFFFFFu;
            // MSB-first bisection on one register per lane, from the first bit in which lo and hi differ
            int remaining = k - below, nc = ncand;
            int bit = 31 - __clz((int)(lo ^ hi));
            uint32_t prefix = (bit == 31) ? 0u : (lo & (0xFFFFFFFFu << (bit + 1)));
            for (; bit >= 0; --bit) {
                const uint32_t hm = (bit == 31) ? 0u : (0xFFFFFFFFu << (bit + 1));
                const int c0 = wave_count(have && ((ck & hm) == prefix) && !((ck >> bit) & 1u));
                if (remaining > c0) {
                    remaining -= c0;
                    prefix |= 1u << bit;
                    nc -= c0;
                } else {
                    nc = c0;
                }
            }
            kstar = prefix;
            need = remaining;
            n_equal = nc;
            grp_sync<NW>();
        }
    }

    PCG_STAMP(2);
    // ---- 3. compaction of kept ids, ascending: in place into keys[] (one wave), into ckeys[] (workgroup: the
    //         survivors are dead by now, and a separate target lets every wave own a contiguous stretch of the row) ---
    uint32_t *selbuf = (NW > 1) ? ckeys : keys;
    auto nbr_id = [&](int i) { return CSR_IDS ? (uint32_t)nbr[i] : ids[i]; };
    int ns = 0;
    const bool ranked_ties = n_equal != need;   // some, not all, of the equal keys are kept
    if (keep_all) {
        ns = d;
        for (int i = tid; i < d; i += NT) selbuf[i] = nbr_id(i);
    } else if (NW == 1 || ranked_ties) {
        int ties_seen = 0;
        for (int base = 0; base < d; base += NT) {
            const int i = base + tid;
            const bool in = i < d;
            const uint32_t key = in ? keys[i] : 0u;
            const uint32_t id = in ? nbr_id(i) : 0u;
            bool sel = in && key <= kstar;
            if (ranked_ties) {
                const bool tie = in && key == kstar;
                const uint64_t tm = __ballot(tie);
                int tpre, ttot;
                grp_scan<NW>(__popcll(tm), wave, lane, red, tpre, ttot);
                const int trank = ties_seen + tpre + __popcll(tm & lanemask_lt());
                sel = in && (key < kstar || (tie && trank < need));
                ties_seen += ttot;
            }
            const uint64_t sm = __ballot(sel);
            int spre, stot;
            grp_scan<NW>(__popcll(sm), wave, lane, red, spre, stot);
            if (sel) selbuf[ns + spre + __popcll(sm & lanemask_lt())] = id;   // one wave, in place: slot <= i, read above
            ns += stot;
        }
    } else {
        // every key <= kstar is kept: count per contiguous stretch, one scan, then write - 2 barriers per row
        const int seg = (((d + NW - 1) / NW) + PCG_WAVE - 1) & ~(PCG_WAVE - 1);
        const int b0 = wave * seg;
        const int e0 = b0 + seg < d ? b0 + seg : d;
        int mine = 0;
        for (int i0 = b0; i0 < e0; i0 += PCG_WAVE) {
            const int i = i0 + lane;
            mine += wave_count(i < e0 && keys[i] <= kstar);
        }
        int run;
        grp_scan<NW>(mine, wave, lane, red, run, ns);
        constexpr int CU = 4;     // ids of 4 iterations in flight (CSR_IDS: global loads, one latency per batch)
        for (int i0 = b0; i0 < e0; i0 += CU * PCG_WAVE) {
            uint32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                idv[u] = i < e0 ? nbr_id(i) : 0u;
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                const bool sel = i < e0 && keys[i] <= kstar;
                const uint64_t sm = __ballot(sel);
                if (sel) selbuf[run + __popcll(sm & lanemask_lt())] = idv[u];
                run += __popcll(sm);
            }
        }
    }
    grp_sync<NW>();
    const uint32_t *sel = selbuf;
    PCG_STAMP(3);
    // (the kept ids are written to the list at the very end: vmcnt orders loads behind older stores, and the
    //  minority search below is a chain of dependent loads that must not queue behind ~ns stores to fresh lines)

    // ---- 4. minority over-sampling (layers.py:675-691): slots out[ns .. ns + mt) -----------
    int mt = 0;        // slots used
    int valid = 0;     // per-thread count of non-duplicate minority picks
    if (m > 0) {
        const uint64_t *__restrict__ pk = a.pos_keys;
        const int P = a.g.n_pos;
        int L, R, L2, R2, tau = INT_MAX, need_t = 0;
        if (m >= P) {
            L = L2 = 0;
            R = R2 = P;
        } else {
            // window [lo, lo+m) of the m nearest: first lo whose left end is not farther than the element right of the window
            const int lo = wave_partition_point(0, P - m, lane, [&](int x) {
                return (c - pos_score(pk, x)) > (pos_score(pk, x + m) - c);
            });
            // one batch of six independent loads decides the usual tie-free case
            const uint32_t NOKEY = 0xFFFFFFFEu;    // never equals a distance key (keys have bit 31 clear)
            const uint32_t ka = pos_dkey(pk, lo, c), kb = pos_dkey(pk, lo + m - 1, c);
            const uint32_t ka1 = m > 1 ? pos_dkey(pk, lo + 1, c) : NOKEY;
            const uint32_t kb1 = m > 1 ? pos_dkey(pk, lo + m - 2, c) : NOKEY;
            const uint32_t kl = lo > 0 ? pos_dkey(pk, lo - 1, c) : NOKEY;
            const uint32_t kr = lo + m < P ? pos_dkey(pk, lo + m, c) : NOKEY;
            const uint32_t ks = ka > kb ? ka : kb;  // m-th smallest distance
            const bool tie_l = ka == ks, tie_r = kb == ks;
            const bool none_outside = kl != ks && kr != ks;
            if (none_outside && m == 1) {                  // the window is one element; it is the single tie
                L2 = lo;
                L = R = R2 = lo + 1;
            } else if (none_outside && !(tie_l && tie_r) && (tie_l ? ka1 != ks : kb1 != ks)) {
                // exactly one element at distance ks, at one end of the window; no tie outside it
                L2 = L = tie_l ? lo + 1 : lo;
                R = R2 = tie_l ? lo + m : lo + m - 1;
                if (tie_l) L2 = lo; else R2 = lo + m;      // that one element is the (single) tie, and it is taken
            } else {
                L = run_end_fwd(pk, c, ks, lo, lo + m, lane);
                R = (L == lo + m) ? L : run_begin_bwd(pk, c, ks, lo + m - 1, L, lane);
                L2 = run_begin_bwd(pk, c, ks, lo - 1, 0, lane);
                R2 = run_end_fwd(pk, c, ks, lo + m, P, lane);
            }
            // ties are [L2, L) and [R, R2); strictly nearer ones are [L, R)
            need_t = m - (R - L);
            const int T = (L - L2) + (R2 - R);
            if (T == need_t) {               // every tie is taken (the usual case): one contiguous, fully parallel range
                L = L2;
                R = R2;
                need_t = 0;
            } else if (T > PCG_WAVE) {       // many ties: threshold on the train_pos position by bisection
                int plo = 0, phi = P - 1;
                while (plo < phi) {
                    const int mid = (plo + phi) >> 1;
                    int cn = 0;
                    for (int i0 = L2; i0 < L; i0 += PCG_WAVE) {
                        const int i = i0 + lane;
                        cn += wave_count(i < L && (int)(uint32_t)pk[i] <= mid);
                    }
                    for (int i0 = R; i0 < R2; i0 += PCG_WAVE) {
                        const int i = i0 + lane;
                        cn += wave_count(i < R2 && (int)(uint32_t)pk[i] <= mid);
                    }
                    if (cn >= need_t) phi = mid;
                    else plo = mid + 1;
                }
                tau = plo;
            } else {                          // a few ties: one per lane, the need_t smallest positions by in-register ranking
                const int nl = L - L2;
                const int ti = lane < nl ? L2 + lane : R + (lane - nl);
                const bool tv = lane < T;
                const int tp = tv ? (int)(uint32_t)pk[ti] : INT_MAX;
                int rank = 0;
                for (int j = 0; j < T; ++j) rank += __builtin_amdgcn_readlane(tp, j) < tp;
                // tau = the need_t-th smallest position among the ties (positions are distinct)
                const uint64_t hit = __ballot(tv && rank == need_t - 1);
                const int src = __ffsll((unsigned long long)hit) - 1;
                tau = __builtin_amdgcn_readlane(tp, src < 0 ? 0 : src);
            }
        }
        PCG_STAMP(4);
        // strictly nearer ones: slot = i - L, every thread of the group strides over them
        const int n_strict = R - L;
        for (int base = tid; base < n_strict; base += NT * KEY_UNROLL) {
            uint32_t pos[KEY_UNROLL], u[KEY_UNROLL];
#pragma unroll
            for (int x = 0; x < KEY_UNROLL; ++x) {
                const int j = base + x * NT;
                pos[x] = j < n_strict ? (uint32_t)pk[L + j] : 0u;
            }
#pragma unroll
            for (int x = 0; x < KEY_UNROLL; ++x) u[x] = (uint32_t)a.g.train_pos[pos[x]];
#pragma unroll
            for (int x = 0; x < KEY_UNROLL; ++x) {
                const int j = base + x * NT;
                if (j < n_strict) {
                    const bool dup = sorted_contains(sel, ns, u[x]) || (a.add_self && u[x] == (uint32_t)node);  // set(), :694
                    out[ns + j] = dup ? -1 : (int32_t)u[x];
                    valid += !dup;
                }
            }
        }
        mt = n_strict;
        // the (rare) ties at the m-th distance: first wave, in window order
        if (need_t > 0) {
            int taken = 0;
            for (int part = 0; part < 2; ++part) {
                const int s0i = part == 0 ? L2 : R, e0i = part == 0 ? L : R2;
                for (int i0 = s0i; i0 < e0i; i0 += PCG_WAVE) {
                    const int i = i0 + lane;
                    bool take = false;
                    uint32_t u = 0;
                    if (i < e0i) {
                        const uint32_t pos = (uint32_t)pk[i];
                        take = (int)pos <= tau;
                        if (take) u = (uint32_t)a.g.train_pos[pos];
                    }
                    const uint64_t tmk = __ballot(take);
                    if (take && wave == 0) {
                        const bool dup = sorted_contains(sel, ns, u) || (a.add_self && u == (uint32_t)node);
                        out[ns + n_strict + taken + __popcll(tmk & lanemask_lt())] = dup ? -1 : (int32_t)u;
                        valid += !dup;
                    }
                    taken += __popcll(tmk);
                }
            }
            mt += taken;
        }
    }
    PCG_STAMP(5);
    // GCN-style self union (graphsage.py:78-79, 214): the centre joins its own set
    int n_self = 0;
    if (a.add_self && !sorted_contains(sel, ns, (uint32_t)node)) {
        n_self = 1;
        if (tid == 0) out[ns + mt] = node;
    }
    // |set| = kept + non-duplicate minority picks + self
    int vsum = valid;
    for (int o = 1; o < PCG_WAVE; o <<= 1) vsum += __shfl_xor(vsum, o);
    int vpre, vtot;
    grp_scan<NW>(vsum, wave, lane, red, vpre, vtot);
    for (int i = tid; i < ns; i += NT) out[i] = (int32_t)sel[i];
    const int used = ns + mt + n_self;
    const int cap = (int)(a.w.row_begin[row + 1] - a.w.row_begin[row]);
    for (int i = used + tid; i < cap; i += NT) out[i] = -1;   // unused tail of the region: the whole list prefix stays clean
    if (tid == 0) {
        a.w.len[row] = used;
        a.cnt[row] = ns + vtot + n_self;
    }
    grp_sync<NW>();
    PCG_STAMP(6);
}

// One persistent launch selects every row of the batch, longest rows first:
//   1. hub rows (deg > 4096): the whole 8-wave workgroup per row, neighbour ids re-read from the CSR row; up to 6144
//      neighbours keys + first-pass survivors in LDS, up to 12288 keys in LDS and survivors in this workgroup's global
//      scratch, everything in scratch beyond that
//   2. mid rows (512 < deg <= 4096): the whole workgroup per row, keys + ids + survivors in LDS
//   3. rows of 129..512, then rows of <= 128 neighbours: one wave per row
// Every workgroup walks the four queues in this order, so the long poles start first and the short rows fill in
// behind them on whatever wave slots are free - without the cross-stream fork / join this used to take (inside a
// captured step the branches cost more than they overlapped: 0.43 -> 0.36 ms per step on the 2 M-node graph).
__global__ void __launch_bounds__(SEL_NW *PCG_WAVE) __attribute__((amdgpu_waves_per_eu(6, 8))) select_rows(const ChooseArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t *cand = lds + SEL_LDS_WORDS;                        // [SEL_NW][64]
    int *red = reinterpret_cast<int *>(cand + SEL_NW * PCG_WAVE);
    const int wave = threadIdx.x >> 6;
    // Rows are assigned statically - a queue head bumped by every wave is thousands of same-address device-scope
    // atomics from 8 XCDs, which serialise at the memory side (measured: 590 us instead of 60).
    // wide rows: the virtual queue [hub rows | mid rows], strided over the workgroups from the first one on
    const int n16 = (int)a.w.counters[C_N16], n4 = (int)a.w.counters[C_N4];
    const size_t md = (size_t)a.g.max_degree;
    uint32_t *gk = a.w.scratch + (size_t)blockIdx.x * 2 * md;     // only touched when hub rows exist (then it is allocated)
    for (int j = (int)blockIdx.x; j < n16 + n4; j += SEL_BLOCKS) {
        if (j < n16) {
            const int row = __builtin_amdgcn_readfirstlane(a.w.q16[j]);     // one row per workgroup: scalar
            const int d = a.w.recs[row].d;
            if (d <= HUB_LDS_CAP) select_row<SEL_NW, true>(a, row, lds, nullptr, lds + HUB_LDS_CAP, cand, red);       // keys + survivors in LDS
            else if (d <= SEL_LDS_WORDS) select_row<SEL_NW, true>(a, row, lds, nullptr, gk, cand, red);            // keys in LDS
            else select_row<SEL_NW, true>(a, row, gk, nullptr, gk + md, cand, red);                                // all in scratch
        } else {
            const int row = __builtin_amdgcn_readfirstlane(a.w.q4[j - n16]);
            select_row<SEL_NW>(a, row, lds, lds + T4_CAP, lds + 2 * T4_CAP, cand, red);
        }
    }
    // single-wave rows: the virtual queue [129..512 | <= 128], strided over the waves from the LAST workgroup on, so that
    // with few wide rows the workgroups holding those are not the ones holding short rows too.
    // No compaction buffer here (rows <= 512: later passes just re-filter the keys).
    const int n1 = (int)a.w.counters[C_N1], n0 = (int)a.w.counters[C_N0];
    uint32_t *wk = lds + wave * (2 * T1_CAP);
    for (int j = (SEL_BLOCKS - 1 - (int)blockIdx.x) * SEL_NW + wave; j < n1 + n0; j += SEL_BLOCKS * SEL_NW) {
        const int row = __builtin_amdgcn_readfirstlane(j < n1 ? a.w.q1[j] : a.w.q0[j - n1]);
        select_row<1>(a, row, wk, wk + T1_CAP, nullptr, cand + wave * PCG_WAVE, nullptr);
    }
}

static size_t select_smem_bytes() {
    return sizeof(uint32_t) * (SEL_LDS_WORDS + SEL_NW * PCG_WAVE) + sizeof(int) * (2 * SEL_NW + 2);
}

// ---------------------------------------------------------------------------------------------
// gather + combine
// ---------------------------------------------------------------------------------------------
struct AggArgs {
    const float *X;
    int32_t feat_dim, feat_stride;
    int32_t n_rows;             // n_rel * B
    const int64_t *row_begin;
    const int32_t *chunk_begin;
    const int32_t *len;
    const int32_t *cnt;
    const int4 *chunk_desc;
    int32_t chunk_cap;
    const int32_t *list;
    const uint32_t *n_chunks;   // device word
    float *partial;
    float *agg;                 // [n_rows, agg_stride]
    int32_t agg_stride, norm;
};

struct RowGeom {  // how one wave-instruction covers feature rows
    int lpr, rpw, slot, sub, nch;
};

__device__ __forceinline__ RowGeom row_geom(int stride, int lane) {
    RowGeom q;
    q.lpr = lanes_per_row(stride);
    q.rpw = PCG_WAVE / q.lpr;
    q.slot = lane / q.lpr;
    q.sub = lane % q.lpr;
    q.nch = stride >> 2;
    return q;
}

template <int NACC>
__device__ __forceinline__ void store_row(float *out, const float4 (&acc)[NACC], const RowGeom &q, int feat_dim,
                                          float den) {
#pragma unroll
    for (int x = 0; x < NACC; ++x) {
        const int ch = x * q.lpr + q.sub;
        if (ch >= q.nch) continue;
        const int f = 4 * ch;
        if (f + 0 < feat_dim) out[f + 0] = acc[x].x / den;
        if (f + 1 < feat_dim) out[f + 1] = acc[x].y / den;
        if (f + 2 < feat_dim) out[f + 2] = acc[x].z / den;
        if (f + 3 < feat_dim) out[f + 3] = acc[x].w / den;
    }
}

template <int NACC>
__global__ void __launch_bounds__(256) gather_chunks(const AggArgs a) {
    const int lane = lane_id();
    const RowGeom q = row_geom(a.feat_stride, lane);
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t ch0 = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // the chunk's descriptor (written by the plan) is requested together with the chunk count: one load level instead of
    // four (count -> row -> offsets -> list).  n is the chunk's share of the row's REGION: entries past the ones the
    // select wrote are -1 (it fills the tail), so they add nothing.
    int4 desc = a.chunk_desc[ch0 < (uint32_t)a.chunk_cap ? ch0 : 0u];
    const uint32_t total = *a.n_chunks;
    for (uint32_t ch = ch0; ch < total; ch += nwaves) {
        if (ch != ch0) desc = a.chunk_desc[ch];
        const int row = desc.x, n = desc.z, nch_row = desc.w;
        const int32_t *__restrict__ list = a.list + desc.y;
        const int cnt = nch_row == 1 ? a.cnt[row] : 1;
        float4 acc[NACC];
#pragma unroll
        for (int x = 0; x < NACC; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int per_iter = q.rpw * UNROLL;
        for (int base = 0; base < n; base += per_iter) {
            float4 v[UNROLL][NACC];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int i = base + u * q.rpw + q.slot;
                const int id = i < n ? list[i] : -1;                  // -1: hole left by a duplicate
                const bool ok = id >= 0;
                const float *rowp = a.X + (size_t)(ok ? id : 0) * a.feat_stride;
#pragma unroll
                for (int x = 0; x < NACC; ++x) {
                    const int c4 = x * q.lpr + q.sub;
                    v[u][x] = (ok && c4 < q.nch) ? *reinterpret_cast<const float4 *>(rowp + 4 * c4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int x = 0; x < NACC; ++x) {
                    acc[x].x += v[u][x].x;
                    acc[x].y += v[u][x].y;
                    acc[x].z += v[u][x].z;
                    acc[x].w += v[u][x].w;
                }
        }
#pragma unroll
        for (int x = 0; x < NACC; ++x)
            for (int o = q.lpr; o < PCG_WAVE; o <<= 1) {
                acc[x].x += __shfl_xor(acc[x].x, o);
                acc[x].y += __shfl_xor(acc[x].y, o);
                acc[x].z += __shfl_xor(acc[x].z, o);
                acc[x].w += __shfl_xor(acc[x].w, o);
            }
        if (lane < q.lpr) {
            if (nch_row == 1) {
                const float den = a.norm == PCG_NORM_SQRT_COUNT ? sqrtf((float)cnt) : (float)cnt;
                store_row<NACC>(a.agg + (size_t)row * a.agg_stride, acc, q, a.feat_dim, den);
            } else {
                float *pp = a.partial + (size_t)ch * a.feat_stride;
#pragma unroll
                for (int x = 0; x < NACC; ++x) {
                    const int c4 = x * q.lpr + q.sub;
                    if (c4 < q.nch) *reinterpret_cast<float4 *>(pp + 4 * c4) = acc[x];
                }
            }
        }
    }
}

// rows longer than one chunk: add the partial sums in chunk order
template <int NACC>
__global__ void __launch_bounds__(256) combine_rows(const AggArgs a) {
    const int lane = lane_id();
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= a.n_rows || *a.n_chunks == 0) return;
    const int cb = a.chunk_begin[row], nch_row = a.chunk_begin[row + 1] - cb;
    if (nch_row <= 1) {
        if (nch_row == 0 && lane < a.feat_dim) {   // cap == 0: empty set -> 0/0 like the reference's mask.div
            for (int f = lane; f < a.feat_dim; f += PCG_WAVE) a.agg[(size_t)row * a.agg_stride + f] = 0.f / 0.f;
        }
        return;
    }
    const RowGeom q = row_geom(a.feat_stride, lane);
    if (lane >= q.lpr) return;
    const int cnt = a.cnt[row];
    float4 acc[NACC];
#pragma unroll
    for (int x = 0; x < NACC; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    // the partial sums are added in chunk order; their loads are issued a batch at a time (the row with the most chunks
    // sets this kernel's duration: one load latency per batch instead of one per chunk)
    constexpr int CB = NACC == 1 ? 16 : 8;
    for (int j0 = 0; j0 < nch_row; j0 += CB) {
        float4 t[CB][NACC];
#pragma unroll
        for (int u = 0; u < CB; ++u) {
            const float *pp = a.partial + (size_t)(cb + j0 + u) * a.feat_stride;
#pragma unroll
            for (int x = 0; x < NACC; ++x) {
                const int c4 = x * q.lpr + q.sub;
                t[u][x] = (j0 + u < nch_row && c4 < q.nch) ? *reinterpret_cast<const float4 *>(pp + 4 * c4)
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < CB; ++u)
#pragma unroll
            for (int x = 0; x < NACC; ++x) {
                if (j0 + u < nch_row) {
                    acc[x].x += t[u][x].x; acc[x].y += t[u][x].y; acc[x].z += t[u][x].z; acc[x].w += t[u][x].w;
                }
            }
    }
    const float den = a.norm == PCG_NORM_SQRT_COUNT ? sqrtf((float)cnt) : (float)cnt;
    store_row<NACC>(a.agg + (size_t)row * a.agg_stride, acc, q, a.feat_dim, den);
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
    hipLaunchKernelGGL(select_rows, dim3(SEL_BLOCKS), dim3(SEL_NW * PCG_WAVE), select_smem_bytes(), st, a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

template <int NACC>
static int launch_aggregate(const AggArgs &g, hipStream_t st) {
    hipLaunchKernelGGL(gather_chunks<NACC>, dim3(GATHER_BLOCKS), dim3(256), 0, st, g);
    PCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(combine_rows<NACC>, dim3((g.n_rows + 3) / 4), dim3(256), 0, st, g);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

static unsigned long long *g_stamps = nullptr;

}  // namespace pcg

extern "C" {

/* diagnostic: per-row phase timestamps of the select kernels ([rows][8] uint64, wall_clock64 ticks
 * = 10 ns); pass NULL to switch off.  Not part of the product path. */
void pcg_debug_set_stamps(void *ptr) { pcg::g_stamps = static_cast<unsigned long long *>(ptr); }

int64_t pcg_choose_workspace_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity) {
    if (!g || B < 0 || list_capacity < 0 || list_capacity >= (1ll << 31)) return PCG_E_ARG;
    return pcg::carve(g, B, list_capacity, nullptr, nullptr);
}

int64_t pcg_choose_workspace_offset(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, int32_t which) {
    if (!g || B < 0 || list_capacity < 0) return PCG_E_ARG;
    pcg::Workspace w;
    unsigned char *base = reinterpret_cast<unsigned char *>(4096);   // fake base, only differences are used
    pcg::carve(g, B, list_capacity, base, &w);
    switch (which) {
        case 0: return reinterpret_cast<unsigned char *>(w.row_begin) - base;
        case 1: return reinterpret_cast<unsigned char *>(w.len) - base;
        case 2: return reinterpret_cast<unsigned char *>(w.list) - base;
        case 3: return reinterpret_cast<unsigned char *>(w.chunk_begin) - base;
        case 4: return reinterpret_cast<unsigned char *>(w.chunk_desc) - base;
        case 5: return reinterpret_cast<unsigned char *>(w.counters) - base;
        case 6: return reinterpret_cast<unsigned char *>(w.partial) - base;
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
                       int64_t list_capacity, uint32_t *status, bool plan_only = false) {
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
    a.pos_keys = pos_keys;
    for (int r = 0; r < PCG_MAX_REL; ++r) a.thr[r] = r < g->n_rel ? thresholds[r] : 0.0;
    for (int r = 0; r < PCG_MAX_REL; ++r) a.rho[r] = (r < g->n_rel && rho) ? rho[r] : 0.0;
    a.train_flag = train_flag;
    a.add_self = add_self;
    a.cnt = cnt;
    a.status = status;
    a.stamps = pcg::g_stamps;
    pcg::carve(g, B, list_capacity, static_cast<unsigned char *>(workspace), &a.w);
    return PCG_OK;
}

static int choose_select(bool planned, const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const float *s0, const float *center_s0, const uint64_t *pos_keys, const double *thresholds,
                         const double *rho, int32_t train_flag, int32_t add_self, int32_t *cnt, void *workspace,
                         int64_t list_capacity, uint32_t *status, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;  // empty trailing batch (model_handler.py:134 produces one): nothing to do
    if (!cnt) return PCG_E_ARG;
    pcg::ChooseArgs a;
    const int rc = choose_args(a, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                               workspace, list_capacity, status);
    if (rc != PCG_OK) return rc;
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
                              int64_t list_capacity, uint32_t *status, void *stream) {
    return choose_select(true, g, nodes, labels, B, s0, center_s0, pos_keys, thresholds, rho, train_flag, add_self, cnt,
                         workspace, list_capacity, status, stream);
}

/* first half: class-0 logits of rows [row_begin, row_end) -> s0_out[row]  ||  plan pass 1 */
int pcg_step_front_a(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end,
                     float *s0_out, uint64_t *pos_keys, const int32_t *nodes, const int32_t *labels, int32_t B,
                     const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *workspace,
                     int64_t list_capacity, uint32_t *status, void *stream) {
    if (!g || !g->X || !W || !b || !s0_out || B < 0) return PCG_E_ARG;
    if (g->feat_dim < 1 || g->feat_stride < g->feat_dim || g->feat_stride % 4 != 0) return PCG_E_ARG;
    if ((reinterpret_cast<uintptr_t>(g->X) & 15u) != 0) return PCG_E_ARG;
    if (row_begin < 0 || row_end > g->n_nodes || row_begin > row_end) return PCG_E_ARG;
    if (B == 0) return pcg_score_table(g, W, b, row_begin, row_end, s0_out, stream);   // (then _b gathers its keys itself)
    pcg::ChooseArgs a;
    const int rc = choose_args(a, g, nodes, labels, B, nullptr, nullptr, nullptr, thresholds, rho, train_flag, add_self, nullptr,
                               workspace, list_capacity, status, true);
    if (rc != PCG_OK) return rc;
    const int rows = g->n_rel * B;
    pcg::PlanTotals *tot = reinterpret_cast<pcg::PlanTotals *>(a.w.plan_totals);
    const int n_count = (rows + pcg::FRONT_COUNT_THREADS - 1) / pcg::FRONT_COUNT_THREADS;
    const int n_score = (int)pcg::score_table_blocks(row_end - row_begin, g->feat_stride);
    // the train positives' unsorted keys go to the scratch half of pos_keys (rank-sort sizes only)
    const bool raw = pos_keys && train_flag && g->n_pos > 0 && g->n_pos <= pcg::RANK_MAX && g->train_pos;
    const int rows_per_block = 4 * (PCG_WAVE / pcg::lanes_per_row(g->feat_stride));
    int n_key = raw ? (g->n_pos + rows_per_block - 1) / rows_per_block : 0;
    if (n_key > 256) n_key = 256;
    uint64_t *raw_keys = raw ? pos_keys + pcg_pos_sort_capacity(g->n_pos) / 2 : nullptr;
    hipLaunchKernelGGL(pcg::front_a_kernel, dim3(n_count + n_key + n_score), dim3(pcg::FRONT_COUNT_THREADS), 0,
                       static_cast<hipStream_t>(stream), a, tot, n_count, n_key, raw_keys, W, b, row_begin, row_end, s0_out);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

/* second half: train-pos sort by s0 (if train_flag and n_pos > 0)  ||  plan pass 2 */
int pcg_step_front_b(const pcg_graph_desc *g, const float *s0, uint64_t *pos_keys, int32_t raw_keys_ready,
                     const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                     int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status,
                     void *stream) {
    if (!g || !s0 || B < 0) return PCG_E_ARG;
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
                       tot, n_write, n_count, pos_keys, (int)cap, raw_keys);
    PCG_LAUNCH_CHECK();
    if (sort && !rank) return pcg_pos_sort(g, s0, pos_keys, stream);    // many positives: the chunk sort's own launches
    return PCG_OK;
}

int pcg_step_front(const pcg_graph_desc *g, const float *W, const float *b, float *s0, uint64_t *pos_keys,
                   const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds, const double *rho,
                   int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status,
                   void *stream) {
    if (!g) return PCG_E_ARG;
    const int rc = pcg_step_front_a(g, W, b, 0, g->n_nodes, s0, pos_keys, nodes, labels, B, thresholds, rho, train_flag, add_self,
                                    workspace, list_capacity, status, stream);
    if (rc != PCG_OK) return rc;
    return pcg_step_front_b(g, s0, pos_keys, B > 0 ? 1 : 0, nodes, labels, B, thresholds, rho, train_flag, add_self, workspace,
                            list_capacity, status, stream);
}

int pcg_aggregate_lists(const float *X, int32_t feat_dim, int32_t feat_stride, int32_t n_rows, const int32_t *cnt,
                        const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, int32_t norm,
                        float *agg, int32_t agg_stride, void *stream) {
    if (!X || !cnt || !g || !workspace || !agg || n_rows < 0 || B < 0) return PCG_E_ARG;
    if (n_rows == 0) return PCG_OK;
    if (feat_stride % 4 != 0 || feat_stride < feat_dim || agg_stride < feat_dim) return PCG_E_ARG;
    if (feat_stride > 512) return PCG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) & 15u) != 0) return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w);
    pcg::AggArgs a;
    a.X = X;
    a.feat_dim = feat_dim;
    a.feat_stride = feat_stride;
    a.n_rows = n_rows;
    a.row_begin = w.row_begin;
    a.chunk_begin = w.chunk_begin;
    a.len = w.len;
    a.cnt = cnt;
    a.chunk_desc = w.chunk_desc;
    a.chunk_cap = (int32_t)w.chunk_cap;
    a.list = w.list;
    a.n_chunks = w.counters + pcg::C_NCHUNK;
    a.partial = w.partial;
    a.agg = agg;
    a.agg_stride = agg_stride;
    a.norm = norm;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return feat_stride <= 256 ? pcg::launch_aggregate<1>(a, st) : pcg::launch_aggregate<2>(a, st);
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
    return pcg_aggregate_lists(g->X, g->feat_dim, g->feat_stride, g->n_rel * B, cnt, g, B, workspace, list_capacity,
                               norm, agg, agg_stride, stream);
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
                                             add_self, cnt, workspace, list_capacity, status, stream);
    if (rc != PCG_OK) return rc;
    return pcg_aggregate_lists(g->X, g->feat_dim, g->feat_stride, g->n_rel * B, cnt, g, B, workspace, list_capacity,
                               norm, agg, agg_stride, stream);
}

}  // extern "C"
