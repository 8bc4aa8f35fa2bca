// Segmented mean over the selection lists (src/layers.py:594-624; src/graphsage.py:82-95, 216-231) for gfx950.
//
//   gather   chip-wide balanced: one wave per 128-entry chunk of a row's list, feature rows gathered 64/lpr per
//            wave-instruction (128-B rows: 8 rows = 1 KiB), 8 in flight, f32 segmented sum; single-chunk rows are
//            finished here
//   combine  rows longer than one chunk: partial sums added in chunk order (bitwise reproducible), divided by |set|
//            (or its sqrt)
#include "choose.h"
#include "halo_map.h"

namespace pcg {

constexpr int UNROLL = 8;        // row-gather instructions in flight per wave
constexpr int GATHER_BLOCKS = 2048;

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
    int64_t table_rows;         // rows of X: a list entry outside [0, table_rows) is not gathered (a hole) and reported
    uint32_t *status;           // device status word (PCG_ST_LIST_ID_RANGE), or null
    HaloMap hm;                 // hm.keys != null (a partitioned rank's extended table): the list holds node ids - translated here
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

// workgroup `block` of `n_blocks` (256 threads each)
template <int NACC, bool MAP = false>
__device__ __forceinline__ void gather_chunks_body(const AggArgs &a, uint32_t block, uint32_t n_blocks) {
    const int lane = lane_id();
    const RowGeom q = row_geom(a.feat_stride, lane);
    const uint32_t nwaves = n_blocks * (blockDim.x >> 6);
    const uint32_t ch0 = block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    // the chunk's descriptor (written by the plan) is requested together with the chunk count: one load level instead of
    // four (count -> row -> offsets -> list).  n = the entries of the chunk the select kernel filled (it overwrites the
    // capacity share the plan put there; nothing fills a region's unused tail).
    int4 desc = a.chunk_desc[ch0 < (uint32_t)a.chunk_cap ? ch0 : 0u];
    uint32_t total = *a.n_chunks;
    total = total < (uint32_t)a.chunk_cap ? total : (uint32_t)a.chunk_cap;      // (a device-side counter never indexes unchecked)
    for (uint32_t ch = ch0; ch < total; ch += nwaves) {
        if (ch != ch0) desc = a.chunk_desc[ch];
        const int row = desc.x, n = desc.z, nch_row = desc.w;
        const int32_t *__restrict__ list = a.list + desc.y;
        const int cnt = a.cnt[row];                                    // (only single-chunk rows use it; requested up front)
        float4 acc[NACC];
#pragma unroll
        for (int x = 0; x < NACC; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
        // Every load below is UNCONDITIONAL (index clamped, value discarded by a select afterwards): a load inside a
        // conditional makes the compiler branch around it and wait for everything outstanding first, which serialises the
        // row gathers - one in flight per wave instead of UNROLL.  The ids of one iteration (rpw * UNROLL <= 64) are one
        // coalesced load, dealt out to the row slots by cross-lane reads; the next iteration's ids are requested before
        // this iteration's rows.
        const int per_iter = q.rpw * UNROLL;
        bool bad = false;                                              // an entry that names no row of the table
        // (negative = nothing there, or a hole left by a duplicate.  The sign bit is OR-ed in rather than the value replaced:
        //  a value that is only used under a condition gets its load sunk into a branch again)
        bool miss = false;                                             // MAP: an id that is nowhere in this rank's table
        int my = list[lane < n ? lane : (n > 0 ? n - 1 : 0)];
        my |= (lane >= n || lane >= per_iter) ? (int)0x80000000 : 0;
        if constexpr (MAP) my = halo_translate(a.hm, my, miss);        // node id -> row of the extended table (halo_map.h)
        for (int base = 0; base < n; base += per_iter) {
            const int ln = base + per_iter + lane;
            int nxt = list[ln < n ? ln : n - 1];
            nxt |= (ln >= n || lane >= per_iter) ? (int)0x80000000 : 0;
            float4 v[UNROLL][NACC];
            int ids[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                ids[u] = __shfl(my, u * q.rpw + q.slot);
                // (an id beyond the table is turned into a hole before it can form an address; the check is two VALU ops
                //  in front of a load that is unconditional either way)
                const bool beyond = ids[u] >= 0 && (int64_t)ids[u] >= a.table_rows;
                bad |= beyond;
                ids[u] |= beyond ? (int)0x80000000 : 0;
                const float *rowp = a.X + (size_t)(ids[u] >= 0 ? ids[u] : 0) * a.feat_stride;
#pragma unroll
                for (int x = 0; x < NACC; ++x) {
                    const int c4 = x * q.lpr + q.sub;
                    v[u][x] = *reinterpret_cast<const float4 *>(rowp + 4 * (c4 < q.nch ? c4 : q.nch - 1));
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u)
#pragma unroll
                for (int x = 0; x < NACC; ++x) {
                    const bool ok = ids[u] >= 0 && x * q.lpr + q.sub < q.nch;
                    acc[x].x += ok ? v[u][x].x : 0.f;
                    acc[x].y += ok ? v[u][x].y : 0.f;
                    acc[x].z += ok ? v[u][x].z : 0.f;
                    acc[x].w += ok ? v[u][x].w : 0.f;
                }
            // (the next iteration's ids are translated behind this iteration's rows: owned ids - most - cost a compare, fetched
            //  ones one or two probes of the window's hash table)
            if constexpr (MAP) nxt = halo_translate(a.hm, nxt, miss);
            my = nxt;
        }
        if constexpr (MAP) {
            if (miss && a.hm.overflow) atomicOr(a.hm.overflow, 4u);
        }
#pragma unroll
        for (int x = 0; x < NACC; ++x)
            for (int o = q.lpr; o < PCG_WAVE; o <<= 1) {
                acc[x].x += __shfl_xor(acc[x].x, o);
                acc[x].y += __shfl_xor(acc[x].y, o);
                acc[x].z += __shfl_xor(acc[x].z, o);
                acc[x].w += __shfl_xor(acc[x].w, o);
            }
        if (bad && a.status) atomicOr(a.status, (uint32_t)PCG_ST_LIST_ID_RANGE);
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

template <int NACC>
__global__ void __launch_bounds__(256) gather_chunks(const AggArgs a) {
    gather_chunks_body<NACC>(a, blockIdx.x, gridDim.x);
}
// The partitioned path's gather: the lists hold node ids, translated to rows of the rank's extended table as they are read (no
// look-up launch, no rewritten list); the launch's first workgroup also refreshes the label classifier's snapshot (its parameters
// and Adam state as the launches of this step see them: what the NEXT step's score pass recomputes the classifier's update from,
// pcg_step_scores_dist)
struct SnapCopy {
    const float *src[3];        // theta, m, v at the classifier's offset
    float *dst;                 // [3][n] or null
    int32_t n;
};
template <int NACC>
__global__ void __launch_bounds__(256) gather_chunks_dist(const AggArgs a, const SnapCopy sc) {
    if (sc.dst && blockIdx.x == 0)
        for (int i = (int)threadIdx.x; i < 3 * sc.n; i += 256) sc.dst[i] = sc.src[i / sc.n][i % sc.n];
    gather_chunks_body<NACC, true>(a, blockIdx.x, gridDim.x);
}

// The gather launch of a TRAINING step, with what else fits beside it (SideJob, choose.h): workgroups
//   [gather chunks | the previous step's deferred Adam update (all parameters but the label classifier's) |
//    the next step's train-pos keys | the next step's score pass]
// The dense kernel that follows needs the first two; the select kernel of the NEXT step needs the last two, which read the
// classifier the select launch of THIS step has just stepped (ClfStep) - nothing here waits for anything inside the launch.
#ifndef PCG_GT_WPE          // (A/B switch: 1 = cap the registers at the gather's own 80 - 6 waves per SIMD - at the price of a few spills)
#define PCG_GT_WPE 0
#endif
#if PCG_GT_WPE == 2      // five waves per SIMD for the 128-B-row instantiation (94 registers, no spill; the weight-gradient workgroups' code took it to 100 + 4)
#define PCG_GT_ATTR __attribute__((amdgpu_waves_per_eu(NACC == 1 ? 5 : 4, 8)))
#elif PCG_GT_WPE
#define PCG_GT_ATTR __attribute__((amdgpu_waves_per_eu(6, 8)))
#else
#define PCG_GT_ATTR
#endif
template <int NACC>
__global__ void __launch_bounds__(256) PCG_GT_ATTR gather_train_kernel(const AggArgs a, const SideJob s, int n_gather_blocks, int64_t n_nodes,
                                                           const int32_t *__restrict__ train_pos, int n_pos) {
    __shared__ float part[4][256];                            // adam_reduce_body uses [4][64] of it, wgrad_adam_body all
    __shared__ unsigned short sel[4 * MARK_GROUP];            // score_marked_body
    int b = (int)blockIdx.x;
    if (s.zero_word && b == 0 && threadIdx.x == 0) s.zero_word[0] = 0u;     // the select kernel's arrival counter, for its next launch
    // the previous step's weight gradients (GEMMs over its batch) + Adam (wgrad.h) come FIRST in the launch's workgroup order:
    // two memory round trips + a few dozen matrix instructions per wave is the longest chain in the launch - dispatched behind
    // the gather's workgroups it started when they ended
    if (b < s.n_wgrad_blocks) {
        if (s.wg_prio) __builtin_amdgcn_s_setprio(3);             // (A/B knob PCG_WGRAD_PRIO)
        wgrad_adam_body(s.wg, b, part);
        return;
    }
    b -= s.n_wgrad_blocks;
    if (b < n_gather_blocks) {
        gather_chunks_body<NACC>(a, (uint32_t)b, (uint32_t)n_gather_blocks);
        return;
    }
    b -= n_gather_blocks;
    if (b < s.n_adam_blocks) {
        if (s.ad.pending[0] == 1u)                            // (one word, the same for every thread)
            adam_reduce_body(s.ad.theta, s.ad.m, s.ad.v, s.ad.slabs, (int)s.ad.pending[1], s.ad.n_params, 0, s.ad.p_end,
                             s.ad.step_counter, s.ad.h, nullptr, 1, b, reinterpret_cast<float (*)[PCG_WAVE]>(&part[0][0]));
        return;
    }
    b -= s.n_adam_blocks;
    if (b < s.n_key_blocks) {
        pos_key_body(a.X, a.feat_dim, a.feat_stride, s.W, s.bias, train_pos, n_pos, s.raw_keys, b, s.n_key_blocks);
        return;
    }
    b -= s.n_key_blocks;
    if (s.touched) score_marked_body<NACC>(a.X, a.feat_dim, a.feat_stride, s.W, s.bias, n_nodes, s.s0, s.touched, b, s.n_score_blocks, sel);
    else score_table_body<NACC>(a.X, a.feat_dim, a.feat_stride, s.W, s.bias, 0, n_nodes, s.s0, b, s.n_score_blocks);
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

template <int NACC>
static int launch_aggregate(const AggArgs &g, hipStream_t st, bool combine) {
    hipLaunchKernelGGL(gather_chunks<NACC>, dim3(GATHER_BLOCKS), dim3(256), 0, st, g);
    PCG_LAUNCH_CHECK();
    if (combine) {
        hipLaunchKernelGGL(combine_rows<NACC>, dim3((g.n_rows + 3) / 4), dim3(256), 0, st, g);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

static void fill_agg_args(AggArgs &a, const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                          const int32_t *cnt, const Workspace &w, int32_t norm, float *agg, int32_t agg_stride, uint32_t *status) {
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
    a.n_chunks = w.counters + C_NCHUNK;
    a.partial = w.partial;
    a.agg = agg;
    a.agg_stride = agg_stride;
    a.norm = norm;
    a.table_rows = table_rows;
    a.status = status;
    a.hm = HaloMap{};
}

int launch_gather_train(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, const int32_t *cnt,
                        const pcg_graph_desc *g, int32_t B, const Workspace &w, float *agg, int32_t agg_stride, uint32_t *status,
                        const SideJob &side, hipStream_t st) {
    if (!X || !cnt || !g || !agg || B < 1 || table_rows < 1) return PCG_E_ARG;
    if (feat_stride % 4 != 0 || feat_stride < feat_dim || agg_stride < feat_dim) return PCG_E_ARG;
    if (feat_stride > 512) return PCG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) & 15u) != 0) return PCG_E_ARG;
    AggArgs a;
    fill_agg_args(a, X, feat_dim, feat_stride, table_rows, g->n_rel * B, cnt, w, PCG_NORM_COUNT, agg, agg_stride, status);
    const int extra = side.n_adam_blocks + side.n_wgrad_blocks + side.n_key_blocks + side.n_score_blocks;
    // workgroups of the gather group: half the stand-alone launch's - the riders' workgroups are dispatched behind them, and at
    // dataset scale most of the 2048 found no chunk (measured, 2048 / 1024 / 512: YelpChi-like 53.3 / 51.9 / 52.8 us per step,
    // power-law 2 M 149.2 / 146.4 / - )
#ifndef PCG_GT_BLOCKS
#define PCG_GT_BLOCKS (GATHER_BLOCKS / 2)
#endif
    const int gb = PCG_GT_BLOCKS;
    if (feat_stride <= 256)
        hipLaunchKernelGGL(gather_train_kernel<1>, dim3(gb + extra), dim3(256), 0, st, a, side, gb, g->n_nodes, g->train_pos, g->n_pos);
    else
        hipLaunchKernelGGL(gather_train_kernel<2>, dim3(gb + extra), dim3(256), 0, st, a, side, gb, g->n_nodes, g->train_pos, g->n_pos);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // namespace pcg

extern "C" {

static int aggregate(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows, const int32_t *cnt,
                     const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, int32_t norm,
                     float *agg, int32_t agg_stride, bool combine, uint32_t *status, void *stream, const void *plan = nullptr) {
    if (!X || !cnt || !g || !workspace || !agg || n_rows < 0 || B < 0 || table_rows < 1) return PCG_E_ARG;
    if (n_rows != g->n_rel * B) return PCG_E_ARG;       // the lists are those of the plan in `workspace`: n_rel * B rows
    if (n_rows == 0) return PCG_OK;
    if (feat_stride % 4 != 0 || feat_stride < feat_dim || agg_stride < feat_dim) return PCG_E_ARG;
    if (feat_stride > 512) return PCG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) & 15u) != 0) return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve1(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w, static_cast<unsigned char *>(const_cast<void *>(plan)));
    pcg::AggArgs a;
    pcg::fill_agg_args(a, X, feat_dim, feat_stride, table_rows, n_rows, cnt, w, norm, agg, agg_stride, status);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return feat_stride <= 256 ? pcg::launch_aggregate<1>(a, st, combine) : pcg::launch_aggregate<2>(a, st, combine);
}

int pcg_aggregate_lists(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                        const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity,
                        int32_t norm, float *agg, int32_t agg_stride, uint32_t *status, void *stream) {
    return aggregate(X, feat_dim, feat_stride, table_rows, n_rows, cnt, g, B, workspace, list_capacity, norm, agg, agg_stride, true,
                     status, stream);
}

/* gather only: rows of one chunk are finished (mean in agg), rows of several are left as per-chunk partial sums in the
 * workspace for pcg_train_dense to add up while it stages its tile (no combine launch) */
int pcg_gather_lists(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                     const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, float *agg,
                     int32_t agg_stride, uint32_t *status, void *stream) {
    return aggregate(X, feat_dim, feat_stride, table_rows, n_rows, cnt, g, B, workspace, list_capacity, PCG_NORM_COUNT, agg,
                     agg_stride, false, status, stream);
}

/* the same with the batch's plan part outside the workspace (pcg_plan_batches; plan == NULL: inside, as pcg_gather_lists) */
int pcg_gather_lists_planned(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                             const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                             int64_t list_capacity, float *agg, int32_t agg_stride, uint32_t *status, void *stream) {
    return aggregate(X, feat_dim, feat_stride, table_rows, n_rows, cnt, g, B, workspace, list_capacity, PCG_NORM_COUNT, agg,
                     agg_stride, false, status, stream, plan);
}

/* pcg_gather_lists_planned for a partitioned rank (pc-gnn_amd/dist.py): the lists hold NODE ids; the gather translates them to rows
 * of the extended table [ owned | train-pos | halo ] as it reads them (the window's hash table of pcg_halo_collect; pos_ids /
 * pos_idx as pcg_halo_lookup) - the list itself is left as it is; an id that is in none of the three is skipped and sets
 * overflow bit 4 in counts[128].  snap_dst != NULL: the launch also copies the label classifier's parameters and Adam state
 * (theta / m / v at clf_offset, clf_n floats each) to snap_dst [3 * clf_n] (pcg_step_scores_dist reads them in the next step). */
int pcg_gather_lists_dist(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                          const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                          int64_t list_capacity, float *agg, int32_t agg_stride, uint32_t *status, int32_t lo, int32_t hi,
                          int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos, const uint32_t *table,
                          int64_t table_slots, uint32_t *counts, int32_t halo_cap, int32_t halo_base, const float *theta,
                          const float *m, const float *v, int64_t clf_offset, int32_t clf_n, float *snap_dst, void *stream) {
    if (!X || !cnt || !g || !workspace || !agg || n_rows < 0 || B < 0 || table_rows < 1) return PCG_E_ARG;
    if (n_rows != g->n_rel * B) return PCG_E_ARG;
    if (n_rows == 0) return PCG_OK;
    if (feat_stride % 4 != 0 || feat_stride < feat_dim || agg_stride < feat_dim) return PCG_E_ARG;
    if (feat_stride > 512) return PCG_E_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(X) & 15u) != 0) return PCG_E_ARG;
    if (!table || !counts || table_slots < 1024 || (table_slots & (table_slots - 1)) != 0 || lo > hi || n_pos < 0 ||
        (n_pos > 0 && (!pos_ids || !pos_idx)))
        return PCG_E_ARG;
    if (snap_dst && (!theta || !m || !v || clf_n < 1 || clf_offset < 0)) return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve1(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w, static_cast<unsigned char *>(const_cast<void *>(plan)));
    pcg::AggArgs a;
    pcg::fill_agg_args(a, X, feat_dim, feat_stride, table_rows, n_rows, cnt, w, PCG_NORM_COUNT, agg, agg_stride, status);
    a.hm.keys = table;
    a.hm.vals = table + table_slots;
    a.hm.mask = (uint32_t)(table_slots - 1);
    a.hm.lo = lo; a.hm.hi = hi; a.hm.n_local = n_local;
    a.hm.pos_ids = pos_ids; a.hm.pos_idx = pos_idx; a.hm.n_pos = n_pos;
    a.hm.halo_cap = halo_cap; a.hm.halo_base = halo_base;
    a.hm.overflow = counts + 128;
    pcg::SnapCopy sc;
    sc.src[0] = theta ? theta + clf_offset : nullptr;
    sc.src[1] = m ? m + clf_offset : nullptr;
    sc.src[2] = v ? v + clf_offset : nullptr;
    sc.dst = snap_dst;
    sc.n = clf_n;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (feat_stride <= 256) hipLaunchKernelGGL(pcg::gather_chunks_dist<1>, dim3(pcg::GATHER_BLOCKS), dim3(256), 0, st, a, sc);
    else hipLaunchKernelGGL(pcg::gather_chunks_dist<2>, dim3(pcg::GATHER_BLOCKS), dim3(256), 0, st, a, sc);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

/* gather + combine (finished means) with the plan part outside the workspace */
int pcg_aggregate_lists_planned(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                                const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                                int64_t list_capacity, int32_t norm, float *agg, int32_t agg_stride, uint32_t *status, void *stream) {
    return aggregate(X, feat_dim, feat_stride, table_rows, n_rows, cnt, g, B, workspace, list_capacity, norm, agg, agg_stride, true,
                     status, stream, plan);
}

}  // extern "C"
