// Shared between the plan / front kernels (choose.hip), the select kernel (select.hip) and the gather kernel
// (gather.hip): workspace layout, per-row records, launch arguments.
#pragma once
#include "common.h"
#include "wgrad.h"

namespace pcg {

// ---- degree tiers of the select kernel --------------------------------------------------------------------------------
constexpr int TA_CAP = 16;       // rows of <= 16 neighbours WITHOUT minority picks / self union: four per wave, 16 lanes each, DPP ranking
constexpr int TB_CAP = 64;       // rows of <= 64: one wave, one key per lane, ranked lane against lane
constexpr int T1_CAP = 512;      // rows of <= 512: one wave, up to 8 keys per lane in registers, 256-bin LDS histogram rounds
constexpr int T4_CAP = 4096;     // workgroup rows are queued in two classes (> 4096 first) so that the longest start first
constexpr int WG_KEYCAP = 10240; // workgroup rows up to this length keep their distance keys in LDS; longer ones recompute them
constexpr int BIG_KEYCAP = 16384; // ... on hub-heavy graphs (maximum degree beyond WG_KEYCAP) select_rows runs two workgroups per CU with this
                                  // many keys of LDS each, so that rows up to it need no launch of their own
constexpr int LONG_NW = 16;       // waves per workgroup of select_long_rows (rows beyond WG_KEYCAP: a launch of their own, one workgroup per CU)
constexpr int LONG_KEYCAP = 32768; // ... which keep the keys of rows up to this length in LDS (128 KB of the CU's 160); longer still: global scratch
constexpr int SEL_NW = 8;        // waves per select workgroup
constexpr int SEL_BLOCKS = 768;  // persistent select workgroups: 3 per CU (<= 80 VGPRs: 6 waves per SIMD; 49.5 KB of LDS each)
constexpr int HIST_WG = 2048;    // histogram bins of a workgroup row
constexpr int HIST_W = 256;      // histogram bins of a single-wave row
constexpr int WAVE_AREA = WG_KEYCAP / SEL_NW;   // LDS words a wave owns while it works on single-wave rows
constexpr int CHUNK = 128;       // list entries per gather work item
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_PER = 4;      // rows per plan thread per tile
constexpr int FRONT_COUNT_THREADS = 256;

// counters (uint32) at the head of the workspace
enum { C_N1 = 0, C_N4 = 1, C_N16 = 2, C_NCHUNK = 5, C_N0 = 8, C_NA = 9 };

struct RowRec {            // 32 bytes, written by the plan, read by select: one 32-B load instead of a 3-deep chain
    int64_t start;         // offset of the row in indices[r]
    int32_t node, d, k, m; // centre, degree, ceil(d * threshold), minority picks (0 unless a positive centre in training)
    int32_t lbeg;          // first entry of the row's region in the selection list (list_capacity < 2^31)
    int32_t chunk0;        // first gather chunk of the row
};
__host__ __device__ __forceinline__ bool rec_keep_all(const RowRec &p) { return !(p.d > p.k + 1); }   // layers.py:662
__host__ __device__ __forceinline__ int rec_cap(const RowRec &p, int add_self) {
    return (rec_keep_all(p) ? p.d : p.k) + p.m + (add_self ? 1 : 0);
}

struct Workspace {
    uint32_t *counters;    // [64]
    uint32_t *heads;       // [8 * 16] work-queue heads of the select kernel, one per shard, 64 bytes apart; the plan zeroes them
    int64_t *row_begin;    // [rows + 1] start of every row's region in list
    int32_t *chunk_begin;  // [rows + 1]
    int32_t *len;          // [rows]     entries (holes included) actually written
    int32_t *q0, *q1, *q4, *q16, *qa;  // [rows] each: the rows of every degree tier
    struct RowRec *recs;   // [rows] what the plan worked out per row
    unsigned char *plan_totals;   // [blocks of the two-pass plan] PlanTotals
    int4 *chunk_desc;      // [chunk_cap] per gather chunk: {row, first list entry, entries in use (select), chunks of the row}
    float *partial;        // [chunk_cap, feat_stride]
    uint32_t *row_ticket;  // [rows] arrivals of a multi-chunk row's chunks (gather; reset by the row's last chunk)
    int32_t *list;         // [list_capacity]  chosen ids; -1 = hole
    uint32_t *key_scratch; // [scratch_cap] distance keys of rows too long for the select kernel's LDS: max_degree entries for
                           // each workgroup of select_long_rows
    int64_t list_capacity, chunk_cap, scratch_cap;
};

static inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

// The workspace is two parts.  PLAN part: what the plan writes for one batch (counters, row records, offsets, tier queues, the
// gather's chunk table) - it depends only on the batch's ids / labels and the CSR degrees, so an epoch's batches are planned
// together, one plan part ("slot") each.  DATA part: what a step writes (selection list, per-chunk partial sums, key scratch) -
// one per engine, shared by every slot.  A legacy single-buffer workspace is [plan | data] (carve1).
struct CarveSizes {
    int64_t plan_bytes, data_bytes;
};
static inline CarveSizes carve(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, unsigned char *plan_base,
                               unsigned char *data_base, Workspace *w) {
    const int64_t rows = (int64_t)g->n_rel * B;
    const int64_t chunk_cap = list_capacity / CHUNK + rows + 1;
    // key scratch: only graphs with rows beyond select_long_rows' LDS key capacity need it; max_degree entries for each of its
    // workgroups (up to 256 of them; fewer if that would be more than 1 GiB)
    int64_t scratch_cap = 0;
    if (g->max_degree > LONG_KEYCAP) {
        int64_t nb = (1ll << 28) / g->max_degree;
        nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
        scratch_cap = nb * (int64_t)g->max_degree;
    }
    int64_t off = 0;
    unsigned char *base = plan_base;
    auto take = [&](int64_t bytes) {
        const int64_t o = off;
        off += align256(bytes);
        return base ? base + o : nullptr;
    };
    unsigned char *p;
    p = take(256);                                 if (w) w->counters = reinterpret_cast<uint32_t *>(p);
    p = take(4 * 8 * 16);                          if (w) w->heads = reinterpret_cast<uint32_t *>(p);
    p = take(8 * (rows + 1));                      if (w) w->row_begin = reinterpret_cast<int64_t *>(p);
    p = take(4 * (rows + 1));                      if (w) w->chunk_begin = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->len = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q0 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q1 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q4 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->q16 = reinterpret_cast<int32_t *>(p);
    p = take(4 * rows);                            if (w) w->qa = reinterpret_cast<int32_t *>(p);
    p = take(32 * rows);                           if (w) w->recs = reinterpret_cast<RowRec *>(p);
    p = take(64 * (rows / 256 + 2));               if (w) w->plan_totals = p;   // PlanTotals (<= 64 B) per 256 rows
    p = take(16 * chunk_cap);                      if (w) w->chunk_desc = reinterpret_cast<int4 *>(p);
    CarveSizes sz;
    sz.plan_bytes = off;
    off = 0;
    base = data_base;
    p = take(4 * chunk_cap * g->feat_stride);      if (w) w->partial = reinterpret_cast<float *>(p);
    p = take(4 * rows);                            if (w) w->row_ticket = reinterpret_cast<uint32_t *>(p);
    p = take(4 * list_capacity);                   if (w) w->list = reinterpret_cast<int32_t *>(p);
    p = take(4 * scratch_cap);                     if (w) w->key_scratch = reinterpret_cast<uint32_t *>(p);
    sz.data_bytes = off;
    if (w) {
        w->list_capacity = list_capacity;
        w->chunk_cap = chunk_cap;
        w->scratch_cap = scratch_cap;
    }
    return sz;
}
// plan == null: the single-buffer layout [plan | data] at `workspace`; else the plan part at `plan`, the data part at `workspace`
static inline int64_t carve1(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, unsigned char *workspace, Workspace *w,
                             unsigned char *plan = nullptr) {
    const CarveSizes sz = carve(g, B, list_capacity, nullptr, nullptr, nullptr);
    if (plan) carve(g, B, list_capacity, plan, workspace, w);
    else carve(g, B, list_capacity, workspace, workspace ? workspace + sz.plan_bytes : nullptr, w);
    return sz.plan_bytes + sz.data_bytes;
}
// move the plan part of a carved workspace by `bytes` (slot s of an epoch's plans lies s * stride behind slot 0)
template <typename T>
__host__ __device__ __forceinline__ void shift_ptr(T *&ptr, int64_t bytes) {
    ptr = reinterpret_cast<T *>(reinterpret_cast<unsigned char *>(ptr) + bytes);
}
__host__ __device__ __forceinline__ void shift_plan(Workspace &w, int64_t bytes) {
    shift_ptr(w.counters, bytes); shift_ptr(w.heads, bytes); shift_ptr(w.row_begin, bytes); shift_ptr(w.chunk_begin, bytes);
    shift_ptr(w.len, bytes); shift_ptr(w.q0, bytes); shift_ptr(w.q1, bytes); shift_ptr(w.q4, bytes); shift_ptr(w.q16, bytes);
    shift_ptr(w.qa, bytes); shift_ptr(w.recs, bytes); shift_ptr(w.plan_totals, bytes); shift_ptr(w.chunk_desc, bytes);
}

// The label classifier's own training step (src/layers.py:230-243 forward, src/model.py:54-61 its loss term, model_handler.py:153
// its Adam update), riding in the select launch.  Its 2F + 2 parameters get gradient from nothing but the batch centres' feature
// rows - not from the selection, the aggregates or the GNN weights - so step t's update can be worked out at the START of step t,
// and the scores that step t + 1 selects by can be formed beside step t's gather instead of behind its dense kernel.
//   clf_next [2F + 2]: in: the classifier step t scores / selects by; out: the classifier after step t's update.  theta_clf gets
//   the in-value (the dense kernel of step t reads it for the loss); m, v: the classifier's Adam state; t = step_counter[0] + 1.
struct ClfStep {
    float *clf_next;           // null: off
    float *theta_clf, *m, *v;
    const int32_t *step_counter;
    float scale;               // lambda_1 / global batch size: what a row's d loss / d logit is multiplied by
    AdamHyper h;
    // a batch of more than 1024 rows is shared by n_wg workgroups (a slice of rows each): every one leaves its gradient in
    // part[w * part_stride ..] (write-through), the one whose ticket is the last adds them up in workgroup order and applies Adam
    int32_t n_wg;
    float *part;
    int64_t part_stride;
    uint32_t *ticket;          // device word, zero between launches
};

struct ChooseArgs {
    pcg_graph_desc g;
    const int32_t *nodes;
    const int32_t *labels;
    int32_t B;
    const float *s0;
    const float *center_s0;
    int64_t center_off;        // center_s0 == null: centre b's score is s0[nodes[b] + center_off] (the partitioned path: `nodes` are
                               // table rows of owned nodes, the scores are indexed by global node id)
    const uint64_t *pos_keys;
    // in-kernel sort of the train positives (select_rows; null / 0 = pos_keys is sorted already): the keys come in groups of 64
    // (n_sort groups); group g's ranks are worked out by sort_slices workgroups, each against its share of all the keys, added
    // up in rank_acc; the group's last workgroup (group_ticket) stores the group's keys at their ranks in sort_out (= pos_keys,
    // sort_cap entries) and counts the group in sort_done (zero at launch); a row with minority picks waits for
    // sort_done == n_sort.  rank_acc [n_sort * 64] and group_ticket [n_sort]: zero at launch, left zero.
    const uint64_t *raw_keys;
    uint64_t *sort_out;
    uint32_t *sort_done, *rank_acc, *group_ticket;
    int32_t n_sort, sort_cap, sort_slices, sort_slice_len;
    int32_t key_cap;           // LDS words of a select_rows workgroup's key area (WG_KEYCAP or BIG_KEYCAP: launch_select_rows sets it)
    uint32_t *pending_clear;   // device word the select kernel zeroes ("the deferred Adam update has been applied"), or null
    ClfStep clf;               // the label classifier's step for this batch (one workgroup), or clf.clf_next == null
    // the previous step's weight gradients + Adam (wgrad.h) by the first n_wg_units row workgroups of the select launch, two 16 x 16
    // tiles each (four waves a tile), before they start on rows - they take the LAST of the first units (short rows), like the
    // sorting workgroups: small batches only (one K part); n_wg_units == 0: the gather launch's riders do it
    WgradArgs wg;
    int32_t n_wg_units;
    double thr[PCG_MAX_REL];
    double rho[PCG_MAX_REL];
    int32_t train_flag, add_self;
    int32_t *cnt;          // [rows] |chosen set|
    uint32_t *status;
    unsigned long long *stamps;   // diagnostic only (pcg_debug_set_stamps): [rows][8] wall-clock ticks per phase, else null
    Workspace w;
};

// DPP-selected lane value (full-rate VALU, no LDS crossbar round trip as __shfl would make)
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v, uint32_t old = 0u) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}

// wave-wide inclusive scan: row_shr 1, 2, 4, 8 inside every 16-lane row (a lane without a source adds 0), then the row
// totals through scalar registers
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    v += (int)dpp_u32<0x111>((uint32_t)v);
    v += (int)dpp_u32<0x112>((uint32_t)v);
    v += (int)dpp_u32<0x114>((uint32_t)v);
    v += (int)dpp_u32<0x118>((uint32_t)v);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}

// select.hip
int launch_select_rows(const ChooseArgs &a, hipStream_t st);
// sort.hip: the one-launch bucket sort over raw keys (RANK_MAX < n_pos <= 131072)
int launch_bk_onepass(const uint64_t *raw, int n_pos, uint64_t *keys, int cap, uint32_t *status, hipStream_t st);

// What rides along the gather launch of a training step (gather.hip: gather_train_kernel): the deferred Adam update of every
// parameter but the label classifier's (from the previous step's slabs; the dense kernel that follows reads the result), and the
// NEXT step's score pass + unsorted train-pos keys with the classifier the select launch has just stepped.
struct SideJob {
    DeferredAdam ad;
    int32_t n_adam_blocks;     // 0: no update from slabs
    WgradArgs wg;              // the deferred update from the dense kernel's transposed activations instead (wgrad.h) ...
    int32_t n_wgrad_blocks;    // ... by this many workgroups (0: off)
    int32_t wg_prio;           // A/B knob: the weight-gradient workgroups raise their wave priority
    const float *W, *bias;     // the classifier to score with (ClfStep::clf_next), or null: no score pass
    float *s0;
    const unsigned char *touched;   // byte map of the rows the next batch reads, or null: the whole table
    uint64_t *raw_keys;        // the train positives' unsorted keys, or null
    int32_t n_key_blocks, n_score_blocks;
    uint32_t *zero_word;       // the select kernel's arrival counter, zeroed for its next launch
};
int launch_gather_train(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, const int32_t *cnt,
                        const pcg_graph_desc *g, int32_t B, const Workspace &w, float *agg, int32_t agg_stride, uint32_t *status,
                        const SideJob &side, hipStream_t st);

}  // namespace pcg
