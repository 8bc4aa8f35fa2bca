/*
 * pcgnn.h - C ABI of libpcgnn_hip.so: the MI355X (gfx950) PC-GNN hot path.
 *
 * The reference (h22hyeon/PC-GNN) is 100 % Python and has no FFI of its own; its
 * "plugin API" for this path is the forward() of PCALayer / InterAgg* / IntraAgg
 * and pick_step().  This header is what a ctypes binding inside those methods
 * calls instead of the Python set / torch.sort / dense-mask code.  Each entry
 * point names the reference lines it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the reference-side ctypes stub.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. a torch tensor
 *     that outlives the call) unless the parameter is documented "host";
 *   - work is enqueued on `stream` (a hipStream_t passed as void*); no entry point
 *     synchronises the stream, allocates, or frees device memory;
 *   - return value: 0 = enqueued, <0 = rejected before anything was enqueued
 *     (PCG_E_*).  Conditions only knowable on the device (a selection buffer
 *     too small) are reported through the caller-provided `status` word;
 *   - one host thread per device; re-entrant across devices.
 *
 * Data layout in HBM (built once per graph, resident for the whole run - mirrors
 * the frozen nn.Embedding + adj_lists the reference keeps, model_handler.py:85-87):
 *   X          float  [n_nodes, feat_stride]  row-major, feat_stride % 4 == 0,
 *                     columns feat_dim..feat_stride-1 are zero, 16-byte aligned
 *   indptr[r]  int64  [n_nodes + 1]           CSR row offsets of relation r
 *   indices[r] int32  [nnz_r]                 neighbour ids, ASCENDING inside a row,
 *                                             self-loops kept (utils.py:226-239)
 *   train_pos  int32  [n_pos]                 ids of training positives, no duplicates
 */
#ifndef PCGNN_H
#define PCGNN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCG_MAX_REL 8

enum {
    PCG_OK = 0,
    PCG_E_ARG = -1,        /* null / misaligned pointer, bad size                   */
    PCG_E_UNSUPPORTED = -2,/* shape outside what the kernels handle                 */
    PCG_E_LAUNCH = -3      /* hipLaunchKernel reported an error                     */
};

/* bits of the device-side status word */
enum {
    PCG_ST_SEL_OVERFLOW = 1,  /* sel_indices capacity too small; nothing was written past it */
    PCG_ST_LIST_ID_RANGE = 2, /* a selection-list entry named no row of the table handed to the gather; it was skipped (a hole) */
    PCG_ST_SYNC_TIMEOUT = 4,  /* a bounded in-kernel wait (the select kernel's wait for its own train-pos sort) ran out */
    PCG_ST_SORT_OVERFLOW = 8  /* the one-launch bucket sort of the train positives met a bucket of more than 4096 keys (eight times
                                 the mean): its surplus keys were dropped - minority picks of that step may be wrong */
};

enum { PCG_NORM_COUNT = 0, PCG_NORM_SQRT_COUNT = 1 };

typedef struct pcg_graph_desc {
    int64_t n_nodes;
    int32_t feat_dim;
    int32_t feat_stride;
    int32_t n_rel;
    int32_t n_pos;
    int32_t max_degree;                 /* max over relations and rows (host-computed) */
    int32_t _pad;
    const float *X;
    const int32_t *train_pos;
    const int64_t *indptr[PCG_MAX_REL];
    const int32_t *indices[PCG_MAX_REL];
} pcg_graph_desc;

/* library / build identification: "pcgnn_hip gfx950 <abi>" (host pointer, static) */
const char *pcg_version(void);
int pcg_abi_version(void);

/* ---- label-aware scores -------------------------------------------------------
 * Replaces  batch_scores = self.label_clf(self.features(unique_nodes))
 *           (src/layers.py:230-237) by scoring the whole table once per step.
 * s0[i] = b[0] + sum_f X[i,f] * W[0,f]   (class-0 logit; the only column the
 * choose step reads, layers.py:649-650).  W is label_clf.weight [2, feat_dim]
 * row-major, b is label_clf.bias [2].  Rows [row_begin, row_end). */
int pcg_score_table(const pcg_graph_desc *g, const float *W, const float *b,
                    int64_t row_begin, int64_t row_end, float *s0, void *stream);

/* center_scores = batch_scores[nodes]  (layers.py:243): both logits of the given
 * rows, bit-identical in column 0 to pcg_score_table.  out is [n_ids, 2]. */
int pcg_score_rows(const pcg_graph_desc *g, const float *W, const float *b,
                   const int32_t *ids, int32_t n_ids, float *out, void *stream);

/* Sort the training positives by class-0 logit once per step; replaces the
 * per-centre torch.sort over all of pos_scores (layers.py:683-688).
 * keys [pcg_pos_sort_capacity(n_pos)] uint64; on return its first ceil_pow2(max(n_pos, 4096)) entries hold
 * (orderable(s0[train_pos[p]]) << 32) | p ascending, padded with UINT64_MAX; the capacity is twice that: the second
 * half is scratch (the bucket sort's scatter buffer and splitters for n_pos > 16384; the unsorted keys pcg_step_front forms beside the score pass). */
int64_t pcg_pos_sort_capacity(int32_t n_pos);
int pcg_pos_sort(const pcg_graph_desc *g, const float *s0, uint64_t *keys, void *stream);

/* ---- choose + aggregate (the hot path) ----------------------------------------------
 * Replaces, for every relation r and batch centre b:
 *   neighbour lookup + score slicing            layers.py:217-219, 246-253
 *   num_sample = ceil(deg * threshold[r])        layers.py:260-262
 *   choose_step_neighs / choose_step_test        layers.py:633-738
 *   dense-mask mean  mask.div(n).mm(X[unique])   layers.py:594-624
 * Selection rule: if deg > k+1 keep the k neighbours with the smallest
 * |s0[centre] - s0[j]|, ties by ascending position in the (ascending-id) row;
 * else keep all.  If train_flag and labels[b]==1 additionally take the
 * m = min(int(k*rho[r]), n_pos) training positives nearest in the same metric, ties
 * by position in train_pos; the union is de-duplicated (set(), layers.py:694).
 *
 * pcg_choose_select writes every row's chosen ids into the workspace's selection list
 * (row = r * B + b; region [row_begin[row], row_begin[row] + len[row]); -1 marks a slot
 * whose candidate was a duplicate) and |set| into cnt; pcg_aggregate_lists gathers and
 * averages those lists from a feature table (the graph's X, or on a multi-GPU run a
 * table extended by the rows fetched from other ranks after the list was re-indexed);
 * pcg_choose_aggregate = both, for one GPU.
 *
 *   nodes   int32 [B]   batch centre ids (duplicates allowed)
 *   labels  int32 [B]   batch labels; may be NULL when train_flag == 0
 *   s0      float [n_nodes] from pcg_score_table;  pos_keys from pcg_pos_sort
 *   center_s0 float [B] or NULL: the centres' class-0 logits if they are not to be
 *           read from s0[nodes[b]] (IntraAgg.forward called with explicit
 *           batch_scores, layers.py:562)
 *   thresholds, rho  HOST double [n_rel]
 *   add_self 1: the centre joins its own set (GCN / SAGE-gcn, graphsage.py:78-79, 214)
 *   cnt     int32 [n_rel * B]             |chosen set|
 *   agg     float [n_rel * B, agg_stride] mean of the chosen rows (cols < feat_dim);
 *           norm = PCG_NORM_COUNT | PCG_NORM_SQRT_COUNT (graphsage.py:224-226)
 *   workspace: pcg_choose_workspace_bytes(g, B, list_capacity) bytes, 256-byte aligned;
 *           list_capacity (< 2^31 entries) bounds sum over rows of
 *           pcg_sel_capacity_row(...); if a batch needs more, nothing is selected and
 *           PCG_ST_SEL_OVERFLOW is OR-ed into *status (uint32 device word; zero it yourself).
 *   pcg_choose_workspace_offset(..., which): byte offset inside the workspace of
 *           0 row_begin int64 [rows+1] | 1 len int32 [rows] | 2 list int32 [list_capacity]
 *           (3 chunk_begin, 4 chunk descriptors, 5 counters, 6 partial: internal, exposed for tests).
 *
 * The workspace has two parts: the PLAN part (what the plan works out for one batch: row records, list / chunk offsets, tier
 * queues, the gather's chunk table - a function of the batch's ids, labels and CSR degrees only) and the DATA part (what a
 * step writes: selection list, per-chunk partial sums, key scratch).  pcg_choose_workspace_bytes = both, the plan part first
 * (the single-buffer layout every entry point without a `plan` argument uses).  pcg_choose_plan_bytes / _data_bytes: the
 * parts on their own, for callers that keep one plan part per batch of an epoch (pcg_plan_batches) and ONE data part. */
int64_t pcg_choose_workspace_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity);
int64_t pcg_choose_plan_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity);
int64_t pcg_choose_data_bytes(const pcg_graph_desc *g, int32_t B, int64_t list_capacity);
int64_t pcg_choose_workspace_offset(const pcg_graph_desc *g, int32_t B, int64_t list_capacity, int32_t which);
int pcg_choose_select(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                      const float *s0, const float *center_s0, const uint64_t *pos_keys,
                      const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self,
                      int32_t *cnt, void *workspace, int64_t list_capacity, uint32_t *status, void *stream);
/* pcg_aggregate_lists: the gather + mean half on its own, over the lists (and the plan) a pcg_choose_select* call left in
 * `workspace`, reading feature rows from X [table_rows, feat_stride] - g->X, or a table with further rows behind it (the
 * partitioned path's [owned | train-pos | halo] table, whose lists pcg_halo_lookup has re-indexed).  n_rows = the rows of the
 * plan = g->n_rel * B (anything else is rejected).  A list entry outside [0, table_rows) is skipped like a hole and
 * PCG_ST_LIST_ID_RANGE is OR-ed into *status (may be NULL: then only skipped). */
int pcg_aggregate_lists(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                        const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity,
                        int32_t norm, float *agg, int32_t agg_stride, uint32_t *status, void *stream);
int pcg_choose_aggregate(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const float *s0, const float *center_s0, const uint64_t *pos_keys,
                         const double *thresholds, const double *rho, int32_t train_flag,
                         int32_t norm, int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt,
                         void *workspace, int64_t list_capacity, uint32_t *status, void *stream);

/* The front of a training step in two launches instead of four.  Equivalent to
 *   pcg_score_table(g, W, b, 0, n_nodes, s0);  pcg_pos_sort(g, s0, pos_keys)  [if train_flag and n_pos > 0];
 *   + the plan (row records, list offsets, tier queues, chunk table) that pcg_choose_select / pcg_choose_aggregate
 *     would compute first for this (nodes, labels, B, thresholds, rho, train_flag, add_self, workspace, list_capacity)
 * with the plan's two passes riding along the score pass and the sort (it needs neither's result; both leave most
 * CUs idle at dataset scale).  Follow it with pcg_choose_aggregate_planned / pcg_choose_select_planned, which take the
 * arguments of their namesakes - the same values as given here - and skip the plan.  Results are bit-identical to the
 * separate calls.  Replaces src/layers.py:230-243 (scores) + :683-688 (sort) + the per-batch bookkeeping of :246-262. */
int pcg_step_front(const pcg_graph_desc *g, const float *W, const float *b, float *s0, uint64_t *pos_keys,
                   const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds,
                   const double *rho, int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity,
                   uint32_t *status, void *stream);
/* The two halves of pcg_step_front on their own, for callers that need something between them (the partitioned
 * path all-gathers the scores): _a = scores of rows [row_begin, row_end) into s0_out[row] (as pcg_score_table) || plan
 * pass 1;  _b = train-pos sort by s0 || plan pass 2.  Same plan arguments in both.
 * _a with row_ids != NULL (int32 [n_nodes] on the device; pos_keys must be NULL then): row r's score goes to s0_out[row_ids[r]],
 * rows with row_ids[r] < 0 are skipped - the partitioned path, whose table rows are owned / train-pos / halo rows while scores
 * are looked up by global node id (every rank scores the rows it holds; no score all-gather).
 * _b with center_s0_out != NULL (float [B]): also center_s0_out[i] = s0[nodes[i] + center_id_offset] - the centres' own
 * scores for pcg_choose_select_planned's center_s0, when `nodes` are table rows but s0 is indexed by global id (a rank of the
 * partitioned path owns the ids [offset, offset + n_local)).
 * _a with pos_keys != NULL (training, 0 < n_pos <= 16384, every train-pos row present in g->X) also forms the unsorted keys
 * from the feature rows in the scratch half of pos_keys (a third group of workgroups); pass raw_keys_ready = 1 to _b then,
 * and its sort stages them with coalesced loads instead of gathering n_pos scores in every sort workgroup. */
int pcg_step_front_a(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end,
                     float *s0_out, const int32_t *row_ids, uint64_t *pos_keys, const int32_t *nodes,
                     const int32_t *labels, int32_t B, const double *thresholds, const double *rho, int32_t train_flag,
                     int32_t add_self, void *workspace, int64_t list_capacity, uint32_t *status, void *stream);
int pcg_step_front_b(const pcg_graph_desc *g, const float *s0, uint64_t *pos_keys, int32_t raw_keys_ready,
                     const int32_t *nodes, const int32_t *labels, int32_t B, const double *thresholds,
                     const double *rho, int32_t train_flag, int32_t add_self, void *workspace, int64_t list_capacity,
                     uint32_t *status, float *center_s0_out, int64_t center_id_offset, void *stream);
/* `plan` (here and below): NULL = the plan part lies inside `workspace` (pcg_step_front / _a + _b made it); else `workspace`
 * is a DATA part (pcg_choose_data_bytes) and `plan` the batch's plan part (a slot pcg_plan_batches filled). */
int pcg_choose_select_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                              const float *s0, const float *center_s0, const uint64_t *pos_keys,
                              const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self,
                              int32_t *cnt, void *workspace, const void *plan, int64_t list_capacity, uint32_t *status,
                              uint32_t *sync_words, int64_t center_id_offset, void *stream);
/* (sync_words: as pcg_choose_gather_planned - the select kernel sorts the unsorted train-pos keys itself; center_id_offset > 0
 *  with center_s0 == NULL: centre b's score is s0[nodes[b] + center_id_offset] - the partitioned path, whose `nodes` are table rows
 *  of owned nodes while s0 is indexed by global node id) */
/* The plans of ALL batches of an epoch in ONE launch - off every step's critical path (a plan depends on the picked ids, their
 * labels and the CSR degrees, never on a parameter; the reference does this bookkeeping per batch, src/layers.py:217-219,
 * 246-262).  Batch s = nodes[s * B, min((s + 1) * B, n_total)) (the last one may be shorter) is planned into
 * plans + s * plan_stride (plan_stride >= pcg_choose_plan_bytes(g, B, list_capacity), a multiple of 256); thresholds .. add_self
 * and list_capacity as the *_planned calls that follow will be given.  An overflowing batch sets PCG_ST_SEL_OVERFLOW and
 * selects nothing.  bump_counter (may be NULL): a device uint64 incremented by the launch (the sampler's epoch number -
 * pcg_pick_shuffled with bump = 0 in front of it). */
int pcg_plan_batches(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t n_total, int32_t B,
                     const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *plans,
                     int64_t plan_stride, int64_t list_capacity, uint32_t *status, uint64_t *bump_counter, void *stream);
/* pcg_plan_batches for n_epochs epochs at once - epoch e's n_total picks at nodes + e * n_total (labels likewise), its batches in
 * slots e * ceil(n_total / B) ..., every epoch's last batch the shorter one: the launch's latency chain (two dependent load
 * levels and a hand-off) is paid once per n_epochs epochs.  bump_counter += n_epochs. */
int pcg_plan_epochs(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t n_total, int32_t n_epochs, int32_t B,
                    const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self, void *plans,
                    int64_t plan_stride, int64_t list_capacity, uint32_t *status, uint64_t *bump_counter, void *stream);
/* select + gather WITHOUT the combine launch (two launches): rows whose list fits one 128-entry gather chunk are finished
 * (their mean is in agg), longer rows are left as per-chunk partial sums in the workspace; pcg_train_dense, given the same
 * workspace and cnt, adds them up in chunk order (bit-identical to pcg_choose_aggregate_planned's agg) while it stages its
 * tile.  pcg_gather_lists is the gather half on its own (norm = PCG_NORM_COUNT), pcg_gather_lists_planned the same with the
 * plan part outside the workspace.
 * sync_words != NULL (the four words of pcg_train_dense / pcg_step_scores_train): pos_keys' scratch half holds the UNSORTED
 * train-pos keys of this step (pcg_step_scores_train) and - if pcg_pos_sort_in_select(n_pos) - the select kernel sorts them
 * itself, the work shared by its workgroups before they start on the rows (ranks accumulated in the words behind sync_words[3],
 * key groups counted in sync_words[3] - zero on entry; pcg_step_scores_train zeroes it); only a row with minority picks waits
 * for that count (a bounded wait: PCG_ST_SYNC_TIMEOUT if it ever ran out).  The launch also clears sync_words[1].  Results are bit-identical
 * to sorting first (pcg_pos_sort) and passing sync_words = NULL. */
int pcg_choose_gather_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                              const float *s0, const float *center_s0, const uint64_t *pos_keys,
                              const double *thresholds, const double *rho, int32_t train_flag, int32_t add_self,
                              float *agg, int32_t agg_stride, int32_t *cnt, void *workspace, const void *plan,
                              int64_t list_capacity, uint32_t *status, uint32_t *sync_words, void *stream);
int32_t pcg_pos_sort_in_select(int32_t n_pos);      /* 1: 0 < n_pos <= 16384 (host helper) */
/* 16384 < n_pos <= 131072 train positives whose UNSORTED keys exist already (pos_keys' scratch half: pcg_choose_gather_train forms
 * them beside the score pass for these sizes too): the bucket sort in ONE launch - workgroup b sorts the same small sample, takes
 * its two splitters, streams all keys once (counting those below its range: the bucket's offset; collecting its own in LDS),
 * ranks them and stores them in place; no counts, no scatter, no hand-off between workgroups (pcg_pos_sort's four launches took
 * 35 us at 40 K keys).  A bucket of more than 4096 keys sets PCG_ST_SORT_OVERFLOW in *status (may be NULL).  src/layers.py:683-691's order. */
int32_t pcg_pos_sort_one_launch(int32_t n_pos);
int pcg_pos_sort_raw(const pcg_graph_desc *g, uint64_t *pos_keys, uint32_t *status, void *stream);
/* scores of rows [row_begin, row_end) (-> s0_out[row], or s0_out[row_ids[row]]) || the UNSORTED train-pos keys into pos_keys'
 * scratch half (pos_keys may be NULL; pos_row_base >= 0: train positive i's feature row is table row pos_row_base + i - required
 * with row_ids -, else row train_pos[i]); zeroes sync_words[3].  ONE launch, no plan, no parameter update.
 * touched (may be NULL; needs row_ids == NULL, row_begin == 0): only the rows the byte map marks (pcg_mark_touched) are scored.
 * Train positives by node id (no row_ids / pos_row_base) and more than 16384 of them: the bucket sort's launches follow and
 * pos_keys is sorted on return. */
int pcg_step_scores(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end, float *s0_out,
                    const int32_t *row_ids, uint64_t *pos_keys, int64_t pos_row_base, uint32_t *sync_words, const uint8_t *touched,
                    void *stream);
/* select + gather of a TRAINING step with the label classifier stepped on its own: the THREE-launch step
 *       pcg_choose_gather_train (select_rows, gather_train_kernel)  ->  pcg_train_dense(adam_clf = 2).
 * The label classifier (src/layers.py:230-243; its loss term src/model.py:54-61) gets gradient from nothing but the batch
 * centres' feature rows and labels - not from the selection, the aggregates or the GNN weights.  So
 *   - ONE workgroup of the select launch does the classifier's whole step for THIS batch: logits of the centres' rows, the
 *     lambda_1 / global-batch weighted cross-entropy gradient summed over the batch in a fixed order, torch.optim.Adam's update
 *     (t = step_counter[0] + 1; m, v in place at the classifier's offset).  clf_next [2 * feat_dim + 2] (W row-major, then b):
 *     in: the classifier s0 was scored with; out: the updated one.  The in-value is copied to theta's classifier (the dense
 *     kernel computes this step's loss term with it);
 *   - the gather launch carries, in extra workgroups, the previous step's deferred Adam update of every OTHER parameter (from
 *     slabs, if sync_words[1] is set - the dense kernel that follows reads the result) and, if score_next, the NEXT step's score
 *     pass -> s0 and unsorted train-pos keys -> pos_keys' scratch half, computed with the new clf_next (next_touched != NULL:
 *     only the rows that byte map marks; n_pos > 16384: the bucket sort's launches follow).  It zeroes sync_words[3].
 * Nothing waits inside a launch for any of this.  On entry s0 / pos_keys hold THIS step's scores / unsorted keys (from the
 * previous step's call with score_next = 1, or from pcg_step_scores(W = clf_next, b = clf_next + 2 * feat_dim)).
 * pcg_adam_flush(..., clf_next) afterwards brings theta up to date (deferred update + the stepped classifier).
 * acts != NULL (with act_ld; `slabs` is then only the classifier step's scratch, >= 8 * n_params floats): the previous step ran
 * pcg_train_dense(adam_clf = 3) - no gradient slabs exist; the deferred update's workgroups are the weight-gradient GEMMs over
 * that step's batch, each 16 x 16 output tile's workgroup applying Adam to its own parameters (pcg_wgrad below).
 * keys_sorted != 0: pos_keys' first half holds THIS step's train-pos keys SORTED already (the previous step's
 * pcg_train_dense(adam_clf = 3, sort_keys) sorted them on CUs its tiles leave idle, or pcg_pos_sort behind pcg_step_scores): the
 * select launch sorts nothing, no row waits, and a positive hub row's minority window search runs on one of its waves beside the
 * others' key pass.  Same lists either way, bit for bit.
 * inv_count = 1 / global batch size.  Selection, lists and aggregates: exactly pcg_choose_gather_planned(train_flag = 1). */
int pcg_choose_gather_train(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B, float *s0,
                            uint64_t *pos_keys, const double *thresholds, const double *rho, int32_t add_self, float *agg,
                            int32_t agg_stride, int32_t *cnt, void *workspace, const void *plan, int64_t list_capacity,
                            uint32_t *status, uint32_t *sync_words, float *theta, float *m, float *v, int32_t emb, float *clf_next,
                            const float *slabs, const int32_t *step_counter, float lambda_1, float inv_count, double lr, double beta1,
                            double beta2, double eps, double weight_decay, int32_t score_next, const uint8_t *next_touched,
                            const float *acts, int32_t act_ld, float *wg_scratch, int32_t keys_sorted, void *stream);
int32_t pcg_sync_words_count(void);                 /* uint32 words of a `sync_words` buffer (zero-initialised ONCE by the caller; the
                                                        kernels leave every word but [1], [2] zero between launches) */
int pcg_aggregate_lists_planned(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                                const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                                int64_t list_capacity, int32_t norm, float *agg, int32_t agg_stride, uint32_t *status,
                                void *stream);       /* pcg_aggregate_lists with the plan part outside the workspace */
int pcg_gather_lists_planned(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                             const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                             int64_t list_capacity, float *agg, int32_t agg_stride, uint32_t *status, void *stream);
int pcg_gather_lists(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                     const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, float *agg,
                     int32_t agg_stride, uint32_t *status, void *stream);
int pcg_choose_aggregate_planned(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                                 const float *s0, const float *center_s0, const uint64_t *pos_keys,
                                 const double *thresholds, const double *rho, int32_t train_flag,
                                 int32_t norm, int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt,
                                 void *workspace, int64_t list_capacity, uint32_t *status, void *stream);

/* diagnostic only: per-row phase timestamps of the select kernels ([rows][8] uint64, 10-ns ticks); NULL = off */
void pcg_debug_set_stamps(void *ptr);
void pcg_debug_set_dense_stamps(void *ptr);   /* [tiles][16] uint64 per dense_step tile */

/* Upper bound on |chosen set| of one row (host helper, pure arithmetic):
 * (deg > k+1 ? k : deg) + m [+1 if add_self]. */
int64_t pcg_sel_capacity_row(int64_t deg, double threshold, double rho, int32_t positive_train,
                             int32_t n_pos, int32_t add_self);

/* ---- segmented mean over explicit index lists -----------------------------------
 * out[i,:] = sum_{j in idx[begin[i] .. begin[i]+count[i])} X[j,:] / norm(count[i])
 * Replaces mask.div(num_neigh).mm(embed_matrix) for a given samp_neighs
 * (layers.py:599-624; graphsage.py:82-95 and, with PCG_NORM_SQRT_COUNT, 216-231). */
int pcg_segment_mean(const pcg_graph_desc *g, const int64_t *begin, const int32_t *count,
                     const int32_t *idx, int32_t n_rows, int32_t norm,
                     float *out, int32_t out_stride, void *stream);

/* ---- pick (label-balanced sampler) ------------------------------------------------
 * Replaces random.choices(idx_train, weights=deg/LF, k) (src/utils.py:274-278):
 * out[i] = idx_train[bisect_right(cum, u[i] * cum[n-1], 0, n-1)].
 * cum is the sequential fp64 running sum of the weights (host-computed once, it
 * is epoch-invariant).  uniforms: fp64 [k] in [0,1); if NULL they are drawn on
 * the device from Philox4x32-10(seed, counter = draw index). */
int pcg_pick(const double *cum, const int32_t *idx_train, int32_t n_train,
             const double *uniforms, uint64_t seed, uint64_t epoch, int32_t k, int32_t *out, void *stream);

/* One epoch's picks, shuffled, with their labels, in one launch: pick (src/utils.py:274-278) + random.shuffle
 * (src/model_handler.py:131-133) + the label lookup of the batch loop (:147).  Draw i is pcg_pick's draw i of
 * (seed, epoch); it is stored at position sigma(i), sigma a keyed pseudo-random permutation of [0, k) (an invertible
 * integer mix on the next power of two, cycle-walked into [0, k)).  epoch = epoch_base + (epoch_counter[0] if given); with
 * bump != 0 a second, one-thread launch increments epoch_counter[0] afterwards, so a captured graph replays a new epoch
 * each time.  epoch_counter points to TWO zero-initialised uint64 words ([1] is reserved).
 * labels_all: int32 label of every node (or NULL with out_labels NULL). */
int pcg_pick_shuffled(const double *cum, const int32_t *idx_train, int32_t n_train, uint64_t seed, uint64_t epoch_base,
                      uint64_t *epoch_counter, int32_t bump, int32_t k, const int32_t *labels_all, int32_t *out_ids,
                      int32_t *out_labels, void *stream);

/* pcg_pick_shuffled for n_epochs consecutive epochs in one launch: epoch e (= epoch_base + epoch_counter[0] + e) is drawn and
 * shuffled exactly as a call of its own would, into out_ids + e * k / out_labels + e * k; bump != 0: epoch_counter[0] += n_epochs. */
int pcg_pick_shuffled_epochs(const double *cum, const int32_t *idx_train, int32_t n_train, uint64_t seed, uint64_t epoch_base,
                             uint64_t *epoch_counter, int32_t bump, int32_t n_epochs, int32_t k, const int32_t *labels_all,
                             int32_t *out_ids, int32_t *out_labels, void *stream);

/* ---- dense tail: relation / inter GEMMs, classifier, loss, backward, Adam ---------------
 * Parameters live in ONE flat f32 buffer `theta` in this order (offsets from
 * pcg_dense_param_offset; shapes are the reference's state-dict shapes, row-major):
 *   which 0  weight                      [2, E]          src/model.py:29
 *   which 1  inter1.weight               [F + R*E, E]    src/layers.py:196
 *   which 2  inter1.intra_agg{rel+1}.weight [2F, E]      src/layers.py:559
 *   which 3  inter1.label_clf.weight     [2, F]          src/layers.py:200
 *   which 4  inter1.label_clf.bias       [2]
 * pcg_dense_step replaces, for one batch (B rows, tiles of 16 rows):
 *   self_feats gather, cat/mm/relu per relation   layers.py:273-277, 625-629
 *   cat/mm/relu of the inter aggregator           layers.py:284-289 ([B,E], not transposed)
 *   scores = W_cls . embeds, label-aware logits   model.py:38, layers.py:236-243
 *   both CrossEntropyLoss terms + loss.backward() model.py:54-61, model_handler.py:152
 * Outputs: logits [B,2], center [B,2]; optional combined [B,E], row_loss [B]
 * (per-row  xent(gnn) + lambda_1*xent(label); the batch loss is inv_count * sum).
 * slabs != NULL (training): labels required; slabs [pcg_dense_n_tiles(B), n_params]
 * receives per-tile partial gradients (already scaled by inv_count = 1/global batch),
 * and *step_counter (int32, may be NULL) is incremented.  slabs == NULL: inference.
 * pcg_adam_step sums the slabs in tile order (bitwise reproducible) and, if `apply`,
 * performs torch.optim.Adam's update with coupled weight decay on theta/m/v using
 * t = *step_counter (model_handler.py:124,153); grad_out [n_params] (optional)
 * receives the summed gradient. */
int64_t pcg_dense_n_params(int32_t feat_dim, int32_t emb, int32_t n_rel);
int64_t pcg_dense_param_offset(int32_t feat_dim, int32_t emb, int32_t n_rel, int32_t which, int32_t rel);
int32_t pcg_dense_n_tiles(int32_t B);
int pcg_dense_step(const pcg_graph_desc *g, const float *theta, int32_t emb,
                   const int32_t *ids, const int32_t *labels, int32_t B,
                   const float *agg, int32_t agg_stride, float lambda_1, float inv_count,
                   float *logits, float *center, float *combined, float *row_loss,
                   float *slabs, int32_t *step_counter, void *stream);
int pcg_adam_step(float *theta, float *m, float *v, const float *slabs, int32_t n_slabs, int64_t n_params,
                  const int32_t *step_counter, double lr, double beta1, double beta2, double eps,
                  double weight_decay, float *grad_out, int32_t apply, void *stream);

/* The training step's tail and front with Adam taken off the critical path (src/model_handler.py:149-153).
 *
 * pcg_train_dense = pcg_dense_step, plus
 *   - workspace != NULL (with cnt, plan, list_capacity as given to pcg_choose_gather_planned): aggregates of rows the gather
 *     left as partial sums are added up here (no combine launch);
 *   - adam_clf == 2 (training; slabs, sync_words required): gradient slabs only, marked as waiting (sync_words[1] = 1,
 *     sync_words[2] = #slabs) for the deferred update of the next pcg_choose_gather_train / pcg_adam_flush; the label
 *     classifier's own step is pcg_choose_gather_train's (its share of the slabs is written but not used);
 *   - adam_clf == 3 (training; acts, act_ld, sync_words required; slabs unused): NO gradient slabs.  The launch - one workgroup
 *     per 16 batch rows - runs forward, loss and the activation gradients and leaves, TRANSPOSED (one batch row per column,
 *     act_ld floats per row, act_ld >= B rounded up to 16, a multiple of 4; acts 16-byte aligned, pcg_wgrad_act_rows(...) rows):
 *     [self | h_r] , agg_r, dcomb, dh_r, combined, dlogits, dcentre - what every weight gradient is a GEMM over the batch of
 *     (src/layers.py:625-629, 284-289; src/model.py:54-61).  It marks them as waiting (sync_words[1] = 2, sync_words[2] = B / 16
 *     rounded up) for the next pcg_choose_gather_train(acts) / pcg_adam_flush(acts), whose workgroups run those GEMMs (f32 MFMA,
 *     fixed summation order) and apply Adam tile by tile.  adam_clf == 4: the same without marking (pcg_wgrad follows).
 *     sort_keys != NULL (adam_clf == 3 only; the pos_keys buffer whose scratch half holds the NEXT step's unsorted train-pos keys,
 *     formed by the pcg_choose_gather_train(score_next) before this launch): if pcg_dense_sorts_keys(B, n_pos) the launch carries
 *     ceil(n_pos / 64) more workgroups that rank-sort them into the first half (src/layers.py:683-691's order) - one workgroup
 *     per 16 rows leaves most CUs of a batch of <= ~3000 rows idle; the next pcg_choose_gather_train is then told keys_sorted;
 *   - adam_clf == 1 (training: slabs, m, v, sync_words required): the workgroup that arrives last (device-scope ticket,
 *     write-through partial gradients) sums the label classifier's gradient over the tiles in tile order and applies
 *     Adam to those 2 * feat_dim + 2 parameters - the only ones the next step's score pass reads - and the launch marks the
 *     slabs as holding a gradient the OTHER parameters have not seen yet: sync_words[1] = 1, sync_words[2] = #slabs.
 *   sync_words: pcg_sync_words_count() zero-initialised uint32 device words owned by the caller ([0] arrival ticket, 0 between
 *     launches; [3] and the words behind it belong to pcg_choose_gather_planned's in-kernel sort).
 * pcg_step_front_train = pcg_step_front(train_flag = 1) reading the label classifier from theta, with that deferred update
 *   (parameters [0, offset of label_clf.weight), same arithmetic and summation order as pcg_adam_step) applied by extra
 *   workgroups beside the score pass when sync_words[1] is set; its second launch clears sync_words[1].
 * pcg_step_scores_train = the front of a training step whose plan exists already (pcg_plan_batches), ONE launch:
 *   [train-pos keys from their feature rows -> pos_keys' scratch half || that deferred update || score pass -> s0]; it zeroes
 *   sync_words[3]; the sort is left to pcg_choose_gather_planned(..., sync_words) (n_pos > 16384: the bucket sort's launches
 *   follow here instead and pos_keys is sorted on return).  Replaces src/layers.py:230-237.
 *   touched (may be NULL = score every row): this batch's byte map from pcg_mark_touched - only rows whose byte is set are
 *   scored (bit for bit the scores pcg_score_table gives them); the other entries of s0 keep whatever they held - the batch's
 *   selection never reads them.  For graphs whose feature table is far larger than what a batch touches.
 * pcg_adam_flush applies a still-deferred update now (two or three small launches) - before parameters are read or saved
 *   (slabs may be NULL when acts is given: a step of adam_clf = 3 is applied by the weight-gradient GEMMs, sync_words[1] == 2);
 *   clf_next != NULL (the pcg_choose_gather_train step): the stepped label classifier is copied into theta [p_end, n_params) too,
 *   if an update was pending. */
int pcg_step_scores_train(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, float *s0,
                          uint64_t *pos_keys, const float *slabs, const int32_t *step_counter, uint32_t *sync_words,
                          double lr, double beta1, double beta2, double eps, double weight_decay, const uint8_t *touched,
                          void *stream);
/* Which rows the batches nodes[s * B, min((s + 1) * B, n_total)) can read the score of (src/layers.py:226-237 scores exactly
 * `unique_nodes` = batch + neighbours): one byte map per batch at maps + s * map_stride (map_stride >= pcg_touched_bytes(n_nodes),
 * a multiple of 16; maps 16-byte aligned) - zeroed, then 1 for every centre of the batch, every neighbour it has in any
 * relation, and every train positive (their scores feed pcg_pos_sort when there are more than 16384 of them).  Two launches for ALL batches: per epoch, like pcg_plan_batches - it depends on the picks and the CSR only. */
int64_t pcg_touched_bytes(int64_t n_nodes);
/* queue: uint32 [4 + n_rel * n_total] scratch (may be NULL if g->max_degree <= 4096): rows of more than 4096 neighbours - a
 * hub's - are queued by the per-row pass and marked by the whole grid in a third launch (one wave per row set the duration by
 * the longest row: 1.6 ms per epoch at 10 M nodes / 200 M edges). */
int pcg_mark_touched(const pcg_graph_desc *g, const int32_t *nodes, int32_t n_total, int32_t B, uint8_t *maps,
                     int64_t map_stride, uint32_t *queue, void *stream);
/* pcg_mark_touched from the batches' PLANS (pcg_plan_batches has run on `plans` for the same nodes / n_total / B / list_capacity;
 * an epoch of a pcg_plan_epochs group: pass that epoch's first slot and its n_total): the plan slots hold every row's record and the
 * degree-tier queues, so the hub rows (> 4096 neighbours) are swept, one id range at a time, into LDS bitmaps that are expanded
 * into the byte maps with coalesced stores - which also write every other byte zero (no zeroing pass) - and the other rows, the
 * centres and the train positives are marked afterwards.  Two launches; the same maps as pcg_mark_touched, bit for bit. */
int pcg_mark_touched_planned(const pcg_graph_desc *g, const int32_t *nodes, int32_t n_total, int32_t B, const void *plans,
                             int64_t plan_stride, int64_t list_capacity, uint8_t *maps, int64_t map_stride, void *stream);
int pcg_train_dense(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, const int32_t *ids,
                    const int32_t *labels, int32_t B, const float *agg, int32_t agg_stride, const int32_t *cnt,
                    const void *workspace, const void *plan, int64_t list_capacity, float lambda_1, float inv_count, float *logits,
                    float *center, float *combined, float *row_loss, float *slabs, int32_t *step_counter,
                    uint32_t *sync_words, double lr, double beta1, double beta2, double eps, double weight_decay,
                    int32_t adam_clf, float *acts, int32_t act_ld, uint64_t *sort_keys, void *stream);
int32_t pcg_dense_sorts_keys(int32_t B, int32_t n_pos);   /* 1: the launch above (adam_clf = 3, sort_keys) sorts n_pos keys at batch B */
int pcg_step_front_train(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, float *s0,
                         uint64_t *pos_keys, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const double *thresholds, const double *rho, int32_t add_self, void *workspace,
                         int64_t list_capacity, uint32_t *status, const float *slabs, const int32_t *step_counter,
                         uint32_t *sync_words, double lr, double beta1, double beta2, double eps, double weight_decay,
                         void *stream);
/* The optimizer of a data-parallel / partitioned step, whose gradient passes through an all-reduce: pcg_grad_reduce sums the
 * slabs (tile order) into grad_out and sets flag[0] ("a gradient is waiting"); after the collective, pcg_adam_apply_pending - at
 * the head of the NEXT step's launches, or on its own before parameters are read - applies torch.optim.Adam's update from grad to
 * every parameter if flag[0] is set.  The flag is cleared by a later launch: clear != 0 adds a one-thread launch that does it;
 * clear == 0 leaves it to the caller's next launch (flag = sync_words + 1: pcg_choose_select_planned(sync_words) clears it).
 * flag: one zero-initialised uint32 device word. */
int pcg_grad_reduce(const float *slabs, int32_t n_slabs, int64_t n_params, float *grad_out, uint32_t *flag, void *stream);
int pcg_adam_apply_pending(float *theta, float *m, float *v, const float *grad, int64_t n_params, const int32_t *step_counter,
                           uint32_t *flag, int32_t clear, double lr, double beta1, double beta2, double eps, double weight_decay,
                           void *stream);
/* The partitioned step's front and gather (pc-gnn_amd/dist.py; the reference has no distributed code).
 * pcg_step_scores_dist = pcg_step_scores with the optimizer riding in it, ONE launch behind the gradient all-reduce: if
 *   sync_words[1] == 1 (pcg_wgrad(flag_set) / pcg_grad_reduce) torch.optim.Adam's update from grad [n_params] (the all-reduced
 *   gradient) is applied to EVERY parameter by some workgroups while the others score rows [row_begin, row_end) ->
 *   s0_out[row_ids[row]] and form the train positives' unsorted keys (rows pos_row_base + i; pos_keys may be NULL) with the label
 *   classifier AFTER that update - each works it out for itself from clf_snap [3 * (2 feat_dim + 2)] (the classifier's
 *   parameters, m, v as of the last applied update) and the gradient, same arithmetic, same bits.  The flag is cleared by the
 *   step's select launch (pcg_choose_select_planned(sync_words)); the launch zeroes sync_words[3].
 * pcg_gather_lists_dist = pcg_gather_lists_planned over lists of NODE ids: translated to rows of the extended table
 *   [ owned | train-pos | halo ] as they are read (table / counts / pos_ids / pos_idx as pcg_halo_lookup; the list is left as it
 *   is; an id in none of the three is skipped and sets overflow bit 4 in counts[128]); snap_dst != NULL: the launch also copies
 *   theta / m / v [clf_offset, clf_offset + clf_n) to snap_dst [3 * clf_n] - the snapshot the NEXT step's pcg_step_scores_dist reads. */
int pcg_step_scores_dist(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, const float *grad,
                         const float *clf_snap, int64_t row_begin, int64_t row_end, float *s0_out, const int32_t *row_ids,
                         uint64_t *pos_keys, int64_t pos_row_base, const int32_t *step_counter, uint32_t *sync_words, double lr,
                         double beta1, double beta2, double eps, double weight_decay, void *stream);
int pcg_gather_lists_dist(const float *X, int32_t feat_dim, int32_t feat_stride, int64_t table_rows, int32_t n_rows,
                          const int32_t *cnt, const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan,
                          int64_t list_capacity, float *agg, int32_t agg_stride, uint32_t *status, int32_t lo, int32_t hi,
                          int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos, const uint32_t *table,
                          int64_t table_slots, uint32_t *counts, int32_t halo_cap, int32_t halo_base, const float *theta,
                          const float *m, const float *v, int64_t clf_offset, int32_t clf_n, float *snap_dst, void *stream);
int pcg_adam_flush(float *theta, float *m, float *v, const float *slabs, int32_t n_slabs, int64_t n_params, int64_t p_end,
                   const int32_t *step_counter, uint32_t *sync_words, double lr, double beta1, double beta2, double eps,
                   double weight_decay, const float *clf_next, const float *acts, int32_t act_ld, int32_t feat_dim, int32_t emb,
                   int32_t n_rel, float *wg_scratch, void *stream);
/* The weight gradients of pcg_train_dense(adam_clf = 3 / 4)'s transposed activations as GEMMs over the batch (B rows; one
 * 256-thread workgroup per 16 x 16 tile of a weight matrix, its four waves a quarter of the batch each, partial tiles added in
 * wave order): grad_out [n_params] (may be NULL) gets the gradient, apply != 0 applies torch.optim.Adam's update (coupled weight
 * decay; t = step_counter[0]) to the tile's parameters.  with_clf == 0 leaves the label classifier's 2 * feat_dim + 2 entries
 * alone (its step is pcg_choose_gather_train's).  Batches beyond 1024 rows: a tile is shared by one workgroup per 1024 rows, whose
 * partial tiles meet in wg_scratch (pcg_wgrad_scratch_bytes(feat_dim, emb, n_rel, largest B), zero-initialised ONCE by the caller:
 * arrival tickets, left zero by every launch, then the partial tiles) and are added in part order by the one that arrives last;
 * wg_scratch may be NULL for B <= 1024 (act_ld <= 1024 where the batch size is only known on the device: pcg_adam_flush,
 * pcg_choose_gather_train).  act_ld: a multiple of 16.  flag_set (may be NULL): a device word the launch sets to 1 - "a gradient
 * is waiting" for pcg_step_scores_dist / pcg_adam_apply_pending.  Replaces src/model_handler.py:152-153 for those parameters. */
int64_t pcg_wgrad_act_rows(int32_t feat_dim, int32_t emb, int32_t n_rel);
int64_t pcg_wgrad_scratch_bytes(int32_t feat_dim, int32_t emb, int32_t n_rel, int32_t B);
int pcg_wgrad(const float *acts, int32_t act_ld, int32_t B, int32_t feat_dim, int32_t emb, int32_t n_rel, float *theta, float *m,
              float *v, const int32_t *step_counter, double lr, double beta1, double beta2, double eps, double weight_decay,
              float *grad_out, int32_t apply, int32_t with_clf, float *wg_scratch, uint32_t *flag_set, void *stream);

/* ---- multi-GPU halo exchange helpers (no counterpart in the reference) ----------
 * A rank of a partitioned run holds the table [ owned rows | train-pos rows | halo ]; CSR rows and selection lists hold
 * GLOBAL ids.  Feature rows never change, so a fetched row stays valid: the exchange runs once per WINDOW of steps over every
 * neighbour the window's centres have; a step only looks rows up.  Nothing here is sized by the node-id space.
 *   bounds   int32 [world + 1]: rank r owns ids [bounds[r], bounds[r + 1])          (device)
 *   pos_ids / pos_idx  int32 [n_pos]: the train-pos ids ascending / their row in the replicated block   (device)
 *   table    uint32 [2 * table_slots], table_slots = pcg_halo_table_slots(halo_cap)
 *   counts   uint32 [131], zeroed ONCE by the caller; after a collect [0, world) = unique remote ids per owner, [128] =
 *            overflow bits (1: table full, 2: more unique remote ids than halo_cap / than an owner's pitch, 4: a lookup
 *            missed; STICKY across calls - the caller clears them after looking, so one look per epoch sees every step),
 *            [129] / [130] = the largest number of unique remote ids a window needed so far / needed from one owner
 *   uniq     int32 [halo_cap]: the request list - the unique remote ids grouped by owner in rank order; it doubles as the
 *            halo rows' id column (row halo_base + i holds node uniq[i], -1 = unused)
 *   owner_pitch  0: packed (owner o's ids start where owner o - 1's end; the all-to-all's split sizes are the counts)
 *                > 0 (halo_cap == max(world - 1, 1) * owner_pitch): the j-th OTHER rank's ids (ranks in order, self_rank left
 *                out) sit at [j * pitch, (j + 1) * pitch), unused entries are -1 - the all-to-alls' split sizes are known in
 *                advance, and no count has to reach the host before they are issued
 * pcg_halo_collect: reset (table, per-window counts, request list) + walk the CSR rows (all relations) of `centres` (local
 *   row numbers [n_centres], duplicates allowed) inserting their remote non-train-pos neighbours + assign slots / fill the
 *   request list (an id that does not fit gets no slot: the overflow bit).  Then all-to-all #1 (ids), pcg_halo_serve on the
 *   owner: out[i, :] = X[req[i] - lo, :] (whole padded rows; req[i] < 0 or not owned: row i untouched), all-to-all #2 (rows).
 * pcg_halo_lookup: per step - the list's global ids -> rows of the table: owned id -> id - lo; train-pos id -> n_local + its
 *   row; remote id -> halo_base + slot; an id the window did not collect becomes a hole (-1) and sets bit 4.
 * A rank then holds the feature row of every node its window can touch and scores them itself (pcg_step_front_a with
 * row_ids): the step needs no score exchange - its only collective is the gradient all-reduce. */
int64_t pcg_halo_table_slots(int32_t halo_cap);
int pcg_halo_collect(const pcg_graph_desc *g, const int32_t *centres, int32_t n_centres, int32_t lo, int32_t hi,
                     int32_t n_local, const int32_t *pos_ids, int32_t n_pos, const int32_t *bounds, int32_t world,
                     uint32_t *table, int64_t table_slots, uint32_t *counts, int32_t *uniq, int32_t halo_cap,
                     int32_t halo_base, int32_t owner_pitch, int32_t self_rank, void *stream);
int pcg_halo_serve(const pcg_graph_desc *g, const int32_t *req, int32_t n_req, int32_t lo, int32_t n_local, float *out,
                   int32_t out_stride, void *stream);
int pcg_halo_lookup(const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan, int64_t list_capacity, int32_t lo, int32_t hi,
                    int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos, uint32_t *table,
                    int64_t table_slots, uint32_t *counts, int32_t halo_cap, int32_t halo_base, void *stream);

/* gather rows: out[i, :feat_dim] = X[ids[i], :feat_dim]  (self_feats, layers.py:273-277) */
int pcg_gather_rows(const pcg_graph_desc *g, const int32_t *ids, int32_t n_ids,
                    float *out, int32_t out_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PCGNN_H */
