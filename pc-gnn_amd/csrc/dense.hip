// Dense tail of the PC-GNN step for gfx950: relation GEMMs, inter GEMM, classifier, the two cross-entropy terms,
// their backward and Adam.
//
//   dense_step : one 1024-thread workgroup per tile of 16 batch rows (up to 4 of them when the batch has few tiles: all
//                run the forward pass, the weight-gradient tiles are dealt out among them) does the whole forward for its
//                rows, the loss gradients, and this tile's partial weight gradients (split-K over the batch: one partial
//                "slab" per tile, no atomics => bitwise reproducible).  GEMMs run on the f32 matrix cores
//                (v_mfma_f32_16x16x4_f32: exact fmaf chains).  Rows whose selection list the gather left as several
//                partial sums are summed (in chunk order) while the tile is staged - no combine launch.  Optionally the
//                workgroup whose label-classifier gradient arrives last applies Adam to those 2F + 2 parameters - the only
//                ones the next step's score pass needs - so that the update of all others can ride along that pass.
//   adam_reduce: sums the slabs in tile order and applies torch.optim.Adam's update (coupled L2 weight decay) to a
//                range of the flat parameter buffer.
//
// f32 MFMA runs at the f32 vector rate (256 FLOP / clk / CU): one tile is ~2.6 MFLOP forward + backward, i.e. >= 4 us of
// matrix time on ONE CU, which is why a tile's work is spread over several workgroups rather than several tiles looped
// over by one (the slabs that costs are summed by the Adam pass, beside the next step's score pass).
//
// Reference lines replaced: src/layers.py:273-289, 625-629; src/model.py:34-62;
// src/model_handler.py:124,149-153 (optimizer.zero_grad / loss.backward / optimizer.step).
// Gradients never flow into the gathered features or the selection (features frozen,
// model_handler.py:86; selection is index-only), so backward is dense GEMMs only.
#include "choose.h"
#include "wgrad.h"

namespace pcg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16;          // batch rows per workgroup (one MFMA M-tile)
constexpr int DENSE_WAVES = 16;
constexpr int DENSE_THREADS = DENSE_WAVES * PCG_WAVE;
constexpr int WSTAGE = 8;       // weight float4 loads in flight per thread while staging

struct DenseArgs {
    const float *X;
    int32_t feat_dim, feat_stride, n_rel, emb;
    const int32_t *ids;
    const int32_t *labels;      // null => inference (no loss, no gradients)
    int32_t B;
    const float *agg;           // [R, B, agg_stride]
    int32_t agg_stride;
    // optional: rows of more than one gather chunk are summed here from the gather's partial sums (no combine launch)
    const int32_t *chunk_begin; // [R * B + 1] or null (agg is complete)
    const float *partial;       // [chunks, partial_stride]
    const int32_t *cnt;         // [R * B]
    int32_t partial_stride;
    const float *W_cls;         // [2, E]
    const float *W_inter;       // [F + R*E, E]
    const float *W_intra[PCG_MAX_REL];  // [2F, E]
    const float *W_clf;         // [2, F]
    const float *b_clf;         // [2]
    float lambda_1, inv_count;
    float *logits;              // [B, 2]
    float *center;              // [B, 2]
    float *combined;            // [B, E] or null
    float *row_loss;            // [B] or null
    float *slabs;               // [n_tiles, n_params] or null
    float *acts;                // [wgrad_act_rows][act_ld] or null: the step's activations / activation gradients, transposed
    int32_t act_ld;             //   (wgrad.h) INSTEAD of weight-gradient slabs - the weight gradients are GEMMs of a later launch
    int64_t n_params;
    int32_t *step_counter;      // incremented once per training launch (Adam's t), or null
    int32_t n_split;            // training: workgroups per 16-row tile; they all run the forward pass, the weight-gradient tiles are dealt out
    // optional: Adam for the label classifier's parameters by the workgroup whose gradient arrives last
    float *theta, *m, *v;       // null => off
    uint32_t *ticket;           // device word, 0 between launches
    uint32_t *staged;           // device word, 0 between launches: workgroups that take no ticket (sp != 0) and have read the classifier
    uint32_t *pending;          // two device words: [0] = 1 "the slabs hold a gradient not yet applied to the other parameters", [1] = its slab count
    AdamHyper h;
    unsigned long long *stamps; // diagnostic only (pcg_debug_set_dense_stamps): [tiles][16] wall-clock ticks, else null
    // optional riders: the NEXT step's train-pos sort (rank sort of the unsorted keys the gather launch before this one formed), by
    // workgroups behind the tiles' - only when tiles and sort together leave no CU with two workgroups (a batch of <= ~3000 rows)
    const uint64_t *sort_raw;   // null: off
    uint64_t *sort_out;
    int32_t sort_n, sort_cap, n_tile_blocks;
};
constexpr int DENSE_SORT_TILE = 4096;     // keys per LDS tile of the riding sort (32 KB of the launch's dynamic LDS)
#define DENSE_STAMP(slot) do { if (a.stamps && threadIdx.x == 0 && sp == 0) a.stamps[(size_t)tile_id * 16 + (slot)] = wall_clock64(); } while (0)

// flat parameter / gradient order: W_cls | W_inter | W_intra[0..R) | W_clf | b_clf
__host__ __device__ inline int64_t off_cls(int F, int E, int R) { return 0; }
__host__ __device__ inline int64_t off_inter(int F, int E, int R) { return 2 * (int64_t)E; }
__host__ __device__ inline int64_t off_intra(int F, int E, int R, int r) {
    return off_inter(F, E, R) + (int64_t)(F + R * E) * E + (int64_t)r * 2 * F * E;
}
__host__ __device__ inline int64_t off_clf(int F, int E, int R) { return off_intra(F, E, R, R); }
__host__ __device__ inline int64_t off_bias(int F, int E, int R) { return off_clf(F, E, R) + 2 * (int64_t)F; }
__host__ __device__ inline int64_t n_params_of(int F, int E, int R) { return off_bias(F, E, R) + 2; }

// One accumulator chain of n_steps v_mfma_f32_16x16x4_f32 (n_steps a multiple of MU: every K dimension is padded with
// zeros to a multiple of 4 * MU), software-pipelined by hand: the operands of the next MU steps are requested before the
// MFMAs of the current MU steps issue (two register sets), so that a wave's LDS reads overlap its own matrix work.
// (hipcc does not unroll these chains by itself; one step at a time every MFMA waits for its own two operand reads.  The
// fetched values go into the MFMAs untouched - any arithmetic on them would be scheduled, with its wait, ahead of the
// MFMAs and undo the prefetch.)  fa(s) / fb(s): this lane's A / B operand of step s.
constexpr int MU = 4;
constexpr int KPAD = 4 * MU;
template <class FA, class FB>
__device__ __forceinline__ void mfma_fetch(float (&av)[MU], float (&bv)[MU], int s, FA &fa, FB &fb) {
#pragma unroll
    for (int u = 0; u < MU; ++u) {
        av[u] = fa(s + u);
        bv[u] = fb(s + u);
    }
}
template <class FA, class FB>
__device__ __forceinline__ f32x4 mfma_chain(int n_steps, FA fa, FB fb) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a0[MU], b0[MU], a1[MU], b1[MU];
    if (n_steps <= 0) return acc;
    mfma_fetch(a0, b0, 0, fa, fb);
    for (int s = 0; s < n_steps; s += 2 * MU) {
        if (s + MU < n_steps) mfma_fetch(a1, b1, s + MU, fa, fb);          // (wave-uniform)
#pragma unroll
        for (int u = 0; u < MU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u], acc, 0, 0, 0);
        if (s + MU >= n_steps) break;
        if (s + 2 * MU < n_steps) mfma_fetch(a0, b0, s + 2 * MU, fa, fb);
#pragma unroll
        for (int u = 0; u < MU; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[u], b1[u], acc, 0, 0, 0);
    }
    return acc;
}

// C[16x16] = A * Bt^T with Bt in GLOBAL memory, row-major [n][k] (a weight matrix used transposed): lane (n = r, kq) would read
// Bt[n][4s + kq] for step s - 4 bytes from each of 16 rows per load instruction.  Instead a lane reads the float4
// Bt[n][16u + 4kq .. + 3] (16 rows x 64 contiguous bytes per instruction, a quarter of the instructions) and the four MFMA
// steps of block u take k = 16u + 4kq + i, i = 0..3: a permutation of the k order that the A operand (LDS, any pattern is
// cheap there) follows.  All loads of eight blocks are in flight before the first MFMA.
__device__ __forceinline__ f32x4 tile_lds_globT4(const float *ap /* A + r*lda + 4*kq */, const float *__restrict__ bp /* Bt + n*ldb + 4*kq */,
                                                 int n_blocks) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int NB = 8;
    for (int u0 = 0; u0 < n_blocks; u0 += NB) {
        f4 b[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int u = u0 + j < n_blocks ? u0 + j : n_blocks - 1;          // (clamped: unconditional loads)
            b[j] = *reinterpret_cast<const f4 *>(bp + 16 * u);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (u0 + j >= n_blocks) break;                                    // (wave-uniform)
            const float *aj = ap + 16 * (u0 + j);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[0], b[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[1], b[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[2], b[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aj[3], b[j].w, acc, 0, 0, 0);
        }
    }
    return acc;
}

// C[16x16] += A[16 x k-steps] (LDS, row-major, leading dim lda) * B (global, ld ldb, column n0..n0+15), k in [k_lo, k_hi)
__device__ __forceinline__ f32x4 tile_lds_glob(const float *A, int lda, const float *__restrict__ Bg, int ldb, int n0,
                                               int K, int k_lo, int k_hi, int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const float *ap = A + r * lda + k_lo + kq;
    // rows k >= K of B do not exist: A's pad columns are zero, so any finite value will do there - the last row's
    return mfma_chain((k_hi - k_lo) >> 2, [&](int s) { return ap[4 * s]; },
                      [&](int s) {
                          const int k = k_lo + 4 * s + kq;
                          return Bg[(size_t)(k < K ? k : K - 1) * ldb + n0 + r];
                      });
}

// the same with B an LDS copy of the weight matrix (leading dim ldb; rows beyond K are zero)
__device__ __forceinline__ f32x4 tile_lds_lds(const float *A, int lda, const float *Bl, int ldb, int n0, int k_lo, int k_hi,
                                              int lane) {
    const int r = lane & 15, kq = lane >> 4;
    const float *ap = A + r * lda + k_lo + kq;
    const float *bp = Bl + (k_lo + kq) * ldb + n0 + r;
    return mfma_chain((k_hi - k_lo) >> 2, [&](int s) { return ap[4 * s]; }, [&](int s) { return bp[4 * s * ldb]; });
}

// C[16x16] = At^T * Bt with both operands row tiles in LDS: C[m][n] = sum_t At[t][m0+m] * Bt[t][n0+n], t < 16
__device__ __forceinline__ f32x4 tile_ldsT_lds(const float *At, int lda, int m0, int M, const float *Bt, int ldb, int n0,
                                               int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    const bool mok = m0 + r < M;
    const int mr = mok ? m0 + r : M - 1;
    float av[TB / 4], bv[TB / 4];
#pragma unroll
    for (int j = 0; j < TB / 4; ++j) {
        const int t = 4 * j + kq;
        const float x = At[t * lda + mr];
        av[j] = mok ? x : 0.f;
        bv[j] = Bt[t * ldb + n0 + r];
    }
#pragma unroll
    for (int j = 0; j < TB / 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc, 0, 0, 0);
    return acc;
}

// sum over the 16 lanes of a DPP row (every lane gets it): quad permutes, then the half-row / row mirrors
__device__ __forceinline__ float row16_sum(float p) {
    p = dpp_add<0xB1>(p);
    p = dpp_add<0x4E>(p);
    p = dpp_add<0x141>(p);
    p = dpp_add<0x140>(p);
    return p;
}

// WLDS: the weight matrices are staged in LDS once per workgroup (when they fit), so every MFMA operand
// is an LDS read; otherwise the B operands stream from global memory / L2.
// Phases (one barrier between them): stage -> h_r for all relations -> combined (K split over the waves) -> logits + loss
// grads -> dcomb + small dW -> {dh_r for all r, dW_inter} -> dW_r for all r.
// F_, E_, R_ > 0: the shape is a compile-time constant (the datasets' shapes are instantiated below): every LDS offset is
// then an immediate and the index arithmetic folds away - with run-time shapes the kernel issues ~1400 vector and ~750
// scalar instructions per wave, most of them address arithmetic, and that issue time (not the matrix cores, 14 % busy,
// nor the LDS, 20 % busy) is what it is bound by.  0: run-time shape (any F, E % 16 == 0, R <= 8).
template <bool WLDS, int F_, int E_, int R_>
__global__ void __launch_bounds__(DENSE_THREADS) dense_step_kernel(const DenseArgs a) {
    extern __shared__ __align__(16) float sm[];
    if (a.sort_raw && (int)blockIdx.x >= a.n_tile_blocks) {
        // the next step's sorted train-pos keys (pos_rank_sort's body: a workgroup ranks 64 keys against all of them): the select
        // launch that follows finds them sorted - no in-kernel sort, no row waiting for it, and a positive hub row's window
        // search can run beside its key pass
        uint64_t *sh = reinterpret_cast<uint64_t *>(sm);
        int *part = reinterpret_cast<int *>(sh + DENSE_SORT_TILE);
        rank_sort_body<DENSE_SORT_TILE, DENSE_WAVES, false>(nullptr, nullptr, a.sort_n, a.sort_cap, a.sort_out,
                                                            (int)blockIdx.x - a.n_tile_blocks, sh, part, a.sort_raw);
        return;
    }
    const int F = F_ > 0 ? F_ : a.feat_dim, E = E_ > 0 ? E_ : a.emb, R = R_ > 0 ? R_ : a.n_rel;
    const int K1 = 2 * F, K1p = (K1 + KPAD - 1) / KPAD * KPAD, K2 = F + R * E, K2p = (K2 + KPAD - 1) / KPAD * KPAD;
    const int ld1 = K1p + 1, ld2 = K2p + 1, ldE = E + 1, ldW = E + 4;   // ldW: rows stay 16-B aligned (ds_write_b128)
    const int ntile_e = E / 16;
    const int kparts = ntile_e <= DENSE_WAVES ? DENSE_WAVES / ntile_e : 1;      // waves sharing one output tile of `combined`
    float *s_wi = sm;                               // WLDS: [K2p][ldW] copy of W_inter   (first: 16-B aligned)
    float *s_wr = s_wi + (WLDS ? K2p * ldW : 0);    // WLDS: [R][K1p][ldW] copies of W_intra; later the K-split partial tiles
    float *s_part = WLDS ? s_wr : s_wr;             // [kparts][TB][E] partial sums of `combined` (W_intra is dead by then)
    float *s_catr = s_wr + (WLDS ? R * K1p * ldW : kparts * TB * E);   // [R][TB][ld1]  [self | agg_r]
    float *s_cat = s_catr + R * TB * ld1;           // [TB][ld2]  [self | h_1 .. h_R]
    float *s_comb = s_cat + TB * ld2;               // [TB][ldE]
    float *s_dcomb = s_comb + TB * ldE;             // [TB][ldE]
    float *s_dh = s_dcomb + TB * ldE;               // [R][TB][ldE]
    float *s_dlog = s_dh + R * TB * ldE;            // [TB][2] d loss / d gnn logits
    float *s_dcl = s_dlog + TB * 2;                 // [TB][2] d loss / d centre scores (already times lambda_1)
    float *s_wc = s_dcl + TB * 2;                   // [2][E] W_cls, [2][F] W_clf, [2] b_clf
    int *s_flag = reinterpret_cast<int *>(s_wc + 2 * E + 2 * F + 4);   // [4] "this workgroup's classifier gradient arrived last"

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.n_split, tile_id = (int)blockIdx.x / S, sp = (int)blockIdx.x % S;
    const int row0 = tile_id * TB;
    const bool acts_mode = a.acts != nullptr;
    const bool train = a.slabs != nullptr || acts_mode;
    if (train && blockIdx.x == 0 && tid == 0) {
        // (an agent-scope atomic: the workgroup that applies the classifier's Adam reads the new count from another XCD)
        if (a.step_counter) __hip_atomic_fetch_add(a.step_counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a.pending) {
            a.pending[0] = acts_mode ? 2u : 1u;                 // the slabs (1) / acts (2) hold a gradient the other parameters still need
            a.pending[1] = (unsigned)(a.n_tile_blocks / S);           // ... in this many slabs / blocks of 16 batch rows
        }
    }
    // wave t's row of the loss phase: its label is requested now, not when the logits are ready
    const int my_b = row0 + wave;
    const int my_label = (a.labels && my_b < a.B) ? a.labels[my_b] : 0;
    DENSE_STAMP(0);
    if (a.stamps && threadIdx.x == 0 && sp == 0) a.stamps[(size_t)tile_id * 16 + 12] = clock64();     // shader cycles (diagnostic)

    // ---- stage ----------------------------------------------------------------------------------------------------
    // every global load of the prologue is requested before the first LDS store: weights (<= WSTAGE float4 per thread in
    // flight), the tile's self rows (ids -> rows), its aggregated rows (or their partial sums)
    {
        // self rows and aggregates: element e of [TB][F] (self) and [R][TB][F] (agg), one or two per thread.  Every load is
        // unconditional (indices clamped, the value discarded afterwards): a load inside a branch makes the compiler wait
        // for it at the branch's end, which turns the prologue into a chain of L2 round trips
        const int n_self = TB * F, n_agg = R * TB * F;
        const bool do_self = tid < n_self;                                 // (TB * F <= 1024 for F <= 64; a loop covers the rest)
        const int self_i = do_self ? tid : n_self - 1;
        const int self_t = self_i / F, self_f = self_i - self_t * F;
        const int self_b = row0 + self_t;
        const int self_id = a.ids[self_b < a.B ? self_b : a.B - 1];
        constexpr int NAGG = 2;
        float v_agg[NAGG];
        int agg_at[NAGG], agg_nch[NAGG], agg_cb[NAGG];
        size_t agg_row[NAGG];
        int agg_f[NAGG];
        bool agg_ok[NAGG];
#pragma unroll
        for (int u = 0; u < NAGG; ++u) {
            const int i0 = tid + u * DENSE_THREADS;
            const int i = i0 < n_agg ? i0 : n_agg - 1;
            const int r = i / (TB * F), j = i - r * TB * F, t = j / F, f = j - t * F;
            const int b = row0 + t;
            agg_at[u] = i0 < n_agg ? (r * TB + t) * ld1 + F + f : -1;
            agg_ok[u] = i0 < n_agg && b < a.B;
            agg_row[u] = (size_t)r * a.B + (b < a.B ? b : a.B - 1);
            agg_f[u] = f;
            agg_cb[u] = 0;
            agg_nch[u] = 1;
            if (a.chunk_begin) {                                           // (uniform: a kernel argument)
                agg_cb[u] = a.chunk_begin[agg_row[u]];
                agg_nch[u] = a.chunk_begin[agg_row[u] + 1] - agg_cb[u];
            }
            v_agg[u] = a.agg[agg_row[u] * a.agg_stride + f];
        }
        float v_self = a.X[(size_t)self_id * a.feat_stride + self_f];
        if (!do_self || self_b >= a.B) v_self = 0.f;
#pragma unroll
        for (int u = 0; u < NAGG; ++u) {
            if (!agg_ok[u]) v_agg[u] = 0.f;
            else if (agg_nch[u] == 0) v_agg[u] = 0.f / 0.f;       // empty set: 0 / 0 like the reference's mask.div (layers.py:612-614)
            else if (agg_nch[u] > 1) {                             // sum of the gather's partial sums, in chunk order, / |set|  (== combine_rows)
                float acc = 0.f;
                const float *pp = a.partial + (size_t)agg_cb[u] * a.partial_stride + agg_f[u];
                const int nch = agg_nch[u];
                // (eight loads in flight - clamped index, the extra values not added -, the adds in chunk order: a hub row has
                //  dozens of chunks, and this loop sits in front of everything else the workgroup stages)
                for (int jx = 0; jx < nch; jx += 8) {
                    float pv[8];
#pragma unroll
                    for (int x = 0; x < 8; ++x) pv[x] = pp[(size_t)(jx + x < nch ? jx + x : nch - 1) * a.partial_stride];
#pragma unroll
                    for (int x = 0; x < 8; ++x) acc = jx + x < nch ? acc + pv[x] : acc;
                }
                v_agg[u] = acc / (float)a.cnt[agg_row[u]];
            }
        }
        if constexpr (WLDS) {
            // the weight matrices as one list of rows [W_inter | W_intra[0] | ...] (contiguous in theta); thread -> (row, 16-B column chunk)
            const int c4 = E >> 2;
            const int cc = (tid % c4) * 4, r0 = tid / c4, rstep = DENSE_THREADS / c4;   // DENSE_THREADS % c4 == 0 (host-checked)
            const int n_rows = K2 + R * K1;
            // every workgroup streams the same 100+ KB out of L2 at the same moment: each starts at another row, so that they
            // do not all queue on the same L2 channels in the same order
            const int rot = (int)((blockIdx.x * 29u) % (unsigned)n_rows);
            for (int base = r0; base < n_rows; base += WSTAGE * rstep) {
                float4 wv[WSTAGE];
                int rows_[WSTAGE];
#pragma unroll
                for (int u = 0; u < WSTAGE; ++u) {
                    const int rl = base + u * rstep;
                    int rr = (rl < n_rows ? rl : n_rows - 1) + rot;
                    rr = rr >= n_rows ? rr - n_rows : rr;
                    rows_[u] = rl < n_rows ? rr : -1;
                    wv[u] = *reinterpret_cast<const float4 *>(a.W_inter + (size_t)rr * E + cc);
                }
#pragma unroll
                for (int u = 0; u < WSTAGE; ++u) {
                    const int rr = rows_[u];
                    if (rr >= 0) {
                        float *dst = rr < K2 ? s_wi + rr * ldW
                                             : s_wr + ((rr - K2) / K1) * K1p * ldW + ((rr - K2) % K1) * ldW;
                        *reinterpret_cast<float4 *>(dst + cc) = wv[u];
                    }
                }
            }
            for (int i = tid; i < (K2p - K2) * E; i += DENSE_THREADS) s_wi[(K2 + i / E) * ldW + i % E] = 0.f;
            for (int i = tid; i < R * (K1p - K1) * E; i += DENSE_THREADS) {
                const int r = i / ((K1p - K1) * E), j = i - r * (K1p - K1) * E;
                s_wr[r * K1p * ldW + (K1 + j / E) * ldW + j % E] = 0.f;
            }
        }
        for (int i = tid; i < 2 * E; i += DENSE_THREADS) s_wc[i] = a.W_cls[i];
        for (int i = tid; i < 2 * F; i += DENSE_THREADS) s_wc[2 * E + i] = a.W_clf[i];
        if (tid < 2) s_wc[2 * E + 2 * F + tid] = a.b_clf[tid];
        if (tid == 0) s_flag[0] = 0;
        // activations: self into [self | .] of every concatenation, aggregates, zero pad columns
        if (do_self) {
            s_cat[self_t * ld2 + self_f] = v_self;
            for (int r = 0; r < R; ++r) s_catr[(r * TB + self_t) * ld1 + self_f] = v_self;
        }
        for (int i = tid + DENSE_THREADS; i < n_self; i += DENSE_THREADS) {           // F > 64
            const int t = i / F, f = i - t * F, b = row0 + t;
            const float v = b < a.B ? a.X[(size_t)a.ids[b] * a.feat_stride + f] : 0.f;
            s_cat[t * ld2 + f] = v;
            for (int r = 0; r < R; ++r) s_catr[(r * TB + t) * ld1 + f] = v;
        }
#pragma unroll
        for (int u = 0; u < NAGG; ++u)
            if (agg_at[u] >= 0) s_catr[agg_at[u]] = v_agg[u];
        for (int i = tid + NAGG * DENSE_THREADS; i < n_agg; i += DENSE_THREADS) {     // R * F > 128
            const int r = i / (TB * F), j = i - r * TB * F, t = j / F, f = j - t * F, b = row0 + t;
            float v = 0.f;
            if (b < a.B) {
                const size_t row = (size_t)r * a.B + b;
                int cb = 0, nch = 1;
                if (a.chunk_begin) {
                    cb = a.chunk_begin[row];
                    nch = a.chunk_begin[row + 1] - cb;
                }
                if (nch == 0) v = 0.f / 0.f;
                else if (nch > 1) {
                    float acc = 0.f;
                    for (int jx = 0; jx < nch; ++jx) acc += a.partial[(size_t)(cb + jx) * a.partial_stride + f];
                    v = acc / (float)a.cnt[row];
                } else {
                    v = a.agg[row * a.agg_stride + f];
                }
            }
            s_catr[(r * TB + t) * ld1 + F + f] = v;
        }
        for (int i = tid; i < R * TB * (ld1 - K1); i += DENSE_THREADS) {              // pad columns K1 .. ld1
            const int rt = i / (ld1 - K1), c = K1 + i % (ld1 - K1);
            s_catr[rt * ld1 + c] = 0.f;
        }
        for (int i = tid; i < TB * (ld2 - K2); i += DENSE_THREADS) s_cat[(i / (ld2 - K2)) * ld2 + K2 + i % (ld2 - K2)] = 0.f;
    }
    __syncthreads();
    DENSE_STAMP(1);
    // (a workgroup that takes no ticket says here that it has read the classifier's weights - the last ticket holder overwrites
    //  them.  These arrivals are ~10 us ahead of their only reader: their serialisation on the counter costs nobody anything)
    if (a.theta && sp != 0 && tid == 0) __hip_atomic_fetch_add(a.staged, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- forward: h_r = relu([self | agg_r] W_r) for every relation   (layers.py:625-629) ---------
    for (int tile = wave; tile < R * ntile_e; tile += DENSE_WAVES) {
        const int r = tile / ntile_e, ct = tile - r * ntile_e;
        const float *A = s_catr + r * TB * ld1;
        const f32x4 c = WLDS ? tile_lds_lds(A, ld1, s_wr + r * K1p * ldW, ldW, ct * 16, 0, K1p, lane)
                             : tile_lds_glob(A, ld1, a.W_intra[r], E, ct * 16, K1, 0, K1p, lane);
        const int col = ct * 16 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) s_cat[(rq + i) * ld2 + F + r * E + col] = fmaxf(c[i], 0.f);
    }
    __syncthreads();
    DENSE_STAMP(2);
    // ---- combined = relu(cat W)   (layers.py:284-289): the K dimension of every output tile split over `kparts` waves,
    //      their partial tiles added in a fixed order ----------------------------------------------------------------
    {
        const int ksteps = K2p / 4;
        const int per = ((ksteps + kparts - 1) / kparts + MU - 1) / MU * MU;      // steps per part: a multiple of MU
        for (int item = wave; item < ntile_e * kparts; item += DENSE_WAVES) {
            const int ct = item % ntile_e, kp = item / ntile_e;
            const int k_lo = kp * per * 4, k_hi = (kp + 1) * per * 4 < K2p ? (kp + 1) * per * 4 : K2p;
            const f32x4 c = WLDS ? tile_lds_lds(s_cat, ld2, s_wi, ldW, ct * 16, k_lo, k_hi, lane)
                                 : tile_lds_glob(s_cat, ld2, a.W_inter, E, ct * 16, K2, k_lo, k_hi, lane);
            const int col = ct * 16 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) s_part[(kp * TB + rq + i) * E + col] = c[i];
        }
        __syncthreads();
        DENSE_STAMP(11);
        for (int i = tid; i < TB * E; i += DENSE_THREADS) {
            const int t = i / E, e = i - t * E;
            float acc = s_part[t * E + e];
            for (int kp = 1; kp < kparts; ++kp) acc += s_part[(kp * TB + t) * E + e];
            const float v = fmaxf(acc, 0.f);
            s_comb[t * ldE + e] = v;
            const int b = row0 + t;
            if (a.combined && b < a.B && sp == 0) a.combined[(size_t)b * E + e] = v;
        }
    }
    __syncthreads();
    DENSE_STAMP(3);
    // ---- logits, centre scores, loss gradients (model.py:38, layers.py:243, model.py:54-61): wave t has row t; its four
    //      dot products run on 16 lanes each and are added with DPP row operations; no LDS, no barrier in between ----
    {
        const int t = wave, which = lane >> 4, part = lane & 15, b = row0 + t;
        float acc = 0.f;
        if (which < 2) {
            const float *wv = s_wc + which * E;
            for (int e = part; e < E; e += 16) acc = fmaf(s_comb[t * ldE + e], wv[e], acc);
        } else {
            const float *wv = s_wc + 2 * E + (which - 2) * F;
            for (int f = part; f < F; f += 16) acc = fmaf(s_cat[t * ld2 + f], wv[f], acc);
        }
        acc = row16_sum(acc);
        const float g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 0));
        const float g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 16));
        const float c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 32)) + s_wc[2 * E + 2 * F];
        const float c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc), 48)) + s_wc[2 * E + 2 * F + 1];
        if (lane == 0) {
            float dg0 = 0.f, dg1 = 0.f, dc0 = 0.f, dc1 = 0.f;
            if (b < a.B) {
                if (sp == 0) {
                    a.logits[2 * b] = g0;
                    a.logits[2 * b + 1] = g1;
                    a.center[2 * b] = c0;
                    a.center[2 * b + 1] = c1;
                }
                if (a.labels) {
                    const int y = my_label;
                    float lg, lc;
                    xent2(g0, g1, y, lg, dg0, dg1);
                    xent2(c0, c1, y, lc, dc0, dc1);
                    if (a.row_loss && sp == 0) a.row_loss[b] = lg + a.lambda_1 * lc;
                    dg0 *= a.inv_count; dg1 *= a.inv_count;
                    dc0 *= a.inv_count * a.lambda_1; dc1 *= a.inv_count * a.lambda_1;
                }
            }
            s_dlog[2 * t] = dg0; s_dlog[2 * t + 1] = dg1;
            s_dcl[2 * t] = dc0; s_dcl[2 * t + 1] = dc1;
        }
    }
    __syncthreads();
    DENSE_STAMP(4);
    if (!train) return;

    float *slab = acts_mode ? nullptr : a.slabs + (size_t)tile_id * a.n_params;
    const bool adam_clf = a.theta != nullptr;
    // ---- backward ----------------------------------------------------------------------------------------------------
    // dcomb = (dlogits W_cls) * relu'(combined);  dW_cls, dW_clf, db_clf
    for (int i = tid; i < TB * E; i += DENSE_THREADS) {
        const int t = i / E, e = i - t * E;
        const float g = s_dlog[2 * t] * s_wc[e] + s_dlog[2 * t + 1] * s_wc[E + e];
        s_dcomb[t * ldE + e] = s_comb[t * ldE + e] > 0.f ? g : 0.f;
    }
    DENSE_STAMP(8);
    for (int i = tid; i < ((sp == 0 && !acts_mode) ? 2 * E : 0); i += DENSE_THREADS) {
        const int cidx = i / E, e = i - cidx * E;
        float sacc = 0.f;
        for (int t = 0; t < TB; ++t) sacc = fmaf(s_dlog[2 * t + cidx], s_comb[t * ldE + e], sacc);
        slab[off_cls(F, E, R) + i] = sacc;
    }
    DENSE_STAMP(9);
    // the label classifier's partial gradient, by ONE wave: write-through (sc1) stores when another workgroup of this launch
    // will read it - that wave's own vmcnt wait, a phase later, then covers every one of them.  The wave chosen has no other
    // global store in between (the last of the waves that only compute a dh_r tile in the next phase), so that wait is free.
    const int clf_wave = (R * ntile_e - 1) & (DENSE_WAVES - 1);
    if (wave == clf_wave && sp == 0 && !acts_mode) {
        for (int i = lane; i < 2 * F + 2; i += PCG_WAVE) {
            float sacc = 0.f;
            if (i < 2 * F) {
                const int cidx = i / F, f = i - cidx * F;
                for (int t = 0; t < TB; ++t) sacc = fmaf(s_dcl[2 * t + cidx], s_cat[t * ld2 + f], sacc);
            } else {
                for (int t = 0; t < TB; ++t) sacc += s_dcl[2 * t + (i - 2 * F)];
            }
            float *dst = slab + off_clf(F, E, R) + i;
            if (adam_clf) __hip_atomic_store(dst, sacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = sacc;
        }
    }
    DENSE_STAMP(10);
    __syncthreads();
    DENSE_STAMP(5);
    // one phase: dh_r = (dcomb W[F+rE.., :]^T) * relu'(h_r) for every r   and   dW_inter = cat^T dcomb
    {
        const int mt2 = (K2 + 15) / 16;
        const int n_dh = R * ntile_e, n_all = n_dh + mt2 * ntile_e;
        float *dst = slab ? slab + off_inter(F, E, R) : nullptr;
        // every workgroup of the tile needs all of dh_r; the dW_inter tiles are dealt out over the tile's S workgroups
        // (acts_mode: dh_r only - the weight gradients are a later launch's)
        for (int t0 = wave; t0 < (acts_mode ? n_dh : n_dh + (n_all - n_dh + S - 1) / S); t0 += DENSE_WAVES) {
            const int tile = t0 < n_dh ? t0 : n_dh + (t0 - n_dh) * S + sp;
            if (tile >= n_all) continue;
            if (tile < n_dh) {
                const int r = tile / ntile_e, ct = tile - r * ntile_e;
                const float *Wr = a.W_inter + (size_t)(F + r * E) * E;   // rows of W_inter that multiply h_r
                const int rr = lane & 15, kq = lane >> 4;
                const float *ap = s_dcomb + rr * ldE + kq;                 // out[t][j] = sum_e dcomb[t][e] * Wr[j][e]
                const float *bl = s_wi + (F + r * E + ct * 16 + rr) * ldW + kq;
                const float *bg = Wr + (size_t)(ct * 16 + rr) * E + kq;
                // (E is a multiple of 16.  Without the LDS copy the transposed weight rows come from L2 as float4 per lane)
                const f32x4 acc = WLDS ? mfma_chain(E >> 2, [&](int s) { return ap[4 * s]; }, [&](int s) { return bl[4 * s]; })
                                       : tile_lds_globT4(s_dcomb + rr * ldE + 4 * kq, bg + 3 * kq, E >> 4);
                const int col = ct * 16 + rr, rq = kq * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    s_dh[(r * TB + rq + i) * ldE + col] = s_cat[(rq + i) * ld2 + F + r * E + col] > 0.f ? acc[i] : 0.f;
            } else {
                const int tl = tile - n_dh;
                const int m0 = (tl / ntile_e) * 16, n0 = (tl % ntile_e) * 16;
                const f32x4 c = tile_ldsT_lds(s_cat, ld2, m0, K2, s_dcomb, ldE, n0, lane);
                const int col = n0 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (m0 + rq + i < K2) dst[(size_t)(m0 + rq + i) * E + col] = c[i];
            }
        }
    }
    __syncthreads();
    DENSE_STAMP(6);
    if (acts_mode) {
        // everything the weight gradients are made of, transposed (a batch row per column: the GEMMs over the batch then read
        // both operands contiguously - wgrad.h); rows beyond the batch in this tile are zero in every array (staged as zeros,
        // no loss gradient).  A thread stores 16-float runs (one act row of this tile: 64 bytes); nothing reads them in this launch.
        const int n_act = K2 + R * F + E + R * E + E + 4;
        float *__restrict__ out = a.acts + row0;
        for (int i = tid; i < n_act * TB; i += DENSE_THREADS) {
            const int rho = i >> 4, t = i & 15;
            int q = rho;
            float v;
            if (q < K2) v = s_cat[t * ld2 + q];
            else if ((q -= K2) < R * F) {
                const int r = q / F, f = q - r * F;
                v = s_catr[(r * TB + t) * ld1 + F + f];
            } else if ((q -= R * F) < E) v = s_dcomb[t * ldE + q];
            else if ((q -= E) < R * E) {
                const int r = q / E, e = q - r * E;
                v = s_dh[(r * TB + t) * ldE + e];
            } else if ((q -= R * E) < E) v = s_comb[t * ldE + q];
            else if ((q -= E) < 2) v = s_dlog[2 * t + q];
            else v = s_dcl[2 * t + (q - 2)];
            out[(size_t)rho * a.act_ld + t] = v;
        }
        DENSE_STAMP(7);
        return;
    }
    // arrival ticket (the workgroup's classifier gradient - its clf wave's stores, issued a phase ago - is write-through and
    // drained; and a workgroup that has arrived has long read the classifier's weights).  The answer is not needed before the
    // end of the kernel, so nobody waits for it here.
    // Only the workgroups that stored a share of the classifier's gradient (sp == 0: one per tile) arrive: the other slab
    // entries are read by the NEXT launch.  (Every workgroup arriving was 256 same-address atomics at B = 1024 - a counter
    // takes ~88 per microsecond, so the last arriver learned that it was the last ~3 us after the first one asked.)
    unsigned ticket_old = 0u;
    if (adam_clf && wave == clf_wave && sp == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) ticket_old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // dW_r = [self|agg_r]^T dh_r for every r
    {
        const int mt1 = (K1 + 15) / 16, per_r = mt1 * ntile_e;
        for (int tile = sp + S * wave; tile < R * per_r; tile += S * DENSE_WAVES) {
            const int r = tile / per_r, tl = tile - r * per_r;
            const int m0 = (tl / ntile_e) * 16, n0 = (tl % ntile_e) * 16;
            const f32x4 c = tile_ldsT_lds(s_catr + r * TB * ld1, ld1, m0, K1, s_dh + r * TB * ldE, ldE, n0, lane);
            float *dst = slab + off_intra(F, E, R, r);
            const int col = n0 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (m0 + rq + i < K1) dst[(size_t)(m0 + rq + i) * E + col] = c[i];
        }
    }
    DENSE_STAMP(7);
    if (a.stamps && threadIdx.x == 0 && sp == 0) a.stamps[(size_t)tile_id * 16 + 13] = clock64();
    // ---- the workgroup whose ticket was the last: sum of every tile's share of the classifier gradient (tile order), Adam
    //      for those 2F + 2 parameters (model_handler.py:153) - the only ones the next step's score pass reads ----------
    if (!adam_clf) return;
    if (wave == clf_wave && lane == 0) s_flag[0] = sp == 0 && ticket_old == (unsigned)a.n_tile_blocks / (unsigned)S - 1u;
    __syncthreads();
    if (s_flag[0]) {
        if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        DENSE_STAMP(14);
        const int n_tiles = a.n_tile_blocks / S;
        unsigned staged_seen = (tid == 0 && S > 1) ? __hip_atomic_load(a.staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        const int64_t oc = off_clf(F, E, R);
        const int NC = 2 * F + 2;
        // G threads per parameter: thread (g, i) adds up the tiles s = g, g + G, ... of parameter i in that order (the loads of
        // a batch of eight are all in flight: one memory round trip per batch instead of one per tile), the G partial sums
        // are added in group order - a fixed order for a given batch size, like everything else here.  s_dh is free by now.
        int G = NC <= DENSE_THREADS ? (DENSE_THREADS / NC < 16 ? DENSE_THREADS / NC : 16) : 1;
        const int room = (R * TB * ldE) / NC;                  // what s_dh can hold
        G = G < room ? G : room;
        G = G < 1 ? 1 : G;
        float *s_red = s_dh;                                   // [G][NC] (only when G > 1)
        for (int i0 = 0; i0 < NC; i0 += DENSE_THREADS) {       // (one pass unless there are more parameters than threads)
            const int g = G > 1 ? tid / NC : 0, i = i0 + (G > 1 ? tid - g * NC : tid);
            float acc = 0.f;
            if (g < G && i < NC) {
                const float *src = a.slabs + oc + i;
                // (every batch of eight loads is issued whole - indices clamped, the surplus not added: at 64 tiles and fifteen
                //  threads per parameter a thread has five tiles, and a loop of single agent-scope loads waited for each of them in
                //  turn: five memory round trips on the step's critical path instead of one)
                for (int s2 = g; s2 < n_tiles; s2 += 8 * G) {
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int t2 = s2 + u * G;
                        x[u] = __hip_atomic_load(src + (size_t)(t2 < n_tiles ? t2 : n_tiles - 1) * a.n_params, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = (s2 + u * G < n_tiles) ? acc + x[u] : acc;
                }
            }
            if (G > 1) {
                if (g < G && i < NC) s_red[g * NC + i] = acc;
                __syncthreads();
                if (tid < NC) {
                    acc = s_red[tid];
                    for (int gg = 1; gg < G; ++gg) acc += s_red[gg * NC + tid];
                }
            }
            const int ip = G > 1 ? tid : i;
            if (i0 == 0 && S > 1) {
                // every workgroup without a ticket has long staged the old classifier (it said so ~10 us ago); the count was
                // requested before the gradient's loads, so the check costs nothing in all but pathological schedules; bounded
                if (tid == 0)
                    for (int spins = 0; staged_seen < (unsigned)a.n_tile_blocks - (unsigned)n_tiles && spins < (1 << 20); ++spins) {
                        __builtin_amdgcn_s_sleep(4);
                        staged_seen = __hip_atomic_load(a.staged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                __syncthreads();
            }
            if (ip < NC) {
                // t: the step this launch counted (block 0 incremented the counter at its start; read it past the L1)
                const float t = (float)__hip_atomic_load(a.step_counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                adam_apply_one(a.theta, a.m, a.v, oc + ip, acc, t, a.h);
            }
        }
        if (tid == 0) {
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.staged, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        DENSE_STAMP(15);
    }
}

__global__ void __launch_bounds__(256) adam_reduce_kernel(float *__restrict__ theta, float *__restrict__ m,
                                                          float *__restrict__ v, const float *__restrict__ slabs,
                                                          int n_slabs, int64_t n_params, int64_t p_begin, int64_t p_end,
                                                          const int32_t *__restrict__ step_counter, AdamHyper h,
                                                          float *__restrict__ grad_out, int apply, const uint32_t *pending) {
    __shared__ float part[4][PCG_WAVE];
    if (pending) {                                    // the deferred update: nothing to do unless a gradient is waiting in the slabs
        if (pending[0] != 1u) return;                 // (wave-uniform: one word; 2 = it waits in `acts`: wgrad_adam_kernel's)
        n_slabs = (int)pending[1];
    }
    adam_reduce_body(theta, m, v, slabs, n_slabs, n_params, p_begin, p_end, step_counter, h, grad_out, apply, (int)blockIdx.x, part);
}

// the weight gradients from the dense kernel's transposed activations + Adam (wgrad.h), as a launch of its own
__global__ void __launch_bounds__(256) wgrad_adam_kernel(const WgradArgs a) {
    __shared__ float red[4][256];
    wgrad_adam_body(a, (int)blockIdx.x, red);
}

__global__ void clear_word_kernel(uint32_t *w) { w[0] = 0u; }
// dst[i] = src[i] if an update is pending (pcg_adam_flush: the stepped label classifier -> the parameter buffer)
__global__ void __launch_bounds__(256) copy_if_pending_kernel(float *__restrict__ dst, const float *__restrict__ src, int n,
                                                              const uint32_t *__restrict__ pending) {
    if (pending[0] == 0u) return;
    for (int i = (int)threadIdx.x; i < n; i += (int)blockDim.x) dst[i] = src[i];
}

// The partitioned path's two optimizer launches (its gradient goes through an all-reduce between them):
//   grad_reduce   : slabs summed in tile order -> grad; marks "a gradient is waiting" (flag[0] = 1)
//   apply_pending : if a gradient is waiting: Adam on every parameter from grad (after the all-reduce, at the head of the NEXT step's
//                   graph - no launch of its own between the collective and the next score pass).  The flag is cleared by a LATER
//                   launch (the step's select kernel: pcg_choose_select_planned(sync_words) clears sync_words[1]; or the
//                   one-thread launch pcg_adam_apply_pending(clear = 1) adds) - a departure ticket in this kernel was 420
//                   same-address atomics, 4.8 us
__global__ void __launch_bounds__(256) grad_reduce_kernel(const float *__restrict__ slabs, int n_slabs, int64_t n_params,
                                                          float *__restrict__ grad_out, uint32_t *flag) {
    __shared__ float part[4][PCG_WAVE];
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = 1u;
    const AdamHyper none = {0.f, 0.f, 0.f, 0.f, 0.f};
    adam_reduce_body(nullptr, nullptr, nullptr, slabs, n_slabs, n_params, 0, n_params, nullptr, none, grad_out, 0, (int)blockIdx.x, part);
}
__global__ void __launch_bounds__(256) apply_pending_kernel(float *__restrict__ theta, float *__restrict__ m, float *__restrict__ v,
                                                            const float *__restrict__ grad, int64_t n_params,
                                                            const int32_t *__restrict__ step_counter, AdamHyper h,
                                                            const uint32_t *__restrict__ flag) {
    __shared__ float part[4][PCG_WAVE];
    if (flag[0] == 0u) return;                                    // (one word, the same for every thread)
    adam_reduce_body(theta, m, v, grad, 1, n_params, 0, n_params, step_counter, h, nullptr, 1, (int)blockIdx.x, part);
}

static size_t dense_smem_bytes(int F, int E, int R, bool wlds) {
    const int K1p = (2 * F + KPAD - 1) / KPAD * KPAD, K2p = (F + R * E + KPAD - 1) / KPAD * KPAD;
    const int ntile_e = E / 16, kparts = ntile_e <= DENSE_WAVES ? DENSE_WAVES / ntile_e : 1;
    size_t fl = (size_t)(R * TB * (K1p + 1) + TB * (K2p + 1) + (2 + R) * TB * (E + 1) + 4 * TB + 2 * E + 2 * F + 4 + 4);
    if (wlds) fl += (size_t)(K2p + R * K1p) * (E + 4);
    else fl += (size_t)kparts * TB * E;
    return sizeof(float) * fl;
}

static bool dense_wlds(int F, int E, int R) {
    const int K1p = (2 * F + KPAD - 1) / KPAD * KPAD;
    const int ntile_e = E / 16, kparts = ntile_e <= DENSE_WAVES ? DENSE_WAVES / ntile_e : 1;
    // the K-split partial tiles of `combined` live where the W_intra copies were
    return dense_smem_bytes(F, E, R, true) <= 160 * 1024 && DENSE_THREADS % (E / 4) == 0 && (size_t)kparts * TB * E <= (size_t)R * K1p * (E + 4);
}

static unsigned long long *g_dense_stamps = nullptr;

}  // namespace pcg
/* 1: pcg_train_dense(adam_clf = 3, sort_keys) for a batch of B rows also sorts the next step's train-pos keys (n_pos of them: the
 * rank sort's sizes) - when its tiles' workgroups and the sort's together leave no CU with two of them (up to ~3000 rows): the
 * sort is then free.  (PCG_PRESORT_MAX_TILES=n also allows batches of up to n tiles, the sorting workgroups running behind the
 * tiles'.  Measured at 256 tiles: power-law 2 M / 8000 keys - where the in-kernel sort publishes 13 us into the select launch and
 * every positive row waits for it - the call 123.3 -> 117.3 us but the step 136.5 -> 142.8: 125 rank-sorting workgroups behind the
 * tiles cost the dense launch more than the select launch gains; emb 128 / 2670 keys: 94.1 -> 93.5.  Off.)  Host helper. */
extern "C" int32_t pcg_dense_sorts_keys(int32_t B, int32_t n_pos) {
    if (B < 1 || n_pos < 1 || n_pos > pcg::RANK_MAX) return 0;
    static int max_tiles = -1;
    if (max_tiles < 0) {
        const char *e = getenv("PCG_PRESORT_MAX_TILES");
        max_tiles = e ? atoi(e) : 0;
    }
    const int tiles = (B + pcg::TB - 1) / pcg::TB;
    if (tiles + (n_pos + PCG_WAVE - 1) / PCG_WAVE <= 256) return 1;
    return tiles <= max_tiles ? 1 : 0;
}
namespace pcg {

struct DenseExtra {          // the optional parts of a launch
    const int32_t *chunk_begin = nullptr;
    const float *partial = nullptr;
    const int32_t *cnt = nullptr;
    int32_t partial_stride = 0;
    float *theta_rw = nullptr, *m = nullptr, *v = nullptr;
    uint32_t *ticket = nullptr, *pending = nullptr, *staged = nullptr;
    float *acts = nullptr;
    int32_t act_ld = 0;
    uint64_t *sort_keys = nullptr;          // the riding sort of the next step's train-pos keys (pcg_train_dense)
    AdamHyper h = {0.f, 0.f, 0.f, 0.f, 0.f};
};

static int launch_dense(const pcg_graph_desc *g, const float *theta, int32_t emb, const int32_t *ids, const int32_t *labels,
                        int32_t B, const float *agg, int32_t agg_stride, float lambda_1, float inv_count, float *logits,
                        float *center, float *combined, float *row_loss, float *slabs, int32_t *step_counter,
                        const DenseExtra &x, void *stream) {
    if (!g || !g->X || !theta || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!ids || !agg || !logits || !center) return PCG_E_ARG;
    if (emb < 16 || emb % 16 != 0 || g->n_rel < 1 || g->n_rel > PCG_MAX_REL) return PCG_E_UNSUPPORTED;
    if ((slabs || x.acts) && !labels) return PCG_E_ARG;
    if (x.acts && (x.act_ld < (B + TB - 1) / TB * TB || x.act_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(x.acts) & 15u) != 0)) return PCG_E_ARG;
    if (x.theta_rw && (!slabs || !x.m || !x.v || !x.ticket || !step_counter)) return PCG_E_ARG;
    if (x.chunk_begin && (!x.partial || !x.cnt)) return PCG_E_ARG;
    const int F = g->feat_dim, E = emb, R = g->n_rel;
    const bool wlds = dense_wlds(F, E, R);
    const size_t smem = dense_smem_bytes(F, E, R, wlds);
    if (smem > 160 * 1024) return PCG_E_UNSUPPORTED;
    DenseArgs a;
    a.X = g->X;
    a.feat_dim = F;
    a.feat_stride = g->feat_stride;
    a.n_rel = R;
    a.emb = E;
    a.ids = ids;
    a.labels = labels;
    a.B = B;
    a.agg = agg;
    a.agg_stride = agg_stride;
    a.chunk_begin = x.chunk_begin;
    a.partial = x.partial;
    a.cnt = x.cnt;
    a.partial_stride = x.partial_stride;
    a.W_cls = theta + off_cls(F, E, R);
    a.W_inter = theta + off_inter(F, E, R);
    for (int r = 0; r < PCG_MAX_REL; ++r) a.W_intra[r] = r < R ? theta + off_intra(F, E, R, r) : nullptr;
    a.W_clf = theta + off_clf(F, E, R);
    a.b_clf = theta + off_bias(F, E, R);
    a.lambda_1 = lambda_1;
    a.inv_count = inv_count;
    a.logits = logits;
    a.center = center;
    a.combined = combined;
    a.row_loss = row_loss;
    a.slabs = x.acts ? nullptr : slabs;
    a.acts = x.acts;
    a.act_ld = x.act_ld;
    a.n_params = n_params_of(F, E, R);
    a.step_counter = step_counter;
    a.theta = x.theta_rw;
    a.m = x.m;
    a.v = x.v;
    a.ticket = x.ticket;
    a.staged = x.staged;
    a.pending = x.pending;
    a.h = x.h;
    // few tiles (small batches): up to 4 workgroups per tile, so that the weight-gradient tiles of a 16-row tile are not one
    // CU's serial work while most of the chip idles
    const int n_tiles = (B + TB - 1) / TB;
    // (acts instead of slabs: no weight-gradient tiles to deal out - one workgroup per tile)
    int n_split = (slabs && !x.acts) ? 256 / n_tiles : 1;
    a.n_split = n_split < 1 ? 1 : (n_split > 4 ? 4 : n_split);
    a.stamps = g_dense_stamps;
    a.sort_raw = nullptr;
    a.sort_out = nullptr;
    a.sort_n = a.sort_cap = 0;
    a.n_tile_blocks = n_tiles * a.n_split;
    int n_sort_blocks = 0;
    if (x.sort_keys && pcg_dense_sorts_keys(B, g->n_pos)) {
        const int64_t cap = pcg_pos_sort_capacity(g->n_pos) / 2;
        a.sort_out = x.sort_keys;
        a.sort_raw = x.sort_keys + cap;
        a.sort_n = g->n_pos;
        a.sort_cap = (int32_t)cap;
        n_sort_blocks = (g->n_pos + PCG_WAVE - 1) / PCG_WAVE;
    }
    // the instantiated shapes: YelpChi (F 32) and Amazon (F 25) at emb 64 and 128, three relations; anything else: run-time shape
    typedef void (*kern_t)(const DenseArgs);
    kern_t kern;
    if (R == 3 && F == 32 && E == 64 && wlds) kern = dense_step_kernel<true, 32, 64, 3>;
    else if (R == 3 && F == 25 && E == 64 && wlds) kern = dense_step_kernel<true, 25, 64, 3>;
    else if (R == 3 && F == 32 && E == 128 && !wlds) kern = dense_step_kernel<false, 32, 128, 3>;
    else if (R == 3 && F == 25 && E == 128 && !wlds) kern = dense_step_kernel<false, 25, 128, 3>;
    else kern = wlds ? dense_step_kernel<true, 0, 0, 0> : dense_step_kernel<false, 0, 0, 0>;
    static kern_t attr_done[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool seen = false;
    for (kern_t k : attr_done) seen = seen || k == kern;
    if (!seen) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
            hipSuccess)
            return PCG_E_LAUNCH;
        for (kern_t &k : attr_done)
            if (!k) {
                k = kern;
                break;
            }
    }
    const dim3 grid(n_tiles * a.n_split + n_sort_blocks), block(DENSE_THREADS);
    const size_t sort_smem = n_sort_blocks ? sizeof(uint64_t) * DENSE_SORT_TILE + sizeof(int) * DENSE_THREADS : 0;
    hipLaunchKernelGGL(kern, grid, block, smem > sort_smem ? smem : sort_smem, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // namespace pcg

extern "C" {

void pcg_debug_set_dense_stamps(void *ptr) { pcg::g_dense_stamps = static_cast<unsigned long long *>(ptr); }

int64_t pcg_dense_n_params(int32_t feat_dim, int32_t emb, int32_t n_rel) {
    if (feat_dim < 1 || emb < 1 || n_rel < 1 || n_rel > PCG_MAX_REL) return PCG_E_ARG;
    return pcg::n_params_of(feat_dim, emb, n_rel);
}

int64_t pcg_dense_param_offset(int32_t feat_dim, int32_t emb, int32_t n_rel, int32_t which, int32_t rel) {
    switch (which) {
        case 0: return pcg::off_cls(feat_dim, emb, n_rel);
        case 1: return pcg::off_inter(feat_dim, emb, n_rel);
        case 2: return pcg::off_intra(feat_dim, emb, n_rel, rel);
        case 3: return pcg::off_clf(feat_dim, emb, n_rel);
        case 4: return pcg::off_bias(feat_dim, emb, n_rel);
        default: return PCG_E_ARG;
    }
}

int32_t pcg_dense_n_tiles(int32_t B) { return B < 0 ? PCG_E_ARG : (B + pcg::TB - 1) / pcg::TB; }

int pcg_dense_step(const pcg_graph_desc *g, const float *theta, int32_t emb, const int32_t *ids, const int32_t *labels,
                   int32_t B, const float *agg, int32_t agg_stride, float lambda_1, float inv_count, float *logits,
                   float *center, float *combined, float *row_loss, float *slabs, int32_t *step_counter, void *stream) {
    return pcg::launch_dense(g, theta, emb, ids, labels, B, agg, agg_stride, lambda_1, inv_count, logits, center, combined,
                             row_loss, slabs, step_counter, pcg::DenseExtra(), stream);
}

int pcg_train_dense(const pcg_graph_desc *g, float *theta, float *m, float *v, int32_t emb, const int32_t *ids,
                    const int32_t *labels, int32_t B, const float *agg, int32_t agg_stride, const int32_t *cnt,
                    const void *workspace, const void *plan, int64_t list_capacity, float lambda_1, float inv_count, float *logits, float *center,
                    float *combined, float *row_loss, float *slabs, int32_t *step_counter, uint32_t *sync_words, double lr,
                    double beta1, double beta2, double eps, double weight_decay, int32_t adam_clf, float *acts, int32_t act_ld,
                    uint64_t *sort_keys, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    pcg::DenseExtra x;
    if (workspace) {
        if (!cnt || list_capacity < 1) return PCG_E_ARG;
        pcg::Workspace w;
        pcg::carve1(g, B, list_capacity, static_cast<unsigned char *>(const_cast<void *>(workspace)), &w,
                    static_cast<unsigned char *>(const_cast<void *>(plan)));
        x.chunk_begin = w.chunk_begin;
        x.partial = w.partial;
        x.cnt = cnt;
        x.partial_stride = g->feat_stride;
    }
    if (adam_clf == 3) {                     // no slabs: transposed activations for the weight-gradient GEMMs of a later launch
        if (!acts || !sync_words) return PCG_E_ARG;
        x.pending = sync_words + 1;
        x.acts = acts;
        x.act_ld = act_ld;
        x.sort_keys = sort_keys;             // (only this mode has one workgroup per tile: CUs to spare for the riding sort)
    } else if (adam_clf == 4) {              // the same without marking anything as waiting (pcg_wgrad follows: gradients only)
        if (!acts) return PCG_E_ARG;
        x.acts = acts;
        x.act_ld = act_ld;
    } else if (adam_clf == 2) {              // the label classifier is stepped elsewhere (pcg_choose_gather_train): slabs + "pending" only
        if (!slabs || !sync_words) return PCG_E_ARG;
        x.pending = sync_words + 1;
    } else if (adam_clf) {
        if (!slabs || !m || !v || !sync_words) return PCG_E_ARG;
        x.theta_rw = theta;
        x.m = m;
        x.v = v;
        x.ticket = sync_words;
        x.staged = sync_words + pcg_sync_words_count() - 4;
        x.pending = sync_words + 1;
        x.h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    }
    return pcg::launch_dense(g, theta, emb, ids, labels, B, agg, agg_stride, lambda_1, inv_count, logits, center, combined,
                             row_loss, slabs, step_counter, x, stream);
}

int pcg_adam_step(float *theta, float *m, float *v, const float *slabs, int32_t n_slabs, int64_t n_params,
                  const int32_t *step_counter, double lr, double beta1, double beta2, double eps, double weight_decay,
                  float *grad_out, int32_t apply, void *stream) {
    if (!slabs || n_slabs < 0 || n_params < 1) return PCG_E_ARG;
    if (apply && (!theta || !m || !v || !step_counter)) return PCG_E_ARG;
    if (!apply && !grad_out) return PCG_E_ARG;
    const pcg::AdamHyper h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    hipLaunchKernelGGL(pcg::adam_reduce_kernel, dim3((unsigned)((n_params + PCG_WAVE - 1) / PCG_WAVE)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), theta, m, v, slabs, n_slabs, n_params, (int64_t)0, n_params,
                       step_counter, h, grad_out, apply, (const uint32_t *)nullptr);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_grad_reduce(const float *slabs, int32_t n_slabs, int64_t n_params, float *grad_out, uint32_t *flag, void *stream) {
    if (!slabs || n_slabs < 0 || n_params < 1 || !grad_out || !flag) return PCG_E_ARG;
    hipLaunchKernelGGL(pcg::grad_reduce_kernel, dim3((unsigned)((n_params + PCG_WAVE - 1) / PCG_WAVE)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), slabs, n_slabs, n_params, grad_out, flag);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_adam_apply_pending(float *theta, float *m, float *v, const float *grad, int64_t n_params, const int32_t *step_counter,
                           uint32_t *flag, int32_t clear, double lr, double beta1, double beta2, double eps, double weight_decay,
                           void *stream) {
    if (!theta || !m || !v || !grad || n_params < 1 || !step_counter || !flag) return PCG_E_ARG;
    const pcg::AdamHyper h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pcg::apply_pending_kernel, dim3((unsigned)((n_params + PCG_WAVE - 1) / PCG_WAVE)), dim3(256), 0, st, theta, m, v,
                       grad, n_params, step_counter, h, flag);
    PCG_LAUNCH_CHECK();
    if (clear) {
        hipLaunchKernelGGL(pcg::clear_word_kernel, dim3(1), dim3(1), 0, st, flag);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

static int wgrad_args(pcg::WgradArgs &w, const float *acts, int32_t act_ld, int32_t feat_dim, int32_t emb, int32_t n_rel,
                      int32_t n_kblocks, float *scratch) {
    if (!acts || act_ld < 16 || act_ld % 16 != 0 || (reinterpret_cast<uintptr_t>(acts) & 15u) != 0) return PCG_E_ARG;
    if (feat_dim < 1 || emb < 16 || emb % 16 != 0 || n_rel < 1 || n_rel > PCG_MAX_REL) return PCG_E_UNSUPPORTED;
    if (n_kblocks < 1 || n_kblocks * 16 > act_ld) return PCG_E_ARG;
    w.acts = acts;
    w.ld = act_ld;
    w.F = feat_dim;
    w.E = emb;
    w.R = n_rel;
    w.n_kblocks = n_kblocks;
    w.kparts = pcg::wgrad_kparts(n_kblocks);
    w.tickets = nullptr;
    w.partials = nullptr;
    if (w.kparts > 1) {
        if (!scratch) return PCG_E_ARG;
        const int t = pcg::wgrad_tiles(feat_dim, emb, n_rel, 1);
        w.tickets = reinterpret_cast<uint32_t *>(scratch);
        w.partials = scratch + (t + 63) / 64 * 64;
    }
    return PCG_OK;
}

int64_t pcg_wgrad_act_rows(int32_t feat_dim, int32_t emb, int32_t n_rel) {
    if (feat_dim < 1 || emb < 1 || n_rel < 1 || n_rel > PCG_MAX_REL) return PCG_E_ARG;
    return pcg::wgrad_act_rows(feat_dim, emb, n_rel);
}

int64_t pcg_wgrad_scratch_bytes(int32_t feat_dim, int32_t emb, int32_t n_rel, int32_t B) {
    if (feat_dim < 1 || emb < 16 || emb % 16 != 0 || n_rel < 1 || n_rel > PCG_MAX_REL || B < 1) return PCG_E_ARG;
    return 4 * pcg::wgrad_scratch_floats(feat_dim, emb, n_rel, (B + 15) / 16);
}

int pcg_wgrad(const float *acts, int32_t act_ld, int32_t B, int32_t feat_dim, int32_t emb, int32_t n_rel, float *theta, float *m,
              float *v, const int32_t *step_counter, double lr, double beta1, double beta2, double eps, double weight_decay,
              float *grad_out, int32_t apply, int32_t with_clf, float *scratch, uint32_t *flag_set, void *stream) {
    pcg::WgradArgs w;
    if (B < 1) return PCG_E_ARG;
    const int rc = wgrad_args(w, acts, act_ld, feat_dim, emb, n_rel, (B + 15) / 16, scratch);
    if (rc != PCG_OK) return rc;
    if (apply && (!theta || !m || !v || !step_counter)) return PCG_E_ARG;
    if (!apply && !grad_out) return PCG_E_ARG;
    w.theta = theta; w.m = m; w.v = v;
    w.step_counter = step_counter;
    w.h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    w.pending = nullptr;
    w.grad_out = grad_out;
    w.flag_set = flag_set;
    w.apply = apply;
    w.with_clf = with_clf ? 1 : 0;
    hipLaunchKernelGGL(pcg::wgrad_adam_kernel, dim3((unsigned)(pcg::wgrad_tiles(feat_dim, emb, n_rel, w.with_clf) * w.kparts)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_adam_flush(float *theta, float *m, float *v, const float *slabs, int32_t n_slabs, int64_t n_params, int64_t p_end,
                   const int32_t *step_counter, uint32_t *sync_words, double lr, double beta1, double beta2, double eps,
                   double weight_decay, const float *clf_next, const float *acts, int32_t act_ld, int32_t feat_dim, int32_t emb,
                   int32_t n_rel, float *wg_scratch, void *stream) {
    if (!theta || !m || !v || (!slabs && !acts) || !step_counter || !sync_words || n_slabs < 0 || n_params < 1 || p_end < 0 ||
        p_end > n_params)
        return PCG_E_ARG;
    const pcg::AdamHyper h = {(float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay};
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (acts) {                              // a step of pcg_train_dense(adam_clf = 3) waiting (sync_words[1] == 2): weight-gradient GEMMs + Adam
        pcg::WgradArgs w;
        const int rc = wgrad_args(w, acts, act_ld, feat_dim, emb, n_rel, act_ld / 16, wg_scratch);
        if (rc != PCG_OK) return rc;
        if (pcg::n_params_of(feat_dim, emb, n_rel) != n_params) return PCG_E_ARG;
        w.theta = theta; w.m = m; w.v = v;
        w.step_counter = step_counter;
        w.h = h;
        w.pending = sync_words + 1;
        w.grad_out = nullptr;
        w.flag_set = nullptr;
        w.apply = 1;
        w.with_clf = p_end > pcg::off_clf(feat_dim, emb, n_rel) ? 1 : 0;
        hipLaunchKernelGGL(pcg::wgrad_adam_kernel, dim3((unsigned)(pcg::wgrad_tiles(feat_dim, emb, n_rel, w.with_clf) * w.kparts)), dim3(256), 0, st, w);
        PCG_LAUNCH_CHECK();
    }
    if (p_end > 0 && slabs) {
        hipLaunchKernelGGL(pcg::adam_reduce_kernel, dim3((unsigned)((p_end + PCG_WAVE - 1) / PCG_WAVE)), dim3(256), 0, st, theta, m,
                           v, slabs, n_slabs, n_params, (int64_t)0, p_end, step_counter, h, (float *)nullptr, 1,
                           (const uint32_t *)(sync_words + 1));
        PCG_LAUNCH_CHECK();
    }
    if (clf_next && p_end < n_params) {      // (pcg_choose_gather_train keeps the stepped label classifier outside theta until now)
        hipLaunchKernelGGL(pcg::copy_if_pending_kernel, dim3(1), dim3(256), 0, st, theta + p_end, clf_next, (int)(n_params - p_end),
                           (const uint32_t *)(sync_words + 1));
        PCG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(pcg::clear_word_kernel, dim3(1), dim3(1), 0, st, sync_words + 1);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
