// Dense tail of the PC-GNN step for gfx950: relation GEMMs, inter GEMM, classifier,
// the two cross-entropy terms, their backward and Adam - two launches per step.
//
//   dense_step : one 256-thread workgroup per tile of 16 batch rows does the whole
//                forward for its rows, the loss gradients, and this tile's partial
//                weight gradients (split-K over the batch: one partial "slab" per
//                tile, no atomics => bitwise reproducible).  GEMMs run on the f32
//                matrix cores (v_mfma_f32_16x16x4_f32: exact fmaf chains).
//   adam_reduce: sums the slabs in tile order and applies torch.optim.Adam's update
//                (coupled L2 weight decay) to the flat parameter buffer.
//
// Reference lines replaced: src/layers.py:273-289, 625-629; src/model.py:34-62;
// src/model_handler.py:124,149-153 (optimizer.zero_grad / loss.backward / optimizer.step).
// Gradients never flow into the gathered features or the selection (features frozen,
// model_handler.py:86; selection is index-only), so backward is dense GEMMs only.
#include "common.h"

namespace pcg {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TB = 16;          // batch rows per workgroup (one MFMA M-tile)
constexpr int DENSE_WAVES = 16;

struct DenseArgs {
    const float *X;
    int32_t feat_dim, feat_stride, n_rel, emb;
    const int32_t *ids;
    const int32_t *labels;      // null => inference (no loss, no gradients)
    int32_t B;
    const float *agg;           // [R, B, agg_stride]
    int32_t agg_stride;
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
    int64_t n_params;
    int32_t *step_counter;      // incremented once per training launch (Adam's t), or null
    int32_t n_split;            // training: workgroups per 16-row tile; they all run the forward pass, the weight-gradient tiles are dealt out
    unsigned long long *stamps; // diagnostic only (pcg_debug_set_dense_stamps): [tiles][16] wall-clock ticks, else null
};
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

// C[16x16] += A[16 x K] (LDS, row-major, leading dim lda, zero-padded to Kp) * B[K x ..] (global, ld ldb, column n0..n0+15)
__device__ __forceinline__ f32x4 tile_lds_glob(const float *A, int lda, const float *__restrict__ Bg, int ldb, int n0,
                                               int K, int Kp, int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll 4
    for (int k0 = 0; k0 < Kp; k0 += 4) {
        const int k = k0 + kq;
        const float a = A[r * lda + k];
        const float b = (k < K) ? Bg[(size_t)k * ldb + n0 + r] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// C[16x16] += A[16 x K] (LDS) * B[K x ..] (LDS copy of a weight matrix, leading dim ldb, column n0..n0+15)
__device__ __forceinline__ f32x4 tile_lds_lds(const float *A, int lda, const float *Bl, int ldb, int n0, int Kp,
                                              int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
#pragma unroll 8
    for (int k0 = 0; k0 < Kp; k0 += 4) {
        const int k = k0 + kq;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[r * lda + k], Bl[k * ldb + n0 + r], acc, 0, 0, 0);
    }
    return acc;
}

// cooperative copy of a row-major [rows x cols] weight matrix into LDS with leading dim ld; rows..rows_pad-1 zeroed
__device__ __forceinline__ void stage_weights(float *dst, int ld, const float *__restrict__ src, int rows, int rows_pad,
                                              int cols) {
    const int c4 = cols >> 2;                       // cols % 4 == 0 (E % 16 == 0)
    for (int i = threadIdx.x; i < rows * c4; i += blockDim.x) {
        const int rr = i / c4, cc = (i - rr * c4) * 4;
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)rr * cols + cc);
        float *d = dst + rr * ld + cc;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    for (int i = threadIdx.x; i < (rows_pad - rows) * cols; i += blockDim.x) {
        const int rr = rows + i / cols, cc = i % cols;
        dst[rr * ld + cc] = 0.f;
    }
}

// C[16x16] = At^T * Bt with both operands row tiles in LDS: C[m][n] = sum_t At[t][m0+m] * Bt[t][n0+n], t < 16
__device__ __forceinline__ f32x4 tile_ldsT_lds(const float *At, int lda, int m0, int M, const float *Bt, int ldb, int n0,
                                               int lane) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int r = lane & 15, kq = lane >> 4;
    const bool mok = m0 + r < M;
#pragma unroll
    for (int t0 = 0; t0 < TB; t0 += 4) {
        const int t = t0 + kq;
        const float a = mok ? At[t * lda + m0 + r] : 0.f;
        const float b = Bt[t * ldb + n0 + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

__device__ __forceinline__ void xent2(float a, float b, int y, float &loss, float &da, float &db) {
    const float mx = fmaxf(a, b);
    const float ea = expf(a - mx), eb = expf(b - mx);
    const float s = ea + eb;
    const float lse = mx + logf(s);
    loss = lse - (y == 1 ? b : a);
    da = ea / s - (y == 0 ? 1.f : 0.f);
    db = eb / s - (y == 1 ? 1.f : 0.f);
}

// WLDS: the weight matrices are staged in LDS once per workgroup (when they fit), so every MFMA operand
// is an LDS read; otherwise the B operands stream from global memory / L2.
// Phases (one barrier between them): stage -> h_r for all relations -> combined -> logits ->
// loss grads -> dcomb + small dW -> {dh_r for all r, dW_inter} -> dW_r for all r.
template <bool WLDS>
__global__ void __launch_bounds__(DENSE_WAVES *PCG_WAVE) dense_step_kernel(const DenseArgs a) {
    extern __shared__ __align__(16) float sm[];
    const int F = a.feat_dim, E = a.emb, R = a.n_rel;
    const int K1 = 2 * F, K1p = (K1 + 3) & ~3, K2 = F + R * E, K2p = (K2 + 3) & ~3;
    const int ld1 = K1p + 1, ld2 = K2p + 1, ldE = E + 1, ldW = E + 4;   // ldW: rows stay 16-B aligned (ds_write_b128)
    float *s_wi = sm;                               // WLDS: [K2p][ldW] copy of W_inter   (first: 16-B aligned)
    float *s_wr = s_wi + (WLDS ? K2p * ldW : 0);    // WLDS: [R][K1p][ldW] copies of W_intra
    float *s_catr = s_wr + (WLDS ? R * K1p * ldW : 0);   // [R][TB][ld1]  [self | agg_r]
    float *s_cat = s_catr + R * TB * ld1;           // [TB][ld2]  [self | h_1 .. h_R]
    float *s_comb = s_cat + TB * ld2;               // [TB][ldE]
    float *s_dcomb = s_comb + TB * ldE;             // [TB][ldE]
    float *s_dh = s_dcomb + TB * ldE;               // [R][TB][ldE]
    float *s_dlog = s_dh + R * TB * ldE;            // [TB][2] d loss / d gnn logits
    float *s_dcl = s_dlog + TB * 2;                 // [TB][2] d loss / d centre scores (already times lambda_1)
    float *s_wc = s_dcl + TB * 2;                   // [2][E] W_cls, [2][F] W_clf, [2] b_clf
    float *s_tmp = s_wc + 2 * E + 2 * F + 4;        // [TB][4] logits scratch

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = a.n_split, tile_id = (int)blockIdx.x / S, sp = (int)blockIdx.x % S;
    const int row0 = tile_id * TB;
    const bool train = a.slabs != nullptr;
    if (train && a.step_counter && blockIdx.x == 0 && tid == 0) a.step_counter[0] += 1;
    DENSE_STAMP(0);

    // ---- stage: weights (16-B global loads -> 16-B LDS stores), classifier weights, zeroed tiles, self rows
    if constexpr (WLDS) {
        // thread -> (row, 16-B column chunk) fixed once: no per-element division, 4 rows in flight
        const int c4 = E >> 2;
        const int cc = (tid % c4) * 4, r0 = tid / c4, rstep = blockDim.x / c4;   // blockDim.x % c4 == 0 (host-checked)
        auto stage = [&](float *dst, const float *__restrict__ src, int rows, int rows_pad) {
            int rr = r0;
            for (; rr + 3 * rstep < rows; rr += 4 * rstep) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4 *>(src + (size_t)(rr + u * rstep) * E + cc);
#pragma unroll
                for (int u = 0; u < 4; ++u) *reinterpret_cast<float4 *>(dst + (rr + u * rstep) * ldW + cc) = v[u];
            }
            for (; rr < rows; rr += rstep)
                *reinterpret_cast<float4 *>(dst + rr * ldW + cc) = *reinterpret_cast<const float4 *>(src + (size_t)rr * E + cc);
            for (int i = tid; i < (rows_pad - rows) * E; i += blockDim.x) dst[(rows + i / E) * ldW + i % E] = 0.f;
        };
        stage(s_wi, a.W_inter, K2, K2p);
        for (int r = 0; r < R; ++r) stage(s_wr + r * K1p * ldW, a.W_intra[r], K1, K1p);
    }
    for (int i = tid; i < 2 * E; i += blockDim.x) s_wc[i] = a.W_cls[i];
    for (int i = tid; i < 2 * F; i += blockDim.x) s_wc[2 * E + i] = a.W_clf[i];
    if (tid < 2) s_wc[2 * E + 2 * F + tid] = a.b_clf[tid];
    for (int i = tid; i < TB * ld2; i += blockDim.x) s_cat[i] = 0.f;
    for (int i = tid; i < R * TB * ld1; i += blockDim.x) s_catr[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < TB * F; i += blockDim.x) {
        const int t = i / F, f = i - t * F;
        const int b = row0 + t;
        if (b < a.B) {
            const float v = a.X[(size_t)a.ids[b] * a.feat_stride + f];
            s_cat[t * ld2 + f] = v;
            for (int r = 0; r < R; ++r) s_catr[(r * TB + t) * ld1 + f] = v;
        }
    }
    for (int i = tid; i < R * TB * F; i += blockDim.x) {
        const int r = i / (TB * F), j = i - r * TB * F, t = j / F, f = j - t * F;
        const int b = row0 + t;
        if (b < a.B) s_catr[(r * TB + t) * ld1 + F + f] = a.agg[((size_t)r * a.B + b) * a.agg_stride + f];
    }
    __syncthreads();
    DENSE_STAMP(1);

    // ---- forward: h_r = relu([self | agg_r] W_r) for every relation   (layers.py:625-629) ---------
    const int ntile_e = E / 16;
    for (int tile = wave; tile < R * ntile_e; tile += DENSE_WAVES) {
        const int r = tile / ntile_e, ct = tile - r * ntile_e;
        const float *A = s_catr + r * TB * ld1;
        const f32x4 c = WLDS ? tile_lds_lds(A, ld1, s_wr + r * K1p * ldW, ldW, ct * 16, K1p, lane)
                             : tile_lds_glob(A, ld1, a.W_intra[r], E, ct * 16, K1, K1p, lane);
        const int col = ct * 16 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) s_cat[(rq + i) * ld2 + F + r * E + col] = fmaxf(c[i], 0.f);
    }
    __syncthreads();
    DENSE_STAMP(2);
    // ---- combined = relu(cat W)   (layers.py:284-289) -------------------------------------
    for (int ct = wave; ct < ntile_e; ct += DENSE_WAVES) {
        const f32x4 c = WLDS ? tile_lds_lds(s_cat, ld2, s_wi, ldW, ct * 16, K2p, lane)
                             : tile_lds_glob(s_cat, ld2, a.W_inter, E, ct * 16, K2, K2p, lane);
        const int col = ct * 16 + (lane & 15), rq = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = fmaxf(c[i], 0.f);
            s_comb[(rq + i) * ldE + col] = v;
            const int b = row0 + rq + i;
            if (a.combined && b < a.B && sp == 0) a.combined[(size_t)b * E + col] = v;
        }
    }
    __syncthreads();
    DENSE_STAMP(3);
    // ---- logits, centre scores, loss gradients (model.py:38, layers.py:243, model.py:54-61) ---
    if (tid < TB * 16) {   // 16 rows x 4 dot products, each split over 4 lanes, combined by a 2-step butterfly
        const int part = tid & 3, which = (tid >> 2) & 3, t = tid >> 4;
        float acc = 0.f;
        if (which < 2) {
            const float *wv = s_wc + which * E;
            for (int e = part; e < E; e += 4) acc = fmaf(s_comb[t * ldE + e], wv[e], acc);
        } else {
            const float *wv = s_wc + 2 * E + (which - 2) * F;
            for (int f = part; f < F; f += 4) acc = fmaf(s_cat[t * ld2 + f], wv[f], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0) s_tmp[t * 4 + which] = acc + (which >= 2 ? s_wc[2 * E + 2 * F + (which - 2)] : 0.f);
    }
    __syncthreads();
    if (tid < TB) {
        const int t = tid, b = row0 + t;
        const float g0 = s_tmp[t * 4 + 0], g1 = s_tmp[t * 4 + 1], c0 = s_tmp[t * 4 + 2], c1 = s_tmp[t * 4 + 3];
        float dg0 = 0.f, dg1 = 0.f, dc0 = 0.f, dc1 = 0.f;
        if (b < a.B) {
            if (sp == 0) {
                a.logits[2 * b] = g0;
                a.logits[2 * b + 1] = g1;
                a.center[2 * b] = c0;
                a.center[2 * b + 1] = c1;
            }
            if (a.labels) {
                const int y = a.labels[b];
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
    __syncthreads();
    DENSE_STAMP(4);
    if (!train) return;

    float *slab = a.slabs + (size_t)tile_id * a.n_params;
    // ---- backward ----------------------------------------------------------------------------
    // dcomb = (dlogits W_cls) * relu'(combined);  dW_cls, dW_clf, db_clf
    for (int i = tid; i < TB * E; i += blockDim.x) {
        const int t = i / E, e = i - t * E;
        const float g = s_dlog[2 * t] * s_wc[e] + s_dlog[2 * t + 1] * s_wc[E + e];
        s_dcomb[t * ldE + e] = s_comb[t * ldE + e] > 0.f ? g : 0.f;
    }
    DENSE_STAMP(8);
    for (int i = tid; i < (sp == 0 ? 2 * E : 0); i += blockDim.x) {
        const int cidx = i / E, e = i - cidx * E;
        float sacc = 0.f;
        for (int t = 0; t < TB; ++t) sacc = fmaf(s_dlog[2 * t + cidx], s_comb[t * ldE + e], sacc);
        slab[off_cls(F, E, R) + i] = sacc;
    }
    DENSE_STAMP(9);
    for (int i = tid; i < (sp == 0 ? 2 * F : 0); i += blockDim.x) {
        const int cidx = i / F, f = i - cidx * F;
        float sacc = 0.f;
        for (int t = 0; t < TB; ++t) sacc = fmaf(s_dcl[2 * t + cidx], s_cat[t * ld2 + f], sacc);
        slab[off_clf(F, E, R) + i] = sacc;
    }
    DENSE_STAMP(10);
    if (tid < 2 && sp == 0) {
        float sacc = 0.f;
        for (int t = 0; t < TB; ++t) sacc += s_dcl[2 * t + tid];
        slab[off_bias(F, E, R) + tid] = sacc;
    }
    __syncthreads();
    DENSE_STAMP(5);
    // one phase: dh_r = (dcomb W[F+rE.., :]^T) * relu'(h_r) for every r   and   dW_inter = cat^T dcomb
    {
        const int mt2 = (K2 + 15) / 16;
        const int n_dh = R * ntile_e, n_all = n_dh + mt2 * ntile_e;
        float *dst = slab + off_inter(F, E, R);
        // every workgroup of the tile needs all of dh_r; the dW_inter tiles are dealt out over the tile's S workgroups
        for (int t0 = wave; t0 < n_dh + (n_all - n_dh + S - 1) / S; t0 += DENSE_WAVES) {
            const int tile = t0 < n_dh ? t0 : n_dh + (t0 - n_dh) * S + sp;
            if (tile >= n_all) continue;
            if (tile < n_dh) {
                const int r = tile / ntile_e, ct = tile - r * ntile_e;
                const float *Wr = a.W_inter + (size_t)(F + r * E) * E;   // rows of W_inter that multiply h_r
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int rr = lane & 15, kq = lane >> 4;
#pragma unroll 4
                for (int e0 = 0; e0 < E; e0 += 4) {                       // out[t][j] = sum_e dcomb[t][e] * Wr[j][e]
                    const float av = s_dcomb[rr * ldE + e0 + kq];
                    const float bv = WLDS ? s_wi[(F + r * E + ct * 16 + rr) * ldW + e0 + kq]
                                          : Wr[(size_t)(ct * 16 + rr) * E + e0 + kq];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
                }
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
    __syncthreads();
    DENSE_STAMP(7);
}

// g = sum over slabs in a fixed order (8 interleaved partial sums, then a fixed tree), so 8 slab
// reads are in flight per thread; torch.optim.Adam step with coupled weight decay.
constexpr int ADAM_ACC = 8;
__global__ void __launch_bounds__(256) adam_reduce_kernel(float *__restrict__ theta, float *__restrict__ m,
                                                          float *__restrict__ v, const float *__restrict__ slabs,
                                                          int n_slabs, int64_t n_params,
                                                          const int32_t *__restrict__ step_counter, float lr,
                                                          float beta1, float beta2, float eps, float wd,
                                                          float *__restrict__ grad_out, int apply) {
    // 64 parameters per workgroup; the slabs are split over its 4 waves (each: 8 interleaved accumulators = 8 loads in
    // flight), the four partial sums are added in wave order: a fixed order, and a quarter of the dependent load batches
    __shared__ float part[4][PCG_WAVE];
    const int lane = threadIdx.x & (PCG_WAVE - 1), w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * PCG_WAVE + lane;
    const bool ok = i < n_params;
    const int per = (n_slabs + 3) / 4;
    const int s_begin = w * per, s_end = (s_begin + per < n_slabs) ? s_begin + per : n_slabs;
    // the optimizer state of wave 0's parameters is requested up front, behind nothing
    float p = 0.f, m_old = 0.f, v_old = 0.f, t = 1.f;
    if (w == 0 && ok && apply) {
        p = theta[i];
        m_old = m[i];
        v_old = v[i];
        t = (float)step_counter[0];
    }
    float acc[ADAM_ACC];
#pragma unroll
    for (int u = 0; u < ADAM_ACC; ++u) acc[u] = 0.f;
    int s = s_begin;
    if (ok) {
        for (; s + ADAM_ACC <= s_end; s += ADAM_ACC) {
#pragma unroll
            for (int u = 0; u < ADAM_ACC; ++u) acc[u] += slabs[(size_t)(s + u) * n_params + i];
        }
        for (int u = 0; s < s_end; ++s, ++u) acc[u] += slabs[(size_t)s * n_params + i];
    }
    part[w][lane] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (w != 0 || !ok) return;
    float g = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (grad_out) grad_out[i] = g;
    if (!apply) return;
    g = fmaf(wd, p, g);
    const float mi = beta1 * m_old + (1.f - beta1) * g;
    const float vi = beta2 * v_old + (1.f - beta2) * g * g;
    m[i] = mi;
    v[i] = vi;
    const float bc1 = 1.f - powf(beta1, t), bc2 = 1.f - powf(beta2, t);
    const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
    theta[i] = p - (lr / bc1) * (mi / denom);
}

static size_t dense_smem_bytes(int F, int E, int R, bool wlds) {
    const int K1p = (2 * F + 3) & ~3, K2p = (F + R * E + 3) & ~3;
    size_t fl = (size_t)(R * TB * (K1p + 1) + TB * (K2p + 1) + (2 + R) * TB * (E + 1) + 4 * TB + 2 * E + 2 * F + 4 + 4 * TB);
    if (wlds) fl += (size_t)(K2p + R * K1p) * (E + 4);
    return sizeof(float) * fl;
}

static unsigned long long *g_dense_stamps = nullptr;

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
    if (!g || !g->X || !theta || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;
    if (!ids || !agg || !logits || !center) return PCG_E_ARG;
    if (emb < 16 || emb % 16 != 0 || g->n_rel < 1 || g->n_rel > PCG_MAX_REL) return PCG_E_UNSUPPORTED;
    if (slabs && !labels) return PCG_E_ARG;
    const int F = g->feat_dim, E = emb, R = g->n_rel;
    const bool wlds = pcg::dense_smem_bytes(F, E, R, true) <= 160 * 1024 && (pcg::DENSE_WAVES * PCG_WAVE) % (E / 4) == 0;
    const size_t smem = pcg::dense_smem_bytes(F, E, R, wlds);
    if (smem > 160 * 1024) return PCG_E_UNSUPPORTED;
    pcg::DenseArgs a;
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
    a.W_cls = theta + pcg::off_cls(F, E, R);
    a.W_inter = theta + pcg::off_inter(F, E, R);
    for (int r = 0; r < PCG_MAX_REL; ++r) a.W_intra[r] = r < R ? theta + pcg::off_intra(F, E, R, r) : nullptr;
    a.W_clf = theta + pcg::off_clf(F, E, R);
    a.b_clf = theta + pcg::off_bias(F, E, R);
    a.lambda_1 = lambda_1;
    a.inv_count = inv_count;
    a.logits = logits;
    a.center = center;
    a.combined = combined;
    a.row_loss = row_loss;
    a.slabs = slabs;
    a.n_params = pcg::n_params_of(F, E, R);
    a.step_counter = step_counter;
    // few tiles (small batches): up to 4 workgroups per tile, so that the weight-gradient tiles of a 16-row tile are not one
    // CU's serial work while most of the chip idles
    const int n_tiles = (B + pcg::TB - 1) / pcg::TB;
    int n_split = slabs ? 256 / n_tiles : 1;
    a.n_split = n_split < 1 ? 1 : (n_split > 4 ? 4 : n_split);
    a.stamps = pcg::g_dense_stamps;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(pcg::dense_step_kernel<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void *>(pcg::dense_step_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return PCG_E_LAUNCH;
        attr = true;
    }
    const dim3 grid(n_tiles * a.n_split), block(pcg::DENSE_WAVES * PCG_WAVE);
    if (wlds) hipLaunchKernelGGL(pcg::dense_step_kernel<true>, grid, block, smem, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(pcg::dense_step_kernel<false>, grid, block, smem, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_adam_step(float *theta, float *m, float *v, const float *slabs, int32_t n_slabs, int64_t n_params,
                  const int32_t *step_counter, double lr, double beta1, double beta2, double eps, double weight_decay,
                  float *grad_out, int32_t apply, void *stream) {
    if (!slabs || n_slabs < 0 || n_params < 1) return PCG_E_ARG;
    if (apply && (!theta || !m || !v || !step_counter)) return PCG_E_ARG;
    if (!apply && !grad_out) return PCG_E_ARG;
    hipLaunchKernelGGL(pcg::adam_reduce_kernel, dim3((unsigned)((n_params + PCG_WAVE - 1) / PCG_WAVE)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), theta, m, v, slabs, n_slabs, n_params, step_counter, (float)lr,
                       (float)beta1, (float)beta2, (float)eps, (float)weight_decay, grad_out, apply);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
