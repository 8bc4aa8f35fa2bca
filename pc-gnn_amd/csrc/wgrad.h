// Weight gradients of the dense tail as GEMMs over the BATCH dimension, with the Adam update applied by the workgroup that
// owns the output tile - no per-tile gradient slabs anywhere (src/layers.py:625-629, 284-289; src/model.py:54-61;
// src/model_handler.py:124,149-153).
//
// The dense kernel (dense.hip, adam_clf = 3) leaves the step's activations and activation gradients TRANSPOSED in `acts`
// ([rows][ld], a batch row per column; columns beyond the batch, up to the next multiple of 16, are zero):
//     cat    [F + R E]  self features | h_1 .. h_R (after the ReLU)         -> A of dW_inter, first F rows: A of dW_r / B of dW_clf
//     agg    [R F]      the aggregated neighbour features of every relation -> A of dW_r (rows F .. 2F)
//     dcomb  [E]        d loss / d (cat W_inter), ReLU mask applied         -> B of dW_inter
//     dh     [R E]      d loss / d ([self | agg_r] W_r), ReLU mask applied  -> B of dW_r
//     comb   [E]        combined embeddings                                 -> B of dW_cls
//     dlog   [2]        d loss / d gnn logits                               -> A of dW_cls
//     dcl    [2]        d loss / d label-aware logits (times lambda_1)      -> A of dW_clf, db_clf
// Every weight gradient is then  dW[m][n] = sum_t A[m][t] * B[n][t]  with both operands contiguous in t: a lane of the f32
// matrix core reads float4s (sixteen batch rows per four lanes) instead of four scattered floats.
//
// One 256-thread workgroup per 16 x 16 tile of a weight matrix (batches beyond 1024 rows: one per tile and 1024 rows, the
// partial tiles summed in part order by the part that arrives last): its four waves take a quarter of the rows each (blocks of 16
// rows in ascending order; inside a block the k order is the fixed permutation 4 kq + i the float4 reads imply), their partial
// tiles are added in wave order in LDS, and thread (row, col) applies torch.optim.Adam's update to its parameter - or only
// stores the gradient (grad_out: the parity tests, and the partitioned path's all-reduce).  Sums have a fixed order: results are
// bitwise reproducible run to run.  These workgroups ride in the NEXT step's gather launch (gather.hip) - the dense kernel that
// follows it is the first reader of the updated weights - or run as a launch of their own (pcg_adam_flush, pcg_wgrad).
#pragma once
#include "common.h"

namespace pcg {

struct WgradArgs {
    const float *acts;
    int32_t ld;                // floats between two rows of acts (>= 16 * the K blocks below)
    int32_t F, E, R;
    float *theta, *m, *v;      // apply != 0
    const int32_t *step_counter;
    AdamHyper h;
    const uint32_t *pending;   // device words [0] == 2: acts hold a step whose update is due, [1] = its K blocks; null: n_kblocks
    int32_t n_kblocks;         // batch rows / 16 (rounded up); with `pending` only the EXPECTED count (the engine's batch size): the
                               // first operands are requested for it before the pending words have arrived
    int32_t kparts;            // workgroups sharing a tile's K range (1: none; the grid is tiles * kparts)
    uint32_t *tickets;         // kparts > 1: [tiles] arrival counters, zero between launches
    float *partials;           // kparts > 1: [tiles][kparts][256] partial tiles (write-through)
    float *grad_out;           // [n_params] or null
    uint32_t *flag_set;        // device word set to 1 by the launch ("a gradient is waiting": the partitioned path), or null
    int32_t apply;
    int32_t with_clf;          // also the label classifier's tiles (its step is pcg_choose_gather_train's otherwise)
};

__host__ __device__ inline int64_t wg_off_inter(int E) { return 2 * (int64_t)E; }
__host__ __device__ inline int64_t wg_off_intra(int F, int E, int R, int r) { return wg_off_inter(E) + (int64_t)(F + R * E) * E + (int64_t)r * 2 * F * E; }
__host__ __device__ inline int wgrad_act_rows(int F, int E, int R) { return (F + R * E) + R * F + E + R * E + E + 4; }
__host__ __device__ inline int wgrad_tiles(int F, int E, int R, int with_clf) {
    const int ne = E / 16;
    return ne * (1 + (F + R * E + 15) / 16 + R * ((2 * F + 15) / 16)) + (with_clf ? (F + 1 + 15) / 16 : 0);
}
constexpr int WG_NB = 8;       // K blocks (of 16 batch rows) whose operands are in flight per wave
// a wave's share of the batch is at most two rounds of WG_NB blocks (two memory round trips): batches beyond 4 * 2 * WG_NB * 16
// = 1024 rows are shared by several workgroups per tile (their partial tiles meet in `partials`, the last one in adds them up)
__host__ __device__ inline int wgrad_kparts(int n_kblocks) { return n_kblocks <= 8 * WG_NB ? 1 : (n_kblocks + 8 * WG_NB - 1) / (8 * WG_NB); }
__host__ __device__ inline int64_t wgrad_scratch_floats(int F, int E, int R, int n_kblocks) {
    const int t = wgrad_tiles(F, E, R, 1);
    return (int64_t)((t + 63) / 64 * 64) + (int64_t)t * wgrad_kparts(n_kblocks) * 256;
}

// workgroup `wg` of wgrad_tiles(...) * kparts (256 threads: tid; a group of four waves of a larger workgroup may pass its own
// thread index - then kparts must be 1: the groups of a workgroup share its barriers); red: 4 * 256 floats of LDS; NB: K blocks
// whose operands a wave has in flight
template <int NB = WG_NB>
__device__ __forceinline__ void wgrad_adam_body(const WgradArgs &a, int wg, float (*red)[256], int tid = (int)threadIdx.x) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int KP = a.kparts > 1 ? a.kparts : 1;
    if (a.flag_set && wg == 0 && tid == 0) a.flag_set[0] = 1u;
    int b = wg / KP;
    const int tile = b, kp = wg - b * KP;
    // the words that say whether (and how much) there is to do are requested first, the first operands - for the batch size
    // the host expects - right behind them: one memory round trip instead of two in the usual case
    const int hint = a.n_kblocks;
    uint32_t pend0 = 2u, pend1 = (uint32_t)hint;
    if (a.pending) {
        pend0 = a.pending[0];                                  // (one word, the same for every thread)
        pend1 = a.pending[1];
    }
    const int F = a.F, E = a.E, R = a.R;
    const int K2 = F + R * E, K1 = 2 * F, ne = E / 16;
    const int r_agg = K2, r_dcomb = K2 + R * F, r_dh = r_dcomb + E, r_comb = r_dh + R * E, r_dlog = r_comb + E, r_dcl = r_dlog + 2;
    const int lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, kq = lane >> 4;
    // ---- which tile: section, m0, n0 (workgroup-uniform) ----
    int sec, m0, n0, rel = 0;
    {
        const int mt2 = (K2 + 15) / 16, mt1 = (K1 + 15) / 16;
        if (b < ne) { sec = 0; m0 = 0; n0 = 16 * b; }
        else if ((b -= ne) < mt2 * ne) { sec = 1; m0 = 16 * (b / ne); n0 = 16 * (b % ne); }
        else if ((b -= mt2 * ne) < R * mt1 * ne) {
            sec = 2; rel = b / (mt1 * ne);
            const int bb = b - rel * mt1 * ne;
            m0 = 16 * (bb / ne); n0 = 16 * (bb % ne);
        } else { b -= R * mt1 * ne; sec = 3; m0 = 0; n0 = 16 * b; }
    }
    const int M = sec == 0 ? 2 : (sec == 1 ? K2 : (sec == 2 ? K1 : 2));
    const int N = sec == 3 ? F + 1 : E;
    // the parameter of thread (row, col) of the tile, its optimizer state requested up front
    const int orow = tid >> 4, ocol = tid & 15;
    const int pm = m0 + orow, pn = n0 + ocol;
    const bool pok = pm < M && pn < N;
    int64_t pidx;
    if (sec == 0) pidx = (int64_t)pm * E + pn;
    else if (sec == 1) pidx = wg_off_inter(E) + (int64_t)pm * E + pn;
    else if (sec == 2) pidx = wg_off_intra(F, E, R, rel) + (int64_t)pm * E + pn;
    else pidx = pn < F ? wg_off_intra(F, E, R, R) + (int64_t)pm * F + pn : wg_off_intra(F, E, R, R) + 2 * (int64_t)F + pm;
    float p_old = 0.f, m_old = 0.f, v_old = 0.f, tstep = 1.f;
    if (pok && a.apply) {
        p_old = a.theta[pidx];
        m_old = a.m[pidx];
        v_old = a.v[pidx];
        tstep = (float)a.step_counter[0];
    }
    // ---- operand rows of this lane (clamped: every load is unconditional; rows / columns beyond the matrix only feed outputs
    //      that are never stored) ----
    const int am = m0 + lr < M ? m0 + lr : M - 1;
    const int bn = n0 + lr < N ? n0 + lr : N - 1;
    int arow, brow;
    bool ones = false;
    if (sec == 0) { arow = r_dlog + am; brow = r_comb + bn; }
    else if (sec == 1) { arow = am; brow = r_dcomb + bn; }
    else if (sec == 2) { arow = am < F ? am : r_agg + rel * F + (am - F); brow = r_dh + rel * E + bn; }
    else { arow = r_dcl + am; brow = bn < F ? bn : 0; ones = bn >= F; }      // (column F of the classifier's tile: the bias, B = 1)
    const float *__restrict__ ap = a.acts + (size_t)arow * a.ld + 4 * kq;
    const float *__restrict__ bp = a.acts + (size_t)brow * a.ld + 4 * kq;
    // this wave's blocks: part kp of the batch's blocks, a quarter of that per wave
    auto range = [&](int nkb, int &lo, int &hi) {
        const int pp = (nkb + KP - 1) / KP, p_lo = kp * pp < nkb ? kp * pp : nkb, p_hi = p_lo + pp < nkb ? p_lo + pp : nkb;
        const int per = (p_hi - p_lo + 3) >> 2;
        lo = p_lo + wave * per < p_hi ? p_lo + wave * per : p_hi;
        hi = lo + per < p_hi ? lo + per : p_hi;
    };
    const int ld_blocks = a.ld >> 4;
    int u_lo, u_hi;
    range(hint < ld_blocks ? hint : ld_blocks, u_lo, u_hi);
    f4 av[NB], bv[NB];
    auto fetch = [&](int u0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            int u = u0 + j < u_hi ? u0 + j : u_hi - 1;
            u = u < 0 ? 0 : u;                                     // (an empty range: any block of the buffer; nothing is added)
            av[j] = *reinterpret_cast<const f4 *>(ap + 16 * u);
            bv[j] = *reinterpret_cast<const f4 *>(bp + 16 * u);
        }
    };
    fetch(u_lo);
    if (pend0 != 2u) return;                                       // nothing is waiting (workgroup-uniform)
    if ((int)pend1 != hint) {                                      // another batch size than expected (an epoch's last batch): again
        range((int)pend1 < ld_blocks ? (int)pend1 : ld_blocks, u_lo, u_hi);
        fetch(u_lo);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int u0 = u_lo; u0 < u_hi; u0 += NB) {
        if (u0 != u_lo) fetch(u0);
        if (ones) {
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[j] = f4{1.f, 1.f, 1.f, 1.f};
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (u0 + j >= u_hi) break;                             // (wave-uniform)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].x, bv[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].y, bv[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].z, bv[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j].w, bv[j].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][(4 * kq + i) * 16 + lr] = acc[i];
    __syncthreads();
    float g = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    if (KP > 1) {
        // this part's tile -> its slot (write-through: another workgroup of this launch reads it), then the tile's ticket; the
        // part that arrives last adds the parts up in part order (whichever it is: the sum's order is fixed) and goes on
        float *slot = a.partials + ((size_t)tile * KP + kp) * 256;
        __hip_atomic_store(slot + tid, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                           // every thread's store is complete
        int *flag = reinterpret_cast<int *>(&red[0][0]);           // (the partial tiles have been read)
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)KP - 1u;
        __syncthreads();
        if (!flag[0]) return;
        if (tid == 0) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const float *base = a.partials + (size_t)tile * KP * 256 + tid;
        g = __hip_atomic_load(base, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int q0 = 1; q0 < KP; q0 += 8) {                       // (eight loads in flight, added in part order)
            float x[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) x[q] = __hip_atomic_load(base + (size_t)(q0 + q < KP ? q0 + q : KP - 1) * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int q = 0; q < 8; ++q) g = q0 + q < KP ? g + x[q] : g;
        }
    }
    if (!pok) return;
    if (a.grad_out) a.grad_out[pidx] = g;
    if (!a.apply) return;
    float mi, vi;
    const float p_new = adam_update(p_old, m_old, v_old, g, tstep, a.h, mi, vi);   // torch.optim.Adam, coupled weight decay
    a.m[pidx] = mi;
    a.v[pidx] = vi;
    a.theta[pidx] = p_new;
}

}  // namespace pcg
