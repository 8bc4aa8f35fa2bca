// choose + aggregate: the PC-GNN hot kernel for gfx950.
//
// One *group* (one 64-lane wave for ordinary rows, one 1024-thread workgroup for
// hub rows) owns one (relation r, batch centre b) CSR row and does, without the
// chosen set ever leaving the CU:
//   1. distance keys  |s0[centre] - s0[j]|  for the row's neighbours -> LDS
//   2. exact k-th-smallest by MSB-first radix bisection over the LDS keys,
//      ties broken by row position (ballot prefix counts)
//   3. in-place compaction of the kept neighbour ids (ascending) in LDS
//   4. feature-row gather + segmented sum, 64/lpr rows per wave-instruction
//      (float4 per lane, 128-B rows => 8 rows = 1 KiB per instruction)
//   5. minority over-sampling for positive centres: window search on the
//      per-step sorted train-pos scores, de-duplicated against (3)
//   6. mean -> agg[r, b, :]
// Reference lines replaced: src/layers.py:217-219, 246-262, 587-624, 633-738.
#include <limits.h>

#include "common.h"

namespace pcg {

constexpr int WAVE_CAP = 2048;     // max row length handled by a single wave (keys in LDS)
constexpr int WAVES_PER_BLOCK = 4; // independent row-waves per 256-thread block
constexpr int BLOCK_NW = 16;       // waves cooperating on a hub row
constexpr int BLOCK_CAP = 24576;   // hub-row keys kept in LDS; longer rows use global scratch
constexpr int MAX_ACC = 2;         // float4 accumulators per lane => feat_stride <= 512
constexpr int UNROLL = 4;          // row-gather instructions in flight per wave

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
    int32_t train_flag, norm, add_self;
    float *agg;
    int32_t agg_stride;
    int32_t *cnt;
    const int64_t *sel_begin;
    int32_t *sel_indices;
    int64_t sel_capacity;
    uint32_t *status;
    // workspace
    uint32_t *big_counters;  // [0] = #queued hub rows, [1] = dequeue head
    int32_t *big_queue;      // [n_rel * B]
    uint32_t *big_scratch;   // [n_big_blocks * max_degree] for rows longer than BLOCK_CAP
};

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

struct RowGeom {  // how one wave-instruction covers feature rows
    int lpr, rpw, slot, sub, nch, nacc;
};

__device__ __forceinline__ RowGeom row_geom(int stride, int lane) {
    RowGeom q;
    q.lpr = lanes_per_row(stride);
    q.rpw = PCG_WAVE / q.lpr;
    q.slot = lane / q.lpr;
    q.sub = lane % q.lpr;
    q.nch = stride >> 2;
    q.nacc = (q.nch + q.lpr - 1) / q.lpr;
    return q;
}

// acc += sum of X rows list[first .. n) taken with stride `step` batches by this wave
__device__ __forceinline__ void gather_accumulate(const float *__restrict__ X, int stride, const RowGeom &q,
                                                  const uint32_t *list, int n, int first_batch, int batch_step,
                                                  float4 (&acc)[MAX_ACC]) {
    const int per_iter = q.rpw * UNROLL;
    for (int base = first_batch * per_iter; base < n; base += batch_step * per_iter) {
        float4 v[UNROLL][MAX_ACC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = base + u * q.rpw + q.slot;
            const bool ok = i < n;
            const uint32_t id = ok ? list[i] : 0u;
            const float *row = X + (size_t)id * stride;
#pragma unroll
            for (int a = 0; a < MAX_ACC; ++a) {
                const int ch = a * q.lpr + q.sub;
                v[u][a] = (ok && a < q.nacc && ch < q.nch) ? *reinterpret_cast<const float4 *>(row + 4 * ch)
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int a = 0; a < MAX_ACC; ++a) {
                acc[a].x += v[u][a].x;
                acc[a].y += v[u][a].y;
                acc[a].z += v[u][a].z;
                acc[a].w += v[u][a].w;
            }
    }
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

// ---------------------------------------------------------------------------
// One (relation, centre) row, processed by a group of NW waves.
//   keys : >= deg uint32, LDS (or global scratch for over-long hub rows);
//          holds the distance keys, then - compacted in place - the kept ids
//   red  : NW ints (LDS)            fred : NW * feat_stride floats (LDS), NW > 1 only
//   stage: 64 uint32 per wave (LDS) xcnt : one int (LDS), NW > 1 only
// ---------------------------------------------------------------------------
template <int NW>
__device__ void process_row(const ChooseArgs &a, int row, uint32_t *keys, int *red, float *fred, uint32_t *stage,
                            int *xcnt) {
    const int lane = lane_id();
    const int wave = (NW > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int tid = wave * PCG_WAVE + lane;
    constexpr int NT = NW * PCG_WAVE;

    const int r = row / a.B, b = row - r * a.B;
    const int node = a.nodes[b];
    const int64_t start = a.g.indptr[r][node];
    const int d = (int)(a.g.indptr[r][node + 1] - start);
    const int32_t *__restrict__ nbr = a.g.indices[r] + start;
    const float c = a.center_s0 ? a.center_s0[b] : a.s0[node];
    const int k = (int)ceil((double)d * a.thr[r]);   // layers.py:260
    const bool keep_all = !(d > k + 1);              // layers.py:662
    int m = 0;
    if (a.train_flag && a.labels[b] == 1) {          // layers.py:675
        m = (int)((double)k * a.rho[r]);               // layers.py:681
        if (m > a.g.n_pos) m = a.g.n_pos;
        if (m < 0) m = 0;
    }
    const bool emit = a.sel_indices != nullptr;
    const int64_t ebase = emit ? a.sel_begin[row] : 0;

    // ---- 1. distance keys --------------------------------------------------
    if (!keep_all)
        for (int i = tid; i < d; i += NT) keys[i] = dist_key(c, a.s0[nbr[i]]);
    grp_sync<NW>();

    // ---- 2. k-th smallest key: MSB-first bisection (bit 31 is always 0) -----
    uint32_t kstar = 0xFFFFFFFFu;
    int need = 0;
    if (!keep_all) {
        uint32_t prefix = 0;
        int remaining = k;
        for (int bit = 30; bit >= 0; --bit) {
            const uint32_t hi_mask = 0xFFFFFFFFu << (bit + 1);
            int c0 = 0;
            for (int base = wave * PCG_WAVE; base < d; base += NT) {  // wave-uniform trip count
                const int i = base + lane;
                const uint32_t key = i < d ? keys[i] : 0xFFFFFFFFu;
                const bool p = i < d && ((key & hi_mask) == prefix) && !((key >> bit) & 1u);
                c0 += wave_count(p);
            }
            int pre, tot;
            grp_scan<NW>(c0, wave, lane, red, pre, tot);
            if (remaining > tot) {
                remaining -= tot;
                prefix |= 1u << bit;
            }
        }
        kstar = prefix;
        need = remaining;
    }

    // ---- 3. compaction of kept ids, ascending, in place ----------------------
    int ns = 0;
    {
        int ties_seen = 0;
        for (int base = 0; base < d; base += NT) {
            const int i = base + tid;
            const bool in = i < d;
            const uint32_t key = (in && !keep_all) ? keys[i] : 0u;
            const uint32_t id = in ? (uint32_t)nbr[i] : 0u;
            const bool tie = in && !keep_all && key == kstar;
            const uint64_t tm = __ballot(tie);
            int tpre, ttot;
            grp_scan<NW>(__popcll(tm), wave, lane, red, tpre, ttot);  // also orders reads before writes
            const int trank = ties_seen + tpre + __popcll(tm & lanemask_lt());
            const bool sel = in && (keep_all || key < kstar || (tie && trank < need));
            const uint64_t sm = __ballot(sel);
            int spre, stot;
            grp_scan<NW>(__popcll(sm), wave, lane, red, spre, stot);
            if (sel) keys[ns + spre + __popcll(sm & lanemask_lt())] = id;
            ns += stot;
            ties_seen += ttot;
        }
    }
    grp_sync<NW>();
    const uint32_t *sel = keys;

    // GCN-style self union (graphsage.py:78-79, 214): the centre joins its own set
    bool self_extra = false;
    if (a.add_self) self_extra = !sorted_contains(sel, ns, (uint32_t)node);

    if (emit) {
        if (ebase + ns + (self_extra ? 1 : 0) <= a.sel_capacity) {
            for (int i = tid; i < ns; i += NT) a.sel_indices[ebase + i] = (int32_t)sel[i];
            if (self_extra && tid == 0) a.sel_indices[ebase + ns] = node;
        } else if (tid == 0) {
            atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
        }
    }

    // ---- 4. gather + segmented sum of the kept rows ----------------------------
    const RowGeom q = row_geom(a.g.feat_stride, lane);
    float4 acc[MAX_ACC];
#pragma unroll
    for (int x = 0; x < MAX_ACC; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    gather_accumulate(a.g.X, a.g.feat_stride, q, sel, ns, wave, NW, acc);
    int extras = 0;  // per-wave count of minority rows added beyond the kept neighbours
    const int n_self = self_extra ? 1 : 0;
    if (self_extra && wave == 0) {
        stage[0] = (uint32_t)node;
        gather_accumulate(a.g.X, a.g.feat_stride, q, stage, 1, 0, 1, acc);
    }

    // ---- 5. minority over-sampling (layers.py:675-691) --------------------------
    if (m > 0) {
        const uint64_t *__restrict__ pk = a.pos_keys;
        const int P = a.g.n_pos;
        int L, R, L2, R2, tau = INT_MAX;
        if (m >= P) {
            L = L2 = 0;
            R = R2 = P;
        } else {
            int lo = 0, hi = P - m;  // window [lo, lo+m) of the m nearest
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((c - pos_score(pk, mid)) > (pos_score(pk, mid + m) - c)) lo = mid + 1;
                else hi = mid;
            }
            const uint32_t ka = pos_dkey(pk, lo, c), kb = pos_dkey(pk, lo + m - 1, c);
            const uint32_t ks = ka > kb ? ka : kb;  // m-th smallest distance
            L = run_end_fwd(pk, c, ks, lo, lo + m, lane);
            R = (L == lo + m) ? L : run_begin_bwd(pk, c, ks, lo + m - 1, L, lane);
            L2 = run_begin_bwd(pk, c, ks, lo - 1, 0, lane);
            R2 = run_end_fwd(pk, c, ks, lo + m, P, lane);
            // ties are [L2, L) and [R, R2); strictly nearer ones are [L, R)
            const int need_t = m - (R - L);
            const int T = (L - L2) + (R2 - R);
            if (T > need_t) {  // take the need_t ties with the smallest train_pos position
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
            }
        }
        if (NW > 1) {
            if (tid == 0) *xcnt = 0;
            __syncthreads();
        }
        uint32_t *my_stage = stage + wave * PCG_WAVE;
        int chunk = 0;
        for (int i0 = L2; i0 < R2; i0 += PCG_WAVE, ++chunk) {
            if (NW > 1 && (chunk % NW) != wave) continue;
            const int i = i0 + lane;
            bool take = false;
            uint32_t u = 0;
            if (i < R2) {
                const uint32_t pos = (uint32_t)pk[i];
                take = (i >= L && i < R) || ((int)pos <= tau);
                if (take) {
                    u = (uint32_t)a.g.train_pos[pos];
                    if (sorted_contains(sel, ns, u) || (a.add_self && u == (uint32_t)node)) take = false;  // set(), :694
                }
            }
            const uint64_t tmk = __ballot(take);
            const int nnew = __popcll(tmk);
            if (nnew == 0) continue;
            if (take) my_stage[__popcll(tmk & lanemask_lt())] = u;
            if (emit) {
                int off;
                if (NW > 1) {
                    int o = 0;
                    if (lane == 0) o = atomicAdd(xcnt, nnew);
                    off = __shfl(o, 0);
                } else {
                    off = extras;
                }
                const int64_t at = ebase + ns + n_self + off;
                if (at + nnew <= a.sel_capacity) {
                    if (take) a.sel_indices[at + __popcll(tmk & lanemask_lt())] = (int32_t)u;
                } else if (lane == 0) {
                    atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
                }
            }
            gather_accumulate(a.g.X, a.g.feat_stride, q, my_stage, nnew, 0, 1, acc);
            extras += nnew;
        }
    }

    // ---- 6. reduce partial sums, divide, store ----------------------------------
#pragma unroll
    for (int x = 0; x < MAX_ACC; ++x)
        for (int o = q.lpr; o < PCG_WAVE; o <<= 1) {
            acc[x].x += __shfl_xor(acc[x].x, o);
            acc[x].y += __shfl_xor(acc[x].y, o);
            acc[x].z += __shfl_xor(acc[x].z, o);
            acc[x].w += __shfl_xor(acc[x].w, o);
        }
    int epre, etot;
    grp_scan<NW>(extras, wave, lane, red, epre, etot);
    const int n = ns + n_self + etot;
    if constexpr (NW > 1) {
        if (lane < q.lpr) {
#pragma unroll
            for (int x = 0; x < MAX_ACC; ++x) {
                const int ch = x * q.lpr + q.sub;
                if (x < q.nacc && ch < q.nch) *reinterpret_cast<float4 *>(fred + wave * a.g.feat_stride + 4 * ch) = acc[x];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < q.lpr) {
#pragma unroll
            for (int x = 0; x < MAX_ACC; ++x) {
                const int ch = x * q.lpr + q.sub;
                if (x >= q.nacc || ch >= q.nch) continue;
                float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int w = 0; w < NW; ++w) {
                    const float4 t = *reinterpret_cast<const float4 *>(fred + w * a.g.feat_stride + 4 * ch);
                    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
                }
                acc[x] = s;
            }
        }
    }
    if (wave == 0 && lane < q.lpr) {
        const float den = (a.norm == PCG_NORM_SQRT_COUNT) ? sqrtf((float)n) : (float)n;
        float *out = a.agg + ((size_t)r * a.B + b) * a.agg_stride;
#pragma unroll
        for (int x = 0; x < MAX_ACC; ++x) {
            const int ch = x * q.lpr + q.sub;
            if (x >= q.nacc || ch >= q.nch) continue;
            const int f = 4 * ch;
            if (f + 0 < a.g.feat_dim) out[f + 0] = acc[x].x / den;
            if (f + 1 < a.g.feat_dim) out[f + 1] = acc[x].y / den;
            if (f + 2 < a.g.feat_dim) out[f + 2] = acc[x].z / den;
            if (f + 3 < a.g.feat_dim) out[f + 3] = acc[x].w / den;
        }
        if (a.cnt && lane == 0) a.cnt[(size_t)r * a.B + b] = n;
    }
    grp_sync<NW>();
}

// ordinary rows: one wave per row, 4 rows per block; hub rows are queued
__global__ void __launch_bounds__(WAVES_PER_BLOCK *PCG_WAVE) choose_agg_wave(const ChooseArgs a) {
    __shared__ uint32_t keys[WAVES_PER_BLOCK][WAVE_CAP];
    __shared__ uint32_t stage[WAVES_PER_BLOCK][PCG_WAVE];
    const int w = threadIdx.x >> 6;
    const int row = blockIdx.x * WAVES_PER_BLOCK + w;
    if (row >= a.g.n_rel * a.B) return;
    const int r = row / a.B, b = row - r * a.B;
    const int node = a.nodes[b];
    const int64_t d = a.g.indptr[r][node + 1] - a.g.indptr[r][node];
    if (d > WAVE_CAP) {
        if (lane_id() == 0) a.big_queue[atomicAdd(&a.big_counters[0], 1u)] = row;
        return;
    }
    process_row<1>(a, row, keys[w], nullptr, nullptr, stage[w], nullptr);
}

// hub rows: one 1024-thread workgroup per row, pulled from the queue
__global__ void __launch_bounds__(BLOCK_NW *PCG_WAVE) choose_agg_block(const ChooseArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *keys_lds = reinterpret_cast<uint32_t *>(smem);
    float *fred = reinterpret_cast<float *>(keys_lds + BLOCK_CAP);
    uint32_t *stage = reinterpret_cast<uint32_t *>(fred + BLOCK_NW * a.g.feat_stride);
    int *red = reinterpret_cast<int *>(stage + BLOCK_NW * PCG_WAVE);
    int *xcnt = red + BLOCK_NW;
    int *qslot = xcnt + 1;
    const uint32_t nq = a.big_counters[0];
    for (;;) {
        if (threadIdx.x == 0) *qslot = (int)atomicAdd(&a.big_counters[1], 1u);
        __syncthreads();
        const uint32_t qi = (uint32_t)*qslot;
        __syncthreads();
        if (qi >= nq) break;
        const int row = a.big_queue[qi];
        const int r = row / a.B, b = row - r * a.B;
        const int node = a.nodes[b];
        const int64_t d = a.g.indptr[r][node + 1] - a.g.indptr[r][node];
        uint32_t *keys = (d <= BLOCK_CAP) ? keys_lds : a.big_scratch + (size_t)blockIdx.x * a.g.max_degree;
        process_row<BLOCK_NW>(a, row, keys, red, fred, stage, xcnt);
    }
}

constexpr int N_BIG_BLOCKS = 256;

static size_t block_smem_bytes(int feat_stride) {
    return sizeof(uint32_t) * BLOCK_CAP + sizeof(float) * BLOCK_NW * feat_stride + sizeof(uint32_t) * BLOCK_NW * PCG_WAVE +
           sizeof(int) * (BLOCK_NW + 2);
}

}  // namespace pcg

extern "C" {

int64_t pcg_choose_workspace_bytes(const pcg_graph_desc *g, int32_t B) {
    if (!g || B < 0) return PCG_E_ARG;
    int64_t bytes = 256;                                                  // counters
    bytes += (((int64_t)g->n_rel * B * 4 + 255) / 256) * 256;            // hub-row queue
    if (g->max_degree > pcg::BLOCK_CAP) bytes += (int64_t)pcg::N_BIG_BLOCKS * g->max_degree * 4;
    return bytes;
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

int pcg_choose_aggregate(const pcg_graph_desc *g, const int32_t *nodes, const int32_t *labels, int32_t B,
                         const float *s0, const float *center_s0, const uint64_t *pos_keys,
                         const double *thresholds, const double *rho,
                         int32_t train_flag, int32_t norm, int32_t add_self, float *agg, int32_t agg_stride,
                         int32_t *cnt, const int64_t *sel_begin, int32_t *sel_indices, int64_t sel_capacity,
                         void *workspace, uint32_t *status, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;  // empty trailing batch (model_handler.py:134 produces one): nothing to do
    if (!nodes || !s0 || !thresholds || !agg || !workspace) return PCG_E_ARG;
    if (train_flag && !rho) return PCG_E_ARG;
    if (g->n_rel < 1 || g->n_rel > PCG_MAX_REL || !g->X) return PCG_E_ARG;
    if (g->feat_stride % 4 != 0 || g->feat_stride < g->feat_dim || agg_stride < g->feat_dim) return PCG_E_ARG;
    if (g->feat_stride > 4 * 64 * pcg::MAX_ACC) return PCG_E_UNSUPPORTED;
    if (train_flag && (!labels || (g->n_pos > 0 && (!pos_keys || !g->train_pos)))) return PCG_E_ARG;
    if ((sel_indices != nullptr) != (sel_begin != nullptr)) return PCG_E_ARG;
    if (sel_indices && !status) return PCG_E_ARG;
    for (int r = 0; r < g->n_rel; ++r)
        if (!g->indptr[r] || !g->indices[r]) return PCG_E_ARG;

    pcg::ChooseArgs a;
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
    a.norm = norm;
    a.add_self = add_self;
    a.agg = agg;
    a.agg_stride = agg_stride;
    a.cnt = cnt;
    a.sel_begin = sel_begin;
    a.sel_indices = sel_indices;
    a.sel_capacity = sel_capacity;
    a.status = status;
    unsigned char *ws = static_cast<unsigned char *>(workspace);
    a.big_counters = reinterpret_cast<uint32_t *>(ws);
    a.big_queue = reinterpret_cast<int32_t *>(ws + 256);
    a.big_scratch = reinterpret_cast<uint32_t *>(ws + 256 + (((int64_t)g->n_rel * B * 4 + 255) / 256) * 256);

    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a.big_counters, 0, 256, st) != hipSuccess) return PCG_E_LAUNCH;
    const int rows = g->n_rel * B;
    const int blocks = (rows + pcg::WAVES_PER_BLOCK - 1) / pcg::WAVES_PER_BLOCK;
    hipLaunchKernelGGL(pcg::choose_agg_wave, dim3(blocks), dim3(pcg::WAVES_PER_BLOCK * PCG_WAVE), 0, st, a);
    PCG_LAUNCH_CHECK();
    if (g->max_degree > pcg::WAVE_CAP) {
        const size_t smem = pcg::block_smem_bytes(g->feat_stride);
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(pcg::choose_agg_block),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return PCG_E_LAUNCH;
            attr_set = true;
        }
        const int nb = rows < pcg::N_BIG_BLOCKS ? rows : pcg::N_BIG_BLOCKS;
        hipLaunchKernelGGL(pcg::choose_agg_block, dim3(nb), dim3(pcg::BLOCK_NW * PCG_WAVE), smem, st, a);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

}  // extern "C"
