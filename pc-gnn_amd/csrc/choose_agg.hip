// choose + aggregate: the PC-GNN hot kernel for gfx950.
//
// One *group* owns one (relation r, batch centre b) CSR row; the group size follows
// the row length so the serial latency chain of a row stays short:
//     deg <= 512   : one 64-lane wave           (4 rows per 256-thread block)
//     deg <= 8192  : one 256-thread workgroup   (4 waves)
//     longer       : one 1024-thread workgroup  (16 waves; > 12288 via global scratch)
// and does, without the chosen set ever leaving the CU:
//   1. neighbour ids + distance keys |s0[centre] - s0[j]| -> LDS (4 loads in flight per lane)
//   2. exact k-th-smallest key: MSB-first bisection that starts at the first bit in
//      which the row's keys differ and, once <= 64 candidates remain, finishes on
//      one register per lane; ties broken by row position (ballot prefix counts)
//   3. compaction of the kept neighbour ids (ascending) in LDS
//   4. feature-row gather + segmented sum, 64/lpr rows per wave-instruction
//      (float4 per lane; 128-B rows => 8 rows = 1 KiB per instruction, 8 in flight)
//   5. minority over-sampling for positive centres: 64-ary window search on the
//      per-step sorted train-pos scores, de-duplicated against (3)
//   6. mean -> agg[r, b, :]
// Reference lines replaced: src/layers.py:217-219, 246-262, 587-624, 633-738.
#include <limits.h>

#include "common.h"

namespace pcg {

constexpr int T1_CAP = 512;      // row length handled by a single wave
constexpr int T4_CAP = 8192;     // ... by a 4-wave workgroup
constexpr int T16_CAP = 12288;   // ... by a 16-wave workgroup with ids+keys in LDS; longer rows: global scratch
constexpr int T1_WAVES_PER_BLOCK = 4;
constexpr int N_T4_BLOCKS = 1024;
constexpr int N_T16_BLOCKS = 256;
constexpr int UNROLL = 8;        // row-gather instructions in flight per wave
constexpr int KEY_UNROLL = 4;    // neighbour-score gathers in flight per lane

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
    uint32_t *counters;   // [0] #rows queued for the 4-wave tier, [1] its dequeue head, [2]/[3] same for 16-wave,
                          // [4] blocks of the last kernel that are done; all zero between calls
    int32_t *queue4;      // [n_rel * B]
    int32_t *queue16;     // [n_rel * B]
    uint32_t *scratch;    // [N_T16_BLOCKS * 2 * max_degree] for rows longer than T16_CAP
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

// acc += sum of X rows list[..n); this wave takes batches first_batch, first_batch + batch_step, ...
template <int NACC>
__device__ __forceinline__ void gather_accumulate(const float *__restrict__ X, int stride, const RowGeom &q,
                                                  const uint32_t *list, int n, int first_batch, int batch_step,
                                                  float4 (&acc)[NACC]) {
    const int per_iter = q.rpw * UNROLL;
    for (int base = first_batch * per_iter; base < n; base += batch_step * per_iter) {
        float4 v[UNROLL][NACC];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int i = base + u * q.rpw + q.slot;
            const bool ok = i < n;
            const uint32_t id = ok ? list[i] : 0u;
            const float *row = X + (size_t)id * stride;
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
                const int ch = a * q.lpr + q.sub;
                v[u][a] = (ok && ch < q.nch) ? *reinterpret_cast<const float4 *>(row + 4 * ch)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) {
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

// LDS scratch of one group
struct GroupMem {
    uint32_t *keys;   // >= deg: distance keys, later the kept ids (compacted, ascending)
    uint32_t *ids;    // >= deg: the row's neighbour ids
    int *red;         // 2 * NW + 2 ints (NW > 1)
    uint32_t *cand;   // 64 candidate keys for the register finish
    float *fred;      // NW * feat_stride floats (NW > 1)
    uint32_t *stage;  // 64 per wave
    int *xcnt;        // 1 int (NW > 1)
};

// ---------------------------------------------------------------------------
// One (relation, centre) row, processed by a group of NW waves.
// ---------------------------------------------------------------------------
template <int NW, int NACC>
__device__ void process_row(const ChooseArgs &a, int row, const GroupMem &gm) {
    const int lane = lane_id();
    const int wave = (NW > 1) ? (int)(threadIdx.x >> 6) : 0;
    const int tid = wave * PCG_WAVE + lane;
    constexpr int NT = NW * PCG_WAVE;
    uint32_t *keys = gm.keys;
    uint32_t *ids = gm.ids;
    int *red = gm.red;

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
        m = (int)((double)k * a.rho[r]);             // layers.py:681
        if (m > a.g.n_pos) m = a.g.n_pos;
        if (m < 0) m = 0;
    }
    const bool emit = a.sel_indices != nullptr;
    const int64_t ebase = emit ? a.sel_begin[row] : 0;

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
                ids[i] = id[u];
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

    // ---- 2. k-th smallest key ----------------------------------------------------
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
        int remaining = k, ncand = d;
        uint32_t prefix;
        int bit;                           // next bit to decide
        if (kmin == kmax) {                // every key equal: pure position order
            prefix = kmin;
            bit = -1;
        } else {
            bit = 31 - __clz((int)(kmin ^ kmax));
            prefix = (bit == 31) ? 0u : (kmin & (0xFFFFFFFFu << (bit + 1)));
        }
        // (a) rounds over the LDS keys while more than one wave-ful of candidates remains
        for (; bit >= 0 && ncand > PCG_WAVE; --bit) {
            const uint32_t hi_mask = (bit == 31) ? 0u : (0xFFFFFFFFu << (bit + 1));
            int c0 = 0;
            for (int base = wave * PCG_WAVE; base < d; base += NT) {  // wave-uniform trip count
                const int i = base + lane;
                const uint32_t key = i < d ? keys[i] : 0xFFFFFFFFu;
                c0 += wave_count(i < d && ((key & hi_mask) == prefix) && !((key >> bit) & 1u));
            }
            int pre, tot;
            grp_scan<NW>(c0, wave, lane, red, pre, tot);
            if (remaining > tot) {
                remaining -= tot;
                prefix |= 1u << bit;
                ncand -= tot;
            } else {
                ncand = tot;
            }
        }
        // (b) <= 64 candidates: one register per lane, no more LDS passes / barriers
        if (bit >= 0) {
            const uint32_t hi_mask = (bit == 31) ? 0u : (0xFFFFFFFFu << (bit + 1));
            int seen = 0;
            for (int base = 0; base < d; base += NT) {
                const int i = base + tid;
                const uint32_t key = i < d ? keys[i] : 0u;
                const bool isc = i < d && (key & hi_mask) == prefix;
                const uint64_t bm = __ballot(isc);
                int pre, tot;
                grp_scan<NW>(__popcll(bm), wave, lane, red, pre, tot);
                if (isc) gm.cand[seen + pre + __popcll(bm & lanemask_lt())] = key;
                seen += tot;
            }
            grp_sync<NW>();
            const bool have = lane < ncand;
            const uint32_t ck = have ? gm.cand[lane] : 0u;
            for (; bit >= 0; --bit) {
                const uint32_t hm = (bit == 31) ? 0u : (0xFFFFFFFFu << (bit + 1));
                const int c0 = wave_count(have && ((ck & hm) == prefix) && !((ck >> bit) & 1u));
                if (remaining > c0) {
                    remaining -= c0;
                    prefix |= 1u << bit;
                    ncand -= c0;
                } else {
                    ncand = c0;
                }
            }
            grp_sync<NW>();
        }
        kstar = prefix;
        need = remaining;
        n_equal = ncand;
    }

    // ---- 3. compaction of kept ids, ascending, into keys[] --------------------------
    int ns = 0;
    if (keep_all) {
        ns = d;
        for (int i = tid; i < d; i += NT) keys[i] = ids[i];
    } else {
        const bool ranked_ties = n_equal != need;   // some, not all, of the equal keys are kept
        int ties_seen = 0;
        for (int base = 0; base < d; base += NT) {
            const int i = base + tid;
            const bool in = i < d;
            const uint32_t key = in ? keys[i] : 0u;
            const uint32_t id = in ? ids[i] : 0u;
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
            grp_scan<NW>(__popcll(sm), wave, lane, red, spre, stot);   // its barriers order the reads above before the writes below
            if (sel) keys[ns + spre + __popcll(sm & lanemask_lt())] = id;
            ns += stot;
        }
    }
    grp_sync<NW>();
    const uint32_t *sel = keys;

    // GCN-style self union (graphsage.py:78-79, 214): the centre joins its own set
    bool self_extra = false;
    if (a.add_self) self_extra = !sorted_contains(sel, ns, (uint32_t)node);
    const int n_self = self_extra ? 1 : 0;

    if (emit) {
        if (ebase + ns + n_self <= a.sel_capacity) {
            for (int i = tid; i < ns; i += NT) a.sel_indices[ebase + i] = (int32_t)sel[i];
            if (self_extra && tid == 0) a.sel_indices[ebase + ns] = node;
        } else if (tid == 0) {
            atomicOr(a.status, (uint32_t)PCG_ST_SEL_OVERFLOW);
        }
    }

    // ---- 4. gather + segmented sum of the kept rows ----------------------------------
    const RowGeom q = row_geom(a.g.feat_stride, lane);
    float4 acc[NACC];
#pragma unroll
    for (int x = 0; x < NACC; ++x) acc[x] = make_float4(0.f, 0.f, 0.f, 0.f);
    gather_accumulate<NACC>(a.g.X, a.g.feat_stride, q, sel, ns, wave, NW, acc);
    uint32_t *my_stage = gm.stage + wave * PCG_WAVE;
    if (self_extra && wave == 0) {
        my_stage[0] = (uint32_t)node;
        gather_accumulate<NACC>(a.g.X, a.g.feat_stride, q, my_stage, 1, 0, 1, acc);
    }
    int extras = 0;  // per-wave count of minority rows added beyond the kept neighbours

    // ---- 5. minority over-sampling (layers.py:675-691) ---------------------------------
    if (m > 0) {
        const uint64_t *__restrict__ pk = a.pos_keys;
        const int P = a.g.n_pos;
        int L, R, L2, R2, tau = INT_MAX;
        if (m >= P) {
            L = L2 = 0;
            R = R2 = P;
        } else {
            // window [lo, lo+m) of the m nearest: first lo whose left end is not farther than the element right of the window
            const int lo = wave_partition_point(0, P - m, lane, [&](int x) {
                return (c - pos_score(pk, x)) > (pos_score(pk, x + m) - c);
            });
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
            if (tid == 0) *gm.xcnt = 0;
            __syncthreads();
        }
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
                    if (lane == 0) o = atomicAdd(gm.xcnt, nnew);
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
            gather_accumulate<NACC>(a.g.X, a.g.feat_stride, q, my_stage, nnew, 0, 1, acc);
            extras += nnew;
        }
    }

    // ---- 6. reduce partial sums, divide, store --------------------------------------------
#pragma unroll
    for (int x = 0; x < NACC; ++x)
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
        float *fred = gm.fred;
        if (lane < q.lpr) {
#pragma unroll
            for (int x = 0; x < NACC; ++x) {
                const int ch = x * q.lpr + q.sub;
                if (ch < q.nch) *reinterpret_cast<float4 *>(fred + wave * a.g.feat_stride + 4 * ch) = acc[x];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < q.lpr) {
#pragma unroll
            for (int x = 0; x < NACC; ++x) {
                const int ch = x * q.lpr + q.sub;
                if (ch >= q.nch) continue;
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
        for (int x = 0; x < NACC; ++x) {
            const int ch = x * q.lpr + q.sub;
            if (ch >= q.nch) continue;
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

__device__ __forceinline__ int64_t row_degree(const ChooseArgs &a, int row) {
    const int r = row / a.B, b = row - r * a.B;
    const int node = a.nodes[b];
    return a.g.indptr[r][node + 1] - a.g.indptr[r][node];
}

// tier 1: one wave per row, 4 rows per block; longer rows are queued for the wider tiers
template <int NACC>
__global__ void __launch_bounds__(T1_WAVES_PER_BLOCK *PCG_WAVE) choose_agg_t1(const ChooseArgs a) {
    __shared__ uint32_t keys[T1_WAVES_PER_BLOCK][T1_CAP];
    __shared__ uint32_t ids[T1_WAVES_PER_BLOCK][T1_CAP];
    __shared__ uint32_t cand[T1_WAVES_PER_BLOCK][PCG_WAVE];
    __shared__ uint32_t stage[T1_WAVES_PER_BLOCK][PCG_WAVE];
    const int w = threadIdx.x >> 6;
    const int row = blockIdx.x * T1_WAVES_PER_BLOCK + w;
    if (row >= a.g.n_rel * a.B) return;
    const int64_t d = row_degree(a, row);
    if (d > T1_CAP) {
        if (lane_id() == 0) {
            const uint32_t q = atomicAdd(&a.counters[d <= T4_CAP ? 0 : 2], 1u);
            if (q < (uint32_t)(a.g.n_rel * a.B)) (d <= T4_CAP ? a.queue4 : a.queue16)[q] = row;
        }
        return;
    }
    GroupMem gm{keys[w], ids[w], nullptr, cand[w], nullptr, stage[w], nullptr};
    process_row<1, NACC>(a, row, gm);
}

// tiers 4 / 16: one workgroup per row, rows pulled from the tier's queue
template <int NW, int NACC, int CAP>
__global__ void __launch_bounds__(NW *PCG_WAVE) choose_agg_wide(const ChooseArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *keys_lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t *ids_lds = keys_lds + CAP;
    float *fred = reinterpret_cast<float *>(ids_lds + CAP);
    uint32_t *stage = reinterpret_cast<uint32_t *>(fred + NW * a.g.feat_stride);
    uint32_t *cand = stage + NW * PCG_WAVE;
    int *red = reinterpret_cast<int *>(cand + PCG_WAVE);
    int *xcnt = red + 2 * NW + 2;
    int *qslot = xcnt + 1;
    const int *queue = (NW == 4) ? a.queue4 : a.queue16;
    uint32_t *ctr = a.counters + ((NW == 4) ? 0 : 2);
    uint32_t nq = ctr[0];
    const uint32_t nq_max = (uint32_t)(a.g.n_rel * a.B);
    if (nq > nq_max) nq = nq_max;
    for (;;) {
        if (threadIdx.x == 0) *qslot = (int)atomicAdd(&ctr[1], 1u);
        __syncthreads();
        const uint32_t qi = (uint32_t)*qslot;
        __syncthreads();
        if (qi >= nq) break;
        const int row = queue[qi];
        GroupMem gm{keys_lds, ids_lds, red, cand, fred, stage, xcnt};
        if (NW == 16 && row_degree(a, row) > CAP) {   // over-long hub row: ids + keys in global scratch
            gm.keys = a.scratch + (size_t)blockIdx.x * 2 * a.g.max_degree;
            gm.ids = gm.keys + a.g.max_degree;
        }
        process_row<NW, NACC>(a, row, gm);
    }
    // The 4-wave tier is the last kernel of a call: its last block to leave the dequeue loop
    // zeroes the counters, so the next call (or graph replay) starts clean without a memset.
    if (NW == 4 && threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(&a.counters[4], 1u) == gridDim.x - 1) {
            a.counters[0] = 0; a.counters[1] = 0; a.counters[2] = 0; a.counters[3] = 0; a.counters[4] = 0;
        }
    }
}

static size_t wide_smem_bytes(int nw, int cap, int feat_stride) {
    return sizeof(uint32_t) * 2 * cap + sizeof(float) * nw * feat_stride + sizeof(uint32_t) * (nw + 1) * PCG_WAVE +
           sizeof(int) * (2 * nw + 2 + 2);
}

template <int NACC>
static int launch_all(const ChooseArgs &a, hipStream_t st) {
    const pcg_graph_desc &g = a.g;
    const int rows = g.n_rel * a.B;
    const int blocks = (rows + T1_WAVES_PER_BLOCK - 1) / T1_WAVES_PER_BLOCK;
    hipLaunchKernelGGL(choose_agg_t1<NACC>, dim3(blocks), dim3(T1_WAVES_PER_BLOCK * PCG_WAVE), 0, st, a);
    PCG_LAUNCH_CHECK();
    if (g.max_degree > T4_CAP) {   // the longest rows first
        const size_t smem = wide_smem_bytes(16, T16_CAP, g.feat_stride);
        static bool attr16 = false;
        if (!attr16) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(choose_agg_wide<16, NACC, T16_CAP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return PCG_E_LAUNCH;
            attr16 = true;
        }
        const int nb = rows < N_T16_BLOCKS ? rows : N_T16_BLOCKS;
        hipLaunchKernelGGL((choose_agg_wide<16, NACC, T16_CAP>), dim3(nb), dim3(16 * PCG_WAVE), smem, st, a);
        PCG_LAUNCH_CHECK();
    }
    if (g.max_degree > T1_CAP) {
        const size_t smem = wide_smem_bytes(4, T4_CAP, g.feat_stride);
        static bool attr4 = false;
        if (!attr4) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(choose_agg_wide<4, NACC, T4_CAP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
                return PCG_E_LAUNCH;
            attr4 = true;
        }
        const int nb = rows < N_T4_BLOCKS ? rows : N_T4_BLOCKS;
        hipLaunchKernelGGL((choose_agg_wide<4, NACC, T4_CAP>), dim3(nb), dim3(4 * PCG_WAVE), smem, st, a);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

static int64_t queue_bytes(const pcg_graph_desc *g, int32_t B) { return (((int64_t)g->n_rel * B * 4 + 255) / 256) * 256; }

}  // namespace pcg

extern "C" {

int64_t pcg_choose_workspace_bytes(const pcg_graph_desc *g, int32_t B) {
    if (!g || B < 0) return PCG_E_ARG;
    int64_t bytes = 256 + 2 * pcg::queue_bytes(g, B);   // counters + the two tier queues
    if (g->max_degree > pcg::T16_CAP) bytes += (int64_t)pcg::N_T16_BLOCKS * 2 * g->max_degree * 4;
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
                         const double *thresholds, const double *rho, int32_t train_flag, int32_t norm,
                         int32_t add_self, float *agg, int32_t agg_stride, int32_t *cnt, const int64_t *sel_begin,
                         int32_t *sel_indices, int64_t sel_capacity, void *workspace, uint32_t *status, void *stream) {
    if (!g || B < 0) return PCG_E_ARG;
    if (B == 0) return PCG_OK;  // empty trailing batch (model_handler.py:134 produces one): nothing to do
    if (!nodes || !s0 || !thresholds || !agg || !workspace) return PCG_E_ARG;
    if (train_flag && !rho) return PCG_E_ARG;
    if (g->n_rel < 1 || g->n_rel > PCG_MAX_REL || !g->X) return PCG_E_ARG;
    if (g->feat_stride % 4 != 0 || g->feat_stride < g->feat_dim || agg_stride < g->feat_dim) return PCG_E_ARG;
    if (g->feat_stride > 512) return PCG_E_UNSUPPORTED;
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
    const int64_t qb = pcg::queue_bytes(g, B);
    a.counters = reinterpret_cast<uint32_t *>(ws);
    a.queue4 = reinterpret_cast<int32_t *>(ws + 256);
    a.queue16 = reinterpret_cast<int32_t *>(ws + 256 + qb);
    a.scratch = reinterpret_cast<uint32_t *>(ws + 256 + 2 * qb);

    hipStream_t st = static_cast<hipStream_t>(stream);
    return g->feat_stride <= 256 ? pcg::launch_all<1>(a, st) : pcg::launch_all<2>(a, st);
}

}  // extern "C"
