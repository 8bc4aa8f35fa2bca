// The choose step (src/layers.py:633-738) for gfx950: ONE persistent launch selects every (relation, centre) row of a batch.
//
// Selection rule of a row with d neighbours (ascending ids), k = ceil(d * threshold):
//   d <= k + 1 : keep all;  else keep the k smallest distance keys |s0[centre] - s0[j]|, ties by row position.
// A distance is a non-negative float, so its bit pattern orders like its value: keys are uint32 and everything below is
// integer work.  How a row finds its k-th smallest key depends on its length:
//   <= 16  (four rows per wave, 16 lanes each) every lane ranks its key against the 15 others with DPP row rotations;
//   <= 64  (one wave) one key per lane, ranked lane against lane (v_readlane), both exact with the positional tie-break
//          built in: no k-th value is ever formed;
//   <= 512 (one wave) up to 8 keys per lane stay in registers; rounds of a 256-bin LDS histogram over the bit range
//          [lo, hi] that still holds the k-th key, until <= 64 candidates are left, which are ranked in one wave;
//   longer (one workgroup of 8 waves) keys in LDS (rows <= 10240) or recomputed from the scores on every pass (longer
//          still: no scratch memory anywhere), 2048-bin histogram rounds, same finish.
// A histogram round narrows [lo, hi] by the factor of its bin count whatever the key distribution, so the number of
// rounds is bounded (31 bits / 8 or 11 bits per round) and usually one round + the in-wave finish is all it takes.
// Kept ids are compacted in row order (ascending ids) into the row's region of the selection list; positive centres in
// training add their minority picks (64-ary window search in the per-step sorted train-pos keys, de-duplicated against
// the kept ids by binary search; a duplicate leaves a -1 hole so slots - and sums - keep a fixed order).
// Nothing fills the unused tail of a row's region: the row writes how many entries each of its gather chunks holds.
//
// Work is pulled by whole workgroups (one atomic per workgroup row, one per batch of eight single-wave items), longest
// class first: see select_rows.
#include <limits.h>
#include <stdlib.h>

#include "choose.h"

namespace pcg {

constexpr int KEY_UNROLL = 8;    // neighbour-score gathers in flight per lane
// every gather of a neighbour's score: a plain load.  (Measured: non-temporal gathers - in the workgroup rows only, or in every
// row - cost the power-law 2 M batch +17 us per step and the YelpChi-like one +0.6 / +2.7: neighbours are popular nodes, and
// their score lines are re-used out of the L1 / L2 by the rows that follow.)
template <int LEVEL>
__device__ __forceinline__ float s0_ld(const float *__restrict__ s0, uint32_t id) {
    return s0[id];
}
constexpr int KPT = T1_CAP / PCG_WAVE;   // keys per lane of a single-wave register row

#define PCG_STAMP(slot)                                                                \
    do {                                                                               \
        if (a.stamps && tid == 0) a.stamps[(size_t)row * 8 + (slot)] = wall_clock64(); \
    } while (0)

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

// wave-wide min / max of uint32: butterfly inside every 16-lane row (quad permutes, half-row / row mirrors - after them every
// lane of a row holds the row's result), then the four rows through scalar registers
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t t;
    t = dpp_u32<0xB1>(v); v = t < v ? t : v;
    t = dpp_u32<0x4E>(v); v = t < v ? t : v;
    t = dpp_u32<0x141>(v); v = t < v ? t : v;
    t = dpp_u32<0x140>(v); v = t < v ? t : v;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t x = r0 < r1 ? r0 : r1, y = r2 < r3 ? r2 : r3;
    return x < y ? x : y;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    uint32_t t;
    t = dpp_u32<0xB1>(v); v = t > v ? t : v;
    t = dpp_u32<0x4E>(v); v = t > v ? t : v;
    t = dpp_u32<0x141>(v); v = t > v ? t : v;
    t = dpp_u32<0x140>(v); v = t > v ? t : v;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 0), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 32), r3 = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    const uint32_t x = r0 > r1 ? r0 : r1, y = r2 > r3 ? r2 : r3;
    return x > y ? x : y;
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

// The sorted train-pos keys may be produced INSIDE this launch (select_rows sorts them before the rows start, every workgroup
// doing a share): a wave that needs them waits - once - until every key group has been counted in.  Visibility: the producers
// store the sorted keys write-through (sc1) and count a group in only behind those stores' completion (the count is an
// agent-scope atomic, polled with agent-scope loads).  The consumer side needs no cache invalidate: a CU's L1 - and its XCD's L2 -
// hold no line of the sorted-key buffer before the count is complete (both start the launch clean - the scores, rewritten by the
// launch before this one and read here with plain loads, depend on that too - and in this launch nothing reads the buffer
// before its own wait has returned: the loads are behind the wait in program order and the hardware does not speculate loads),
// so the first touch of a line after the wait fetches what the producers wrote through.  An agent-scope acquire fence here
// (buffer_inv sc1, one per workgroup, three workgroups per CU, all at the same moment) cost ~5 us per positive row: it empties
// the L1 under the other workgroups' gathers and under the window search's own first probes.  Nobody waits on a
// workgroup that itself waits: the sort comes first in every workgroup.  The wait is bounded: if the count never arrives (it
// cannot, short of a lost workgroup) the row goes on with whatever the buffer holds and PCG_ST_SYNC_TIMEOUT is raised - a wrong
// result that is reported, never a hung device.
constexpr int SORT_SPIN_MAX = 1 << 21;       // x >= 0.2 us per poll
// sortw: an LDS block of the workgroup, [0] and [1] zero at kernel start: [0] = "the sorted keys are complete", [1] = "a wave of
// this workgroup is polling", [2] = entries of the search index, [3] = its stride, [4 ..] = the index.  ONE wave per workgroup
// polls the device counter (thousands of waves polling one address every few hundred nanoseconds saturate its L2 channel); the
// others wait on the LDS word.  The polling wave also builds the workgroup's SEARCH INDEX: the score bits of every stride-th
// sorted key.  Every row's window search starts in it (LDS) instead of in the keys themselves: a thousand rows are released at
// the same moment, and the first round of a search over the whole buffer probes the same 64 positions for every one of them -
// the few L2 lines behind those positions took ~4 us to serve them all.
constexpr int KIDX_MAX = 512;
__device__ __forceinline__ uint64_t pk_ld(const uint64_t *pk, int i);
__device__ __forceinline__ void wait_sorted_keys(const ChooseArgs &a, int &keys_ok, int lane, int *sortw) {
    if (a.n_sort == 0 || keys_ok) return;                                  // (wave-uniform)
    if (__hip_atomic_load(&sortw[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        int elected = 0;
        if (lane == 0) elected = atomicCAS(&sortw[1], 0, 1) == 0;
        elected = __builtin_amdgcn_readfirstlane(elected);
        if (elected) {
            for (int spins = 0;; ++spins) {
                const unsigned seen = (unsigned)__builtin_amdgcn_readfirstlane(
                    (int)__hip_atomic_load(a.sort_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                if (seen >= (unsigned)a.n_sort) break;
                if (spins >= SORT_SPIN_MAX) {
                    if (lane == 0 && a.status) atomicOr(a.status, (uint32_t)PCG_ST_SYNC_TIMEOUT);
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");         // (compiler + wave ordering: the key loads come after the wait)
            const int P = a.g.n_pos;
            int stride = 16;
            while ((P + stride - 1) / stride > KIDX_MAX) stride <<= 1;
            const int n_idx = (P + stride - 1) / stride;
            for (int t = lane; t < n_idx; t += PCG_WAVE) sortw[4 + t] = (int)(uint32_t)(pk_ld(a.pos_keys, t * stride) >> 32);
            if (lane == 0) {
                sortw[2] = n_idx;
                sortw[3] = stride;
            }
            if (lane == 0) __hip_atomic_store(&sortw[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            while (__hip_atomic_load(&sortw[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                 // (compiler + wave ordering: the key loads come after the wait)
    keys_ok = 1;
}

// every load of a sorted train-pos key.  The keys may have been sorted INSIDE this launch by other workgroups (sort_share:
// write-through sc1 stores, each storing wave drained, one agent-scope add per key group), so the consumer side must be one of
// the hand-off forms that are valid across CUs and XCDs (per-XCD L2s are not coherent, a CU's L1 is never refreshed):
//   PCG_PK_NT = 2 (default): EVERY load of a key is an sc1 load to registers (raw buffer load, aux 16): served by L2 past the L1,
//     and an sc1-stored line is dropped from the storing XCD's L2 - with the polling wave's relaxed agent poll and the LDS word
//     the other waves wait on this is the guides' "sc1 stores, drained; counter; sc1 loads in place of the acquire" form
//     (MI355X_MICROARCH.md, inter-workgroup visibility, Valid forms).  An ordinary load to the compiler (speculated, batched),
//     unlike an agent-scope ATOMIC load ("x = cond ? load : c" became a branch with a load and a full wait of its own: +5 us per
//     positive row).
//   PCG_PK_NT = 0: plain loads - correct only as long as no line of the sorted keys can sit in this CU's L1 / this XCD's L2
//     before the count is complete (the launch starts with clean caches and nothing reads the buffer before its own wait returns;
//     an sc1 store leaves no line behind in the storing XCD's L2).  Not an architectural guarantee: kept as the A/B reference.
//   PCG_PK_NT = 1: non-temporal loads (L1 bypassed, no sc1).
// An agent-scope acquire fence instead (buffer_inv sc1 per workgroup, three workgroups per CU) cost ~5 us per positive row: it
// empties the L1 under the other workgroups' score gathers.
#ifndef PCG_PK_NT
#define PCG_PK_NT 2
#endif
__device__ __forceinline__ uint64_t pk_ld(const uint64_t *pk, int i) {
#if PCG_PK_NT == 2
    // (the descriptor is wave-uniform - four scalar registers formed from the kernel argument; 0x00020000: raw 32-bit format;
    //  the range is not bounded here: every index is clamped by its caller)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(pk), 0, 0x7FFFFFFF, 0x00020000);
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    const u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, i * 8, 0, 16);        // aux 16 = sc1
    return ((uint64_t)v.y << 32) | v.x;
#elif PCG_PK_NT == 1
    return __builtin_nontemporal_load(pk + i);
#else
    return pk[i];
#endif
}

__device__ __forceinline__ float pos_score(const uint64_t *pk, int i) { return from_orderable((uint32_t)(pk_ld(pk, i) >> 32)); }
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

// The k-th smallest (1-based rank `want`) of <= 64 candidate keys, one per lane (`have` lanes): its value, how many
// candidates equal it and how many of those belong to the `want` smallest.  Every lane gets the same answer.
__device__ __forceinline__ void wave_kth(uint32_t ck, bool have, int n, int want, uint32_t &kstar, int &need, int &n_equal) {
    int lt = 0, eq = 0;
    for (int j = 0; j < n; ++j) {                                            // n is wave-uniform
        const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)ck, j);
        lt += o < ck;
        eq += o == ck;
    }
    const int r = want - 1;
    const uint64_t hit = __ballot(have && lt <= r && r < lt + eq);
    const int src = hit ? __ffsll((unsigned long long)hit) - 1 : 0;
    kstar = (uint32_t)__builtin_amdgcn_readlane((int)ck, src);
    n_equal = __builtin_amdgcn_readlane(eq, src);
    need = want - __builtin_amdgcn_readlane(lt, src);
}

// The window of a positive centre's minority picks in the sorted train-pos keys (layers.py:675-691): the m nearest are
// [L, R) (strictly nearer than the m-th distance) plus need_t of the ties [L2, L) and [R, R2) - those whose train_pos position is
// <= tau.  One wave works it out (every lane gets the result); it needs the centre's score and the sorted keys, nothing of the row.
struct MinorWindow {
    int L, R, L2, R2, tau, need_t;
};
__device__ __forceinline__ MinorWindow minority_window(const ChooseArgs &a, int m, float c, int lane, const int *sortw) {
    const uint64_t *__restrict__ pk = a.pos_keys;
    const int P = a.g.n_pos;
    int L, R, L2, R2, tau = INT_MAX, need_t = 0;
    if (m >= P) {
        L = L2 = 0;
        R = R2 = P;
    } else {
        // window [lo, lo+m) of the m nearest: first lo whose left end is not farther than the element right of the window.
        // With pc = the number of keys whose score is below c: the predicate is false for every x >= pc (the left end is not
        // below c) and true for every x < pc - m (the element right of the window is still below c), so lo is in
        // [pc - m, pc]; the workgroup's search index (every stride-th key's score, in LDS) brackets pc to one stride.
        int lo_min = 0, lo_max = P - m;
        if (a.n_sort > 0 && sortw) {
            const int n_idx = sortw[2], stride = sortw[3];
            const uint32_t *ix = reinterpret_cast<const uint32_t *>(sortw + 4);
            const int t = wave_partition_point(0, n_idx, lane, [&](int x) { return from_orderable(ix[x]) < c; });
            const int pc_lo = t > 0 ? (t - 1) * stride + 1 : 0, pc_hi = t * stride < P ? t * stride : P;
            lo_min = pc_lo - m > 0 ? pc_lo - m : 0;
            lo_max = pc_hi < P - m ? pc_hi : P - m;
            lo_max = lo_max < lo_min ? lo_min : lo_max;
        }
        const int lo = wave_partition_point(lo_min, lo_max, lane, [&](int x) {
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
                    cn += wave_count(i < L && (int)(uint32_t)pk_ld(pk, i) <= mid);
                }
                for (int i0 = R; i0 < R2; i0 += PCG_WAVE) {
                    const int i = i0 + lane;
                    cn += wave_count(i < R2 && (int)(uint32_t)pk_ld(pk, i) <= mid);
                }
                if (cn >= need_t) phi = mid;
                else plo = mid + 1;
            }
            tau = plo;
        } else {                          // a few ties: one per lane, the need_t smallest positions by in-register ranking
            const int nl = L - L2;
            const int ti = lane < nl ? L2 + lane : R + (lane - nl);
            const bool tv = lane < T;
            const int tp = tv ? (int)(uint32_t)pk_ld(pk, ti) : INT_MAX;
            int rank = 0;
            for (int j = 0; j < T; ++j) rank += __builtin_amdgcn_readlane(tp, j) < tp;
            // tau = the need_t-th smallest position among the ties (positions are distinct)
            const uint64_t hit = __ballot(tv && rank == need_t - 1);
            const int src = __ffsll((unsigned long long)hit) - 1;
            tau = __builtin_amdgcn_readlane(tp, src < 0 ? 0 : src);
        }
    }
    MinorWindow w;
    w.L = L; w.R = R; w.L2 = L2; w.R2 = R2; w.tau = tau; w.need_t = need_t;
    return w;
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared tail of every row (layers.py:675-694): minority over-sampling for positive centres, GCN-style self union,
// the kept ids to the list (unless the caller has stored them already), |set|, length and the chunk fill counts.
// sel[0 .. ns): the kept neighbour ids, ascending - LDS, or (over-long rows) the row's own region of the global list.
// NW waves work on the row (tid / NT: thread index and count inside the group); red: 2 * NW + 2 ints of LDS (NW > 1).
// ---------------------------------------------------------------------------------------------------------------------
template <int NW>
__device__ __forceinline__ void finish_row(const ChooseArgs &a, int row, const RowRec &p, float c, const uint32_t *sel, int ns,
                                           bool sel_stored, int wave, int lane, int *red, int &keys_ok, int *sortw,
                                           const int *pre_window = nullptr) {
    constexpr int NT = NW * PCG_WAVE;
    const int tid = wave * PCG_WAVE + lane;
    const int m = p.m, node = p.node;
    int32_t *__restrict__ out = a.w.list + p.lbeg;
    int mt = 0;        // slots used
    int valid = 0;     // per-thread count of non-duplicate minority picks
    if (m > 0) {
        const uint64_t *__restrict__ pk = a.pos_keys;
        int L, R, L2, R2, tau, need_t;
        if (pre_window) {                        // worked out by a wave of this workgroup beside the row's key pass (select_wg_row)
            L = pre_window[0]; R = pre_window[1]; L2 = pre_window[2]; R2 = pre_window[3]; tau = pre_window[4]; need_t = pre_window[5];
        } else {
            wait_sorted_keys(a, keys_ok, lane, sortw);
            PCG_STAMP(7);
            const MinorWindow mw = minority_window(a, m, c, lane, sortw);
            L = mw.L; R = mw.R; L2 = mw.L2; R2 = mw.R2; tau = mw.tau; need_t = mw.need_t;
        }
        PCG_STAMP(4);
        // strictly nearer ones: slot = i - L, every thread of the group strides over them
        const int n_strict = R - L;
        for (int base = tid; base < n_strict; base += NT * KEY_UNROLL) {
            uint32_t pos[KEY_UNROLL], u[KEY_UNROLL];
#pragma unroll
            for (int x = 0; x < KEY_UNROLL; ++x) {
                const int j = base + x * NT;
                pos[x] = (uint32_t)pk_ld(pk, L + (j < n_strict ? j : n_strict - 1));
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
                        const uint32_t pos = (uint32_t)pk_ld(pk, i);
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
    if (!sel_stored)
        for (int i = tid; i < ns; i += NT) out[i] = (int32_t)sel[i];
    const int used = ns + mt + n_self;
    const int nch = (rec_cap(p, a.add_self) + CHUNK - 1) / CHUNK;
    for (int j = tid; j < nch; j += NT) {           // what every gather chunk of the row holds
        const int left = used - j * CHUNK;
        a.w.chunk_desc[p.chunk0 + j].z = left < 0 ? 0 : (left > CHUNK ? CHUNK : left);
    }
    if (tid == 0) {
        a.w.len[row] = used;
        a.cnt[row] = ns + vtot + n_self;
    }
    grp_sync<NW>();
    PCG_STAMP(6);
}

// A row that needs neither minority picks nor the self union: everything it has to report, by one lane.
__device__ __forceinline__ void report_plain_row(const ChooseArgs &a, int row, const RowRec &p, int ns) {
    a.w.len[row] = ns;
    a.cnt[row] = ns;
    const int nch = (rec_cap(p, 0) + CHUNK - 1) / CHUNK;      // (kept <= 512 here: at most 4 chunks)
    for (int j = 0; j < nch; ++j) {
        const int left = ns - j * CHUNK;
        a.w.chunk_desc[p.chunk0 + j].z = left < 0 ? 0 : (left > CHUNK ? CHUNK : left);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// single-wave rows
// ---------------------------------------------------------------------------------------------------------------------
// rotate a (key, position) pair by N lanes inside every 16-lane row
#define PCG_ROR_STEP(N)                                                                                     \
    do {                                                                                                    \
        const uint32_t ok = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x120 + (N), 0xF, 0xF, false); \
        const int op = __builtin_amdgcn_update_dpp(0, pos, 0x120 + (N), 0xF, 0xF, false);                  \
        rank += (ok < key) || (ok == key && op < pos);                                                      \
    } while (0)

// Four rows of <= 16 neighbours, one per 16-lane row of the wave.  Only rows that have nothing to do after the selection
// (no minority picks, no self union) are queued here, so a group runs from its records to its four lists without ever
// leaving the 16-lane rows; the short rows that do go on have a wave of their own (select_lane_row).
__device__ __forceinline__ void select_four_short_rows(const ChooseArgs &a, int q_first, int na, const int32_t *const *t_indices,
                                                       int lane) {
    const int g = lane >> 4, pos = lane & 15;
    const int qi = q_first + g;
    const bool active = qi < na;
    const int row = a.w.qa[active ? qi : na - 1];
    const RowRec p = a.w.recs[row];
    const int r = row / a.B;
    // (every load below is unconditional - index clamped, value discarded afterwards: a load inside a branch is waited for
    //  at the branch's end, one round trip after the other)
    const int32_t *__restrict__ nbr = t_indices[r] + (p.d > 0 ? p.start : 0);
    const bool have = active && pos < p.d;
    const uint32_t id = (uint32_t)nbr[pos < p.d ? pos : (p.d > 0 ? p.d - 1 : 0)];
    const float c = a.center_s0 ? a.center_s0[row - r * a.B] : a.s0[p.node + a.center_off];
    const bool keep_all = rec_keep_all(p);
    const float sc = s0_ld<2>(a.s0, id);
    const uint32_t key = have ? dist_key(c, sc) : 0xFFFFFFFFu;     // (valid keys have bit 31 clear)
    int rank = 0;
    PCG_ROR_STEP(1); PCG_ROR_STEP(2); PCG_ROR_STEP(3); PCG_ROR_STEP(4); PCG_ROR_STEP(5);
    PCG_ROR_STEP(6); PCG_ROR_STEP(7); PCG_ROR_STEP(8); PCG_ROR_STEP(9); PCG_ROR_STEP(10);
    PCG_ROR_STEP(11); PCG_ROR_STEP(12); PCG_ROR_STEP(13); PCG_ROR_STEP(14); PCG_ROR_STEP(15);
    const bool sel = have && (keep_all || rank < p.k);             // stable order: (key, position)
    const uint32_t gm = (uint32_t)(__ballot(sel) >> (16 * g)) & 0xFFFFu;
    if (sel) a.w.list[p.lbeg + __popc(gm & ((1u << pos) - 1u))] = (int32_t)id;
    if (active && pos == 0) report_plain_row(a, row, p, __popc(gm));
}

// One row of 17 .. 64 neighbours on one wave: one key per lane, ranked lane against lane in the stable (key, position)
// order - no k-th value is formed.  area: WAVE_AREA words of LDS (the kept ids go behind the histogram's place).
__device__ __forceinline__ void select_lane_row(const ChooseArgs &a, int row, uint32_t *area, int lane, int &keys_ok, int *sortw) {
    const int tid = lane;
    if (a.stamps && tid == 0) a.stamps[(size_t)row * 8] = wall_clock64() | ((unsigned long long)blockIdx.x << 54);   // + who ran it
    const RowRec p = a.w.recs[row];
    const int d = p.d, k = p.k;
    const bool keep_all = rec_keep_all(p);
    const int r = row / a.B;
    const int32_t *__restrict__ nbr = a.g.indices[r] + (d > 0 ? p.start : 0);      // (d == 0: a node without neighbours that joins its own set)
    const float c = a.center_s0 ? a.center_s0[row - r * a.B] : a.s0[p.node + a.center_off];
    uint32_t *sel_lds = area + HIST_W;
    const bool tail = p.m > 0 || a.add_self;
    const bool have = lane < d;
    const uint32_t id = (uint32_t)nbr[have ? lane : (d > 0 ? d - 1 : 0)];   // unconditional load (clamped)
    const float sc = s0_ld<2>(a.s0, id);
    const uint32_t mine = have ? dist_key(c, sc) : 0xFFFFFFFFu;
    PCG_STAMP(1);
    int rank = 0;
    if (!keep_all) {
        // #keys below mine (two instructions per key); when no two keys are equal - the usual case, recognised by the ranks
        // adding up to d (d - 1) / 2 - that is the rank in the stable (key, position) order already
        for (int j = 0; j < d; ++j)                                              // d is wave-uniform
            rank += (uint32_t)__builtin_amdgcn_readlane((int)mine, j) < mine;
        int sum = have ? rank : 0;
        sum += (int)dpp_u32<0xB1>((uint32_t)sum);
        sum += (int)dpp_u32<0x4E>((uint32_t)sum);
        sum += (int)dpp_u32<0x141>((uint32_t)sum);
        sum += (int)dpp_u32<0x140>((uint32_t)sum);
        const int total = __builtin_amdgcn_readlane(sum, 0) + __builtin_amdgcn_readlane(sum, 16) +
                          __builtin_amdgcn_readlane(sum, 32) + __builtin_amdgcn_readlane(sum, 48);
        if (total != d * (d - 1) / 2)                                            // equal keys: + the equal ones at earlier positions
            for (int j = 0; j < d; ++j)
                rank += ((uint32_t)__builtin_amdgcn_readlane((int)mine, j) == mine) && j < lane;
    }
    const bool s = have && (keep_all || rank < k);
    const uint64_t sm = __ballot(s);
    const int ns = __popcll(sm);
    PCG_STAMP(2);
    if (s) {
        const int at = __popcll(sm & lanemask_lt());
        if (tail) sel_lds[at] = id;
        else a.w.list[p.lbeg + at] = (int32_t)id;
    }
    PCG_STAMP(3);
    if (tail) finish_row<1>(a, row, p, c, sel_lds, ns, false, 0, lane, nullptr, keys_ok, sortw);
    else {
        if (lane == 0) report_plain_row(a, row, p, ns);
        PCG_STAMP(6);
    }
}

// One row of 65 .. 512 neighbours on one wave: keys in registers.  area: WAVE_AREA words of LDS
// (histogram HIST_W | kept ids T1_CAP | candidates 64).
__device__ __forceinline__ void select_wave_row(const ChooseArgs &a, int row, uint32_t *area, int lane, int &keys_ok, int *sortw) {
    const int tid = lane;
    if (a.stamps && tid == 0) a.stamps[(size_t)row * 8] = wall_clock64() | ((unsigned long long)blockIdx.x << 54);   // + who ran it
    const RowRec p = a.w.recs[row];
    const int d = p.d, k = p.k;
    const bool keep_all = rec_keep_all(p);
    const int r = row / a.B;
    const int32_t *__restrict__ nbr = a.g.indices[r] + p.start;
    const float c = a.center_s0 ? a.center_s0[row - r * a.B] : a.s0[p.node + a.center_off];
    uint32_t *hist = area, *sel_lds = area + HIST_W, *cand = area + HIST_W + T1_CAP;
    const bool tail = p.m > 0 || a.add_self;
    int32_t *__restrict__ out = a.w.list + p.lbeg;

    // ---- 1. neighbour ids and distance keys -> registers (position u * 64 + lane) ----
    uint32_t id[KPT], key[KPT];
    {
        float sc[KPT];
#pragma unroll
        for (int u = 0; u < KPT; ++u) {
            const int i = u * PCG_WAVE + lane;
            id[u] = (uint32_t)nbr[i < d ? i : d - 1];                    // unconditional loads (clamped): all in flight together
        }
#pragma unroll
        for (int u = 0; u < KPT; ++u) sc[u] = s0_ld<2>(a.s0, id[u]);
#pragma unroll
        for (int u = 0; u < KPT; ++u) key[u] = (u * PCG_WAVE + lane < d) ? dist_key(c, sc[u]) : 0xFFFFFFFFu;
    }
    PCG_STAMP(1);

    // ---- 2. the k-th smallest key: kstar, and how many of the keys equal to it stay ----
    uint32_t kstar = 0xFFFFFFFEu;       // keep_all: every real key is below it
    int need = 0, n_equal = 0;
    if (!keep_all) {
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
        for (int u = 0; u < KPT; ++u)
            if (u * PCG_WAVE + lane < d) {
                kmin = key[u] < kmin ? key[u] : kmin;
                kmax = key[u] > kmax ? key[u] : kmax;
            }
        uint32_t lo = wave_min_u32(kmin), hi = wave_max_u32(kmax);
        int below = 0, cnt = d;
        // invariant: the k-th smallest key lies in [lo, hi]; below = #keys < lo; cnt = #keys in [lo, hi]
        while (lo < hi && cnt > PCG_WAVE) {
            const uint32_t range = hi - lo;
            const int bits = 32 - __clz((int)range);
            const int shift = bits > 8 ? bits - 8 : 0;                      // (range >> shift) < HIST_W
            *reinterpret_cast<uint4 *>(hist + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int u = 0; u < KPT; ++u)
                if (key[u] >= lo && key[u] <= hi) atomicAdd(&hist[(key[u] - lo) >> shift], 1u);   // (pad keys are > hi)
            const uint4 v = *reinterpret_cast<const uint4 *>(hist + 4 * lane);
            const int s = (int)(v.x + v.y + v.z + v.w);
            const int incl = wave_incl_scan(s, lane);
            const int want = k - below;
            const uint64_t reach = __ballot(incl >= want);
            const int L = __ffsll((unsigned long long)reach) - 1;           // total = cnt >= want: some lane reaches it
            int c0 = incl - s, b = 0, hb = (int)v.x;
            if (c0 + (int)v.x < want) {
                c0 += (int)v.x; b = 1; hb = (int)v.y;
                if (c0 + (int)v.y < want) {
                    c0 += (int)v.y; b = 2; hb = (int)v.z;
                    if (c0 + (int)v.z < want) { c0 += (int)v.z; b = 3; hb = (int)v.w; }
                }
            }
            const int bin = __builtin_amdgcn_readlane(4 * lane + b, L);
            below += __builtin_amdgcn_readlane(c0, L);
            cnt = __builtin_amdgcn_readlane(hb, L);
            lo += (uint32_t)bin << shift;
            const uint32_t top = lo + ((1u << shift) - 1u);
            hi = top < hi ? top : hi;
        }
        kstar = lo;
        need = k - below;
        n_equal = cnt;
        if (lo < hi) {
            // <= 64 candidates in [lo, hi]: one per lane, ranked in-wave
            int at = 0;
#pragma unroll
            for (int u = 0; u < KPT; ++u) {
                const bool isc = key[u] >= lo && key[u] <= hi;
                const uint64_t bm = __ballot(isc);
                if (isc) cand[at + __popcll(bm & lanemask_lt())] = key[u];
                at += __popcll(bm);
            }
            const bool hv = lane < cnt;
            const uint32_t ck = hv ? cand[lane] : 0xFFFFFFFFu;
            wave_kth(ck, hv, cnt, k - below, kstar, need, n_equal);
        }
    }
    PCG_STAMP(2);

    // ---- 3. kept ids, ascending (row order), to LDS (the tail needs them) or straight to the list ----
    const bool ranked = n_equal != need;       // some, not all, of the keys equal to kstar stay: the first `need` by position
    int ns = 0, ties_seen = 0;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        if (u * PCG_WAVE >= d) break;           // wave-uniform
        const bool in = u * PCG_WAVE + lane < d;
        bool s = in && key[u] <= kstar;
        if (ranked) {
            const bool tie = in && key[u] == kstar;
            const uint64_t tm = __ballot(tie);
            s = in && (key[u] < kstar || (tie && ties_seen + __popcll(tm & lanemask_lt()) < need));
            ties_seen += __popcll(tm);
        }
        const uint64_t sm = __ballot(s);
        if (s) {
            const int at = ns + __popcll(sm & lanemask_lt());
            if (tail) sel_lds[at] = id[u];
            else out[at] = (int32_t)id[u];
        }
        ns += __popcll(sm);
    }
    PCG_STAMP(3);
    if (tail) finish_row<1>(a, row, p, c, sel_lds, ns, false, 0, lane, nullptr, keys_ok, sortw);
    else {
        if (lane == 0) report_plain_row(a, row, p, ns);
        PCG_STAMP(6);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// workgroup rows (> 512 neighbours): 8 waves, keys in LDS (LDSK) or - rows beyond its capacity - in global scratch
// ---------------------------------------------------------------------------------------------------------------------
// f(i, key) for every position i in [begin, end) with stride NT from `first`.  LDSK: the keys are in LDS.  Otherwise they are
// in the row's global scratch (gk: written by pass 1, read back coalesced, KEY_UNROLL loads in flight)
template <bool LDSK, class F>
__device__ __forceinline__ void for_keys(const uint32_t *keys, const uint32_t *__restrict__ gk, int first, int end, int step, F f) {
    if constexpr (LDSK) {
        for (int i = first; i < end; i += step) f(i, keys[i]);
    } else {
        for (int base = first; base < end; base += step * KEY_UNROLL) {
            uint32_t kv[KEY_UNROLL];
#pragma unroll
            for (int u = 0; u < KEY_UNROLL; ++u) {
                const int i = base + u * step;
                kv[u] = gk[i < end ? i : end - 1];
            }
#pragma unroll
            for (int u = 0; u < KEY_UNROLL; ++u) {
                const int i = base + u * step;
                if (i < end) f(i, kv[u]);
            }
        }
    }
}

// lds: WG_KEYCAP words (keys, later the kept ids) | hist HIST_WG | cand 64 | red
template <bool LDSK, int NW = SEL_NW>
__device__ __forceinline__ void select_wg_row(const ChooseArgs &a, int row, uint32_t *keys, uint32_t *hist, uint32_t *cand,
                                              int *red, int tid, int &keys_ok, int *sortw, uint32_t *gk = nullptr,
                                              int key_cap = WG_KEYCAP /* words of `keys` */) {
    constexpr int NT = NW * PCG_WAVE;
    constexpr int HBITS = NW == 8 ? 11 : 12;                 // 4 * NT histogram bins: one uint4 of them per thread
    static_assert(4 * NT == (1 << HBITS), "NW is 8 or 16");
    const int lane = tid & (PCG_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.stamps && tid == 0) a.stamps[(size_t)row * 8] = wall_clock64() | ((unsigned long long)blockIdx.x << 54);
    const RowRec p = a.w.recs[row];
    const int d = p.d, k = p.k;
    const bool keep_all = rec_keep_all(p);
    const int r = row / a.B;
    const int32_t *__restrict__ nbr = a.g.indices[r] + p.start;
    const float *__restrict__ s0 = a.s0;
    const float c = a.center_s0 ? a.center_s0[row - r * a.B] : s0[p.node + a.center_off];
    int32_t *__restrict__ out = a.w.list + p.lbeg;
    // res: what one thread / wave found, for everybody: red[2 * NW + 2 ..]
    int *res = red + 2 * NW + 2;

    uint32_t kstar = 0xFFFFFFFFu;
    int need = 0, n_equal = 0;
    // A positive centre's minority window needs the centre's score and the sorted train-pos keys - nothing of the row.  When the
    // keys were sorted BEFORE this launch (a.n_sort == 0: the previous step's dense launch did it on CUs it leaves idle) the
    // group's last wave works the window out (three dependent rounds of key loads, ~2.3 us) while the others form the row's
    // distance keys, and leaves it in LDS for the row's tail: the search is off the row's chain.
    int *win = sortw ? sortw + 4 : nullptr;                       // (the search index's place: unused when nothing is sorted in here)
    const bool early = !keep_all && p.m > 0 && a.n_sort == 0 && win != nullptr && NW > 1;
    // (a row too long for the LDS keeps its keys in global scratch, gk: written by pass 1, read back coalesced by the later passes)
    if (!keep_all) {
        // ---- 1. distance keys (-> LDS), their range ----
        uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
        constexpr int KU1 = 16;     // pass 1 of a workgroup row: sixteen gathers in flight per lane (a 6 000-neighbour row: one round)
        const int nt1 = early ? NT - PCG_WAVE : NT;                // threads that share the key pass
        if (early && wave == NW - 1) {
            const MinorWindow mw = minority_window(a, p.m, c, lane, nullptr);
            if (lane == 0) {
                win[0] = mw.L; win[1] = mw.R; win[2] = mw.L2; win[3] = mw.R2; win[4] = mw.tau; win[5] = mw.need_t;
            }
        } else
        for (int base = tid; base < d; base += nt1 * KU1) {
            uint32_t id[KU1];
            float sc[KU1];
#pragma unroll
            for (int u = 0; u < KU1; ++u) {
                const int i = base + u * nt1;
                id[u] = (uint32_t)nbr[i < d ? i : d - 1];
            }
#pragma unroll
            for (int u = 0; u < KU1; ++u) sc[u] = s0_ld<1>(s0, id[u]);
#pragma unroll
            for (int u = 0; u < KU1; ++u) {
                const int i = base + u * nt1;
                if (i < d) {
                    const uint32_t key = dist_key(c, sc[u]);
                    if constexpr (LDSK) keys[i] = key;
                    else if (gk) gk[i] = key;
                    kmin = key < kmin ? key : kmin;
                    kmax = key > kmax ? key : kmax;
                }
            }
        }
        kmin = wave_min_u32(kmin);
        kmax = wave_max_u32(kmax);
        if (lane == 0) {
            red[wave] = (int)kmin;
            red[NW + wave] = (int)kmax;
        }
        if constexpr (!LDSK) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // (the scratch keys: same CU, same L1)
        __syncthreads();
        if constexpr (!LDSK) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (int w = 0; w < NW; ++w) {
            const uint32_t x = (uint32_t)red[w], y = (uint32_t)red[NW + w];
            kmin = x < kmin ? x : kmin;
            kmax = y > kmax ? y : kmax;
        }
        PCG_STAMP(1);
        // ---- 2. k-th smallest key: histogram rounds over the bit range [lo, hi] ----
        uint32_t lo = kmin, hi = kmax;
        int below = 0, cnt = d;
        while (lo < hi && cnt > PCG_WAVE) {
            const uint32_t range = hi - lo;
            const int bits = 32 - __clz((int)range);
            const int shift = bits > HBITS ? bits - HBITS : 0;              // (range >> shift) < 4 * NT bins
            *reinterpret_cast<uint4 *>(hist + 4 * tid) = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();                                                 // (also: everybody is done with red / res)
            for_keys<LDSK>(keys, gk, tid, d, NT, [&](int, uint32_t key) {
                if (key >= lo && key <= hi) atomicAdd(&hist[(key - lo) >> shift], 1u);
            });
            __syncthreads();
            const uint4 v = *reinterpret_cast<const uint4 *>(hist + 4 * tid);
            const int s = (int)(v.x + v.y + v.z + v.w);
            const int incl_w = wave_incl_scan(s, lane);
            if (lane == PCG_WAVE - 1) red[wave] = incl_w;
            __syncthreads();
            int pre = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) pre += (w < wave) ? red[w] : 0;
            const int incl = pre + incl_w, excl = incl - s;
            const int want = k - below;
            if (excl < want && want <= incl) {                               // exactly one thread
                int c0 = excl, b = 0, hb = (int)v.x;
                if (c0 + (int)v.x < want) {
                    c0 += (int)v.x; b = 1; hb = (int)v.y;
                    if (c0 + (int)v.y < want) {
                        c0 += (int)v.y; b = 2; hb = (int)v.z;
                        if (c0 + (int)v.z < want) { c0 += (int)v.z; b = 3; hb = (int)v.w; }
                    }
                }
                res[0] = 4 * tid + b;
                res[1] = c0;
                res[2] = hb;
            }
            __syncthreads();
            below += res[1];
            cnt = res[2];
            lo += (uint32_t)res[0] << shift;
            const uint32_t top = lo + ((1u << shift) - 1u);
            hi = top < hi ? top : hi;
        }
        kstar = lo;
        need = k - below;
        n_equal = cnt;
        if (lo < hi) {
            // <= 64 candidates in [lo, hi]: gathered (in any order) and ranked by every wave for itself
            __syncthreads();
            if (tid == 0) res[3] = 0;
            __syncthreads();
            for_keys<LDSK>(keys, gk, tid, d, NT, [&](int, uint32_t key) {
                if (key >= lo && key <= hi) {
                    const int at = atomicAdd(&res[3], 1);
                    if (at < PCG_WAVE) cand[at] = key;          // (cnt <= 64 by construction; a counter never indexes unchecked)
                }
            });
            __syncthreads();
            const bool hv = lane < cnt;
            const uint32_t ck = hv ? cand[lane] : 0xFFFFFFFFu;
            wave_kth(ck, hv, cnt, k - below, kstar, need, n_equal);
        }
        __syncthreads();
    }
    PCG_STAMP(2);

    // ---- 3. kept ids, ascending: every wave owns a contiguous stretch of the row ----
    const bool ranked = !keep_all && n_equal != need;
    const int seg = (((d + NW - 1) / NW) + PCG_WAVE - 1) & ~(PCG_WAVE - 1);
    const int b0 = wave * seg < d ? wave * seg : d;
    const int e0 = b0 + seg < d ? b0 + seg : d;
    int tie_base = 0;
    if (ranked) {                                                            // ties before this wave's stretch
        int tc = 0;
        for_keys<LDSK>(keys, gk, b0 + lane, e0, PCG_WAVE, [&](int, uint32_t key) { tc += key == kstar; });
        for (int o = 1; o < PCG_WAVE; o <<= 1) tc += __shfl_xor(tc, o);
        int ttot;
        grp_scan<NW>(tc, wave, lane, red, tie_base, ttot);
    }
    int ns = 0;
    // where the kept ids go: the keys' LDS (dead once every wave has its mask; a scratch-key row never used it) if they fit -
    // the minority picks are de-duplicated against them by binary search, 13 dependent loads per pick: LDS, not global - else
    // straight into the list region
    // A row with nothing to do after the selection (no minority picks, no self union) writes its kept ids straight into its
    // list region and reports itself: no LDS staging, no copy.
    const bool plain = p.m == 0 && !a.add_self;                              // (the same for every thread)
    // (a keep-all row keeps d ids, not k: what must fit is the kept count)
    const bool sel_in_lds = (LDSK || (keep_all ? d : k) <= key_cap) && !plain;
    uint32_t *selbuf = sel_in_lds ? keys : reinterpret_cast<uint32_t *>(out);
    if constexpr (LDSK) {
        // pass A: which positions stay (a bit per iteration in a register; seg <= 1280 -> <= 20 iterations), pass B: their
        // ids, re-read from the CSR row (coalesced, L2-hot), into the keys' own LDS
        uint32_t mask = 0;
        int mine = 0, ties_seen = tie_base;
        int it = 0;
        for (int i0 = b0; i0 < e0; i0 += PCG_WAVE, ++it) {
            const int i = i0 + lane;
            const bool in = i < e0;
            const uint32_t key = (in && !keep_all) ? keys[i] : 0u;
            bool s = in && (keep_all || key <= kstar);
            if (ranked) {
                const bool tie = in && key == kstar;
                const uint64_t tm = __ballot(tie);
                s = in && (key < kstar || (tie && ties_seen + __popcll(tm & lanemask_lt()) < need));
                ties_seen += __popcll(tm);
            }
            mask |= (uint32_t)s << it;
            mine += wave_count(s);
        }
        int run;
        grp_scan<NW>(mine, wave, lane, red, run, ns);                        // (its barriers part pass A's reads from pass B's writes)
        constexpr int CU = 4;
        it = 0;
        for (int i0 = b0; i0 < e0; i0 += CU * PCG_WAVE, it += CU) {
            uint32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                idv[u] = (uint32_t)nbr[i < e0 ? i : d - 1];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const bool s = (mask >> (it + u)) & 1u;
                const uint64_t sm = __ballot(s);
                const int at = run + __popcll(sm & lanemask_lt());
                if (s) {                                                     // (two typed stores instead of one through a generic pointer)
                    if (plain) out[at] = (int32_t)idv[u];
                    else keys[at] = idv[u];
                }
                run += __popcll(sm);
            }
        }
        __syncthreads();
    } else {
        // over-long row: count per stretch, scan, then the same pass again writing straight into the list region
        // ids and keys of KEY_UNROLL positions of this wave's stretch (the keys from the row's scratch)
        auto stretch_keys = [&](int i0, uint32_t (&id)[KEY_UNROLL], uint32_t (&key)[KEY_UNROLL]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < KEY_UNROLL; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                id[u] = (uint32_t)nbr[i < e0 ? i : d - 1];
                key[u] = gk[i < e0 ? i : d - 1];
            }
        };
        int mine = 0, ties_seen = tie_base;
        for (int i0 = b0; i0 < e0; i0 += PCG_WAVE * KEY_UNROLL) {
            uint32_t id[KEY_UNROLL], key[KEY_UNROLL];
            stretch_keys(i0, id, key);
#pragma unroll
            for (int u = 0; u < KEY_UNROLL; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                const bool in = i < e0;
                bool s = in && (keep_all || key[u] <= kstar);
                if (ranked) {
                    const bool tie = in && key[u] == kstar;
                    const uint64_t tm = __ballot(tie);
                    s = in && (key[u] < kstar || (tie && ties_seen + __popcll(tm & lanemask_lt()) < need));
                    ties_seen += __popcll(tm);
                }
                mine += wave_count(s);
            }
        }
        int run;
        grp_scan<NW>(mine, wave, lane, red, run, ns);
        ties_seen = tie_base;
        for (int i0 = b0; i0 < e0; i0 += PCG_WAVE * KEY_UNROLL) {
            uint32_t id[KEY_UNROLL], key[KEY_UNROLL];
            stretch_keys(i0, id, key);
#pragma unroll
            for (int u = 0; u < KEY_UNROLL; ++u) {
                const int i = i0 + u * PCG_WAVE + lane;
                const bool in = i < e0;
                bool s = in && (keep_all || key[u] <= kstar);
                if (ranked) {
                    const bool tie = in && key[u] == kstar;
                    const uint64_t tm = __ballot(tie);
                    s = in && (key[u] < kstar || (tie && ties_seen + __popcll(tm & lanemask_lt()) < need));
                    ties_seen += __popcll(tm);
                }
                const uint64_t sm = __ballot(s);
                if (s) selbuf[run + __popcll(sm & lanemask_lt())] = id[u];
                run += __popcll(sm);
            }
        }
        // the kept ids are read back (binary search of the minority picks) by every wave of this workgroup: same CU, same L1
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    PCG_STAMP(3);
    if (plain) {                                                             // everything the row has to report, by wave 0
        if (wave == 0) {
            if (lane == 0) {
                a.w.len[row] = ns;
                a.cnt[row] = ns;
            }
            const int nch = (rec_cap(p, 0) + CHUNK - 1) / CHUNK;
            for (int j = lane; j < nch; j += PCG_WAVE) {
                const int left = ns - j * CHUNK;
                a.w.chunk_desc[p.chunk0 + j].z = left < 0 ? 0 : (left > CHUNK ? CHUNK : left);
            }
        }
        PCG_STAMP(6);
        return;
    }
    // (early: the window is in LDS since before the first barrier of the k-th selection)
    if constexpr (LDSK) finish_row<NW>(a, row, p, c, keys, ns, false, wave, lane, red, keys_ok, sortw, early ? win : nullptr);
    else finish_row<NW>(a, row, p, c, selbuf, ns, !sel_in_lds, wave, lane, red, keys_ok, sortw, early ? win : nullptr);
}

// One persistent launch selects every row of the batch, longest rows first.  The work is one queue of workgroup-sized units,
//   [workgroup rows > 4096 | workgroup rows 513 .. 4096 | batches of eight consecutive single-wave items, one per wave],
// the single-wave items being [rows 65 .. 512 | rows 17 .. 64 and the short rows with a tail | groups of four short rows].
// Unit u belongs to shard u % 8.  Workgroup b starts on unit b (no atomic at all for a batch that fits the grid) and then
// pulls the further units of its shard b % 8 with one returning device-scope atomic per unit, issued a whole unit ahead of
// its use (its latency hides behind the work).  Eight head words on cache lines of their own keep the pulls of the 768
// workgroups from serialising on one address (one word serves ~90 atomics per microsecond); per-wave pulls would be
// thousands.  A workgroup busy with long rows simply pulls fewer units; nothing else is assigned in advance.
#ifndef PCG_WG_ROW_PRIO
#define PCG_WG_ROW_PRIO 2
#endif
constexpr int SEL_SHARDS = 8;
constexpr int SORT_TILE = 4096;          // keys per LDS tile of the in-kernel train-pos sort (32 KB of the 40 KB key area)

// The train positives' sort inside select_rows, shared by (nearly) every workgroup of the launch: the unsorted keys (formed
// beside the score pass, unique) come in groups of 64; workgroup w works on group w % n_sort and on slice w / n_sort of the key
// range: it counts, for each of its group's keys, the keys of its slice that are smaller (keys in LDS, the slice split over
// the waves, broadcast reads: rank_sort_body's inner loop), adds the 64 counts to the group's rank accumulators (device-scope
// atomics) and takes the group's ticket; the workgroup whose ticket is the last reads the complete ranks, stores the group's
// keys at their ranks (write-through), puts accumulators and ticket back to zero and counts the group in.  A few hundred
// compares per lane instead of the thousands a whole-range rank sort by ceil(P / 64) workgroups costs while the other
// workgroups of its CU compete for the same SIMDs: the keys are sorted ~4 us into the launch instead of ~9.
template <int TILE>
__device__ __forceinline__ void sort_share(const ChooseArgs &a, int w, uint64_t *sh, int *part, int tid) {
    constexpr int PER = TILE / (SEL_NW * PCG_WAVE);
    const int lane = tid & (PCG_WAVE - 1), wave = tid >> 6;
    const int P = a.g.n_pos;
    const int group = w % a.n_sort, slice = w / a.n_sort;
    const int i = group * PCG_WAVE + lane;
    const uint64_t *__restrict__ raw = a.raw_keys;
    const uint64_t mine = raw[i < P ? i : P - 1] | (i < P ? 0ull : ~0ull);
    const int jb = slice * a.sort_slice_len;
    const int je = jb + a.sort_slice_len < P ? jb + a.sort_slice_len : P;
    int c = 0;
    for (int t0 = jb; t0 < je; t0 += TILE) {
        const int nt = je - t0 < TILE ? je - t0 : TILE;
        uint64_t kt[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {                                    // (unconditional loads: index clamped, pad = all ones)
            const int t = tid + u * SEL_NW * PCG_WAVE;
            kt[u] = raw[t0 + (t < nt ? t : nt - 1)] | (t < nt ? 0ull : ~0ull);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) sh[tid + u * SEL_NW * PCG_WAVE] = kt[u];
        __syncthreads();
        const int chunk = (((nt + SEL_NW - 1) / SEL_NW) + 7) & ~7;
        const int j0 = wave * chunk;
        const int nt8 = (nt + 7) & ~7;
        const int j1 = j0 + chunk < nt8 ? j0 + chunk : nt8;
        const uint4 *sh4 = reinterpret_cast<const uint4 *>(sh);
        for (int j = j0; j < j1; j += 8) {
            const uint4 q0 = sh4[(j >> 1) + 0], q1 = sh4[(j >> 1) + 1], q2 = sh4[(j >> 1) + 2], q3 = sh4[(j >> 1) + 3];
            const uint64_t a0 = ((uint64_t)q0.y << 32) | q0.x, a1 = ((uint64_t)q0.w << 32) | q0.z;
            const uint64_t a2 = ((uint64_t)q1.y << 32) | q1.x, a3 = ((uint64_t)q1.w << 32) | q1.z;
            const uint64_t a4 = ((uint64_t)q2.y << 32) | q2.x, a5 = ((uint64_t)q2.w << 32) | q2.z;
            const uint64_t a6 = ((uint64_t)q3.y << 32) | q3.x, a7 = ((uint64_t)q3.w << 32) | q3.z;
            c += (a0 < mine) + (a1 < mine) + (a2 < mine) + (a3 < mine) + (a4 < mine) + (a5 < mine) + (a6 < mine) + (a7 < mine);
        }
    }
    part[wave * PCG_WAVE + lane] = c;
    __syncthreads();
    if (wave == 0) {
        int cnt = 0;
#pragma unroll
        for (int x = 0; x < SEL_NW; ++x) cnt += part[x * PCG_WAVE + lane];
        uint32_t ticket = 0;
        if (a.sort_slices > 1) {                                           // (one slice: the count is the rank)
            if (i < P) __hip_atomic_fetch_add(a.rank_acc + i, (uint32_t)cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the adds have been performed before the ticket is taken
            if (lane == 0) ticket = __hip_atomic_fetch_add(a.group_ticket + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
        }
        if (ticket == (uint32_t)a.sort_slices - 1u) {                      // the group's last slice: every share has been added
            if (i < P) {
                uint32_t rank = (uint32_t)cnt;
                if (a.sort_slices > 1) {
                    rank = __hip_atomic_load(a.rank_acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(a.rank_acc + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.sort_out) + (rank < (uint32_t)P ? rank : 0u),
                                   (unsigned long long)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (group == 0)
                for (int t = P + lane; t < a.sort_cap; t += PCG_WAVE)
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.sort_out) + t, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0 && a.sort_slices > 1) __hip_atomic_store(a.group_ticket + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's stores are complete: count the group in
            if (lane == 0) __hip_atomic_fetch_add(a.sort_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.stamps && lane == 0) atomicMax(&a.stamps[(size_t)a.g.n_rel * a.B * 8 + 5], (unsigned long long)wall_clock64());
        }
    }
    __syncthreads();                                                       // (the LDS is the row paths' again)
}

// ---------------------------------------------------------------------------------------------------------------------
// The label classifier's own training step for this batch (ClfStep, choose.h), by ONE workgroup of the select launch:
//   logits of the centres' feature rows (layers.py:230-243) -> cross entropy against the labels (model.py:54-61, the term
//   weighted lambda_1) -> d loss / d W, d b summed over the batch in a fixed order -> Adam (model_handler.py:153).
// The classifier's parameters get no gradient from anything this launch or the two after it compute, so the update of step t
// does not have to wait for step t's dense kernel - and the scores of step t + 1 can be formed beside step t's gather.
// Geometry as in the score pass: lanes_per_row lanes share a feature row, each its float4 chunk(s); row slots x waves x
// iterations add up in registers, then slots (butterfly), then waves (LDS, wave order).  lds: (SEL_NW + 1) * (2F + 2) floats.
// ---------------------------------------------------------------------------------------------------------------------
// (handed its few arguments by value: a reference to the kernel's argument block would be a stack copy of it.  Inlined: as a
// function of its own - callee-saved registers, arguments on the stack - it ran twice as long)
// K: float4 chunks of a row per lane (2 only for rows of more than 256 floats), U: rows in flight per lane.  Everything the
// step reads that depends on nothing else (the classifier, its Adam state, the step count, the batch's ids and labels) is
// requested up front; the ids and labels are staged in LDS a block of rows at a time, so that the feature rows of half a batch
// of 1024 are one memory round trip; every (wave, row slot) leaves its partial gradient in LDS and 2F + 2 threads add them up in
// (wave, slot) order - no cross-lane traffic.  lds: NC4 + SEL_NW * rpw * NC4 + 2 * RB words (NC4 = 2F + 2 rounded up to 4).
template <int K, int U>
__device__ __forceinline__ void clf_step_body(const ClfStep c, const float *__restrict__ X, int F, int stride,
                                           const int32_t *__restrict__ nodes, const int32_t *__restrict__ labels, int B,
                                           uint32_t *lds_u, int tid, unsigned long long *st, int cw) {
    // nodes / labels / B: THIS workgroup's slice of the batch (workgroup cw of c.n_wg)
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int NT = SEL_NW * PCG_WAVE;
    const int NC = 2 * F + 2, NC4 = (NC + 3) & ~3;
    const int lane = tid & (PCG_WAVE - 1), wave = tid >> 6;
    const int lpr = lanes_per_row(stride), rpw = PCG_WAVE / lpr, slot = lane / lpr, sub = lane % lpr, nch = stride >> 2;
    const int n_part = SEL_NW * rpw;                            // partial gradients: one per (wave, row slot)
    float *wl = reinterpret_cast<float *>(lds_u);              // [NC] the classifier this step scores / selects by
    float *red = wl + NC4;                                      // [n_part][NC4]
    int *idl = reinterpret_cast<int *>(red + n_part * NC4);     // [RB] a block of the batch's ids, [RB] its labels
    int RB = ((WG_KEYCAP - (n_part + 1) * NC4) / 2) & ~(PCG_WAVE - 1);          // (>= 2048: feat_stride <= 512, host-checked)
    RB = RB < 2 * NT ? RB : 2 * NT;                             // (a workgroup's slice is <= 1024 rows up to batches of 8192)
    int *yl = idl + RB;
    // this thread's parameter (threads < NC; a loop covers NC > NT), its Adam state, the step count: requested now
    const bool mine = tid < NC;
    const int ic = mine ? tid : 0;
    const float p_old = c.clf_next[ic], m_old = c.m[ic], v_old = c.v[ic];
    const float t = (float)(c.step_counter[0] + 1);             // this step's dense launch counts it; it has not run yet
    constexpr int NPRE = 2;                                     // ids / labels per thread of a block: RB <= NPRE * NT
    int nb = B < RB ? B : RB;
    int id_pre[NPRE], y_pre[NPRE];
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {                            // the first block's ids / labels (clamped: unconditional loads)
        const int i = tid + q * NT;
        id_pre[q] = nodes[i < nb ? i : nb - 1];
        y_pre[q] = labels[i < nb ? i : nb - 1];
    }
    for (int i = tid; i < NC; i += NT) {
        const float w = i == tid ? p_old : c.clf_next[i];
        wl[i] = w;
        if (cw == 0) c.theta_clf[i] = w;                        // what this step's dense kernel computes the loss term with
    }
#pragma unroll
    for (int q = 0; q < NPRE; ++q) {
        const int i = tid + q * NT;
        if (i < nb) {
            idl[i] = id_pre[q];
            yl[i] = y_pre[q];
        }
    }
    __syncthreads();
    if (st && tid == 0) st[0] = wall_clock64();
    bool has[K];
    int chc[K];
    float w0[K][4], w1[K][4];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int ch = sub + k * lpr;
        has[k] = ch < nch;
        chc[k] = has[k] ? ch : nch - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 4 * chc[k] + j;
            const bool ok = has[k] && f < F;
            w0[k][j] = ok ? wl[f] : 0.f;
            w1[k][j] = ok ? wl[F + f] : 0.f;
        }
    }
    const float b0 = wl[2 * F], b1 = wl[2 * F + 1];
    float g0[K][4], g1[K][4], gb0 = 0.f, gb1 = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) g0[k][j] = g1[k][j] = 0.f;
    const int per_iter = SEL_NW * rpw;                          // rows the workgroup covers per load round
    for (int blk = 0; blk < B; blk += RB) {
        if (blk > 0) {                                          // (batches beyond one block: staged here, one more round trip each)
            nb = B - blk < RB ? B - blk : RB;
            __syncthreads();
            for (int i = tid; i < nb; i += NT) {
                idl[i] = nodes[blk + i];
                yl[i] = labels[blk + i];
            }
            __syncthreads();
        }
        for (int base = wave * rpw; base < nb; base += per_iter * U) {
            f4 x[U][K];
            int y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                       // (unconditional loads: index clamped, the result not counted)
                const int b = base + u * per_iter + slot;
                const int bc = b < nb ? b : nb - 1;
                y[u] = b < nb ? yl[bc] : -1;
                const float *xrow = X + (size_t)idl[bc] * stride;
#pragma unroll
                for (int k = 0; k < K; ++k) x[u][k] = *reinterpret_cast<const f4 *>(xrow + 4 * chc[k]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    p0 = fmaf(x[u][k].x, w0[k][0], p0); p0 = fmaf(x[u][k].y, w0[k][1], p0);
                    p0 = fmaf(x[u][k].z, w0[k][2], p0); p0 = fmaf(x[u][k].w, w0[k][3], p0);
                    p1 = fmaf(x[u][k].x, w1[k][0], p1); p1 = fmaf(x[u][k].y, w1[k][1], p1);
                    p1 = fmaf(x[u][k].z, w1[k][2], p1); p1 = fmaf(x[u][k].w, w1[k][3], p1);
                }
                p0 = score_reduce(p0, lpr);
                p1 = score_reduce(p1, lpr);
                float loss, d0, d1;
                xent2(p0 + b0, p1 + b1, y[u], loss, d0, d1);
                d0 = y[u] >= 0 ? d0 * c.scale : 0.f;
                d1 = y[u] >= 0 ? d1 * c.scale : 0.f;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    g0[k][0] = fmaf(d0, x[u][k].x, g0[k][0]); g0[k][1] = fmaf(d0, x[u][k].y, g0[k][1]);
                    g0[k][2] = fmaf(d0, x[u][k].z, g0[k][2]); g0[k][3] = fmaf(d0, x[u][k].w, g0[k][3]);
                    g1[k][0] = fmaf(d1, x[u][k].x, g1[k][0]); g1[k][1] = fmaf(d1, x[u][k].y, g1[k][1]);
                    g1[k][2] = fmaf(d1, x[u][k].z, g1[k][2]); g1[k][3] = fmaf(d1, x[u][k].w, g1[k][3]);
                }
                gb0 += d0;
                gb1 += d1;
            }
        }
    }
    if (st && tid == 0) st[1] = wall_clock64();
    {
        float *mine_red = red + (wave * rpw + slot) * NC4;
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int f = 4 * chc[k] + j;
                if (has[k] && f < F) {
                    mine_red[f] = g0[k][j];
                    mine_red[F + f] = g1[k][j];
                }
            }
        if (sub == 0) {
            mine_red[2 * F] = gb0;
            mine_red[2 * F + 1] = gb1;
        }
    }
    __syncthreads();
    if (st && tid == 0) st[2] = wall_clock64();
    if (c.n_wg > 1) {
        // this workgroup's gradient -> its slot (write-through: another workgroup of this launch reads it), then the ticket
        float *slot_out = c.part + (size_t)cw * c.part_stride;
        for (int i = tid; i < NC; i += NT) {
            float g = red[i];
            for (int w = 1; w < n_part; ++w) g += red[w * NC4 + i];
            __hip_atomic_store(slot_out + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // every thread's stores are complete
        int *flag = reinterpret_cast<int *>(red);                // (the partial sums have been read)
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(c.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)c.n_wg - 1u;
        __syncthreads();
        const bool last = flag[0] != 0;
        __syncthreads();                                         // (red is written again below)
        if (!last) return;
        // the last one in: all gradients, in workgroup order (loads past the L1: agent scope; all of a parameter's in flight)
        for (int i = tid; i < NC; i += NT) {
            float x[8];
#pragma unroll
            for (int w = 0; w < 8; ++w)
                x[w] = __hip_atomic_load(c.part + (size_t)(w < c.n_wg ? w : c.n_wg - 1) * c.part_stride + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            float g = x[0];
#pragma unroll
            for (int w = 1; w < 8; ++w) g = w < c.n_wg ? g + x[w] : g;
            red[i] = g;
        }
        if (tid == 0) __hip_atomic_store(c.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const float bc1 = 1.f - powf(c.h.beta1, t), bc2 = 1.f - powf(c.h.beta2, t);
    for (int i = tid; i < NC; i += NT) {
        float g = red[i];
        if (c.n_wg == 1)
            for (int w = 1; w < n_part; ++w) g += red[w * NC4 + i];
        const bool first = i == tid;                             // (its state came with the first loads)
        const float p = first ? p_old : wl[i];
        const float mo = first ? m_old : c.m[i], vo = first ? v_old : c.v[i];
        g = fmaf(c.h.wd, p, g);                                 // torch.optim.Adam, coupled weight decay (as adam_apply_one)
        const float mi = c.h.beta1 * mo + (1.f - c.h.beta1) * g;
        const float vi = c.h.beta2 * vo + (1.f - c.h.beta2) * g * g;
        c.m[i] = mi;
        c.v[i] = vi;
        const float denom = sqrtf(vi) / sqrtf(bc2) + c.h.eps;
        c.clf_next[i] = p - (c.h.lr / bc1) * (mi / denom);
    }
    __syncthreads();                                             // (the LDS is the row paths' again)
}

// CLF: the launch carries the label classifier's step (training: pcg_choose_gather_train): 1 = feature rows of up to 256 floats
// (one float4 chunk per lane: the datasets' 128-B rows), 2 = wider rows.  Instantiation 0 holds none of that code - nor the
// few bytes of scratch it spills under the row paths' register budget -, so every other caller's launch is the row paths' alone.
template <int CLF>
__global__ void __launch_bounds__(SEL_NW *PCG_WAVE) __attribute__((amdgpu_waves_per_eu(6, 8))) select_rows(const ChooseArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t *hist = lds + a.key_cap;
    uint32_t *cand = hist + HIST_WG;
    int *red = reinterpret_cast<int *>(cand + PCG_WAVE);                   // 2 * SEL_NW + 2 + 4 ints, then 2 claim slots
    int *claim = red + 2 * SEL_NW + 6;
    // per-relation neighbour arrays in LDS: a per-lane relation index (four short rows per wave) then costs one ds_read
    // instead of a waterfall over the kernel arguments
    const int32_t **t_indices = reinterpret_cast<const int32_t **>(red + 2 * SEL_NW + 8);
    int *sortw = reinterpret_cast<int *>(t_indices + PCG_MAX_REL);         // 4 + KIDX_MAX words: wait_sorted_keys
    if (threadIdx.x < PCG_MAX_REL) t_indices[threadIdx.x] = a.g.indices[threadIdx.x < (unsigned)a.g.n_rel ? threadIdx.x : 0];
    const bool leader = threadIdx.x == 0;
    if (leader) sortw[0] = sortw[1] = 0;
    const int n16 = (int)a.w.counters[C_N16], n4 = (int)a.w.counters[C_N4];
    const int n1 = (int)a.w.counters[C_N1], n0 = (int)a.w.counters[C_N0], na = (int)a.w.counters[C_NA];
    const int n_wg = n16 + n4, n_items = n1 + n0 + (na + 3) / 4;
    // single-wave items per unit: eight (one per wave) when there is plenty of them, fewer when the batch is so small that
    // the items can be spread over more workgroups (more CUs' load paths) than eight per workgroup would use
    // (training: workgroup 0 does the label classifier's step and nothing else; the row workgroups are the others)
    const int clf_wg = (CLF && a.clf.clf_next) ? a.clf.n_wg : 0;
    const int grid = (int)gridDim.x - clf_wg, bid = (int)blockIdx.x - clf_wg;
    const int avail = grid - n_wg > grid / 4 ? grid - n_wg : grid / 4;
    int bs = (n_items + avail - 1) / avail;
    bs = bs < 1 ? 1 : (bs > SEL_NW ? SEL_NW : bs);
    const int n_units = n_wg + (n_items + bs - 1) / bs;
    const bool pull = n_units > grid;                                      // otherwise unit = workgroup: no atomics at all
    const int shard = (bid < 0 ? 0 : bid) % SEL_SHARDS;
    uint32_t *head = a.w.heads + 16 * shard;                               // (64 bytes apart)
    if (a.pending_clear && bid == 0 && leader) a.pending_clear[0] = 0u;   // the launch before has applied the deferred update
    __syncthreads();
    // the label classifier's step for this batch (training): one workgroup, before its share of the rows
    if (CLF && bid < 0) {
        // (one workgroup, as long as the launch's longest rows, sharing its CU with two workgroups of rows: it goes first)
        __builtin_amdgcn_s_setprio(3);
        if (a.stamps && leader) a.stamps[(size_t)a.g.n_rel * a.B * 8 + 2] = wall_clock64();
        // this workgroup's slice of the batch: whole multiples of 64 rows (every workgroup has some: n_wg <= ceil(B / 1024))
        const int cw = (int)blockIdx.x;
        const int per = ((a.B + a.clf.n_wg - 1) / a.clf.n_wg + PCG_WAVE - 1) & ~(PCG_WAVE - 1);
        const int r0 = cw * per < a.B ? cw * per : a.B, r1 = r0 + per < a.B ? r0 + per : a.B;
        unsigned long long *cst = (a.stamps && cw == 0) ? a.stamps + (size_t)a.g.n_rel * a.B * 8 + 8 : nullptr;
        if constexpr (CLF == 2) clf_step_body<2, 2>(a.clf, a.g.X, a.g.feat_dim, a.g.feat_stride, a.nodes + r0, a.labels + r0, r1 - r0, lds, (int)threadIdx.x, cst, cw);
        else if constexpr (CLF == 1) clf_step_body<1, 3>(a.clf, a.g.X, a.g.feat_dim, a.g.feat_stride, a.nodes + r0, a.labels + r0, r1 - r0, lds, (int)threadIdx.x, cst, cw);
        if (a.stamps && leader) a.stamps[(size_t)a.g.n_rel * a.B * 8 + 3] = wall_clock64();
        return;
    }

    int u = bid;
    // The previous step's weight gradients (GEMMs over its batch) + Adam, two tiles per workgroup, before its rows (CLF launches
    // of small batches: choose.h).  Nothing in this launch reads what they write (the dense launch, two launches on, does).
    if (CLF && a.n_wg_units > 0) {
        if (u < a.n_wg_units) {
            float(*wred)[256] = reinterpret_cast<float(*)[256]>(lds + ((int)threadIdx.x >> 8) * 1024);
            const int n_tiles = wgrad_tiles(a.wg.F, a.wg.E, a.wg.R, 0);
            const int tile = 2 * u + ((int)threadIdx.x >> 8);
            // (an odd tile count: the last unit's second half works on the last tile again - same result, stored twice)
            wgrad_adam_body<4>(a.wg, tile < n_tiles ? tile : n_tiles - 1, wred, (int)threadIdx.x & 255);
            __syncthreads();                                               // (the LDS is the row paths' again)
        }
        u = u < a.n_wg_units ? grid - a.n_wg_units + u : u - a.n_wg_units;
    }
    // The train positives' sort, inside this launch and shared by its workgroups (sort_share), before the rows: only rows with
    // minority picks ever wait for the result (wait_sorted_keys), and they are busy with their own distance keys meanwhile.
    if (a.n_sort > 0) {
        const int helpers = a.n_sort * a.sort_slices;
        if (u < helpers) {
            if (a.stamps && leader && u == 0) a.stamps[(size_t)a.g.n_rel * a.B * 8 + 4] = wall_clock64();
            __builtin_amdgcn_s_setprio(3);                                 // (the other workgroups of this CU are busy with their rows)
            if (a.sort_slice_len <= 512) sort_share<512>(a, u, reinterpret_cast<uint64_t *>(lds), reinterpret_cast<int *>(hist), (int)threadIdx.x);
            else sort_share<SORT_TILE>(a, u, reinterpret_cast<uint64_t *>(lds), reinterpret_cast<int *>(hist), (int)threadIdx.x);
            __builtin_amdgcn_s_setprio(0);
        }
        // the sorting workgroups start late: they take the LAST of the first `grid` units (short rows), the others move up -
        // the longest rows (the first units) start at once
        u = u < helpers ? grid - helpers + u : u - helpers;
    }
    int keys_ok = 0;                                                       // this wave has seen the sorted keys (wave-uniform)
    int pending = 0, slot = 0;
    while (u < n_units) {
        // the unit after this one: behind a batch of single-wave items (a few microseconds) its claim is in flight while the
        // batch runs; behind a workgroup row (up to tens of microseconds) it is made afterwards - a busy workgroup must not sit
        // on a unit that an idle one could take
        const bool ahead = u >= n_wg;
        if (leader && pull && ahead) pending = grid + shard + SEL_SHARDS * (int)atomicAdd(head, 1u);
        // the thread index goes through an opaque move once per unit: everything the row paths derive from it (lane masks,
        // quarter-wave indices, LDS offsets) is then recomputed per unit - a few VALU ops - instead of being hoisted out of
        // this loop and kept alive across every path, which costs more registers than the 80 the occupancy allows (spills)
        int tid = (int)threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & (PCG_WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        uint32_t *area = lds + wave * WAVE_AREA;
        if (u < n_wg) {
            const int row = __builtin_amdgcn_readfirstlane(u < n16 ? a.w.q16[u] : a.w.q4[u - n16]);     // one row per workgroup: scalar
            const int d = a.w.recs[row].d;
            // (a workgroup row is the launch's long pole and shares its CU with two workgroups of short rows: it goes first)
            static_assert(true, "");
            if (PCG_WG_ROW_PRIO) __builtin_amdgcn_s_setprio(PCG_WG_ROW_PRIO);
            if (d <= a.key_cap) select_wg_row<true>(a, row, lds, hist, cand, red, tid, keys_ok, sortw, nullptr, a.key_cap);      // (longer rows: select_long_rows)
            if (PCG_WG_ROW_PRIO) __builtin_amdgcn_s_setprio(0);
        } else {
            const int j = wave < bs ? (u - n_wg) * bs + wave : n_items;
            if (j < n1) {
                select_wave_row(a, __builtin_amdgcn_readfirstlane(a.w.q1[j]), area, lane, keys_ok, sortw);
            } else if (j < n1 + n0) {
                select_lane_row(a, __builtin_amdgcn_readfirstlane(a.w.q0[j - n1]), area, lane, keys_ok, sortw);
            } else if (j < n_items) {
                select_four_short_rows(a, 4 * (j - n1 - n0), na, t_indices, lane);
            }
        }
        if (!pull) break;
        if (leader) claim[slot] = ahead ? pending : grid + shard + SEL_SHARDS * (int)atomicAdd(head, 1u);
        __syncthreads();
        u = claim[slot];
        slot ^= 1;
    }
    // the last workgroup out puts the heads back to zero, so that the kernel can be launched again on the same plan
    if (leader && pull) {
        const unsigned done = atomicAdd(a.w.heads + 15, 1u);
        if (done == (unsigned)grid - 1u) {
            for (int i = 0; i < SEL_SHARDS; ++i) atomicExch(a.w.heads + 16 * i, 0u);
            atomicExch(a.w.heads + 15, 0u);
        }
    }
}

// Rows beyond the LDS key capacity (> WG_KEYCAP neighbours), in a launch of their own: the multi-pass selection over keys kept
// in global scratch needs more registers than select_rows' occupancy allows, and only hub-heavy graphs have such rows (the
// launch is skipped when the graph's maximum degree rules them out).  Persistent workgroups walk the > 4096 queue and take
// the rows that select_rows left alone; workgroup b keeps its keys in scratch[b * per_wg ..].
// (LONG_NW = 16 waves per long row: half the gathers per lane; LONG_KEYCAP keys in LDS - one workgroup per CU - so that a row
//  of up to 32768 neighbours never leaves the CU: at 10 M nodes / 200 M edges the rows of 10 - 18 K neighbours, ~170 per batch,
//  took 60 us each with their keys in global scratch - written by pass 1, re-read by every histogram round and by both
//  compaction passes)
constexpr int LONG_BLOCKS = 256;
__global__ void __launch_bounds__(LONG_NW *PCG_WAVE) select_long_rows(const ChooseArgs a, int64_t per_wg) {
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *lds = reinterpret_cast<uint32_t *>(smem);                   // LONG_KEYCAP words: keys, then the kept ids | 4096 bins | cand | red
    uint32_t *hist = lds + LONG_KEYCAP;
    uint32_t *cand = hist + 4 * LONG_NW * PCG_WAVE;
    int *red = reinterpret_cast<int *>(cand + PCG_WAVE);
    const int n16 = (int)a.w.counters[C_N16];
    uint32_t *gk = a.w.key_scratch ? a.w.key_scratch + (size_t)blockIdx.x * per_wg : nullptr;     // (no scratch: no row beyond LONG_KEYCAP)
    // The queue is in row order, the long rows are anywhere in it: a static stride would hand some workgroups three or four of
    // them and most none.  Every workgroup therefore pulls queue positions from one cursor (heads[14]; a few hundred atomics
    // in all) - AFTER it is done with its unit, not a unit ahead: units cost nothing or 60 us here, and a workgroup busy with
    // a long row must not sit on a claim.  The last workgroup out puts the cursor back to zero.
    int *claim = red + 2 * LONG_NW + 6;
    uint32_t *cursor = a.w.heads + 14, *done = a.w.heads + 13;
    const bool leader = threadIdx.x == 0;
    int u = (int)blockIdx.x, slot = 0;
    int keys_ok = 1;                                                        // (select_rows, the launch before, has sorted them)
    while (u < n16) {
        const int row = __builtin_amdgcn_readfirstlane(a.w.q16[u]);
        const int d = a.w.recs[row].d;                                      // (workgroup-uniform)
        if (d > LONG_KEYCAP) select_wg_row<false, LONG_NW>(a, row, lds, hist, cand, red, (int)threadIdx.x, keys_ok, nullptr, gk, LONG_KEYCAP);
        else if (d > a.key_cap) select_wg_row<true, LONG_NW>(a, row, lds, hist, cand, red, (int)threadIdx.x, keys_ok, nullptr, nullptr, LONG_KEYCAP);
        __syncthreads();
        if (leader) claim[slot] = (int)gridDim.x + (int)atomicAdd(cursor, 1u);
        __syncthreads();
        u = claim[slot];
        slot ^= 1;
    }
    if (leader) {
        const unsigned fin = atomicAdd(done, 1u);
        if (fin == gridDim.x - 1) {
            atomicExch(cursor, 0u);
            atomicExch(done, 0u);
        }
    }
}

static size_t select_smem_bytes(int key_cap) {
    return sizeof(uint32_t) * (key_cap + HIST_WG + PCG_WAVE) + sizeof(int) * (2 * SEL_NW + 8) + sizeof(void *) * PCG_MAX_REL +   // (claim[2] = red[22..23])
           sizeof(int) * (4 + KIDX_MAX);                                                                                            // sortw
}

int launch_select_rows(const ChooseArgs &a, hipStream_t st) {
    static_assert(HIST_WG == 4 * SEL_NW * PCG_WAVE, "one uint4 of bins per thread");
    static_assert(HIST_W == 4 * PCG_WAVE, "one uint4 of bins per lane");
    static_assert(WAVE_AREA >= HIST_W + T1_CAP + PCG_WAVE, "a wave's LDS area: histogram | kept ids | candidates");
    static_assert(((BIG_KEYCAP / SEL_NW + PCG_WAVE - 1) / PCG_WAVE) <= 32, "pass A keeps one bit per iteration in a uint32");
    static_assert((2 * SEL_NW + 8) % 2 == 0, "the pointer table behind red stays 8-byte aligned");
    static_assert(SORT_TILE * 2 <= WG_KEYCAP && SEL_NW * PCG_WAVE <= HIST_WG, "the in-kernel sort's tile and partial counts fit the row paths' LDS");
    static int blocks = 0;
    if (!blocks) {                      // (tuning knob: PCG_SEL_BLOCKS = persistent workgroups, a multiple of 8, at most 3 per CU)
        const char *e = getenv("PCG_SEL_BLOCKS");
        const int v = e ? atoi(e) : 0;
        blocks = (v >= SEL_SHARDS && v <= SEL_BLOCKS && v % SEL_SHARDS == 0) ? v : SEL_BLOCKS;
    }
    ChooseArgs as = a;
    // Option PCG_SEL_BIG=1 (off: measured slower): on hub-heavy graphs (rows beyond WG_KEYCAP exist) two workgroups per CU with
    // BIG_KEYCAP keys of LDS each instead of three with WG_KEYCAP - rows up to 16384 neighbours are then done HERE, by whichever
    // workgroup pulls them, and select_long_rows only sees longer rows still.  10 M nodes / 200 M edges, batch 4096: select_rows
    // 59.6 -> 84.0 us (a third fewer waves for the short rows), select_long_rows 60.4 -> 57.2 (its longest rows are beyond 16384
    // as well): 141 vs 120 us (profiles/r04/x_select_big_lds_powerlaw_10m_kernel_stats.csv).
    static int big_knob = -1;
    if (big_knob < 0) {
        const char *e = getenv("PCG_SEL_BIG");
        big_knob = e ? atoi(e) : 0;
    }
    const bool big = big_knob && a.g.max_degree > WG_KEYCAP;
    as.key_cap = big ? BIG_KEYCAP : WG_KEYCAP;
    const int nblk = big ? (blocks > 512 ? 512 : blocks) : blocks;
    const size_t smem = select_smem_bytes(as.key_cap);
    if (big) {
        static bool attr_big = false;
        if (!attr_big) {
            const void *ks[3] = {reinterpret_cast<const void *>(select_rows<0>), reinterpret_cast<const void *>(select_rows<1>),
                                 reinterpret_cast<const void *>(select_rows<2>)};
            for (const void *k : ks)
                if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return PCG_E_LAUNCH;
            attr_big = true;
        }
    }
    const int row_blocks = nblk - (a.clf.clf_next ? a.clf.n_wg : 0);      // (training: one of the workgroups steps the label classifier)
    if (a.n_sort > 0) {                      // how the sort is shared: slices per key group, keys per slice (>= 128)
        if (a.n_sort > row_blocks) return PCG_E_ARG;
        // (PCG_SORT_SLICES: tuning knob.  More slices = fewer compares per workgroup, but accumulator atomics and a ticket hop,
        //  and more workgroups that start on their rows late; measured on the YelpChi-like batch: see DESIGN.md)
        static int knob = -1;
        if (knob < 0) {
            const char *e = getenv("PCG_SORT_SLICES");
            knob = e ? atoi(e) : 0;
        }
        int slices = knob > 0 ? knob : 3;   // (measured, batch 4096: power-law 2 M / 8000 keys 1: 142.1, 2: 140.2, 3: 133.3, 4: 136.8, 6: 143.0 us per step; emb 128 / 2670 keys 2: 93.4, 3: 93.5, 4: 94.1, 6: 95.4)
        if (slices > row_blocks / a.n_sort) slices = row_blocks / a.n_sort;
        const int most = (a.g.n_pos + 127) / 128;
        slices = slices > most ? most : slices;
        as.sort_slices = slices < 1 ? 1 : slices;
        as.sort_slice_len = (a.g.n_pos + as.sort_slices - 1) / as.sort_slices;
    }
    if (a.clf.clf_next && a.g.feat_stride > 256) hipLaunchKernelGGL(select_rows<2>, dim3(nblk), dim3(SEL_NW * PCG_WAVE), smem, st, as);
    else if (a.clf.clf_next) hipLaunchKernelGGL(select_rows<1>, dim3(nblk), dim3(SEL_NW * PCG_WAVE), smem, st, as);
    else hipLaunchKernelGGL(select_rows<0>, dim3(nblk), dim3(SEL_NW * PCG_WAVE), smem, st, as);
    PCG_LAUNCH_CHECK();
    if (a.g.max_degree > as.key_cap) {       // rows too long for the LDS keys: their own launch (hub-heavy graphs only)
        const int64_t per_wg = a.g.max_degree;
        int64_t nb = LONG_BLOCKS;                // (rows beyond LONG_KEYCAP keep their keys in scratch: as many workgroups as it holds)
        if (a.g.max_degree > LONG_KEYCAP) {
            nb = a.w.scratch_cap / per_wg;
            nb = nb < LONG_BLOCKS ? nb : LONG_BLOCKS;
            if (nb < 1) return PCG_E_ARG;
        }
        static_assert(LONG_KEYCAP == LONG_NW * PCG_WAVE * 32, "pass A of an LDS row keeps one bit per iteration in a uint32");
        const size_t long_smem = sizeof(uint32_t) * (LONG_KEYCAP + 4 * LONG_NW * PCG_WAVE + PCG_WAVE) + sizeof(int) * (2 * LONG_NW + 8);
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(select_long_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)long_smem) != hipSuccess)
                return PCG_E_LAUNCH;
            attr_done = true;
        }
        ChooseArgs al = as;                  // (the keys are sorted by now: nothing to sort, nothing to wait for)
        al.n_sort = 0;
        al.pending_clear = nullptr;
        hipLaunchKernelGGL(select_long_rows, dim3((int)nb), dim3(LONG_NW * PCG_WAVE), long_smem, st, al, per_wg);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

}  // namespace pcg
