// Shared device helpers for the gfx950 PC-GNN kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pcgnn.h"

#define PCG_WAVE 64

#define PCG_LAUNCH_CHECK()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return PCG_E_LAUNCH; \
    } while (0)

namespace pcg {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (PCG_WAVE - 1); }

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// number of lanes in the wave for which pred holds (wave-uniform result)
__device__ __forceinline__ int wave_count(bool pred) { return __popcll(__ballot(pred)); }

// |c - s| as an unsigned key: for non-negative floats the bit pattern orders like the value.
__device__ __forceinline__ uint32_t dist_key(float c, float s) { return __float_as_uint(fabsf(c - s)); }

// total order on floats as uint32 (negative < positive), used for the train-pos sort
__device__ __forceinline__ uint32_t orderable(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_orderable(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// lanes-per-row for a row of `stride` floats read as float4: power of two in [8, 64]
__host__ __device__ __forceinline__ int lanes_per_row(int stride) {
    int q = stride >> 2;
    int l = 8;
    while (l < q && l < 64) l <<= 1;
    return l;
}

// Partial dot of one feature row with a weight vector, in a fixed order: lane `sub`
// of `lpr` walks float4 chunks sub, sub+lpr, ...; inside a chunk a 4-long fma chain.
// The cross-lane butterfly that finishes it is in score_reduce().  Both score
// kernels use exactly these two functions, so their column 0 agrees bit for bit.
template <bool STREAM = false>
__device__ __forceinline__ float score_partial(const float *__restrict__ xrow, const float *__restrict__ w,
                                               int feat_dim, int stride, int sub, int lpr) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    float p = 0.f;
    for (int ch = sub; ch < (stride >> 2); ch += lpr) {
        // STREAM: the whole table is read once per step - non-temporal, so it does not evict what the
        // following kernels re-use from L2 / Infinity Cache
        const f4 xv = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xrow + 4 * ch))
                             : *reinterpret_cast<const f4 *>(xrow + 4 * ch);
        const float4 x = make_float4(xv.x, xv.y, xv.z, xv.w);
        const int f = 4 * ch;
        const float w0 = (f + 0 < feat_dim) ? w[f + 0] : 0.f;
        const float w1 = (f + 1 < feat_dim) ? w[f + 1] : 0.f;
        const float w2 = (f + 2 < feat_dim) ? w[f + 2] : 0.f;
        const float w3 = (f + 3 < feat_dim) ? w[f + 3] : 0.f;
        p = fmaf(x.x, w0, p);
        p = fmaf(x.y, w1, p);
        p = fmaf(x.z, w2, p);
        p = fmaf(x.w, w3, p);
    }
    return p;
}
// add the value of a DPP-selected neighbour lane (full-rate VALU, no LDS crossbar round trip)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float p) {
    return p + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(p), CTRL, 0xF, 0xF, false));
}
// Butterfly over the lpr lanes that share a row.  Steps 1, 2 use quad permutes; steps 4 and 8 use the
// half-row / row mirror: after the previous steps every lane of a group holds its group's sum, so adding the
// mirrored lane's value adds the neighbouring group's sum - the same two operands as the xor partner's.
__device__ __forceinline__ float score_reduce(float p, int lpr) {
    p = dpp_add<0xB1>(p);                 // quad_perm [1,0,3,2]  (lane ^ 1)
    p = dpp_add<0x4E>(p);                 // quad_perm [2,3,0,1]  (lane ^ 2)
    p = dpp_add<0x141>(p);                // row_half_mirror      (other quad of the 8-lane group)
    if (lpr > 8) p = dpp_add<0x140>(p);   // row_mirror           (other half of the 16-lane row)
    for (int o = 16; o < lpr; o <<= 1) p += __shfl_xor(p, o);
    return p;
}

// ---- bodies shared by the stand-alone kernels and the fused step-front kernels (same code => same bits) ----------

#ifndef PCG_SCORE_UNROLL        // (tuning knobs of the score stream, overridable for scripts/score_bw.py)
#define PCG_SCORE_UNROLL 8
#endif
#ifndef PCG_SCORE_BLOCKS_PER_CU
#define PCG_SCORE_BLOCKS_PER_CU 4
#endif
#ifndef PCG_SCORE_NT
#define PCG_SCORE_NT 1
#endif
constexpr int SCORE_UNROLL = PCG_SCORE_UNROLL;

// class-0 logit of rows [row_begin, row_end): workgroup `block` of `n_blocks` (256 threads each), grid-stride.
// row_ids == nullptr: s0[row] = score(row);  else: s0[row_ids[row]] = score(row), rows with row_ids[row] < 0 skipped (the
// partitioned path: a rank's table rows are owned / train-pos / halo rows, scores are looked up by global node id)
// SCORE_K: float4 chunks of a row a lane may hold (2 covers rows of up to 512 floats; 1 - rows of up to 256 - halves the
// registers the rows in flight take)
template <int SCORE_K = 2>
__device__ __forceinline__ void score_table_body(const float *__restrict__ X, int feat_dim, int stride,
                                                 const float *__restrict__ W, const float *__restrict__ bias,
                                                 int64_t row_begin, int64_t row_end, float *__restrict__ s0, int block,
                                                 int n_blocks, const int32_t *__restrict__ row_ids = nullptr) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const int64_t wave_global = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)n_blocks * (blockDim.x >> 6);
    const float b0 = bias[0];
    const int64_t rows_per_iter = (int64_t)rpw * SCORE_UNROLL;
    // A lane's share of a row is the same float4 chunk(s) of every row (sub, sub + lpr: feat_stride <= 512 floats => at most
    // two), so its weights are loaded once.  All loads are unconditional - index clamped, value discarded by a select - because
    // a load inside a conditional is compiled into a branch that first waits for every load in flight: the row loads of the
    // unrolled iterations would go out one at a time.  Same fma order as score_partial => the same bits.
    const int nch = stride >> 2;
    bool has[SCORE_K];
    int chc[SCORE_K];
    float wv[SCORE_K][4];
#pragma unroll
    for (int k = 0; k < SCORE_K; ++k) {
        const int ch = sub + k * lpr;
        has[k] = ch < nch;
        chc[k] = has[k] ? ch : nch - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 4 * chc[k] + j;
            // (masked with AND rather than replaced by a select: a value that is only used under a condition gets its load
            //  sunk into a branch again)
            const float wj = W[f < feat_dim ? f : feat_dim - 1];
            wv[k][j] = __int_as_float(__float_as_int(wj) & ((has[k] && f < feat_dim) ? -1 : 0));
        }
    }
    const bool two = SCORE_K > 1 && lpr < nch;                            // (wave-uniform)
    for (int64_t base = row_begin + wave_global * rows_per_iter; base < row_end; base += n_waves * rows_per_iter) {
        f4 xv[SCORE_UNROLL][SCORE_K];
#pragma unroll
        for (int u = 0; u < SCORE_UNROLL; ++u) {
            const int64_t row = base + (int64_t)u * rpw + slot;
            const float *xrow = X + (row < row_end ? row : row_end - 1) * stride;
            xv[u][0] = PCG_SCORE_NT != 0 ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xrow + 4 * chc[0]))
                                         : *reinterpret_cast<const f4 *>(xrow + 4 * chc[0]);
            if constexpr (SCORE_K > 1)
                if (two)
                    xv[u][1] = PCG_SCORE_NT != 0 ? __builtin_nontemporal_load(reinterpret_cast<const f4 *>(xrow + 4 * chc[1]))
                                                 : *reinterpret_cast<const f4 *>(xrow + 4 * chc[1]);
        }
        float p[SCORE_UNROLL];
#pragma unroll
        for (int u = 0; u < SCORE_UNROLL; ++u) {
            float q = 0.f;
            if (has[0]) {
                q = fmaf(xv[u][0].x, wv[0][0], q);
                q = fmaf(xv[u][0].y, wv[0][1], q);
                q = fmaf(xv[u][0].z, wv[0][2], q);
                q = fmaf(xv[u][0].w, wv[0][3], q);
            }
            if constexpr (SCORE_K > 1)
                if (two && has[1]) {
                    q = fmaf(xv[u][1].x, wv[1][0], q);
                    q = fmaf(xv[u][1].y, wv[1][1], q);
                    q = fmaf(xv[u][1].z, wv[1][2], q);
                    q = fmaf(xv[u][1].w, wv[1][3], q);
                }
            p[u] = q;
        }
        // after the butterfly every lane of a row-group holds that row's sum: lane `sub` keeps the result of
        // unrolled row `sub`, so the wave writes its rpw * SCORE_UNROLL consecutive scores in ONE store
        float mine = 0.f;
#pragma unroll
        for (int u = 0; u < SCORE_UNROLL; ++u) {
            const float s = score_reduce(p[u], lpr);
            if (sub == u) mine = s;
        }
        if (sub < SCORE_UNROLL) {
            const int64_t row = base + (int64_t)sub * rpw + slot;
            if (row < row_end) {
                if (row_ids) {
                    const int32_t id = row_ids[row];
                    if (id >= 0) s0[id] = mine + b0;
                } else {
                    s0[row] = mine + b0;
                }
            }
        }
    }
}

// The same scores for the rows a byte map marks (touched[row] != 0), and only for them: what a large graph's training step
// needs - a batch's selection reads the scores of its centres' neighbours, a fraction of the table (10 M nodes, batch 4096: ~15 %
// of the rows; streaming all 1.28 GB was 42 % of the step).  A wave takes 512 rows at a time: their marks are one 8-byte load per
// lane, the marked rows' offsets are compacted into the wave's LDS list (prefix sum over the lanes' counts), and the list is
// worked off like score_table_body works off consecutive rows - the same lanes-per-row geometry, the same fma chain, the same
// butterfly: the score of a marked row is bit for bit score_table_body's.  The map is padded: 8-byte loads up to
// touched_bytes(n) never leave it, and rows beyond the table are unmarked.  sel: 4 x 512 uint16 of LDS per 256-thread workgroup.
constexpr int MARK_GROUP = 512;
__host__ __device__ __forceinline__ int64_t touched_bytes(int64_t n_rows) { return (n_rows + 2 * MARK_GROUP - 1) / MARK_GROUP * MARK_GROUP; }
template <int SCORE_K = 2>
__device__ __forceinline__ void score_marked_body(const float *__restrict__ X, int feat_dim, int stride, const float *__restrict__ W,
                                                  const float *__restrict__ bias, int64_t n_rows, float *__restrict__ s0,
                                                  const unsigned char *__restrict__ touched, int block, int n_blocks,
                                                  unsigned short *sel_all) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const int64_t wave_global = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)n_blocks * (blockDim.x >> 6);
    unsigned short *sel = sel_all + (threadIdx.x >> 6) * MARK_GROUP;
    const float b0 = bias[0];
    const int nch = stride >> 2;
    bool has[SCORE_K];
    int chc[SCORE_K];
    float wv[SCORE_K][4];
#pragma unroll
    for (int k = 0; k < SCORE_K; ++k) {                                   // (as score_table_body: a lane's weights, loaded once)
        const int ch = sub + k * lpr;
        has[k] = ch < nch;
        chc[k] = has[k] ? ch : nch - 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int f = 4 * chc[k] + j;
            const float wj = W[f < feat_dim ? f : feat_dim - 1];
            wv[k][j] = __int_as_float(__float_as_int(wj) & ((has[k] && f < feat_dim) ? -1 : 0));
        }
    }
    const bool two = SCORE_K > 1 && lpr < nch;
    for (int64_t base = wave_global * MARK_GROUP; base < n_rows; base += n_waves * MARK_GROUP) {
        const uint64_t mk = *reinterpret_cast<const uint64_t *>(touched + base + 8 * lane);
        int cnt = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) cnt += ((mk >> (8 * b)) & 0xFFull) != 0ull;
        // inclusive scan of the lanes' counts (DPP row shifts inside every 16-lane row, then the rows' totals)
        int inc = cnt;
        inc += (int)__builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, false);
        inc += (int)__builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, false);
        inc += (int)__builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, false);
        inc += (int)__builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, false);
        const int t0 = __builtin_amdgcn_readlane(inc, 15), t1 = __builtin_amdgcn_readlane(inc, 31), t2 = __builtin_amdgcn_readlane(inc, 47);
        const int rw = lane >> 4;
        inc += (rw > 0 ? t0 : 0) + (rw > 1 ? t1 : 0) + (rw > 2 ? t2 : 0);
        const int total = __builtin_amdgcn_readlane(inc, PCG_WAVE - 1);
        if (total == 0) continue;                                         // (wave-uniform)
        int at = inc - cnt;
#pragma unroll
        for (int b = 0; b < 8; ++b)
            if (((mk >> (8 * b)) & 0xFFull) != 0ull) sel[at++] = (unsigned short)(8 * lane + b);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // (this wave's own LDS list: written above, read below)
        __builtin_amdgcn_wave_barrier();
        for (int i0 = 0; i0 < total; i0 += rpw * SCORE_UNROLL) {
            f4 xv[SCORE_UNROLL][SCORE_K];
            int64_t rid[SCORE_UNROLL];
#pragma unroll
            for (int u = 0; u < SCORE_UNROLL; ++u) {                       // (unconditional loads: a slot beyond the list re-reads its last row)
                const int j = i0 + u * rpw + slot;
                const int64_t row = base + sel[j < total ? j : total - 1];
                rid[u] = j < total ? row : -1;
                const float *xrow = X + row * stride;
                xv[u][0] = *reinterpret_cast<const f4 *>(xrow + 4 * chc[0]);
                if constexpr (SCORE_K > 1)
                    if (two) xv[u][1] = *reinterpret_cast<const f4 *>(xrow + 4 * chc[1]);
            }
#pragma unroll
            for (int u = 0; u < SCORE_UNROLL; ++u) {
                float q = 0.f;
                if (has[0]) {
                    q = fmaf(xv[u][0].x, wv[0][0], q);
                    q = fmaf(xv[u][0].y, wv[0][1], q);
                    q = fmaf(xv[u][0].z, wv[0][2], q);
                    q = fmaf(xv[u][0].w, wv[0][3], q);
                }
                if constexpr (SCORE_K > 1)
                    if (two && has[1]) {
                        q = fmaf(xv[u][1].x, wv[1][0], q);
                        q = fmaf(xv[u][1].y, wv[1][1], q);
                        q = fmaf(xv[u][1].z, wv[1][2], q);
                        q = fmaf(xv[u][1].w, wv[1][3], q);
                    }
                const float sc = score_reduce(q, lpr);
                if (sub == 0 && rid[u] >= 0) s0[rid[u]] = sc + b0;
            }
        }
        __builtin_amdgcn_wave_barrier();                                   // (the list is rewritten by the next group)
    }
}

// number of 256-thread workgroups score_table uses for n rows (8 per CU at most, grid-stride beyond)
__host__ __forceinline__ int64_t score_table_blocks(int64_t n_rows, int stride) {
    const int rpw = PCG_WAVE / lanes_per_row(stride);
    const int64_t rows_per_block = (int64_t)4 * rpw * SCORE_UNROLL;
    int64_t blocks = (n_rows + rows_per_block - 1) / rows_per_block;
    return blocks > 256 * PCG_SCORE_BLOCKS_PER_CU ? 256 * PCG_SCORE_BLOCKS_PER_CU : blocks;
}

__device__ __forceinline__ uint64_t make_pos_key(const float *s0, const int32_t *train_pos, int i, int n_pos) {
    if (i >= n_pos) return ~0ull;
    return ((uint64_t)orderable(s0[train_pos[i]]) << 32) | (uint32_t)i;
}

// raw[i] = make_pos_key(i) computed from the feature rows themselves (the same partial dot + butterfly as the score table,
// so the same bits as s0[train_pos[i]]): lets the keys be formed beside the score pass instead of after it.
// workgroup `block` of `n_blocks`, 256 threads.
__device__ __forceinline__ void pos_key_body(const float *__restrict__ X, int feat_dim, int stride, const float *__restrict__ W,
                                             const float *__restrict__ bias, const int32_t *__restrict__ train_pos, int n_pos,
                                             uint64_t *__restrict__ raw, int block, int n_blocks, int64_t pos_row_base = -1) {
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const float b0 = bias[0];
    const int waves = n_blocks * (int)(blockDim.x >> 6);
    for (int base = (block * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6)) * rpw; base < n_pos; base += waves * rpw) {
        const int i = base + slot;
        const bool ok = i < n_pos;
        // (pos_row_base >= 0: the train positives' rows are a block of the table of their own - the partitioned path's replicated
        //  block - in train_pos order; else row = node id)
        const float *row = X + (size_t)(pos_row_base >= 0 ? pos_row_base + (ok ? i : 0) : (ok ? train_pos[i] : 0)) * stride;
        float p = ok ? score_partial(row, W, feat_dim, stride, sub, lpr) : 0.f;
        p = score_reduce(p, lpr);
        if (ok && sub == 0) raw[i] = ((uint64_t)orderable(p + b0) << 32) | (uint32_t)i;
    }
}

// ---- rank sort of the train-pos keys: one launch, no step barriers (n_pos <= RANK_MAX) -------------------
// Keys are unique, so rank(i) = #{j : key_j < key_i} is a permutation.  Every workgroup owns 64 keys
// (one per lane), walks all keys in LDS tiles of RANK_TILE and splits each tile's j-range over its
// 16 waves; LDS reads are wave-wide broadcasts.  O(P^2) compares, but embarrassingly parallel: it
// beats the many-launch bitonic network up to a few 10^4 keys.
constexpr int RANK_MAX = 16384;
constexpr int RANK_TILE = 8192;
constexpr int RANK_WAVES = 16;

// sh: TILE (default RANK_TILE) uint64, part: WAVES * 64 ints (LDS); workgroup `block` of ceil(n_pos / 64), WAVES * 64 threads
// raw: the unsorted keys (make_pos_key of every i < n_pos), if somebody has formed them already (pos_key_body) - then a
// tile is staged with coalesced 8-byte loads instead of two dependent loads and a random 4-byte gather per key
// PUBLISH: the sorted keys are read by other workgroups of the SAME launch (select_rows): write-through (sc1) stores
template <int TILE = RANK_TILE, int WAVES = RANK_WAVES, bool PUBLISH = false>
__device__ __forceinline__ void rank_sort_body(const float *__restrict__ s0, const int32_t *__restrict__ train_pos, int n_pos,
                                               int cap, uint64_t *__restrict__ keys, int block, uint64_t *sh, int *part,
                                               const uint64_t *__restrict__ raw = nullptr) {
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = block * PCG_WAVE + lane;
    // a key costs two dependent loads (train_pos[i], then s0 of it): this thread's own key and its share of a tile's keys
    // are requested level by level, so the staging of a tile costs two load latencies in all
    constexpr int PER = TILE / (WAVES * PCG_WAVE);
    static_assert(PER >= 1 && PER * WAVES * PCG_WAVE == TILE, "a tile is staged PER keys per thread");
    // (every load here is unconditional - index clamped, the value OR-ed with all-ones where it must not count: a load inside
    //  a conditional is compiled into a branch that waits for every load in flight, and a tile's PER loads per thread would
    //  go out one at a time)
    const int ic = i < n_pos ? i : (n_pos > 0 ? n_pos - 1 : 0);
    const int id_mine = raw ? 0 : train_pos[ic];
    uint64_t mine = (raw ? raw[ic] : 0ull) | (i < n_pos ? 0ull : ~0ull);   // ~0 when i >= n_pos
    int c = 0;
    for (int t0 = 0; t0 < n_pos; t0 += TILE) {
        const int nt = (n_pos - t0 < TILE) ? n_pos - t0 : TILE;
        if (raw) {
            uint64_t kt[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int t = (int)threadIdx.x + u * (int)blockDim.x;
                kt[u] = raw[t0 + (t < nt ? t : nt - 1)] | (t < nt ? 0ull : ~0ull);
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < PER; ++u) sh[(int)threadIdx.x + u * (int)blockDim.x] = kt[u];      // (all-ones beyond nt)
            __syncthreads();
        } else {
        int idt[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = (int)threadIdx.x + u * (int)blockDim.x;
            idt[u] = train_pos[t0 + (t < nt ? t : nt - 1)];
        }
        float st[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) st[u] = s0[idt[u]];
        if (t0 == 0) mine = (((uint64_t)orderable(s0[id_mine]) << 32) | (uint32_t)i) | (i < n_pos ? 0ull : ~0ull);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = (int)threadIdx.x + u * (int)blockDim.x;
            sh[t] = (((uint64_t)orderable(st[u]) << 32) | (uint32_t)(t0 + t)) | (t < nt ? 0ull : ~0ull);
        }
        __syncthreads();
        }
        // this wave's stretch of the tile, eight keys (four 16-byte broadcast reads) per iteration; the tile is padded
        // with all-ones keys (never smaller than anybody's), so the stretches need no tail handling
        const int chunk = (((nt + WAVES - 1) / WAVES) + 7) & ~7;
        const int j0 = wave * chunk;
        int j1 = j0 + chunk;
        const int nt8 = (nt + 7) & ~7;
        j1 = j1 < nt8 ? j1 : nt8;
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
    if (wave == 0 && i < n_pos) {
        int rank = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) rank += part[w * PCG_WAVE + lane];
        if constexpr (PUBLISH) __hip_atomic_store(reinterpret_cast<unsigned long long *>(keys) + rank, (unsigned long long)mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else keys[rank] = mine;
    }
    if (block == 0)
        for (int t = n_pos + threadIdx.x; t < cap; t += blockDim.x) {
            if constexpr (PUBLISH) __hip_atomic_store(reinterpret_cast<unsigned long long *>(keys) + t, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else keys[t] = ~0ull;
        }
}

// ---- Adam update of a range of the flat parameter buffer from per-tile gradient slabs -----------------------------------
// (torch.optim.Adam with coupled L2 weight decay, src/model_handler.py:124,153).  Shared by the stand-alone kernel
// (dense.hip) and the step-front kernel, where the update of the previous step rides along the score pass (choose.hip).
// g = sum over slabs in a fixed order (ADAM_ACC interleaved partial sums per wave, then a fixed tree), so ADAM_ACC slab reads
// are in flight per thread and the result is bitwise reproducible.  (16: a wave's share of a 64-tile batch's slabs is one round
// of loads, not two.)  Workgroup `block` (256 threads) owns 64 parameters
// [p_begin + 64 * block, ...) below p_end; the slabs are split over its 4 waves.  part: 4 * 64 floats of LDS.
struct AdamHyper {
    float lr, beta1, beta2, eps, wd;
};
// torch.optim.Adam's update of one parameter (coupled L2 weight decay; t = the step's number, 1-based): ONE statement of the
// arithmetic, for every place that applies it - two workgroups that work out the same parameter's update from the same inputs
// (the partitioned path's score pass recomputes the label classifier's while other workgroups of its launch store it) must
// arrive at the same bits, so nothing here is left to the compiler's choice of contraction
__device__ __forceinline__ float adam_update(float p, float m_old, float v_old, float g, float t, const AdamHyper &h, float &mi, float &vi) {
#pragma clang fp contract(off)
    g = __builtin_fmaf(h.wd, p, g);
    mi = __builtin_fmaf(h.beta1, m_old, (1.f - h.beta1) * g);
    vi = __builtin_fmaf(h.beta2, v_old, ((1.f - h.beta2) * g) * g);
    const float bc1 = 1.f - powf(h.beta1, t), bc2 = 1.f - powf(h.beta2, t);
    const float denom = sqrtf(vi) / sqrtf(bc2) + h.eps;
    return p - (h.lr / bc1) * (mi / denom);
}
constexpr int ADAM_ACC = 16;
__device__ __forceinline__ void adam_reduce_body(float *__restrict__ theta, float *__restrict__ m, float *__restrict__ v,
                                                 const float *__restrict__ slabs, int n_slabs, int64_t n_params,
                                                 int64_t p_begin, int64_t p_end, const int32_t *__restrict__ step_counter,
                                                 AdamHyper h, float *__restrict__ grad_out, int apply, int block,
                                                 float (*part)[PCG_WAVE]) {
    const int lane = threadIdx.x & (PCG_WAVE - 1), w = threadIdx.x >> 6;
    const int64_t i = p_begin + (int64_t)block * PCG_WAVE + lane;
    const bool ok = i < p_end;
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
    static_assert(ADAM_ACC == 16, "the tree below adds sixteen partial sums");
    part[w][lane] = (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]))) +
                    (((acc[8] + acc[9]) + (acc[10] + acc[11])) + ((acc[12] + acc[13]) + (acc[14] + acc[15])));
    __syncthreads();
    if (w != 0 || !ok) return;
    float g = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    if (grad_out) grad_out[i] = g;
    if (!apply) return;
    float mi, vi;
    const float pn = adam_update(p, m_old, v_old, g, t, h, mi, vi);
    m[i] = mi;
    v[i] = vi;
    theta[i] = pn;
}

// two-class cross entropy and its gradient from the logit difference: one v_exp_f32, one v_log_f32, one v_rcp_f32
// (this runs on a single lane per row, on the critical path between forward and backward: the library expf / logf /
// division sequences are hundreds of dependent instructions there).  Absolute error ~1e-7: far inside the 1e-4 / 2e-5
// the parity tests allow for logits / gradients.
__device__ __forceinline__ void xent2(float a, float b, int y, float &loss, float &da, float &db) {
    const float d = b - a;
    const float e = __expf(-fabsf(d));                   // in (0, 1]
    const float inv = __frcp_rn(1.f + e);
    const float p_hi = inv, p_lo = e * inv;              // softmax of the larger / the smaller logit
    const float pa = d > 0.f ? p_lo : p_hi, pb = d > 0.f ? p_hi : p_lo;
    loss = fmaxf(a, b) + __logf(1.f + e) - (y == 1 ? b : a);
    da = pa - (y == 0 ? 1.f : 0.f);
    db = pb - (y == 1 ? 1.f : 0.f);
}

// the same update for one parameter whose summed gradient g is already known
__device__ __forceinline__ void adam_apply_one(float *theta, float *m, float *v, int64_t i, float g, float t, AdamHyper h) {
    float mi, vi;
    const float pn = adam_update(theta[i], m[i], v[i], g, t, h, mi, vi);
    m[i] = mi;
    v[i] = vi;
    theta[i] = pn;
}

// what the step-front kernel needs to apply the previous step's deferred update (all parameters below p_end)
struct DeferredAdam {
    float *theta, *m, *v;
    const float *slabs;
    int64_t n_params, p_end;
    const int32_t *step_counter;
    uint32_t *pending;       // device words: [0] = 1: the slabs hold a gradient that has not been applied to [0, p_end) yet,
                             // [1] = in how many slabs (pcg_train_dense sets both; the launch after the update clears [0])
    AdamHyper h;
};

}  // namespace pcg
