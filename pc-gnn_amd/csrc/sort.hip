// Per-step sort of the training positives by class-0 logit.
// Replaces the per-centre torch.sort over all pos_scores (src/layers.py:683-688):
// sorted once, every positive centre then finds its m nearest by a window search.
// Keys are unique 64-bit (orderable(score) << 32 | position in train_pos), so the
// result is a total order: equal scores stay in list order.  Small P: one-launch rank sort;
// large P: LDS bitonic sort of 2048-key chunks + one merge-by-ranks pass.
#include "common.h"

namespace pcg {

#ifndef PCG_SORT_CHUNK
#define PCG_SORT_CHUNK 2048     // (measured at 20 K / 40 K / 100 K keys: 48.7 / 67.9 / 108 us; 4096: 67.5 / 79.7 / 105; 1024: 53.7 / 87.0 / 155)
#endif
constexpr int SORT_CHUNK = PCG_SORT_CHUNK;   // keys sorted per workgroup in LDS
constexpr int SORT_MIN_CAP = 4096;           // smallest key buffer (entries)
constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ void cmp_swap(uint64_t &a, uint64_t &b, bool ascending) {
    if ((a > b) == ascending) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// LDS bitonic steps j = j_start .. 1 of merge size k on this block's chunk
__device__ __forceinline__ void lds_steps(uint64_t *sh, int chunk_base, int k, int j_start) {
    for (int j = j_start; j > 0; j >>= 1) {
        for (int p = threadIdx.x; p < SORT_CHUNK / 2; p += SORT_THREADS) {
            const int i = 2 * j * (p / j) + (p % j);
            const bool asc = (((chunk_base + i) & k) == 0);
            cmp_swap(sh[i], sh[i + j], asc);
        }
        __syncthreads();
    }
}

// build the keys and fully sort each SORT_CHUNK-sized chunk (alternating directions)
__global__ void __launch_bounds__(SORT_THREADS) pos_sort_local(const float *__restrict__ s0,
                                                               const int32_t *__restrict__ train_pos, int n_pos,
                                                               uint64_t *__restrict__ keys, int all_ascending) {
    __shared__ uint64_t sh[SORT_CHUNK];
    const int base = blockIdx.x * SORT_CHUNK;
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) sh[t] = make_pos_key(s0, train_pos, base + t, n_pos);
    __syncthreads();
    for (int k = 2; k <= SORT_CHUNK; k <<= 1) lds_steps(sh, all_ascending ? 0 : base, k, k >> 1);
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) keys[base + t] = sh[t];
}

// ---- rank sort (n_pos <= RANK_MAX): rank_sort_body in common.h, shared with the fused step-front kernel ----
__global__ void __launch_bounds__(RANK_WAVES *PCG_WAVE) pos_rank_sort(const float *__restrict__ s0,
                                                                      const int32_t *__restrict__ train_pos,
                                                                      int n_pos, int cap, uint64_t *__restrict__ keys) {
    __shared__ __align__(16) uint64_t sh[RANK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    rank_sort_body(s0, train_pos, n_pos, cap, keys, (int)blockIdx.x, sh, part);
}

// ---- large n_pos: every 2048-key chunk sorted in LDS, then one "merge by ranks" pass ----------------------
// rank(e) = its index in its own chunk + sum over the other chunks of #keys smaller than e (binary search in
// an LDS copy of that chunk).  Keys are unique, so the ranks are a permutation.  Work P * (P/4096) * 12 LDS
// steps instead of the rank sort's P^2 compares.
__global__ void __launch_bounds__(SORT_THREADS) pos_merge_rank(const uint64_t *__restrict__ chunks, int n_pos, int cap,
                                                               uint64_t *__restrict__ out) {
    __shared__ uint64_t sh[SORT_CHUNK];
    const int n_chunks = cap / SORT_CHUNK;
    const int e = blockIdx.x * SORT_THREADS + threadIdx.x;       // element handled by this thread
    const bool real = e < cap;
    const uint64_t mine = real ? chunks[e] : ~0ull;
    const int my_chunk = e / SORT_CHUNK;
    int rank = e - my_chunk * SORT_CHUNK;
    for (int c = 0; c < n_chunks; ++c) {
        __syncthreads();
        for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) sh[t] = chunks[(size_t)c * SORT_CHUNK + t];
        __syncthreads();
        if (c == my_chunk) continue;
        int lo = 0, hi = SORT_CHUNK;                              // #keys of chunk c smaller than mine
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sh[mid] < mine) lo = mid + 1;
            else hi = mid;
        }
        rank += lo;
    }
    if (real && mine != ~0ull) out[rank] = mine;                  // padding keys are not scattered ...
    if (real && e >= n_pos) out[e] = ~0ull;                       // ... the tail [n_pos, cap) is filled directly
}

static int64_t sort_capacity(int32_t n_pos) {
    int64_t c = SORT_MIN_CAP;
    while (c < n_pos) c <<= 1;
    return c;
}

}  // namespace pcg

extern "C" {

int64_t pcg_pos_sort_capacity(int32_t n_pos) {
    if (n_pos < 0) return PCG_E_ARG;
    const int64_t cap = pcg::sort_capacity(n_pos);
    return 2 * cap;     // second half: the chunk-sort path's buffer / the unsorted keys pcg_step_front forms beside the score pass
}

int pcg_pos_sort(const pcg_graph_desc *g, const float *s0, uint64_t *keys, void *stream) {
    if (!g || !s0 || !keys || g->n_pos < 0 || (g->n_pos > 0 && !g->train_pos)) return PCG_E_ARG;
    if (g->n_pos == 0) return PCG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t cap = pcg::sort_capacity(g->n_pos);
    if (cap >= (1ll << 31)) return PCG_E_UNSUPPORTED;
    if (g->n_pos <= pcg::RANK_MAX) {
        hipLaunchKernelGGL(pcg::pos_rank_sort, dim3((g->n_pos + PCG_WAVE - 1) / PCG_WAVE),
                           dim3(pcg::RANK_WAVES * PCG_WAVE), 0, st, s0, g->train_pos, g->n_pos, (int)cap, keys);
        PCG_LAUNCH_CHECK();
        return PCG_OK;
    }
    uint64_t *tmp = keys + cap;                         // second half of the caller's buffer
    const int chunks = (int)(cap / pcg::SORT_CHUNK);
    hipLaunchKernelGGL(pcg::pos_sort_local, dim3(chunks), dim3(pcg::SORT_THREADS), 0, st, s0, g->train_pos, g->n_pos, tmp, 1);
    PCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(pcg::pos_merge_rank, dim3((unsigned)(cap / pcg::SORT_THREADS)), dim3(pcg::SORT_THREADS), 0, st, tmp,
                       g->n_pos, (int)cap, keys);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
