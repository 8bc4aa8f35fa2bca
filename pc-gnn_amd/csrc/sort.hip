// Per-step sort of the training positives by class-0 logit.
// Replaces the per-centre torch.sort over all pos_scores (src/layers.py:683-688):
// sorted once, every positive centre then finds its m nearest by a window search.
// Keys are unique 64-bit (orderable(score) << 32 | position in train_pos), so the
// bitonic network's result is a total order: equal scores stay in list order.
#include "common.h"

namespace pcg {

constexpr int SORT_CHUNK = 4096;    // keys sorted per workgroup in LDS (32 KiB)
constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ uint64_t make_pos_key(const float *s0, const int32_t *train_pos, int i, int n_pos) {
    if (i >= n_pos) return ~0ull;
    return ((uint64_t)orderable(s0[train_pos[i]]) << 32) | (uint32_t)i;
}

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
                                                               uint64_t *__restrict__ keys) {
    __shared__ uint64_t sh[SORT_CHUNK];
    const int base = blockIdx.x * SORT_CHUNK;
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) sh[t] = make_pos_key(s0, train_pos, base + t, n_pos);
    __syncthreads();
    for (int k = 2; k <= SORT_CHUNK; k <<= 1) lds_steps(sh, base, k, k >> 1);
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) keys[base + t] = sh[t];
}

__global__ void __launch_bounds__(256) bitonic_global_step(uint64_t *__restrict__ keys, int64_t n_pairs, int j, int k) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const int64_t i = 2 * (int64_t)j * (p / j) + (p % j);
    uint64_t a = keys[i], b = keys[i + j];
    const bool asc = ((i & k) == 0);
    if ((a > b) == asc) {
        keys[i] = b;
        keys[i + j] = a;
    }
}

__global__ void __launch_bounds__(SORT_THREADS) bitonic_local_merge(uint64_t *__restrict__ keys, int k) {
    __shared__ uint64_t sh[SORT_CHUNK];
    const int base = blockIdx.x * SORT_CHUNK;
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) sh[t] = keys[base + t];
    __syncthreads();
    lds_steps(sh, base, k, SORT_CHUNK >> 1);
    for (int t = threadIdx.x; t < SORT_CHUNK; t += SORT_THREADS) keys[base + t] = sh[t];
}

// ---- rank sort: one launch, no step barriers (n_pos <= RANK_MAX) -------------------
// Keys are unique, so rank(i) = #{j : key_j < key_i} is a permutation.  Every block owns 64 keys
// (one per lane), walks all keys in LDS tiles of RANK_TILE and splits each tile's j-range over its
// 16 waves; LDS reads are wave-wide broadcasts.  O(P^2) compares, but embarrassingly parallel: it
// beats the many-launch bitonic network up to a few 10^4 keys.
constexpr int RANK_MAX = 65536;
constexpr int RANK_TILE = 8192;
constexpr int RANK_WAVES = 16;

__global__ void __launch_bounds__(RANK_WAVES *PCG_WAVE) pos_rank_sort(const float *__restrict__ s0,
                                                                      const int32_t *__restrict__ train_pos,
                                                                      int n_pos, int cap, uint64_t *__restrict__ keys) {
    __shared__ uint64_t sh[RANK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = blockIdx.x * PCG_WAVE + lane;
    const uint64_t mine = make_pos_key(s0, train_pos, i, n_pos);          // ~0 when i >= n_pos
    int c = 0;
    for (int t0 = 0; t0 < n_pos; t0 += RANK_TILE) {
        const int nt = (n_pos - t0 < RANK_TILE) ? n_pos - t0 : RANK_TILE;
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += blockDim.x) sh[t] = make_pos_key(s0, train_pos, t0 + t, n_pos);
        __syncthreads();
        const int chunk = (nt + RANK_WAVES - 1) / RANK_WAVES;
        const int j0 = wave * chunk, j1 = (j0 + chunk < nt) ? j0 + chunk : nt;
        int j = j0;
        for (; j + 4 <= j1; j += 4) {
            const uint64_t a0 = sh[j], a1 = sh[j + 1], a2 = sh[j + 2], a3 = sh[j + 3];
            c += (a0 < mine) + (a1 < mine) + (a2 < mine) + (a3 < mine);
        }
        for (; j < j1; ++j) c += sh[j] < mine;
    }
    part[wave * PCG_WAVE + lane] = c;
    __syncthreads();
    if (wave == 0 && i < n_pos) {
        int rank = 0;
#pragma unroll
        for (int w = 0; w < RANK_WAVES; ++w) rank += part[w * PCG_WAVE + lane];
        keys[rank] = mine;
    }
    if (blockIdx.x == 0)
        for (int t = n_pos + threadIdx.x; t < cap; t += blockDim.x) keys[t] = ~0ull;
}

static int64_t sort_capacity(int32_t n_pos) {
    int64_t c = SORT_CHUNK;
    while (c < n_pos) c <<= 1;
    return c;
}

}  // namespace pcg

extern "C" {

int64_t pcg_pos_sort_capacity(int32_t n_pos) { return n_pos < 0 ? PCG_E_ARG : pcg::sort_capacity(n_pos); }

int pcg_pos_sort(const pcg_graph_desc *g, const float *s0, uint64_t *keys, void *stream) {
    if (!g || !s0 || !keys || g->n_pos < 0 || (g->n_pos > 0 && !g->train_pos)) return PCG_E_ARG;
    if (g->n_pos == 0) return PCG_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t cap = pcg::sort_capacity(g->n_pos);
    if (g->n_pos <= pcg::RANK_MAX) {
        hipLaunchKernelGGL(pcg::pos_rank_sort, dim3((g->n_pos + PCG_WAVE - 1) / PCG_WAVE),
                           dim3(pcg::RANK_WAVES * PCG_WAVE), 0, st, s0, g->train_pos, g->n_pos, (int)cap, keys);
        PCG_LAUNCH_CHECK();
        return PCG_OK;
    }
    const int chunks = (int)(cap / pcg::SORT_CHUNK);
    hipLaunchKernelGGL(pcg::pos_sort_local, dim3(chunks), dim3(pcg::SORT_THREADS), 0, st, s0, g->train_pos, g->n_pos,
                       keys);
    PCG_LAUNCH_CHECK();
    for (int64_t k = 2 * pcg::SORT_CHUNK; k <= cap; k <<= 1) {
        for (int64_t j = k >> 1; j >= pcg::SORT_CHUNK; j >>= 1) {
            const int64_t pairs = cap / 2;
            hipLaunchKernelGGL(pcg::bitonic_global_step, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, keys,
                               pairs, (int)j, (int)k);
            PCG_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(pcg::bitonic_local_merge, dim3(chunks), dim3(pcg::SORT_THREADS), 0, st, keys, (int)k);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

}  // extern "C"
