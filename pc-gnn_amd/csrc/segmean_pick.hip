// Segmented mean over explicit index lists, and the pick sampler.
#include "common.h"

namespace pcg {

// ---- segmented mean -----------------------------------------------------------
// out[i,:] = sum_{t < count[i]} X[idx[begin[i] + t], :] / norm(count[i])
// One wave per output row, 64/lpr feature rows per wave-instruction, 4 in flight.
// (mask.div(num_neigh).mm(embed_matrix): src/layers.py:599-624, graphsage.py:82-95, 216-231)
constexpr int SEG_UNROLL = 4;
constexpr int SEG_MAX_ACC = 2;

__global__ void __launch_bounds__(256) segment_mean_kernel(const float *__restrict__ X, int feat_dim, int stride,
                                                           const int64_t *__restrict__ begin,
                                                           const int32_t *__restrict__ count,
                                                           const int32_t *__restrict__ idx, int n_rows, int norm,
                                                           float *__restrict__ out, int out_stride) {
    const int lane = lane_id();
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= n_rows) return;
    const int lpr = lanes_per_row(stride), rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr, nch = stride >> 2;
    const int nacc = (nch + lpr - 1) / lpr;
    const int32_t *__restrict__ list = idx + begin[row];
    const int n = count[row];
    float4 acc[SEG_MAX_ACC];
#pragma unroll
    for (int a = 0; a < SEG_MAX_ACC; ++a) acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = 0; base < n; base += rpw * SEG_UNROLL) {
        float4 v[SEG_UNROLL][SEG_MAX_ACC];
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u) {
            const int i = base + u * rpw + slot;
            const bool ok = i < n;
            const float *r = X + (size_t)(ok ? list[i] : 0) * stride;
#pragma unroll
            for (int a = 0; a < SEG_MAX_ACC; ++a) {
                const int ch = a * lpr + sub;
                v[u][a] = (ok && a < nacc && ch < nch) ? *reinterpret_cast<const float4 *>(r + 4 * ch)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int u = 0; u < SEG_UNROLL; ++u)
#pragma unroll
            for (int a = 0; a < SEG_MAX_ACC; ++a) {
                acc[a].x += v[u][a].x; acc[a].y += v[u][a].y; acc[a].z += v[u][a].z; acc[a].w += v[u][a].w;
            }
    }
#pragma unroll
    for (int a = 0; a < SEG_MAX_ACC; ++a)
        for (int o = lpr; o < PCG_WAVE; o <<= 1) {
            acc[a].x += __shfl_xor(acc[a].x, o);
            acc[a].y += __shfl_xor(acc[a].y, o);
            acc[a].z += __shfl_xor(acc[a].z, o);
            acc[a].w += __shfl_xor(acc[a].w, o);
        }
    if (lane < lpr) {
        const float den = norm == PCG_NORM_SQRT_COUNT ? sqrtf((float)n) : (float)n;
        float *o = out + (size_t)row * out_stride;
#pragma unroll
        for (int a = 0; a < SEG_MAX_ACC; ++a) {
            const int ch = a * lpr + sub;
            if (a >= nacc || ch >= nch) continue;
            const int f = 4 * ch;
            if (f + 0 < feat_dim) o[f + 0] = acc[a].x / den;
            if (f + 1 < feat_dim) o[f + 1] = acc[a].y / den;
            if (f + 2 < feat_dim) o[f + 2] = acc[a].z / den;
            if (f + 3 < feat_dim) o[f + 3] = acc[a].w / den;
        }
    }
}

// ---- pick ------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter = (draw index, epoch), key = seed.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ double philox_uniform(uint64_t seed, uint64_t epoch, uint32_t draw) {
    uint32_t c[4] = {draw, 0u, (uint32_t)epoch, (uint32_t)(epoch >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    // same 53-bit construction as CPython's random(): (a >> 5, b >> 6)
    return ((double)(c[0] >> 5) * 67108864.0 + (double)(c[1] >> 6)) * (1.0 / 9007199254740992.0);
}

// all four words of the Philox block of (seed, epoch, draw)
__device__ __forceinline__ void philox_block(uint64_t seed, uint64_t epoch, uint32_t draw, uint32_t (&c)[4]) {
    c[0] = draw; c[1] = 0u; c[2] = (uint32_t)epoch; c[3] = (uint32_t)(epoch >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

__device__ __forceinline__ int32_t pick_draw(const double *__restrict__ cum, const int32_t *__restrict__ idx_train, int n, double u) {
    const double x = u * (cum[n - 1] + 0.0);
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (x < cum[mid]) hi = mid;
        else lo = mid + 1;
    }
    return idx_train[lo];
}

// out[i] = idx_train[bisect_right(cum, u * cum[n-1], 0, n-1)]   (random.choices, utils.py:278)
__global__ void __launch_bounds__(256) pick_kernel(const double *__restrict__ cum, const int32_t *__restrict__ idx_train,
                                                   int n, const double *__restrict__ uniforms, uint64_t seed,
                                                   uint64_t epoch, int k, int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const double u = uniforms ? uniforms[i] : philox_uniform(seed, epoch, (uint32_t)i);
    out[i] = pick_draw(cum, idx_train, n, u);
}

// An epoch's picks, shuffled, with their labels, in one launch (utils.py:274-278 + random.shuffle, model_handler.py:131-133).
// Draw i is exactly pcg_pick's draw i of the same (seed, epoch).  Its place in the output is the rank of a shuffle key
// (third Philox word of the same block, ties by i) among all k keys: a uniformly random permutation.  Every workgroup owns
// 64 draws and ranks them against all k keys, recomputed into LDS tiles (rank sort as for the train positives).
constexpr int SHUF_TILE = 8192;
constexpr int SHUF_WAVES = 16;

__device__ __forceinline__ uint64_t shuffle_key(uint64_t seed, uint64_t epoch, int i, int k) {
    if (i >= k) return ~0ull;
    uint32_t c[4];
    philox_block(seed, epoch, (uint32_t)i, c);
    return ((uint64_t)c[2] << 32) | (uint32_t)i;
}

__global__ void __launch_bounds__(SHUF_WAVES *PCG_WAVE) pick_shuffled_kernel(const double *__restrict__ cum, const int32_t *__restrict__ idx_train,
                                                                             int n, uint64_t seed, uint64_t epoch_base,
                                                                             const unsigned long long *__restrict__ epoch_counter, int k,
                                                                             const int32_t *__restrict__ labels_all,
                                                                             int32_t *__restrict__ out_ids, int32_t *__restrict__ out_labels) {
    __shared__ uint64_t sh[SHUF_TILE];
    __shared__ int part[SHUF_WAVES * PCG_WAVE];
    const uint64_t epoch = epoch_base + (epoch_counter ? *epoch_counter : 0ull);
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    const int i = blockIdx.x * PCG_WAVE + lane;
    const uint64_t mine = shuffle_key(seed, epoch, i, k);
    // the owners' draws: a chain of ~log2(n) dependent loads - issued before the ranking so that it overlaps with it
    int32_t id = 0;
    if (wave == 0 && i < k) id = pick_draw(cum, idx_train, n, philox_uniform(seed, epoch, (uint32_t)i));
    int c = 0;
    for (int t0 = 0; t0 < k; t0 += SHUF_TILE) {
        const int nt = (k - t0 < SHUF_TILE) ? k - t0 : SHUF_TILE;
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += blockDim.x) sh[t] = shuffle_key(seed, epoch, t0 + t, k);
        __syncthreads();
        const int chunk = (nt + SHUF_WAVES - 1) / SHUF_WAVES;
        const int j0 = wave * chunk, j1 = (j0 + chunk < nt) ? j0 + chunk : nt;
        for (int j = j0; j < j1; ++j) c += sh[j] < mine;
    }
    part[wave * PCG_WAVE + lane] = c;
    __syncthreads();
    if (wave == 0 && i < k) {
        int rank = 0;
#pragma unroll
        for (int w = 0; w < SHUF_WAVES; ++w) rank += part[w * PCG_WAVE + lane];
        out_ids[rank] = id;
        if (out_labels) out_labels[rank] = labels_all[id];
    }
}

__global__ void bump_counter_kernel(unsigned long long *counter) { *counter += 1ull; }

}  // namespace pcg

extern "C" {

int pcg_segment_mean(const pcg_graph_desc *g, const int64_t *begin, const int32_t *count, const int32_t *idx,
                     int32_t n_rows, int32_t norm, float *out, int32_t out_stride, void *stream) {
    if (!g || !g->X || !begin || !count || !idx || !out || n_rows < 0) return PCG_E_ARG;
    if (g->feat_stride % 4 != 0 || g->feat_stride < g->feat_dim || out_stride < g->feat_dim) return PCG_E_ARG;
    if (g->feat_stride > 4 * 64 * pcg::SEG_MAX_ACC) return PCG_E_UNSUPPORTED;
    if (n_rows == 0) return PCG_OK;
    hipLaunchKernelGGL(pcg::segment_mean_kernel, dim3((n_rows + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g->X, g->feat_dim, g->feat_stride, begin, count, idx, n_rows, norm, out, out_stride);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_pick(const double *cum, const int32_t *idx_train, int32_t n_train, const double *uniforms, uint64_t seed,
             uint64_t epoch, int32_t k, int32_t *out, void *stream) {
    if (!cum || !idx_train || !out || n_train < 1 || k < 0) return PCG_E_ARG;
    if (k == 0) return PCG_OK;
    hipLaunchKernelGGL(pcg::pick_kernel, dim3((k + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), cum,
                       idx_train, n_train, uniforms, seed, epoch, k, out);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_pick_shuffled(const double *cum, const int32_t *idx_train, int32_t n_train, uint64_t seed, uint64_t epoch_base,
                      uint64_t *epoch_counter, int32_t bump, int32_t k, const int32_t *labels_all, int32_t *out_ids,
                      int32_t *out_labels, void *stream) {
    if (!cum || !idx_train || !out_ids || n_train < 1 || k < 0) return PCG_E_ARG;
    if (out_labels && !labels_all) return PCG_E_ARG;
    if (bump && !epoch_counter) return PCG_E_ARG;
    if (k > 131072) return PCG_E_UNSUPPORTED;     // k^2 key compares: beyond this use pcg_pick + a sort-based shuffle
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k > 0) {
        hipLaunchKernelGGL(pcg::pick_shuffled_kernel, dim3((k + PCG_WAVE - 1) / PCG_WAVE), dim3(pcg::SHUF_WAVES * PCG_WAVE), 0, st,
                           cum, idx_train, n_train, seed, epoch_base, reinterpret_cast<const unsigned long long *>(epoch_counter),
                           k, labels_all, out_ids, out_labels);
        PCG_LAUNCH_CHECK();
    }
    if (bump) {
        hipLaunchKernelGGL(pcg::bump_counter_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long *>(epoch_counter));
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

}  // extern "C"
