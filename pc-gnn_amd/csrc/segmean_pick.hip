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
// Draw i is exactly pcg_pick's draw i of the same (seed, epoch); it is stored at position sigma(i), sigma a keyed
// pseudo-random permutation of [0, k).
// A keyed bijection on [0, 2^bits): additions, odd multiplications and xor-shifts modulo 2^bits are each invertible.
__device__ __forceinline__ uint32_t permute_bits(uint32_t x, int bits, uint64_t key) {
    if (bits == 0) return 0u;
    const uint32_t mask = bits >= 32 ? 0xFFFFFFFFu : ((1u << bits) - 1u);
    const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    const int h = (bits + 1) >> 1, t = (bits + 2) / 3;
    x = (x + k0) & mask;
    x = (x * 0x9E3779B1u) & mask;
    x ^= x >> h;
    x = (x * ((k1 << 1) | 1u)) & mask;
    x ^= x >> t;
    x = (x + (k1 ^ 0x85EBCA6Bu)) & mask;
    x = (x * 0xC2B2AE35u) & mask;
    x ^= x >> h;
    x = (x * (((k0 >> 3) << 1) | 1u)) & mask;
    x ^= x >> t;
    return x;
}
// The shuffle: draw i goes to position sigma(i), sigma = the bijection above restricted to [0, k) by cycle walking
// (follow the permutation of [0, 2^bits) from i until it lands below k again: a bijection of [0, k), on average
// < 2 steps).  O(1) per draw, nothing shared between draws - the rank-of-random-keys shuffle it replaces was O(k^2).
__device__ __forceinline__ int shuffle_position(int i, int k, int bits, uint64_t key) {
    uint32_t p = (uint32_t)i;
    do {
        p = permute_bits(p, bits, key);
    } while (p >= (uint32_t)k);
    return (int)p;
}
__device__ __forceinline__ uint64_t shuffle_key_of(uint64_t seed, uint64_t epoch) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (epoch + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// bisect_right(cum, x, 0, n-1) by a group of 16 lanes: 16 probes per step => ceil(log16 n) dependent loads instead of log2 n.
// All 16 lanes of the group must call it with the same x; every lane returns the index.
__device__ __forceinline__ int bisect16(const double *__restrict__ cum, int n, double x, int sub) {
    int lo = 0, hi = n - 1;                      // the answer is the first index in [lo, hi) with x < cum[index], or hi
    while (hi - lo > 0) {
        const int span = hi - lo;
        const int step = (span + 15) >> 4;       // probes at lo + (sub+1)*step - 1, clipped to hi - 1
        int q = lo + (sub + 1) * step - 1;
        if (q > hi - 1) q = hi - 1;
        const bool below = !(x < cum[q]);        // true on a prefix of the probes: the answer lies beyond q
        const uint64_t m = __ballot(below);
        const int shift = lane_id() & ~15;
        const int c = __popc(((unsigned)(m >> shift)) & 0xFFFFu);     // leading `below` probes of this group
        if (c == 16) {
            int ql = lo + 16 * step - 1;
            if (ql > hi - 1) ql = hi - 1;
            lo = ql + 1;                         // beyond the last probe
        } else {
            int qc = lo + (c + 1) * step - 1;    // first probe with x < cum[probe]
            if (qc > hi - 1) qc = hi - 1;
            if (c > 0) {
                int ql = lo + c * step - 1;
                if (ql > hi - 1) ql = hi - 1;
                lo = ql + 1;
            }
            hi = qc;
        }
    }
    return lo;
}

// 16 draws per 256-thread workgroup: 16 lanes per draw run the 16-ary search, lane 0 of the group stores the draw at
// its shuffled position.
__global__ void __launch_bounds__(256) pick_shuffled_kernel(const double *__restrict__ cum, const int32_t *__restrict__ idx_train, int n,
                                                            uint64_t seed, uint64_t epoch_base,
                                                            const unsigned long long *__restrict__ epoch_counter, int k, int bits,
                                                            const int32_t *__restrict__ labels_all, int32_t *__restrict__ out_ids,
                                                            int32_t *__restrict__ out_labels) {
    // (blockIdx.y: one of several epochs drawn in one launch - pcg_pick_shuffled_epochs; its k draws at out + blockIdx.y * k)
    const uint64_t epoch = epoch_base + (epoch_counter ? *epoch_counter : 0ull) + blockIdx.y;
    out_ids += (size_t)blockIdx.y * k;
    if (out_labels) out_labels += (size_t)blockIdx.y * k;
    const int dl = threadIdx.x >> 4, sub = threadIdx.x & 15;          // draw slot 0..15, lane in the group
    const int di = blockIdx.x * 16 + dl;
    const double u = philox_uniform(seed, epoch, (uint32_t)(di < k ? di : 0));
    const int pos = bisect16(cum, n, u * (cum[n - 1] + 0.0), sub);
    if (sub == 0 && di < k) {
        const int32_t id = idx_train[pos];
        const int at = shuffle_position(di, k, bits, shuffle_key_of(seed, epoch));
        out_ids[at] = id;
        if (out_labels) out_labels[at] = labels_all[id];
    }
}

__global__ void bump_counter_kernel(unsigned long long *counter, int by) { *counter += (unsigned long long)by; }

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
    return pcg_pick_shuffled_epochs(cum, idx_train, n_train, seed, epoch_base, epoch_counter, bump, 1, k, labels_all, out_ids, out_labels,
                                    stream);
}

int pcg_pick_shuffled_epochs(const double *cum, const int32_t *idx_train, int32_t n_train, uint64_t seed, uint64_t epoch_base,
                             uint64_t *epoch_counter, int32_t bump, int32_t n_epochs, int32_t k, const int32_t *labels_all,
                             int32_t *out_ids, int32_t *out_labels, void *stream) {
    if (!cum || !idx_train || !out_ids || n_train < 1 || k < 0 || n_epochs < 1 || n_epochs > 65535) return PCG_E_ARG;
    if (out_labels && !labels_all) return PCG_E_ARG;
    if (bump && !epoch_counter) return PCG_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (k > 0) {
        int bits = 0;
        while ((1ll << bits) < (long long)k) ++bits;
        hipLaunchKernelGGL(pcg::pick_shuffled_kernel, dim3((k + 15) / 16, n_epochs), dim3(256), 0, st, cum, idx_train, n_train, seed, epoch_base,
                           reinterpret_cast<const unsigned long long *>(epoch_counter), k, bits, labels_all, out_ids, out_labels);
        PCG_LAUNCH_CHECK();
    }
    if (bump) {       // a second, one-thread launch: a ticket per workgroup would be thousands of same-address atomics
        hipLaunchKernelGGL(pcg::bump_counter_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<unsigned long long *>(epoch_counter), n_epochs);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

}  // extern "C"
