// Per-step sort of the training positives by class-0 logit.
// Replaces the per-centre torch.sort over all pos_scores (src/layers.py:683-688):
// sorted once, every positive centre then finds its m nearest by a window search.
// Keys are unique 64-bit (orderable(score) << 32 | position in train_pos), so the
// result is a total order: equal scores stay in list order.  Small P: one-launch rank sort;
// large P: bucket sort (sampled splitters, scatter, rank sort inside the buckets).
#include "common.h"

namespace pcg {

constexpr int SORT_MIN_CAP = 4096;           // smallest key buffer (entries)
constexpr int SORT_THREADS = 1024;

__device__ __forceinline__ void cmp_swap(uint64_t &a, uint64_t &b, bool ascending) {
    if ((a > b) == ascending) {
        const uint64_t t = a;
        a = b;
        b = t;
    }
}

// ---- rank sort (n_pos <= RANK_MAX): rank_sort_body in common.h, shared with the fused step-front kernel ----
__global__ void __launch_bounds__(RANK_WAVES *PCG_WAVE) pos_rank_sort(const float *__restrict__ s0,
                                                                      const int32_t *__restrict__ train_pos,
                                                                      int n_pos, int cap, uint64_t *__restrict__ keys) {
    __shared__ __align__(16) uint64_t sh[RANK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    rank_sort_body(s0, train_pos, n_pos, cap, keys, (int)blockIdx.x, sh, part);
}

constexpr int BK_MAX = 1024;          // buckets of the bucket sort (n_pos > RANK_MAX)
constexpr int BK_SCRATCH = 2 * BK_MAX; // uint64 entries behind the bucketed keys: splitters | counts, cursors
static int64_t sort_capacity(int32_t n_pos) {
    int64_t c = SORT_MIN_CAP;
    const int64_t need = n_pos > RANK_MAX ? (int64_t)n_pos + BK_SCRATCH : n_pos;
    while (c < need) c <<= 1;
    return c;
}

// ---- n_pos > RANK_MAX: bucket sort -------------------------------------------------------------------------------------------
// splitters from a sorted sample -> every key's bucket (binary search among the splitters in LDS) -> counts -> scatter into
// bucket-major order -> rank sort inside every bucket (rank_sort_body: 64 keys per workgroup against the bucket's keys).  Four
// launches, every one over all CUs (the chunk sort kept P / 2048 workgroups busy for 78 bitonic stages); keys are unique, so
// the buckets' ranges are disjoint and the result is the same total order.
struct BucketArgs {
    const float *s0;
    const int32_t *train_pos;
    int n_pos, n_buckets, n_sample;   // n_sample: a power of two <= SORT_MAX_SAMPLE, a multiple of n_buckets
    uint64_t *keys;                   // [cap]: raw keys (count) -> read by scatter -> final order (sort)
    uint64_t *tmp;                    // [n_pos] bucket-major keys
    uint64_t *splitters;              // [n_buckets] splitters[0] = 0
    uint32_t *counts, *cursors;       // [n_buckets] each
    int cap;
};
constexpr int SORT_MAX_SAMPLE = 4096;

// largest b with spl[b] <= key (spl[0] = 0)
__device__ __forceinline__ int bucket_of(const uint64_t *spl, int nb, uint64_t key) {
    int lo = 0, hi = nb;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (spl[mid] <= key) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one workgroup: sample, sort it, every (n_sample / n_buckets)-th sample is a splitter; zero the counts
__global__ void __launch_bounds__(SORT_THREADS) bk_splitters(const BucketArgs a) {
    __shared__ __align__(16) uint64_t sh[SORT_MAX_SAMPLE];
    for (int j = threadIdx.x; j < a.n_sample; j += SORT_THREADS)
        sh[j] = make_pos_key(a.s0, a.train_pos, (int)(((int64_t)j * a.n_pos) / a.n_sample), a.n_pos);
    __syncthreads();
    if (a.n_sample <= SORT_THREADS) {
        // a small sample: every thread ranks its key against all of them (broadcast reads, no stage barriers)
        __shared__ uint64_t sorted[SORT_THREADS];
        const int t = threadIdx.x;
        if (t < a.n_sample) {
            const uint64_t mine = sh[t];
            int rank = 0;
            const uint4 *sh4 = reinterpret_cast<const uint4 *>(sh);       // (n_sample is a multiple of 8; two keys per 16-byte read)
            for (int j = 0; j < a.n_sample; j += 8) {
                const uint4 q0 = sh4[(j >> 1) + 0], q1 = sh4[(j >> 1) + 1], q2 = sh4[(j >> 1) + 2], q3 = sh4[(j >> 1) + 3];
                rank += ((((uint64_t)q0.y << 32) | q0.x) < mine) + ((((uint64_t)q0.w << 32) | q0.z) < mine) +
                        ((((uint64_t)q1.y << 32) | q1.x) < mine) + ((((uint64_t)q1.w << 32) | q1.z) < mine) +
                        ((((uint64_t)q2.y << 32) | q2.x) < mine) + ((((uint64_t)q2.w << 32) | q2.z) < mine) +
                        ((((uint64_t)q3.y << 32) | q3.x) < mine) + ((((uint64_t)q3.w << 32) | q3.z) < mine);
            }
            sorted[rank] = mine;       // (sample positions are distinct, so the keys - and the ranks - are)
        }
        __syncthreads();
        if (t < a.n_sample) sh[t] = sorted[t];
        __syncthreads();
    } else {
        for (int k = 2; k <= a.n_sample; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int p = threadIdx.x; p < a.n_sample / 2; p += SORT_THREADS) {
                    const int i = 2 * j * (p / j) + (p % j);
                    cmp_swap(sh[i], sh[i + j], (i & k) == 0);
                }
                __syncthreads();
            }
    }
    const int per = a.n_sample / a.n_buckets;
    for (int b = threadIdx.x; b < a.n_buckets; b += SORT_THREADS) {
        a.splitters[b] = b == 0 ? 0ull : sh[b * per];
        a.counts[b] = 0u;
        a.cursors[b] = 0u;
    }
}

// raw keys -> keys[i]; per-bucket counts (LDS histogram per workgroup, then one global add per bucket)
__global__ void __launch_bounds__(SORT_THREADS) bk_count(const BucketArgs a) {
    __shared__ uint64_t spl[BK_MAX];
    __shared__ uint32_t hist[BK_MAX];
    for (int b = threadIdx.x; b < a.n_buckets; b += SORT_THREADS) {
        spl[b] = a.splitters[b];
        hist[b] = 0u;
    }
    __syncthreads();
    const int i = blockIdx.x * SORT_THREADS + threadIdx.x;
    if (i < a.n_pos) {
        const uint64_t key = make_pos_key(a.s0, a.train_pos, i, a.n_pos);
        a.keys[i] = key;
        atomicAdd(&hist[bucket_of(spl, a.n_buckets, key)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < a.n_buckets; b += SORT_THREADS)
        if (hist[b]) atomicAdd(&a.counts[b], hist[b]);
}

// exclusive prefix of v[0 .. n) (n <= BK_MAX <= SORT_THREADS) into out[0 .. n], out[n] = total; all threads call
__device__ __forceinline__ void block_prefix(const uint32_t *v, int n, uint32_t *out, uint32_t *wsum) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t x = t < n ? v[t] : 0u;
    uint32_t inc = x;
    for (int o = 1; o < PCG_WAVE; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o);
        if (lane >= o) inc += y;
    }
    if (lane == PCG_WAVE - 1) wsum[wave] = inc;
    __syncthreads();
    uint32_t pre = 0;
    for (int w = 0; w < wave; ++w) pre += wsum[w];
    if (t < n) out[t] = pre + inc - x;
    if (t == n - 1) out[n] = pre + inc;
    __syncthreads();
}

// keys[i] -> tmp[bucket-major]: a workgroup reserves its share of every bucket with one global atomic per bucket
__global__ void __launch_bounds__(SORT_THREADS) bk_scatter(const BucketArgs a) {
    __shared__ uint64_t spl[BK_MAX];
    __shared__ uint32_t hist[BK_MAX], off[BK_MAX + 1], base[BK_MAX], wsum[SORT_THREADS / PCG_WAVE];
    for (int b = threadIdx.x; b < a.n_buckets; b += SORT_THREADS) {
        spl[b] = a.splitters[b];
        hist[b] = 0u;
    }
    __syncthreads();
    block_prefix(a.counts, a.n_buckets, off, wsum);
    const int i = blockIdx.x * SORT_THREADS + threadIdx.x;
    uint64_t key = 0;
    int b = 0;
    uint32_t r = 0;
    if (i < a.n_pos) {
        key = a.keys[i];
        b = bucket_of(spl, a.n_buckets, key);
        r = atomicAdd(&hist[b], 1u);
    }
    __syncthreads();
    for (int bb = threadIdx.x; bb < a.n_buckets; bb += SORT_THREADS)
        if (hist[bb]) base[bb] = off[bb] + atomicAdd(&a.cursors[bb], hist[bb]);
    __syncthreads();
    if (i < a.n_pos) a.tmp[base[b] + r] = key;
}

// workgroup -> (bucket, group of 64 of its keys): rank sort inside the bucket; the padding [n_pos, cap) by the last workgroups
constexpr int BK_TILE = 4096;         // (buckets hold ~1 K keys: a 32-KB tile lets three workgroups share a CU)
__global__ void __launch_bounds__(RANK_WAVES *PCG_WAVE) bk_sort(const BucketArgs a) {
    __shared__ __align__(16) uint64_t sh[BK_TILE];
    __shared__ int part[RANK_WAVES * PCG_WAVE];
    __shared__ uint32_t off[BK_MAX + 1], grp[BK_MAX + 1], gcount[BK_MAX], wsum[SORT_THREADS / PCG_WAVE];
    static_assert(RANK_WAVES * PCG_WAVE == SORT_THREADS, "block_prefix assumes SORT_THREADS threads");
    block_prefix(a.counts, a.n_buckets, off, wsum);
    for (int b = threadIdx.x; b < a.n_buckets; b += SORT_THREADS) gcount[b] = (a.counts[b] + PCG_WAVE - 1) / PCG_WAVE;
    __syncthreads();
    block_prefix(gcount, a.n_buckets, grp, wsum);
    const uint32_t me = blockIdx.x;
    if (me >= grp[a.n_buckets]) {                       // spare workgroups: the padding behind the keys
        const uint32_t spare = gridDim.x - grp[a.n_buckets], k = me - grp[a.n_buckets];
        for (int64_t t = a.n_pos + (int64_t)k * SORT_THREADS + threadIdx.x; t < a.cap; t += (int64_t)spare * SORT_THREADS) a.keys[t] = ~0ull;
        return;
    }
    int lo = 0, hi = a.n_buckets;                       // largest b with grp[b] <= me
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (grp[mid] <= me) lo = mid;
        else hi = mid;
    }
    const int b = lo, n_b = (int)a.counts[b];
    rank_sort_body<BK_TILE>(nullptr, nullptr, n_b, n_b, a.keys + off[b], (int)(me - grp[b]), sh, part, a.tmp + off[b]);
}

// ---- RANK_MAX < n_pos <= BK1_MAX: the bucket sort in ONE launch over the raw keys (round 3: four launches, 35 us at 40 K keys) ----
// Workgroup b IS bucket b.  Nothing is counted, scattered or handed from one workgroup to another: every workgroup
//   1. sorts the same sample of the raw keys (n_sample <= 1024 evenly spaced positions, ranked in LDS) and takes ITS two splitters;
//   2. streams ALL raw keys once (coalesced 8-byte loads, eight in flight per thread): counts the keys below its lower splitter -
//      that count is the bucket's offset in the sorted order - and collects the keys of its own range in LDS;
//   3. ranks the collected keys against each other and stores every key at offset + rank.
// n_pos keys x n_buckets workgroups of streaming (40 K keys, 128 buckets: 41 MB out of L2) instead of three more launches and
// their hand-offs.  A bucket holds ~n_pos / n_buckets ~ 300 - 512 keys with four samples per bucket; BK1_TILE (4096) of them fit -
// eight to thirteen times the mean: a bucket over that is reported (PCG_ST_SORT_OVERFLOW, its surplus keys dropped), never silent.
constexpr int BK1_MAX = 131072;
constexpr int BK1_TILE = 4096;        // keys of a bucket the workgroup holds in LDS (32 KB)
__global__ void __launch_bounds__(SORT_THREADS) bk_onepass(const uint64_t *__restrict__ raw, int n_pos, int n_buckets, int n_sample,
                                                           uint64_t *__restrict__ keys, int cap, uint32_t *status) {
    extern __shared__ __align__(16) unsigned char bk1_smem[];
    uint64_t *sh = reinterpret_cast<uint64_t *>(bk1_smem);     // [BK1_TILE] the bucket's keys
    uint64_t *samp = sh + BK1_TILE;                            // [SORT_THREADS] the sample
    __shared__ int red[SORT_THREADS / PCG_WAVE];
    __shared__ int n_in;
    const int t = threadIdx.x, b = blockIdx.x;
    if (b >= n_buckets) {                                       // spare workgroups: the padding behind the keys
        const int spare = (int)gridDim.x - n_buckets, k = b - n_buckets;
        for (int64_t i = n_pos + (int64_t)k * SORT_THREADS + t; i < cap; i += (int64_t)spare * SORT_THREADS) keys[i] = ~0ull;
        return;
    }
    // this thread's first keys of the stream are requested with the sample (nothing of them depends on the splitters)
    constexpr int SU = 8;
    uint64_t kv[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
        const int i = t + u * SORT_THREADS;
        kv[u] = raw[i < n_pos ? i : n_pos - 1];
    }
    // the sample: n_sample (a power of two <= 1024) evenly spaced raw keys, sorted in LDS by a bitonic network (45 stages for 512
    // samples, ~0.2 us each - sixteen waves meet at a barrier per stage; ranking every sample against every other cost more in
    // 64-bit compares, one wave running the network alone more in LDS round trips: both measured)
    uint64_t *sorted = samp;
    samp[t] = t < n_sample ? raw[(int)(((int64_t)t * n_pos) / n_sample)] : ~0ull;
    if (t == 0) n_in = 0;
    __syncthreads();
    for (int k = 2; k <= n_sample; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (t < n_sample / 2) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // (j is a power of two)
                cmp_swap(samp[i], samp[i + j], (i & k) == 0);
            }
            __syncthreads();
        }
    const int per = n_sample / n_buckets;
    const uint64_t lo = b == 0 ? 0ull : sorted[b * per], hi = b == n_buckets - 1 ? ~0ull : sorted[(b + 1) * per];
    // ---- the stream: keys below lo are counted, keys in [lo, hi) collected ----
    int below = 0;
    for (int base = 0; base < n_pos; base += SU * SORT_THREADS) {
        uint64_t nx[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {                         // the next round's keys before this round's are looked at
            const int i = base + SU * SORT_THREADS + t + u * SORT_THREADS;
            nx[u] = raw[i < n_pos ? i : n_pos - 1];
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int i = base + t + u * SORT_THREADS;
            if (i < n_pos) {
                below += kv[u] < lo;
                if (kv[u] >= lo && kv[u] < hi) {
                    const int at = atomicAdd(&n_in, 1);
                    if (at < BK1_TILE) sh[at] = kv[u];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) kv[u] = nx[u];
    }
    for (int o = 1; o < PCG_WAVE; o <<= 1) below += __shfl_xor(below, o);
    if ((t & (PCG_WAVE - 1)) == 0) red[t >> 6] = below;
    __syncthreads();
    int off = 0;
#pragma unroll
    for (int w = 0; w < SORT_THREADS / PCG_WAVE; ++w) off += red[w];
    int n = n_in;
    if (n > BK1_TILE) {
        if (t == 0 && status) atomicOr(status, (uint32_t)PCG_ST_SORT_OVERFLOW);
        n = BK1_TILE;
    }
    // ---- the bucket's keys, sorted in LDS (bitonic over the next power of two, padded with all-ones keys), stored at offset + i.
    //      (Measured alternatives, 40 K keys: every key ranked against every other of its bucket - by one thread a key, or by all
    //      threads over slices with LDS atomics - 36 / 44 us a launch against this network's 30; scripts/sort_probe.py) ----
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (int i = n + t; i < np2; i += SORT_THREADS) sh[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int p = t; p < np2 / 2; p += SORT_THREADS) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                cmp_swap(sh[i], sh[i + j], (i & k) == 0);
            }
            __syncthreads();
        }
    for (int i = t; i < n; i += SORT_THREADS) {
        const int dst = off + i;
        if (dst < n_pos) keys[dst] = sh[i];                    // (a counter never indexes unchecked)
    }
}

// the launch geometry of the one-pass sort
static void bk1_geometry(int n_pos, int &n_buckets, int &n_sample) {
    int nb = 32;
    while (nb < 256 && (int64_t)nb * 512 < n_pos) nb <<= 1;    // ~300 - 512 keys per bucket
    n_buckets = nb;
    // four samples a bucket: a bucket's size is then ~Gamma(4) around its mean - BK1_TILE is eight to thirteen times the mean, a
    // bucket beyond it a < 1e-10 event (and reported)
    n_sample = nb * 4;                                         // (<= 1024: a multiple of n_buckets, both powers of two)
}
int launch_bk_onepass(const uint64_t *raw, int n_pos, uint64_t *keys, int cap, uint32_t *status, hipStream_t st) {
    int nb, ns;
    bk1_geometry(n_pos, nb, ns);
    const size_t smem = sizeof(uint64_t) * (BK1_TILE + SORT_THREADS);
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(bk_onepass), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
            return PCG_E_LAUNCH;
        attr_done = true;
    }
    hipLaunchKernelGGL(bk_onepass, dim3(nb + 8), dim3(SORT_THREADS), smem, st, raw, n_pos, nb, ns, keys, cap, status);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // namespace pcg

extern "C" {

/* 1: n_pos train positives are sorted by the one-launch bucket sort when their unsorted keys exist already
 * (pcg_choose_gather_train forms them beside the score pass): 16384 < n_pos <= 131072 (host helper) */
int32_t pcg_pos_sort_one_launch(int32_t n_pos) { return n_pos > pcg::RANK_MAX && n_pos <= pcg::BK1_MAX ? 1 : 0; }

/* the one-launch bucket sort over raw keys that exist already (pos_keys' scratch half) -> pos_keys' first half */
int pcg_pos_sort_raw(const pcg_graph_desc *g, uint64_t *keys, uint32_t *status, void *stream) {
    if (!g || !keys || !pcg_pos_sort_one_launch(g->n_pos)) return PCG_E_ARG;
    const int64_t cap = pcg::sort_capacity(g->n_pos);
    return pcg::launch_bk_onepass(keys + cap, g->n_pos, keys, (int)cap, status, static_cast<hipStream_t>(stream));
}

int64_t pcg_pos_sort_capacity(int32_t n_pos) {
    if (n_pos < 0) return PCG_E_ARG;
    const int64_t cap = pcg::sort_capacity(n_pos);
    return 2 * cap;     // second half: the bucket sort's scatter buffer + splitters / the unsorted keys pcg_step_front forms beside the score pass
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
    // bucket sort: [cap .. cap + n_pos) bucket-major keys | splitters | counts | cursors (sort_capacity left room for them)
    pcg::BucketArgs b;
    b.s0 = s0;
    b.train_pos = g->train_pos;
    b.n_pos = g->n_pos;
    int nb = 16;
    while (nb < pcg::BK_MAX && (int64_t)nb * 1024 < g->n_pos) nb <<= 1;          // ~1024 keys per bucket
    b.n_buckets = nb;
    int ns = nb * 8;                          // (<= 1024 samples are rank-sorted by one workgroup in a few microseconds)
    b.n_sample = ns > pcg::SORT_MAX_SAMPLE ? pcg::SORT_MAX_SAMPLE : ns;
    b.keys = keys;
    b.tmp = keys + cap;
    b.splitters = b.tmp + g->n_pos;
    b.counts = reinterpret_cast<uint32_t *>(b.splitters + pcg::BK_MAX);
    b.cursors = b.counts + pcg::BK_MAX;
    b.cap = (int)cap;
    const int nblk = (g->n_pos + pcg::SORT_THREADS - 1) / pcg::SORT_THREADS;
    hipLaunchKernelGGL(pcg::bk_splitters, dim3(1), dim3(pcg::SORT_THREADS), 0, st, b);
    PCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(pcg::bk_count, dim3(nblk), dim3(pcg::SORT_THREADS), 0, st, b);
    PCG_LAUNCH_CHECK();
    hipLaunchKernelGGL(pcg::bk_scatter, dim3(nblk), dim3(pcg::SORT_THREADS), 0, st, b);
    PCG_LAUNCH_CHECK();
    // a workgroup per 64 keys of a bucket (at most n_pos / 64 + one per bucket), + a few for the padding
    hipLaunchKernelGGL(pcg::bk_sort, dim3((g->n_pos + PCG_WAVE - 1) / PCG_WAVE + nb + 8), dim3(pcg::RANK_WAVES * PCG_WAVE), 0, st, b);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
