// Label-aware scores (src/layers.py:230-243) and plain row gathers.
// score_table streams the whole feature table once per step (weights change every
// step): 64/lpr rows per wave-instruction, float4 per lane, rows adjacent in
// memory => every wave-instruction is one fully coalesced 1 KiB read.
#include "common.h"

namespace pcg {

__global__ void __launch_bounds__(256) score_table_kernel(const float *__restrict__ X, int feat_dim, int stride,
                                                          const float *__restrict__ W, const float *__restrict__ bias,
                                                          int64_t row_begin, int64_t row_end, float *__restrict__ s0) {
    score_table_body(X, feat_dim, stride, W, bias, row_begin, row_end, s0, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void __launch_bounds__(256) score_rows_kernel(const float *__restrict__ X, int feat_dim, int stride,
                                                         const float *__restrict__ W, const float *__restrict__ bias,
                                                         const int32_t *__restrict__ ids, int n_ids,
                                                         float *__restrict__ out) {
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int i = wave_global * rpw + slot;
    const bool ok = i < n_ids;
    const float *row = X + (size_t)(ok ? ids[i] : 0) * stride;
    float p0 = ok ? score_partial(row, W, feat_dim, stride, sub, lpr) : 0.f;
    float p1 = ok ? score_partial(row, W + feat_dim, feat_dim, stride, sub, lpr) : 0.f;
    p0 = score_reduce(p0, lpr);
    p1 = score_reduce(p1, lpr);
    if (ok && sub == 0) {
        out[2 * i + 0] = p0 + bias[0];
        out[2 * i + 1] = p1 + bias[1];
    }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const float *__restrict__ X, int feat_dim, int stride,
                                                          const int32_t *__restrict__ ids, int n_ids,
                                                          float *__restrict__ out, int out_stride) {
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int i = wave_global * rpw + slot;
    if (i >= n_ids) return;
    const float *row = X + (size_t)ids[i] * stride;
    float *o = out + (size_t)i * out_stride;
    for (int ch = sub; ch < (stride >> 2); ch += lpr) {
        const float4 v = *reinterpret_cast<const float4 *>(row + 4 * ch);
        const int f = 4 * ch;
        if (f + 0 < feat_dim) o[f + 0] = v.x;
        if (f + 1 < feat_dim) o[f + 1] = v.y;
        if (f + 2 < feat_dim) o[f + 2] = v.z;
        if (f + 3 < feat_dim) o[f + 3] = v.w;
    }
}

// ---- which rows a batch's selection can read: a byte map per batch, built once per epoch (it depends on the picked ids and the
// CSR rows only, like the plan): the centres and every neighbour they have in any relation
__global__ void __launch_bounds__(256) zero_bytes_kernel(uint4 *__restrict__ p, int64_t n16, uint32_t *__restrict__ queue) {
    if (queue && blockIdx.x == 0 && threadIdx.x == 0) queue[0] = 0u;        // (mark_touched_kernel's queue of long rows)
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
struct MarkArgs {
    const int32_t *nodes;
    int32_t n_total, B, n_rel;
    const int64_t *indptr[PCG_MAX_REL];
    const int32_t *indices[PCG_MAX_REL];
    unsigned char *maps;
    int64_t map_stride, n_nodes;
    const int32_t *train_pos;          // their scores are read too: by the train-pos sort when there are too many of them for the
    int32_t n_pos;                     // front launch to form the keys from the feature rows (pcg_pos_sort reads s0[train_pos])
    uint32_t *queue;                   // [4 + n_rel * n_total]: [0] = how many rows of more than MARK_LONG neighbours, [4 ..] = their items
};
// A row of a hub - a power-law graph has centres with 10^5 .. 10^6 neighbours - is not one wave's work (256 ids per round trip:
// the longest row set the kernel's duration, 1.6 ms per epoch at 10 M nodes / 200 M edges): such rows are queued by the item pass
// and marked by the whole grid in a pass of their own (mark_long_kernel).
constexpr int MARK_LONG = 4096;
__global__ void __launch_bounds__(256) mark_touched_kernel(const MarkArgs a) {
    const int lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    // items in batch-major order: the waves in flight work on two or three batches' maps (a map is one byte per node: 10 MB at
    // 10 M nodes), not on all of them at once - the scattered one-byte stores then mostly find their line in a cache
    const int n_slots = (a.n_total + a.B - 1) / a.B;
    {
        const int64_t nthreads = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (int64_t i = tid; i < (int64_t)a.n_pos * n_slots; i += nthreads) {
            const int slot = (int)(i / a.n_pos);
            const int32_t v = a.train_pos[i - (int64_t)slot * a.n_pos];
            if ((uint32_t)v < (uint64_t)a.n_nodes) a.maps[(int64_t)slot * a.map_stride + v] = 1;
        }
    }
    const int64_t per_slot = (int64_t)a.n_rel * a.B;
    const int64_t items = per_slot * n_slots;
    for (int64_t it = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); it < items; it += nwaves) {
        const int slot = (int)(it / per_slot);
        const int rem = (int)(it - (int64_t)slot * per_slot);
        const int r = rem / a.B;
        const int i = slot * a.B + (rem - r * a.B);
        if (i >= a.n_total) continue;                                       // (the last batch may be shorter)
        const int32_t node = a.nodes[i];
        unsigned char *__restrict__ map = a.maps + (int64_t)(i / a.B) * a.map_stride;
        if (r == 0 && lane == 0) map[node] = 1;                              // the centre's own score is read too
        const int64_t beg = a.indptr[r][node], end = a.indptr[r][node + 1];
        const int32_t *__restrict__ nbr = a.indices[r];
        if (a.queue && end - beg > MARK_LONG) {                             // (wave-uniform) the whole grid's work: mark_long_kernel
            if (lane == 0) a.queue[4 + atomicAdd(a.queue, 1u)] = (uint32_t)it;
            continue;
        }
        constexpr int CU = 4;                                               // four loads of 64 ids in flight (unconditional: clamped)
        for (int64_t j0 = beg; j0 < end; j0 += CU * PCG_WAVE) {
            int32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int64_t j = j0 + u * PCG_WAVE + lane;
                idv[u] = nbr[j < end ? j : end - 1];
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int64_t j = j0 + u * PCG_WAVE + lane;
                if (j < end && (uint32_t)idv[u] < (uint64_t)a.n_nodes) map[idv[u]] = 1;
            }
        }
    }
}

// the queued rows: every workgroup learns where they are (a batch of 256 at a time, one per thread, in LDS), then the grid's
// waves share every row's neighbours, 256 per wave and turn
__global__ void __launch_bounds__(256) mark_long_kernel(const MarkArgs a) {
    __shared__ long long s_beg[256], s_end[256], s_map[256];
    __shared__ int s_rel[256];
    __shared__ long long s_pre[256];                                        // pieces of the rows up to and including this one
    const int lane = lane_id();
    const unsigned n = a.queue[0];
    const int64_t per_slot = (int64_t)a.n_rel * a.B;
    const int64_t wave_g = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * (blockDim.x >> 6);
    constexpr int CU = 4, PIECE = CU * PCG_WAVE;                            // a piece: 256 neighbour ids, four loads of 64 in flight
    for (unsigned e0 = 0; e0 < n; e0 += 256) {
        const unsigned e = e0 + threadIdx.x;
        long long beg = 0, end = 0;
        if (e < n) {
            const int64_t it = (int64_t)a.queue[4 + e];
            const int slot = (int)(it / per_slot);
            const int rem = (int)(it - (int64_t)slot * per_slot);
            const int r = rem / a.B;
            const int i = slot * a.B + (rem - r * a.B);
            const int32_t node = a.nodes[i];
            beg = a.indptr[r][node];
            end = a.indptr[r][node + 1];
            s_map[threadIdx.x] = (long long)slot * a.map_stride;
            s_rel[threadIdx.x] = r;
        }
        s_beg[threadIdx.x] = beg;
        s_end[threadIdx.x] = end;
        s_pre[threadIdx.x] = (end - beg + PIECE - 1) / PIECE;
        __syncthreads();
        if (threadIdx.x == 0)                                               // (256 adds: nothing beside the marking itself)
            for (int q = 1; q < 256; ++q) s_pre[q] += s_pre[q - 1];
        __syncthreads();
        // the batch's pieces, all rows together, dealt out over the grid's waves: a wave finds its piece's row by bisection
        const long long total = s_pre[255];
        for (long long p = wave_g; p < total; p += n_waves) {
            int lo = 0, hi = 255;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s_pre[mid] > p) hi = mid;
                else lo = mid + 1;
            }
            const int q = lo;
            const long long c = p - (q > 0 ? s_pre[q - 1] : 0);
            const int64_t endq = s_end[q], j0 = s_beg[q] + c * PIECE;
            const int r = s_rel[q];
            unsigned char *__restrict__ map = a.maps + s_map[q];
            int32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int64_t j = j0 + u * PCG_WAVE + lane;
                const int64_t jc = j < endq ? j : endq - 1;
                int32_t v = 0;
                for (int rr = 0; rr < PCG_MAX_REL; ++rr)                    // (a kernel-argument array: indexed by a constant)
                    if (rr == r) v = a.indices[rr][jc];
                idv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int64_t j = j0 + u * PCG_WAVE + lane;
                if (j < endq && (uint32_t)idv[u] < (uint64_t)a.n_nodes) map[idv[u]] = 1;
            }
        }
        __syncthreads();
    }
}

static int check_graph_features(const pcg_graph_desc *g) {
    if (!g || !g->X || g->feat_dim < 1 || g->feat_stride < g->feat_dim || g->feat_stride % 4 != 0) return PCG_E_ARG;
    if ((reinterpret_cast<uintptr_t>(g->X) & 15u) != 0) return PCG_E_ARG;
    return PCG_OK;
}

}  // namespace pcg

extern "C" {

int pcg_score_table(const pcg_graph_desc *g, const float *W, const float *b, int64_t row_begin, int64_t row_end,
                    float *s0, void *stream) {
    if (pcg::check_graph_features(g) != PCG_OK || !W || !b || !s0) return PCG_E_ARG;
    if (row_begin < 0 || row_end > g->n_nodes || row_begin > row_end) return PCG_E_ARG;
    if (row_begin == row_end) return PCG_OK;
    const int64_t blocks = pcg::score_table_blocks(row_end - row_begin, g->feat_stride);
    hipLaunchKernelGGL(pcg::score_table_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       g->X, g->feat_dim, g->feat_stride, W, b, row_begin, row_end, s0);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int64_t pcg_touched_bytes(int64_t n_nodes) { return n_nodes < 0 ? PCG_E_ARG : pcg::touched_bytes(n_nodes); }

/* byte maps of the rows the batches nodes[s * B, min((s + 1) * B, n_total)) can read the score of: maps + s * map_stride
 * (map_stride >= pcg_touched_bytes(n_nodes), a multiple of 16) is zeroed, then map[v] = 1 for every centre v of batch s, every
 * neighbour of v in any relation, and every train positive.  Two launches for all batches (per epoch, like pcg_plan_batches). */
int pcg_mark_touched(const pcg_graph_desc *g, const int32_t *nodes, int32_t n_total, int32_t B, uint8_t *maps, int64_t map_stride,
                     uint32_t *queue, void *stream) {
    if (!g || !nodes || n_total < 0 || B < 1 || !maps || g->n_rel < 1 || g->n_rel > PCG_MAX_REL) return PCG_E_ARG;
    if (!queue && g->max_degree > pcg::MARK_LONG) return PCG_E_ARG;
    if (map_stride < pcg::touched_bytes(g->n_nodes) || (map_stride & 15) != 0 || (reinterpret_cast<uintptr_t>(maps) & 15u) != 0)
        return PCG_E_ARG;
    if (n_total == 0) return PCG_OK;
    const int n_slots = (n_total + B - 1) / B;
    pcg::MarkArgs a;
    a.nodes = nodes;
    a.n_total = n_total;
    a.B = B;
    a.n_rel = g->n_rel;
    for (int r = 0; r < PCG_MAX_REL; ++r) {
        a.indptr[r] = r < g->n_rel ? g->indptr[r] : nullptr;
        a.indices[r] = r < g->n_rel ? g->indices[r] : nullptr;
        if (r < g->n_rel && (!a.indptr[r] || !a.indices[r])) return PCG_E_ARG;
    }
    a.maps = maps;
    a.map_stride = map_stride;
    a.n_nodes = g->n_nodes;
    a.train_pos = g->train_pos;
    a.n_pos = g->train_pos ? g->n_pos : 0;
    a.queue = g->max_degree > pcg::MARK_LONG ? queue : nullptr;        // (no such rows in this graph: no queue, no third launch)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t n16 = (int64_t)n_slots * map_stride / 16;
    int zb = (int)((n16 + 255) / 256);
    zb = zb > 4096 ? 4096 : zb;
    hipLaunchKernelGGL(pcg::zero_bytes_kernel, dim3(zb), dim3(256), 0, st, reinterpret_cast<uint4 *>(maps), n16, a.queue);
    PCG_LAUNCH_CHECK();
    int64_t mb = ((int64_t)g->n_rel * n_total + 3) / 4;
    mb = mb > 2048 ? 2048 : mb;
    hipLaunchKernelGGL(pcg::mark_touched_kernel, dim3((int)mb), dim3(256), 0, st, a);
    PCG_LAUNCH_CHECK();
    if (a.queue) {
        hipLaunchKernelGGL(pcg::mark_long_kernel, dim3(2048), dim3(256), 0, st, a);
        PCG_LAUNCH_CHECK();
    }
    return PCG_OK;
}

int pcg_score_rows(const pcg_graph_desc *g, const float *W, const float *b, const int32_t *ids, int32_t n_ids,
                   float *out, void *stream) {
    if (pcg::check_graph_features(g) != PCG_OK || !W || !b || !ids || !out || n_ids < 0) return PCG_E_ARG;
    if (n_ids == 0) return PCG_OK;
    const int rpw = PCG_WAVE / pcg::lanes_per_row(g->feat_stride);
    const int blocks = (n_ids + 4 * rpw - 1) / (4 * rpw);
    hipLaunchKernelGGL(pcg::score_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g->X,
                       g->feat_dim, g->feat_stride, W, b, ids, n_ids, out);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_gather_rows(const pcg_graph_desc *g, const int32_t *ids, int32_t n_ids, float *out, int32_t out_stride,
                    void *stream) {
    if (pcg::check_graph_features(g) != PCG_OK || !ids || !out || n_ids < 0 || out_stride < g->feat_dim)
        return PCG_E_ARG;
    if (n_ids == 0) return PCG_OK;
    const int rpw = PCG_WAVE / pcg::lanes_per_row(g->feat_stride);
    const int blocks = (n_ids + 4 * rpw - 1) / (4 * rpw);
    hipLaunchKernelGGL(pcg::gather_rows_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g->X,
                       g->feat_dim, g->feat_stride, ids, n_ids, out, out_stride);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

const char *pcg_version(void) { return "pcgnn_hip gfx950 abi3"; }
int pcg_abi_version(void) { return 3; }

}  // extern "C"
