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

const char *pcg_version(void) { return "pcgnn_hip gfx950 abi2"; }
int pcg_abi_version(void) { return 2; }

}  // extern "C"
