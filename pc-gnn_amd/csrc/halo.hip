// Halo exchange helpers of the node-partitioned multi-GPU path (pc-gnn_amd/dist.py).
// The selection list of a step holds GLOBAL node ids (-1 = hole).  Before the gather it is
// re-indexed into the rank's extended feature table  [ owned rows | train-pos rows | halo ]:
//   classify : owned id -> row number; train-pos id -> n_local + its position; remote id ->
//              marked (encoded as -(id+2)) and flagged in flag[id]
//   (host side: inclusive scan of flag = slot; per-owner counts; all-to-all of ids and rows)
//   compact  : uniq[slot[id]-1] = id for flagged ids  (ascending = grouped by owner)
//   remap    : marked entries -> halo_base + slot[id]-1; flags cleared for the next step
// The reference has no distributed code; see SURVEY.md section 8(e).
#include "common.h"

namespace pcg {

__global__ void __launch_bounds__(256) halo_classify_kernel(int32_t *__restrict__ list, const int64_t *__restrict__ total,
                                                            int64_t cap, int32_t lo, int32_t hi, int32_t n_local,
                                                            const int32_t *__restrict__ posmap, int32_t *__restrict__ flag) {
    int64_t n = *total;
    if (n > cap) n = cap;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t id = list[i];
        if (id < 0) continue;
        if (id >= lo && id < hi) {
            list[i] = id - lo;
        } else {
            const int32_t pm = posmap[id];
            if (pm >= 0) {
                list[i] = n_local + pm;
            } else {
                flag[id] = 1;              // benign race: every writer stores 1
                list[i] = -(id + 2);
            }
        }
    }
}

__global__ void __launch_bounds__(256) halo_compact_kernel(const int32_t *__restrict__ flag, const int32_t *__restrict__ slot,
                                                           int32_t n_nodes, int32_t *__restrict__ uniq) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < n_nodes; v += stride)
        if (flag[v]) uniq[slot[v] - 1] = (int32_t)v;
}

__global__ void __launch_bounds__(256) halo_remap_kernel(int32_t *__restrict__ list, const int64_t *__restrict__ total,
                                                         int64_t cap, const int32_t *__restrict__ slot, int32_t halo_base,
                                                         int32_t *__restrict__ flag) {
    int64_t n = *total;
    if (n > cap) n = cap;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t e = list[i];
        if (e <= -2) {
            const int32_t id = -e - 2;
            list[i] = halo_base + slot[id] - 1;
            flag[id] = 0;                  // ready for the next step
        }
    }
}

}  // namespace pcg

extern "C" {

int pcg_halo_classify(int32_t *list, const int64_t *total, int64_t list_capacity, int32_t lo, int32_t hi,
                      int32_t n_local, const int32_t *posmap, int32_t *flag, void *stream) {
    if (!list || !total || !posmap || !flag || list_capacity < 0 || lo > hi) return PCG_E_ARG;
    hipLaunchKernelGGL(pcg::halo_classify_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), list, total,
                       list_capacity, lo, hi, n_local, posmap, flag);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_halo_compact(const int32_t *flag, const int32_t *slot, int32_t n_nodes, int32_t *uniq, void *stream) {
    if (!flag || !slot || !uniq || n_nodes < 0) return PCG_E_ARG;
    if (n_nodes == 0) return PCG_OK;
    int blocks = (n_nodes + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcg::halo_compact_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), flag, slot,
                       n_nodes, uniq);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_halo_remap(int32_t *list, const int64_t *total, int64_t list_capacity, const int32_t *slot, int32_t halo_base,
                   int32_t *flag, void *stream) {
    if (!list || !total || !slot || !flag || list_capacity < 0) return PCG_E_ARG;
    hipLaunchKernelGGL(pcg::halo_remap_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), list, total,
                       list_capacity, slot, halo_base, flag);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
