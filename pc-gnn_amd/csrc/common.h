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

}  // namespace pcg
