// Halo exchange helpers of the node-partitioned multi-GPU path (pc-gnn_amd/dist.py).
// The selection list of a step holds GLOBAL node ids (-1 = hole).  Before the gather it is re-indexed into the rank's
// extended feature table  [ owned rows | train-pos rows | halo ].  All of it is work on the list's ENTRIES (the chunk
// table says which entries are in use) - nothing here is sized by, or walks, the node-id space:
//   classify : owned id -> row number; train-pos id -> n_local + its position (binary search in the sorted train-pos ids);
//              remote id -> marked (encoded as -(id+2)) and inserted into an open-addressing hash table (atomicCAS, linear
//              probing); the thread whose insert is the first of an id counts it for the id's owner
//   assign   : every occupied table slot gets a halo slot inside its owner's range (owners in rank order, the order inside
//              an owner is whatever the atomics give - it decides only WHERE a fetched row sits, never a sum's order) and
//              the id goes to the request list
//   remap    : marked entries -> halo_base + the slot found in the table
// The reference has no distributed code; see SURVEY.md section 8(e).
#include "choose.h"

namespace pcg {

constexpr uint32_t HALO_EMPTY = 0xFFFFFFFFu;
constexpr int HALO_MAX_WORLD = 64;

struct HaloArgs {
    int32_t *list;
    const int4 *chunk_desc;
    const uint32_t *n_chunks;        // device word (the plan's chunk count)
    uint32_t chunk_cap;              // entries of chunk_desc (a device-side counter never indexes unchecked)
    int32_t lo, hi, n_local;
    const int32_t *pos_ids;          // [n_pos] train-pos ids ascending
    const int32_t *pos_idx;          // [n_pos] their row in the replicated train-pos block
    int32_t n_pos;
    const int32_t *bounds;           // [world + 1] partition: rank r owns [bounds[r], bounds[r + 1])
    int32_t world;
    uint32_t *keys, *vals;           // hash table, `mask` + 1 slots (a power of two)
    uint32_t mask;
    uint32_t *owner_count;           // [world] unique remote ids per owner (zero on entry)
    uint32_t *owner_fill;            // [world] (zero on entry)
    int32_t *uniq;                   // [halo_cap] request list, grouped by owner
    int32_t halo_cap, halo_base;
    uint32_t *overflow;              // device word: OR-ed with 1 (table full) / 2 (more unique ids than halo_cap)
};

__device__ __forceinline__ uint32_t halo_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

__device__ __forceinline__ int owner_of(const int32_t *s_bounds, int world, int32_t id) {
    int lo = 0, hi = world;          // largest r with bounds[r] <= id
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_bounds[mid] <= id) lo = mid;
        else hi = mid;
    }
    return lo;
}

// one wave per 128-entry chunk of the list (grid-stride)
__global__ void __launch_bounds__(256) halo_classify_kernel(const HaloArgs a) {
    __shared__ int32_t s_bounds[HALO_MAX_WORLD + 1];
    for (int i = threadIdx.x; i <= a.world; i += blockDim.x) s_bounds[i] = a.bounds[i];
    __syncthreads();
    const int lane = lane_id();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t total = *a.n_chunks < a.chunk_cap ? *a.n_chunks : a.chunk_cap;
    for (uint32_t ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ch < total; ch += nwaves) {
        const int4 desc = a.chunk_desc[ch];
        for (int i = lane; i < desc.z; i += PCG_WAVE) {
            int32_t *e = a.list + desc.y + i;
            const int32_t id = *e;
            if (id < 0) continue;
            if (id >= a.lo && id < a.hi) {
                *e = id - a.lo;
                continue;
            }
            int plo = 0, phi = a.n_pos;                      // first train-pos id >= id
            while (plo < phi) {
                const int mid = (plo + phi) >> 1;
                if (a.pos_ids[mid] < id) plo = mid + 1;
                else phi = mid;
            }
            if (plo < a.n_pos && a.pos_ids[plo] == id) {
                *e = a.n_local + a.pos_idx[plo];
                continue;
            }
            *e = -(id + 2);
            uint32_t h = halo_hash((uint32_t)id) & a.mask;
            bool done = false;
            for (uint32_t probe = 0; probe <= a.mask; ++probe) {
                const uint32_t old = atomicCAS(&a.keys[h], HALO_EMPTY, (uint32_t)id);
                if (old == HALO_EMPTY) {                     // first insert of this id
                    atomicAdd(&a.owner_count[owner_of(s_bounds, a.world, id)], 1u);
                    done = true;
                    break;
                }
                if (old == (uint32_t)id) {
                    done = true;
                    break;
                }
                h = (h + 1) & a.mask;
            }
            if (!done) atomicOr(a.overflow, 1u);
        }
    }
}

__global__ void __launch_bounds__(256) halo_assign_kernel(const HaloArgs a) {
    __shared__ int32_t s_bounds[HALO_MAX_WORLD + 1];
    __shared__ uint32_t s_off[HALO_MAX_WORLD + 1];
    for (int i = threadIdx.x; i <= a.world; i += blockDim.x) s_bounds[i] = a.bounds[i];
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int r = 0; r < a.world; ++r) {
            s_off[r] = run;
            run += a.owner_count[r];
        }
        s_off[a.world] = run;
        if (run > (uint32_t)a.halo_cap) atomicOr(a.overflow, 2u);
    }
    __syncthreads();
    const bool fits = s_off[a.world] <= (uint32_t)a.halo_cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h <= a.mask; h += stride) {
        const uint32_t key = a.keys[h];
        if (key == HALO_EMPTY) continue;
        const int o = owner_of(s_bounds, a.world, (int32_t)key);
        const uint32_t slot = s_off[o] + atomicAdd(&a.owner_fill[o], 1u);
        a.vals[h] = slot;
        if (fits && slot < (uint32_t)a.halo_cap) a.uniq[slot] = (int32_t)key;     // (a device-side counter never indexes unchecked)
    }
}

__global__ void __launch_bounds__(256) halo_remap_kernel(const HaloArgs a) {
    const int lane = lane_id();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t total = *a.n_chunks < a.chunk_cap ? *a.n_chunks : a.chunk_cap;
    for (uint32_t ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); ch < total; ch += nwaves) {
        const int4 desc = a.chunk_desc[ch];
        for (int i = lane; i < desc.z; i += PCG_WAVE) {
            int32_t *e = a.list + desc.y + i;
            const int32_t v = *e;
            if (v > -2) continue;
            const uint32_t id = (uint32_t)(-v - 2);
            uint32_t h = halo_hash(id) & a.mask;
            int32_t out = -1;                                // (an id the full table could not take: a hole; the overflow word is set)
            for (uint32_t probe = 0; probe <= a.mask; ++probe) {
                const uint32_t key = a.keys[h];
                if (key == id) {
                    const uint32_t slot = a.vals[h];
                    if (slot < (uint32_t)a.halo_cap) out = a.halo_base + (int32_t)slot;      // (never a row beyond the table)
                    break;
                }
                if (key == HALO_EMPTY) break;
                h = (h + 1) & a.mask;
            }
            *e = out;
        }
    }
}

}  // namespace pcg

extern "C" {

int64_t pcg_halo_table_slots(int32_t halo_cap) {
    if (halo_cap < 0) return PCG_E_ARG;
    int64_t t = 1024;
    while (t < 2 * (int64_t)halo_cap) t <<= 1;
    return t;
}

static int halo_args(pcg::HaloArgs &a, const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, int32_t lo,
                     int32_t hi, int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos,
                     const int32_t *bounds, int32_t world, uint32_t *table, int64_t table_slots, uint32_t *counts, int32_t *uniq,
                     int32_t halo_cap, int32_t halo_base) {
    if (!g || !workspace || B < 1 || list_capacity < 1 || !bounds || world < 1 || world > pcg::HALO_MAX_WORLD || !table || !counts ||
        !uniq || halo_cap < 0 || lo > hi || n_pos < 0 || (n_pos > 0 && (!pos_ids || !pos_idx)))
        return PCG_E_ARG;
    if (table_slots < 1024 || (table_slots & (table_slots - 1)) != 0 || table_slots > (1ll << 31)) return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w);
    a.list = w.list;
    a.chunk_desc = w.chunk_desc;
    a.n_chunks = w.counters + pcg::C_NCHUNK;
    a.chunk_cap = (uint32_t)w.chunk_cap;
    a.lo = lo;
    a.hi = hi;
    a.n_local = n_local;
    a.pos_ids = pos_ids;
    a.pos_idx = pos_idx;
    a.n_pos = n_pos;
    a.bounds = bounds;
    a.world = world;
    a.keys = table;
    a.vals = table + table_slots;
    a.mask = (uint32_t)(table_slots - 1);
    a.owner_count = counts;
    a.owner_fill = counts + pcg::HALO_MAX_WORLD;
    a.overflow = counts + 2 * pcg::HALO_MAX_WORLD;
    a.uniq = uniq;
    a.halo_cap = halo_cap;
    a.halo_base = halo_base;
    return PCG_OK;
}

/* table: uint32 [2 * table_slots] (keys | values), keys all 0xFFFFFFFF on entry;  counts: uint32 [2 * 64 + 1], zero on
 * entry: [0, world) unique remote ids per owner, [64, 64 + world) scratch, [128] overflow bits. */
int pcg_halo_classify(const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, int32_t lo, int32_t hi,
                      int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos, const int32_t *bounds,
                      int32_t world, uint32_t *table, int64_t table_slots, uint32_t *counts, int32_t *uniq, int32_t halo_cap,
                      int32_t halo_base, void *stream) {
    pcg::HaloArgs a;
    const int rc = halo_args(a, g, B, workspace, list_capacity, lo, hi, n_local, pos_ids, pos_idx, n_pos, bounds, world, table,
                             table_slots, counts, uniq, halo_cap, halo_base);
    if (rc != PCG_OK) return rc;
    hipLaunchKernelGGL(pcg::halo_classify_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    int blocks = (int)((table_slots + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcg::halo_assign_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_halo_remap(const pcg_graph_desc *g, int32_t B, void *workspace, int64_t list_capacity, uint32_t *table,
                   int64_t table_slots, int32_t halo_cap, int32_t halo_base, void *stream) {
    if (!g || !workspace || !table || B < 1 || list_capacity < 1 || table_slots < 1024 || (table_slots & (table_slots - 1)) != 0)
        return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w);
    pcg::HaloArgs a = {};
    a.list = w.list;
    a.chunk_desc = w.chunk_desc;
    a.n_chunks = w.counters + pcg::C_NCHUNK;
    a.chunk_cap = (uint32_t)w.chunk_cap;
    a.keys = table;
    a.vals = table + table_slots;
    a.mask = (uint32_t)(table_slots - 1);
    a.halo_cap = halo_cap;
    a.halo_base = halo_base;
    hipLaunchKernelGGL(pcg::halo_remap_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
