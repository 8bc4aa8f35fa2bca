// Halo exchange helpers of the node-partitioned multi-GPU path (pc-gnn_amd/dist.py).
// A rank's feature table is  [ owned rows | train-pos rows (replicated) | halo ];  CSR rows and selection lists hold GLOBAL
// node ids.  Feature rows never change, so a fetched row stays valid: the exchange runs once per WINDOW of steps, over every
// neighbour the window's centres have, and a step itself only looks rows up:
//   collect : walk the CSR rows (all relations) of the window's centres; every remote, non-train-pos neighbour goes into an
//             open-addressing hash table (atomicCAS, linear probing); the first insert of an id counts it for its owner
//   assign  : every occupied table slot gets a halo slot inside its owner's range and the id goes to the request list.  The
//             j-th other rank's range is [j * pitch, (j + 1) * pitch) (packed ranges if pitch == 0); unused request slots
//             hold -1.  Fixed ranges = all-to-alls with split sizes known in advance: no count has to reach the host before
//             they are issued.  The order inside an owner is whatever the atomics give - it decides only WHERE a fetched
//             row sits, never a sum's order
//   serve   : the owner's side - feature rows of the ids a peer asked for (-1 = unused slot: skipped)
//   lookup  : per step - the list's entries -> rows of the table: owned id -> id - lo; train-pos id -> n_local + its
//             position (binary search in the sorted ids); remote id -> halo_base + its slot (found in the hash table)
// As a rank holds the feature row of every node its window can touch, it also scores them itself (pcg_step_front_a with
// row_ids): a step has no score exchange, its only collective is the gradient all-reduce.
// Work and memory are proportional to the window's CSR rows / the list's entries - nothing is sized by, or walks, the
// node-id space.  The reference has no distributed code; see SURVEY.md section 8(e).
#include "choose.h"
#include "halo_map.h"

namespace pcg {

constexpr int HALO_MAX_WORLD = 64;
__device__ __forceinline__ int owner_of(const int32_t *s_bounds, int world, int32_t id) {
    int lo = 0, hi = world;          // largest r with bounds[r] <= id
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_bounds[mid] <= id) lo = mid;
        else hi = mid;
    }
    return lo;
}

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
    int32_t pitch;                   // 0: packed layout; > 0: the j-th OTHER rank's slots are [j * pitch, (j + 1) * pitch)
    int32_t self;                    // this rank (owns nothing remote: no range of its own)
    uint32_t *overflow;              // device word: OR-ed with 1 (table full) / 2 (more unique ids than halo_cap / pitch); sticky
    uint32_t *stats;                 // [2] running maxima: unique remote ids of a step, of one owner in a step
    // collect (window mode)
    const int32_t *centres;          // [n_centres] local row numbers (duplicates allowed)
    int32_t n_centres, n_rel;
    const int64_t *indptr[PCG_MAX_REL];
    const int32_t *indices[PCG_MAX_REL];
};

// insert a remote id (first insert counts it for its owner); false: the table is full
__device__ __forceinline__ bool halo_insert(const HaloArgs &a, const int32_t *s_bounds, int32_t id) {
    uint32_t h = halo_hash((uint32_t)id) & a.mask;
    for (uint32_t probe = 0; probe <= a.mask && probe < HALO_MAX_PROBE; ++probe) {
        const uint32_t seen = __hip_atomic_load(&a.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == (uint32_t)id) return true;
        if (seen == HALO_EMPTY) {
            const uint32_t old = atomicCAS(&a.keys[h], HALO_EMPTY, (uint32_t)id);
            if (old == HALO_EMPTY) {                         // first insert of this id
                atomicAdd(&a.owner_count[owner_of(s_bounds, a.world, id)], 1u);
                return true;
            }
            if (old == (uint32_t)id) return true;
        }
        h = (h + 1) & a.mask;
    }
    return false;
}

// is the id in the table already?  (while inserts are going on a "no" may be stale - the caller then takes the insert path,
// which copes with duplicates)
__device__ __forceinline__ bool halo_has(const HaloArgs &a, uint32_t id) {
    uint32_t h = halo_hash(id) & a.mask;
    for (uint32_t probe = 0; probe <= a.mask && probe < HALO_MAX_PROBE; ++probe) {
        const uint32_t key = __hip_atomic_load(&a.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (key == id) return true;
        if (key == HALO_EMPTY) return false;
        h = (h + 1) & a.mask;
    }
    return false;
}

// slot of a remote id, or HALO_EMPTY
__device__ __forceinline__ uint32_t halo_find(const HaloArgs &a, uint32_t id) {
    uint32_t h = halo_hash(id) & a.mask;
    for (uint32_t probe = 0; probe <= a.mask && probe < HALO_MAX_PROBE; ++probe) {
        const uint32_t key = a.keys[h];
        if (key == id) return a.vals[h];
        if (key == HALO_EMPTY) break;
        h = (h + 1) & a.mask;
    }
    return HALO_EMPTY;
}

// empty table, zero counts, request list all -1 (what classify / assign expect to find)
__global__ void __launch_bounds__(256) halo_reset_kernel(const HaloArgs a, uint32_t n_uniq) {
    const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (uint32_t h = t; h <= a.mask; h += stride) a.keys[h] = HALO_EMPTY;
    for (uint32_t i = t; i < n_uniq; i += stride) a.uniq[i] = -1;
    if (t < (uint32_t)HALO_MAX_WORLD) {
        a.owner_count[t] = 0u;
        a.owner_fill[t] = 0u;
    }
}

__global__ void __launch_bounds__(256) halo_assign_kernel(const HaloArgs a) {
    __shared__ int32_t s_bounds[HALO_MAX_WORLD + 1];
    __shared__ uint32_t s_off[HALO_MAX_WORLD + 1];
    for (int i = threadIdx.x; i <= a.world; i += blockDim.x) s_bounds[i] = a.bounds[i];
    if (threadIdx.x == 0) {
        uint32_t run = 0, most = 0;
        bool over = false;
        for (int r = 0; r < a.world; ++r) {
            const uint32_t c = a.owner_count[r];
            s_off[r] = a.pitch > 0 ? (uint32_t)(r - (r > a.self ? 1 : 0)) * (uint32_t)a.pitch : run;
            run += c;
            most = c > most ? c : most;
            over |= a.pitch > 0 && c > (uint32_t)a.pitch;
        }
        s_off[a.world] = run;
        over |= a.pitch == 0 && run > (uint32_t)a.halo_cap;
        if (blockIdx.x == 0) {
            if (over) atomicOr(a.overflow, 2u);
            if (a.stats) {
                atomicMax(&a.stats[0], run);
                atomicMax(&a.stats[1], most);
            }
        }
    }
    __syncthreads();
    const bool fits = a.pitch > 0 || s_off[a.world] <= (uint32_t)a.halo_cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h <= a.mask; h += stride) {
        const uint32_t key = a.keys[h];
        if (key == HALO_EMPTY) continue;
        const int o = owner_of(s_bounds, a.world, (int32_t)key);
        const uint32_t nth = atomicAdd(&a.owner_fill[o], 1u);
        // (a device-side counter never indexes unchecked: an id beyond its owner's range / the table gets no slot - a hole)
        const bool ok = fits && (a.pitch > 0 ? nth < (uint32_t)a.pitch : s_off[o] + nth < (uint32_t)a.halo_cap);
        const uint32_t slot = ok ? s_off[o] + nth : HALO_EMPTY;
        a.vals[h] = slot;
        if (ok) a.uniq[slot] = (int32_t)key;
    }
}

// one wave per (centre, relation) CSR row, grid-stride; every remote, non-train-pos neighbour goes into the table
__global__ void __launch_bounds__(256) halo_collect_kernel(const HaloArgs a) {
    __shared__ int32_t s_bounds[HALO_MAX_WORLD + 1];
    for (int i = threadIdx.x; i <= a.world; i += blockDim.x) s_bounds[i] = a.bounds[i];
    __syncthreads();
    const int lane = lane_id();
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t rows = (int64_t)a.n_centres * a.n_rel;
    bool full = false;
    for (int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); row < rows; row += nwaves) {
        // the table has been reported full (by this wave or any other): the window is lost anyway, stop walking
        if (full || (__hip_atomic_load(a.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u)) break;
        const int r = (int)(row / a.n_centres);
        const int32_t c = a.centres[row - (int64_t)r * a.n_centres];
        if (c < 0 || c >= a.n_local) continue;               // (not a row this rank owns: nothing to walk)
        const int64_t beg = a.indptr[r][c], end = a.indptr[r][c + 1];
        const int32_t *__restrict__ nbr = a.indices[r];
        // four loads of 64 ids in flight (unconditional: index clamped, the extra lanes masked out afterwards)
        constexpr int CU = 4;
        for (int64_t i0 = beg; i0 < end; i0 += CU * PCG_WAVE) {
            int32_t idv[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int64_t i = i0 + u * PCG_WAVE + lane;
                idv[u] = nbr[i < end ? i : end - 1] | (i < end ? 0 : (int32_t)0x80000000);
            }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int32_t id = idv[u];
                if (id < 0 || (id >= a.lo && id < a.hi)) continue;
                // most remote neighbours have been seen before: one or two probes settle them; only a new id pays for the
                // binary search among the train positives (which are never inserted)
                if (halo_has(a, (uint32_t)id)) continue;
                if (pos_find(a.pos_ids, a.n_pos, id) >= 0) continue;
                full |= !halo_insert(a, s_bounds, id);
            }
            full = __any(full);                              // (wave-uniform from here on)
            if (full) {
                if (lane == 0) atomicOr(a.overflow, 1u);
                break;
            }
        }
    }
}

// per step: list entries (global ids) -> rows of the extended table; a remote id the window did not collect (or
// that got no slot) becomes a hole and raises overflow bit 4
__global__ void __launch_bounds__(256) halo_lookup_kernel(const HaloArgs a) {
    const int lane = lane_id();
    const uint32_t nwaves = gridDim.x * (blockDim.x >> 6);
    const uint32_t total = *a.n_chunks < a.chunk_cap ? *a.n_chunks : a.chunk_cap;
    bool miss = false;
    // a chunk has at most 128 entries: both halves are loaded at once (unconditionally: index clamped, the lanes beyond the
    // chunk marked as holes), and the next chunk's descriptor is requested before this chunk is worked on
    uint32_t ch = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int4 desc = a.chunk_desc[ch < total ? ch : 0u];
    for (; ch < total; ch += nwaves) {
        const int4 next = a.chunk_desc[ch + nwaves < total ? ch + nwaves : ch];
        const int n = desc.z;
        int32_t *base = a.list + desc.y;
        int32_t idv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = lane + u * PCG_WAVE;
            idv[u] = base[i < n ? i : (n > 0 ? n - 1 : 0)] | (i < n ? 0 : (int32_t)0x80000000);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = lane + u * PCG_WAVE;
            int32_t *e = base + i;
            const int32_t id = idv[u];
            if (id < 0) continue;
            int32_t out;
            if (id >= a.lo && id < a.hi) {
                out = id - a.lo;
            } else {
                // the table first (one or two probes for a fetched row, and for a train positive - never in the table - a
                // probe to the first empty slot), the binary search among the train positives only after a miss
                const uint32_t slot = halo_find(a, (uint32_t)id);
                if (slot < (uint32_t)a.halo_cap) {
                    out = a.halo_base + (int32_t)slot;
                } else {
                    const int at = pos_find(a.pos_ids, a.n_pos, id);
                    out = at >= 0 ? a.n_local + a.pos_idx[at] : -1;
                    miss |= out < 0;
                }
            }
            *e = out;
        }
        desc = next;
    }
    if (miss) atomicOr(a.overflow, 4u);
}

// rows of the ids a peer asked for: out[i] = X[req[i] - lo] (req[i] < 0 or not owned: row i is left alone)
__global__ void __launch_bounds__(256) halo_serve_kernel(const float *__restrict__ X, int stride, const int32_t *__restrict__ req,
                                                         int n_req, int lo, int n_local, float *__restrict__ out, int out_stride) {
    const int lane = lane_id();
    const int lpr = lanes_per_row(stride);
    const int rpw = PCG_WAVE / lpr;
    const int slot = lane / lpr, sub = lane % lpr;
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int base = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * rpw; base < n_req; base += nwaves * rpw) {
        const int i = base + slot;
        const int32_t id = i < n_req ? req[i] : -1;
        const int32_t row = id - lo;
        if (id < 0 || row < 0 || row >= n_local) continue;
        const float *src = X + (size_t)row * stride;
        float *o = out + (size_t)i * out_stride;
        for (int ch = sub; ch < (stride >> 2); ch += lpr)
            *reinterpret_cast<float4 *>(o + 4 * ch) = *reinterpret_cast<const float4 *>(src + 4 * ch);
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

/* Window mode (see the header of this file).  pcg_halo_collect = reset + collect + assign for the CSR rows of `centres` (local
 * row numbers, [n_centres], duplicates allowed); table / counts / uniq / halo_cap / owner_pitch as pcg_halo_classify.
 * pcg_halo_lookup re-indexes a step's list (global ids) into the extended table with that table; a remote id the window did
 * not collect becomes a hole and sets overflow bit 4 in counts[128]. */
int pcg_halo_collect(const pcg_graph_desc *g, const int32_t *centres, int32_t n_centres, int32_t lo, int32_t hi, int32_t n_local,
                     const int32_t *pos_ids, int32_t n_pos, const int32_t *bounds, int32_t world, uint32_t *table,
                     int64_t table_slots, uint32_t *counts, int32_t *uniq, int32_t halo_cap, int32_t halo_base,
                     int32_t owner_pitch, int32_t self_rank, void *stream) {
    if (!g || !centres || n_centres < 0 || !bounds || world < 1 || world > pcg::HALO_MAX_WORLD || !table || !counts || !uniq ||
        halo_cap < 0 || lo > hi || n_pos < 0 || (n_pos > 0 && !pos_ids) || g->n_rel < 1 || g->n_rel > PCG_MAX_REL)
        return PCG_E_ARG;
    if (table_slots < 1024 || (table_slots & (table_slots - 1)) != 0 || table_slots > (1ll << 31)) return PCG_E_ARG;
    if (self_rank < 0 || self_rank >= world || owner_pitch < 0) return PCG_E_ARG;
    if (owner_pitch > 0 && (int64_t)owner_pitch * (world > 1 ? world - 1 : 1) != (int64_t)halo_cap) return PCG_E_ARG;
    pcg::HaloArgs a = {};
    a.self = self_rank;
    a.lo = lo;
    a.hi = hi;
    a.n_local = n_local;
    a.pos_ids = pos_ids;
    a.n_pos = n_pos;
    a.bounds = bounds;
    a.world = world;
    a.keys = table;
    a.vals = table + table_slots;
    a.mask = (uint32_t)(table_slots - 1);
    a.owner_count = counts;
    a.owner_fill = counts + pcg::HALO_MAX_WORLD;
    a.overflow = counts + 2 * pcg::HALO_MAX_WORLD;
    a.stats = counts + 2 * pcg::HALO_MAX_WORLD + 1;
    a.uniq = uniq;
    a.halo_cap = halo_cap;
    a.halo_base = halo_base;
    a.pitch = owner_pitch;
    a.centres = centres;
    a.n_centres = n_centres;
    a.n_rel = g->n_rel;
    for (int r = 0; r < g->n_rel; ++r) {
        if (!g->indptr[r] || !g->indices[r]) return PCG_E_ARG;
        a.indptr[r] = g->indptr[r];
        a.indices[r] = g->indices[r];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t work = table_slots > halo_cap ? table_slots : halo_cap;
    int rb = (int)((work + 255) / 256);
    if (rb > 1024) rb = 1024;
    hipLaunchKernelGGL(pcg::halo_reset_kernel, dim3(rb), dim3(256), 0, st, a, (uint32_t)halo_cap);
    PCG_LAUNCH_CHECK();
    if (n_centres > 0) {
        int64_t cb = ((int64_t)n_centres * g->n_rel + 3) / 4;
        if (cb > 4096) cb = 4096;
        hipLaunchKernelGGL(pcg::halo_collect_kernel, dim3((int)cb), dim3(256), 0, st, a);
        PCG_LAUNCH_CHECK();
    }
    int blocks = (int)((table_slots + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcg::halo_assign_kernel, dim3(blocks), dim3(256), 0, st, a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

int pcg_halo_lookup(const pcg_graph_desc *g, int32_t B, void *workspace, const void *plan, int64_t list_capacity, int32_t lo, int32_t hi,
                    int32_t n_local, const int32_t *pos_ids, const int32_t *pos_idx, int32_t n_pos, uint32_t *table,
                    int64_t table_slots, uint32_t *counts, int32_t halo_cap, int32_t halo_base, void *stream) {
    if (!g || !workspace || !table || !counts || B < 1 || list_capacity < 1 || table_slots < 1024 ||
        (table_slots & (table_slots - 1)) != 0 || lo > hi || n_pos < 0 || (n_pos > 0 && (!pos_ids || !pos_idx)))
        return PCG_E_ARG;
    pcg::Workspace w;
    pcg::carve1(g, B, list_capacity, static_cast<unsigned char *>(workspace), &w, static_cast<unsigned char *>(const_cast<void *>(plan)));
    pcg::HaloArgs a = {};
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
    a.keys = table;
    a.vals = table + table_slots;
    a.mask = (uint32_t)(table_slots - 1);
    a.overflow = counts + 2 * pcg::HALO_MAX_WORLD;
    a.halo_cap = halo_cap;
    a.halo_base = halo_base;
    hipLaunchKernelGGL(pcg::halo_lookup_kernel, dim3(1024), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

/* The owner's side of the exchange: out[i, :] = X[req[i] - lo, :] (whole padded rows) for the ids peers asked for;
 * req[i] < 0 (an unused slot of a fixed-pitch request list) or an id this rank does not own leaves row i untouched. */
int pcg_halo_serve(const pcg_graph_desc *g, const int32_t *req, int32_t n_req, int32_t lo, int32_t n_local, float *out,
                   int32_t out_stride, void *stream) {
    if (!g || !g->X || !req || !out || n_req < 0 || n_local < 0 || g->feat_stride % 4 != 0 || out_stride < g->feat_stride ||
        out_stride % 4 != 0)
        return PCG_E_ARG;
    if (n_req == 0) return PCG_OK;
    const int rpw = PCG_WAVE / pcg::lanes_per_row(g->feat_stride);
    int blocks = (n_req + 4 * rpw - 1) / (4 * rpw);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pcg::halo_serve_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g->X, g->feat_stride,
                       req, n_req, lo, n_local, out, out_stride);
    PCG_LAUNCH_CHECK();
    return PCG_OK;
}

}  // extern "C"
