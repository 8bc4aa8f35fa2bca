// Node id -> row of a partitioned rank's extended feature table [ owned rows | train-pos rows | halo ] (halo.hip builds the
// hash table per window of steps; pc-gnn_amd/dist.py).  Shared by the stand-alone look-up launch (pcg_halo_lookup) and the
// gather that translates the selection list's ids as it reads them (pcg_gather_lists_dist).
#pragma once
#include "common.h"

namespace pcg {

constexpr uint32_t HALO_EMPTY = 0xFFFFFFFFu;
// Longest probe sequence any table operation walks.  The table has at least two slots per halo row, so a run of 128 occupied
// slots means it is over-full (a capacity error): the insert then reports "full" (overflow bit 1) instead of walking the whole
// table for every later neighbour - O(ids x slots) probes would look like a hang.  An id that WAS inserted sits within this many
// probes of its hash slot (no deletions), so look-ups bounded the same way find everything that is there.
constexpr uint32_t HALO_MAX_PROBE = 128;

__device__ __forceinline__ uint32_t halo_hash(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}

struct HaloMap {
    const uint32_t *keys, *vals;     // hash table, `mask` + 1 slots (a power of two); keys == null: no translation
    uint32_t mask;
    int32_t lo, hi, n_local;         // this rank owns ids [lo, hi): row = id - lo
    const int32_t *pos_ids;          // [n_pos] train-pos ids ascending
    const int32_t *pos_idx;          // [n_pos] their row in the replicated train-pos block (at n_local)
    int32_t n_pos;
    int32_t halo_cap, halo_base;
    uint32_t *overflow;              // device word: OR-ed with 4 when an id is in none of the three
};

// first train-pos id >= id (binary search in the ascending ids); returns its position or -1
__device__ __forceinline__ int pos_find(const int32_t *__restrict__ pos_ids, int n_pos, int32_t id) {
    int plo = 0, phi = n_pos;
    while (plo < phi) {
        const int mid = (plo + phi) >> 1;
        if (pos_ids[mid] < id) plo = mid + 1;
        else phi = mid;
    }
    return (plo < n_pos && pos_ids[plo] == id) ? plo : -1;
}

// slot of a remote id, or HALO_EMPTY
__device__ __forceinline__ uint32_t halo_map_find(const HaloMap &a, uint32_t id) {
    uint32_t h = halo_hash(id) & a.mask;
    for (uint32_t probe = 0; probe <= a.mask && probe < HALO_MAX_PROBE; ++probe) {
        const uint32_t key = a.keys[h];
        if (key == id) return a.vals[h];
        if (key == HALO_EMPTY) break;
        h = (h + 1) & a.mask;
    }
    return HALO_EMPTY;
}

// a list entry (global id; negative = hole) -> table row (negative = hole); miss: the id is nowhere in the table
__device__ __forceinline__ int32_t halo_translate(const HaloMap &a, int32_t id, bool &miss) {
    if (id < 0) return id;
    if (id >= a.lo && id < a.hi) return id - a.lo;
    // the table first (one or two probes for a fetched row, and for a train positive - never in the table - a probe to the
    // first empty slot), the binary search among the train positives only after a miss
    const uint32_t slot = halo_map_find(a, (uint32_t)id);
    if (slot < (uint32_t)a.halo_cap) return a.halo_base + (int32_t)slot;
    const int at = pos_find(a.pos_ids, a.n_pos, id);
    if (at >= 0) return a.n_local + a.pos_idx[at];
    miss = true;
    return (int32_t)0x80000000;
}

}  // namespace pcg
