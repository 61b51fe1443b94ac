// rt_sort_group.hpp -- the radix sort's scan, folded into the histogram kernels when a pass has few tiles.
//
// With at most kSortGroupedMaxTiles tiles (one histogram workgroup per tile) the per-pass scan launch
// (sort_scan_kernel: offs[d][t] = sum of hist[d][t' < t], totals[d]) is replaced by a two-level form that needs no
// launch of its own:
//   * tiles are grouped by kSortGroup = 16.  A histogram workgroup publishes its column hist[d][tile] with
//     write-through stores, drains them and takes a ticket on its group; the LAST ticket of a group (nobody waits: the
//     hand-off of lbvh_upper_kernel) acquires and scans the group's 16 columns: offs[d][t] = prefix INSIDE the group,
//     group_total[d][g] = the group's sum -- 64 bytes read and written per digit, all groups in parallel;
//   * the down-sweep workgroup of tile t (group g) reads, per digit it owns, group_total[d][0 .. groups) (<= 32 words) and
//     offs[d][t]: digit total = sum over the groups, position of the tile inside the digit's run = sum of the groups
//     before g + offs[d][t]; the digit bases are the block scan it already did.
// The tickets (one word per group) must be zero when a sort starts (the build's first kernel / sah_init_kernel / a
// memset in the stand-alone entry points); the last arriver resets its word, so the passes of one sort reuse them.
#pragma once
#include "rt_device.hpp"
#include "rt_launch.hpp"

namespace rt {

// (store / load of a table word other workgroups of the same launch read: write-through at agent scope, see
//  lbvh_levels.hip store_sc1)
__device__ __forceinline__ void sort_store_sc1(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by ALL threads of a histogram workgroup after it has stored its column with sort_store_sc1.  h_flag: one LDS word.
template <uint32_t BITS, uint32_t NT>
__device__ __forceinline__ void sort_group_tail(const uint32_t* __restrict__ hist, uint32_t* __restrict__ offs,
                                                uint32_t* __restrict__ group_total, uint32_t* __restrict__ arrive,
                                                uint32_t tile, uint32_t num_tiles, uint32_t stride, uint32_t* h_flag)
{
    constexpr uint32_t RADIX = 1u << BITS;
    const uint32_t g = tile / kSortGroup;
    const uint32_t first = g * kSortGroup;
    const uint32_t members = min(kSortGroup, num_tiles - first);
    // every storing wave drains its stores, the workgroup meets, one lane takes the ticket; the last ticket acquires
    // (invalidates this CU's L1 and the stale lines of its L2) before the barrier lets the other waves read
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ticket = __hip_atomic_fetch_add(arrive + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = ticket == members - 1;
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            sort_store_sc1(arrive + g, 0u);    // everybody of the group has arrived: ready for the next pass
        }
        *h_flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*h_flag == 0) return;
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT) {
        const uint4* hrow = reinterpret_cast<const uint4*>(hist + (size_t)d * stride + first);
        uint4* orow = reinterpret_cast<uint4*>(offs + (size_t)d * stride + first);
        uint32_t running = 0;
#pragma unroll
        for (uint32_t q = 0; q < kSortGroup / 4; q++) {
            if (q * 4 < members) {        // (table rows are padded to a multiple of 4 tiles: a started quad is readable)
                uint4 v = hrow[q];
                const uint32_t left = members - q * 4;
                if (left < 4) { v.w = 0u; if (left < 3) v.z = 0u; if (left < 2) v.y = 0u; }
                orow[q] = make_uint4(running, running + v.x, running + v.x + v.y, running + v.x + v.y + v.z);
                running += v.x + v.y + v.z + v.w;
            }
        }
        group_total[(size_t)d * kSortMaxGroups + g] = running;
    }
}

// the down-sweep's side: digit total and this tile's position inside the digit's run
__device__ __forceinline__ void sort_group_lookup(const uint32_t* __restrict__ offs, const uint32_t* __restrict__ group_total,
                                                  uint32_t d, uint32_t tile, uint32_t num_tiles, uint32_t stride,
                                                  uint32_t& total, uint32_t& before)
{
    const uint32_t g = tile / kSortGroup, ng = (num_tiles + kSortGroup - 1) / kSortGroup;
    const uint4* row = reinterpret_cast<const uint4*>(group_total + (size_t)d * kSortMaxGroups);
    uint32_t tot = 0, pre = 0;
#pragma unroll
    for (uint32_t q = 0; q < kSortMaxGroups / 4; q++) {
        if (q * 4 < ng) {
            const uint4 v = row[q];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t gi = q * 4 + k;
                tot += gi < ng ? w[k] : 0u;
                pre += gi < g ? w[k] : 0u;
            }
        }
    }
    total = tot;
    before = pre + offs[(size_t)d * stride + tile];
}

}  // namespace rt
