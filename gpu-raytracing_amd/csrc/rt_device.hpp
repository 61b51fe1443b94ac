// rt_device.hpp -- shared device helpers and launch geometry for the gfx950 kernels.
// Written for wave64 / CDNA4 only (no CUDA / multi-backend paths).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_abi.h"

static_assert(sizeof(rt_triangle) == 36, "Triangle layout (Common.cuh:199)");
static_assert(sizeof(rt_node) == 32, "Node layout (Common.cuh:152)");
static_assert(sizeof(rt_triangle_pair) == 64, "TrianglePair layout (Common.cuh:161)");
static_assert(sizeof(rt_camera) == 64, "Camera layout (Common.cuh:44)");
static_assert(sizeof(rt_attributes) == 72, "Attributes layout (Common.cuh:55)");
static_assert(sizeof(rt_material) == 52, "Material POD mirror");
static_assert(sizeof(rt_texture) == 216, "Texture POD mirror");

namespace rt {

constexpr uint32_t kIndexMask = 0x1FFFFFFFu;  // 29-bit child / parent field (Common.cuh:152-159)

// ---- DeviceUtils.cuh:3-13: monotone float <-> int so integer min/max == float min/max
// (i >= 0) ? i : i ^ 0x7FFFFFFF, written without the select: an arithmetic shift + one v_bitop3 instead of xor + compare +
// select -- these conversions are a visible share of the instructions of the SAH kernels and of the scene-box kernel
__device__ __forceinline__ int float_to_ordered_int(float f)
{
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7FFFFFFF);
}
__device__ __forceinline__ float ordered_int_to_float(int i)
{
    return __int_as_float(i ^ ((i >> 31) & 0x7FFFFFFF));
}

// ---- wave64 helpers
__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-wide reductions: four DPP steps reduce each row of 16 lanes (quad swaps, half-row mirror, row mirror: VALU
// only, no LDS-crossbar permutes), v_readlane picks one lane per row and the scalar unit combines the four rows.
// The result is wave-uniform.  All 64 lanes must be active at the call.
template <int CTRL>
__device__ __forceinline__ int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, dpp_mov<0xB1>(v));    // quad_perm [1,0,3,2]
    v = min(v, dpp_mov<0x4E>(v));    // quad_perm [2,3,0,1]
    v = min(v, dpp_mov<0x141>(v));   // row_half_mirror
    v = min(v, dpp_mov<0x140>(v));   // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, dpp_mov<0xB1>(v));
    v = max(v, dpp_mov<0x4E>(v));
    v = max(v, dpp_mov<0x141>(v));
    v = max(v, dpp_mov<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
    int s = (int)v;
    s += dpp_mov<0xB1>(s);
    s += dpp_mov<0x4E>(s);
    s += dpp_mov<0x141>(s);
    s += dpp_mov<0x140>(s);
    return (uint32_t)(__builtin_amdgcn_readlane(s, 0) + __builtin_amdgcn_readlane(s, 16) +
                      __builtin_amdgcn_readlane(s, 32) + __builtin_amdgcn_readlane(s, 48));
}
// inclusive scan across the 64 lanes of a wave with DPP row shifts / row broadcasts: VALU only (a __shfl_up ladder goes
// through the LDS crossbar: six dependent ds_bpermute round trips).
// CONTRACT: all 64 lanes must be active at the call (full EXEC: call it from wave-uniform control flow only) -- a DPP
// source lane that is switched off contributes 0 / stale data, silently.  The __shfl_up ladder it replaced tolerated
// partial waves; this does not.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int /*lane*/ = 0)
{
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);   // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);   // row_shr:8  -> inclusive inside each row of 16
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1 and 3
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2 and 3
    return (uint32_t)x;
}

// Workgroups are dealt to the 8 XCDs round-robin (b -> XCD b % 8; observed, speed only -- MI355X_MICROARCH.md, dispatch):
// map workgroup b of nb to a work item such that XCD x gets the contiguous range [x * nb/8 ..).  A bijection on [0, nb).
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t nb)
{
    const uint32_t per = nb >> 3, rem = nb & 7u, xcd = b & 7u, loc = b >> 3;
    return xcd * per + min(xcd, rem) + loc;
}

// lanes (among `valid` ones) whose BITS-bit digit equals mine (the radix sort's ballot ranking)
template <uint32_t BITS>
__device__ __forceinline__ uint64_t match_digit(uint32_t d, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (uint32_t b = 0; b < BITS; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

// exclusive scan of one value per thread over a block of NT threads (NT multiple of 64, <= 1024).
// `ws` is NT/64 + 1 words of LDS.  Returns the exclusive prefix; *total gets the block sum.
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* ws, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) ws[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < NW ? ws[lane] : 0u;
        uint32_t wi = wave_incl_scan_u32(w, lane);
        if (lane < NW) ws[lane] = wi - w;
        if (lane == NW - 1) ws[NW] = wi;
    }
    __syncthreads();
    uint32_t r = ws[wave] + incl - v;
    *total = ws[NW];
    __syncthreads();
    return r;
}

// workgroup barrier that orders LDS traffic only: outstanding global loads (the tile's keys, issued first) stay in flight
// across it.  __syncthreads() would drain them (it is a fence for global memory too).
// CONTRACT: it orders LDS ONLY.  A value one wave STORES TO GLOBAL MEMORY before this barrier is not guaranteed visible to a
// load of another wave after it (no vmcnt wait, no cache action): hand data across waves through LDS, or use
// __syncthreads() / an explicit s_waitcnt vmcnt(0) for a global-memory hand-off.  Same for block_excl_scan_lds below.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// block_excl_scan_u32 with LDS-only barriers
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_lds(uint32_t v, uint32_t* ws, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    const uint32_t incl = wave_incl_scan_u32(v);
    if (lane == 63) ws[wave] = incl;
    lds_barrier();
    if (wave == 0) {
        const uint32_t w = lane < NW ? ws[lane] : 0u;
        const uint32_t wi = wave_incl_scan_u32(w);
        if (lane < NW) ws[lane] = wi - w;
        if (lane == NW - 1) ws[NW] = wi;
    }
    lds_barrier();
    const uint32_t r = ws[wave] + incl - v;
    *total = ws[NW];
    lds_barrier();
    return r;
}

}  // namespace rt
