// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Replaces RadixSort() and its kernels CreateHistogramsLM / PrefixSumExclusive / Distribute
// (RadixSort.cu:47-225).  Same contract: ascending, stable, result in keys/values after ping-ponging
// through the temporaries; 4 passes x 8 bits for arbitrary 32-bit keys (the public entry point), 3 passes
// x 10 bits for the builder's 30-bit Morton keys while the tables stay small.  Different machine mapping:
//   * a workgroup owns a TILE of 4096 consecutive keys (the reference fixes 128 segments and gives each to ONE 32-lane
//     warp that ranks its keys with a 32-step serialised LDS atomic).  Its barriers wait for LDS only
//     (`s_waitcnt lgkmcnt(0); s_barrier`), so the key loads issued first stay in flight behind the prologue.  (Groups of
//     2 - 8 consecutive tiles per workgroup, with the next tile prefetched, and 8192-key tiles were measured in round 3:
//     218 - 238 us against 222 us for one tile per workgroup at 10M keys -- the kernel is bound by instruction issue, not
//     by its tables or its run lengths, so they were not kept);
//   * ranks come from wave64 ballots: BITS ballots give the set of lanes holding the same digit, a
//     popcount below the lane gives the stable rank, and only the group leader touches LDS;
//   * a tile is sorted by digit in LDS and written in position order (64-byte runs per 8-bit digit);
//   * tables live in caller scratch (no malloc/free/memset/sync inside the sort), no global atomics.
// Per pass: upsweep (tile digit histogram, 4 B/key read) -> scan (one wave or workgroup per digit)
// -> downsweep (8 B/key read, 8 B/key written).  Algorithmic traffic 20 B/key/pass = 80 B/key.
#include <type_traits>

#include "rt_device.hpp"
#include "rt_launch.hpp"

#ifdef RT_SORT_TUNING
#include <cstdlib>
#endif

namespace rt {

// lanes (among the valid ones) whose BITS-bit digit equals mine, as two 32-bit halves.  Per bit: sign-extend the bit
// (v_bfe_i32), one compare (the ballot), and per half one xnor + one and.
template <uint32_t BITS>
__device__ __forceinline__ void match_digit_halves(uint32_t d, bool valid, uint32_t& mlo, uint32_t& mhi)
{
    const uint64_t vb = __ballot(valid);
    mlo = (uint32_t)vb;
    mhi = (uint32_t)(vb >> 32);
#pragma unroll
    for (uint32_t b = 0; b < BITS; b++) {
        const uint32_t sgn = (uint32_t)__builtin_amdgcn_sbfe((int)d, b, 1u);   // all ones iff bit b of d is set
        const uint64_t bal = __ballot(sgn != 0u);
        mlo &= ~((uint32_t)bal ^ sgn);
        mhi &= ~((uint32_t)(bal >> 32) ^ sgn);
    }
}

// Digit counts of one key per lane into the LDS histogram h.  Spread digits: one LDS atomic per key.  Clustered digits
// (sorted, flat or constant input -- the lanes would queue on one LDS word): group the lanes with ballots and let each
// group's leader add its size.  The wave chooses by looking at how many lanes share the first lane's digit.
__device__ __forceinline__ bool digits_clustered(uint32_t d, bool valid)
{
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
    return __popcll(__ballot(valid && d == d0)) >= 8;
}
template <uint32_t BITS>
__device__ __forceinline__ void hist_add(uint32_t* h, uint32_t d, bool valid, bool clustered)
{
    if (clustered) {
        uint32_t mlo, mhi;
        match_digit_halves<BITS>(d, valid, mlo, mhi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if (valid && below == 0u) atomicAdd(&h[d], (uint32_t)(__popc(mlo) + __popc(mhi)));
    } else if (valid) {
        atomicAdd(&h[d], 1u);
    }
}

// ---- upsweep: hist[d][tile] = number of keys of the tile whose digit is d.  Table rows are `stride` words long (the
// number of tiles rounded up to a multiple of 4, so that a row of four tiles is one aligned 16-byte access).
// TQ tiles per workgroup: 1 while there are few tiles (a pass is then one workgroup's latency chain), 4 when there are
// many -- a workgroup then publishes 16 bytes per digit instead of one scattered dword per digit and tile (the table is
// written transposed: 625 k scattered dwords per pass at 10M keys made this kernel and the scan a third of the sort).
template <uint32_t BITS, uint32_t NT, uint32_t TQ>
__global__ __launch_bounds__(NT) void sort_upsweep_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                           uint32_t shift, uint32_t num_tiles, uint32_t stride,
                                                           uint32_t* __restrict__ hist, const uint32_t* n_dev, int vec_ok)
{
    constexpr uint32_t RADIX = 1u << BITS;
    constexpr uint32_t ITEMS = kSortTile / NT;
    static_assert(TQ == 1 || TQ == 4, "one dword or one uint4 per digit");
    if (n_dev) n = *n_dev;   // device-side count (--pairs): tiles past it see no valid key and publish zeros
    __shared__ uint32_t h[TQ][RADIX];   // (digit-minor: neighbouring digits on neighbouring banks)
    for (uint32_t d = threadIdx.x; d < RADIX * TQ; d += NT) (&h[0][0])[d] = 0;
    __syncthreads();
    const uint32_t tile0 = blockIdx.x * TQ;
    if (vec_ok && (tile0 + TQ) * kSortTile <= n) {
        // full tiles: 16-byte loads, 4 consecutive keys per lane (the order inside a tile is irrelevant to a count), all
        // of the workgroup's loads in flight before the first count
        uint4 q[TQ][ITEMS / 4];
#pragma unroll
        for (uint32_t t = 0; t < TQ; t++)
#pragma unroll
            for (uint32_t i = 0; i < ITEMS / 4; i++)
                q[t][i] = reinterpret_cast<const uint4*>(keys + (size_t)(tile0 + t) * kSortTile)[i * NT + threadIdx.x];
#pragma unroll
        for (uint32_t t = 0; t < TQ; t++)
#pragma unroll
            for (uint32_t i = 0; i < ITEMS / 4; i++) {
                // (one look per four keys: a lane's four keys are neighbours in the input, clustered or spread together)
                const uint32_t dx = (q[t][i].x >> shift) & (RADIX - 1);
                const bool cl = digits_clustered(dx, true);
                hist_add<BITS>(h[t], dx, true, cl);
                hist_add<BITS>(h[t], (q[t][i].y >> shift) & (RADIX - 1), true, cl);
                hist_add<BITS>(h[t], (q[t][i].z >> shift) & (RADIX - 1), true, cl);
                hist_add<BITS>(h[t], (q[t][i].w >> shift) & (RADIX - 1), true, cl);
            }
    } else {
        for (uint32_t t = 0; t < TQ; t++) {
            const uint32_t base = (tile0 + t) * kSortTile;
            uint32_t k[ITEMS];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = base + i * NT + threadIdx.x;
                k[i] = idx < n ? keys[idx] : 0u;
            }
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t d = (k[i] >> shift) & (RADIX - 1);
                const bool valid = base + i * NT + threadIdx.x < n;
                hist_add<BITS>(h[t], d, valid, digits_clustered(d, valid));
            }
        }
    }
    __syncthreads();
    // no global atomics anywhere in the sort.  (tile0 + TQ <= stride: the padding columns of the last group get zeros)
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT) {
        if (TQ == 4) *reinterpret_cast<uint4*>(hist + (size_t)d * stride + tile0) = make_uint4(h[0][d], h[1 % TQ][d], h[2 % TQ][d], h[3 % TQ][d]);
        else hist[(size_t)d * stride + tile0] = h[0][d];
    }
}

// Scan: offs[d][t] = sum(hist[d][0..t)) (position inside the digit's output run), totals[d] = sum over all tiles.  A lane
// takes four consecutive tiles (one 16-byte access).  The digit bases (exclusive scan of totals) are formed by each
// downsweep workgroup in its prologue: one block scan -- cheaper than a launch or RADIX same-address atomics per tile.
// columns [t4 * 4, t4 * 4 + 4) of a table row, with the padding columns (>= num_tiles: never written by the one-tile-per-
// workgroup histogram kernels) read as zero
__device__ __forceinline__ uint4 load4_tiles(const uint4* row, uint32_t t4, uint32_t num_tiles)
{
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (t4 * 4 < num_tiles) {
        v = row[t4];
        const uint32_t left = num_tiles - t4 * 4;
        if (left < 4) { v.w = 0u; if (left < 3) v.z = 0u; if (left < 2) v.y = 0u; }
    }
    return v;
}
__device__ __forceinline__ uint4 excl4(const uint4& v, uint32_t before)
{
    return make_uint4(before, before + v.x, before + v.x + v.y, before + v.x + v.y + v.z);
}
// one WAVE per digit: no barriers (few tiles: 256 per step)
__global__ __launch_bounds__(256) void sort_scan_kernel(const uint32_t* __restrict__ hist, uint32_t num_tiles, uint32_t stride,
                                                        uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t running = 0;
    const uint4* hrow = reinterpret_cast<const uint4*>(hist + (size_t)d * stride);
    uint4* orow = reinterpret_cast<uint4*>(offs + (size_t)d * stride);
    for (uint32_t c = 0; c < stride / 4; c += 64) {
        const uint32_t t = c + lane;
        const uint4 v = load4_tiles(hrow, t, num_tiles);
        const uint32_t s = v.x + v.y + v.z + v.w;
        const uint32_t incl = wave_incl_scan_u32(s);
        if (t < stride / 4) orow[t] = excl4(v, running + incl - s);
        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) totals[d] = running;
}

// one WORKGROUP per digit (block scans of 1024 tiles at a time): for many tiles, where a single wave per digit would run a
// long serial chain
__global__ __launch_bounds__(256) void sort_scan_wide_kernel(const uint32_t* __restrict__ hist, uint32_t num_tiles, uint32_t stride,
                                                             uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t ws[8];
    const uint32_t d = blockIdx.x;
    uint32_t running = 0;
    const uint4* hrow = reinterpret_cast<const uint4*>(hist + (size_t)d * stride);
    uint4* orow = reinterpret_cast<uint4*>(offs + (size_t)d * stride);
    for (uint32_t c = 0; c < stride / 4; c += 256) {
        const uint32_t t = c + threadIdx.x;
        const uint4 v = load4_tiles(hrow, t, num_tiles);
        uint32_t chunk;
        const uint32_t ex = block_excl_scan_u32<256>(v.x + v.y + v.z + v.w, ws, &chunk);
        if (t < stride / 4) orow[t] = excl4(v, running + ex);
        running += chunk;
    }
    if (threadIdx.x == 0) totals[d] = running;
}

// ---- downsweep.  NT threads per 4096-key tile: 512 (8 keys per thread), or 1024 (4 per thread) when a pass is at most
// one workgroup per CU -- one workgroup's dependent chain then IS the kernel's duration, and twice the waves halve its
// ranking rounds.  IDENT: the values of this pass are the identity (the builder's first pass without --pairs:
// GenerateMortonCodes writes values[i] = i, BottomUpBuilder.cu:113), so they are not read -- and the Morton kernel
// does not write them.
//
// The kernel is bound by instruction issue (profiles/r02_build_pmc_10m.txt: 24 M wave-instructions per 10M-key pass at
// ~4.4 cycles each over 1024 SIMDs = the 46 us it took; removing every store changed 6 us, tools/sort_yardstick
// experiments of round 3), so what it does per key is kept short:
//   * keys, values and results go through BUFFER loads / stores: a 32-bit byte offset per access instead of a 64-bit
//     address, immediate offsets for a lane's 8 keys, and the descriptor's range check replaces every `idx < n` branch
//     (a load past the end returns 0, a store past the end is dropped -- which is also the guard that a position formed
//     from the tables in memory can never become a wild store);
//   * a tile that is not the array's last runs without any per-key validity logic (FULL);
//   * the digit match is one v_bfe_i32 + one compare (the ballot) + two v_bitop3 per bit;
//   * the sorted tile is staged as (key, value) pairs: one 64-bit LDS write and one 64-bit LDS read per key.
typedef __amdgpu_buffer_rsrc_t buf_t;
__device__ __forceinline__ buf_t make_buf(const void* p, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <uint32_t BITS, uint32_t NT, bool IDENT, int EXP = 0>   // EXP != 0: timing experiments of the tuning build (wrong output)
__global__ __launch_bounds__(NT, (NT == 512 && BITS == 8) ? 6 : 4) void sort_downsweep_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, uint32_t n,
                                                            uint32_t shift, uint32_t stride,
                                                            const uint32_t* __restrict__ offs,
                                                            const uint32_t* __restrict__ totals,
                                                            const uint32_t* n_dev, uint32_t* exp_table = nullptr)
{
    constexpr uint32_t RADIX = 1u << BITS;
    constexpr uint32_t NW = NT / 64;             // waves
    constexpr uint32_t ITEMS = kSortTile / NT;   // keys per thread
    constexpr uint32_t DPT = (RADIX + NT - 1) / NT;   // digits per thread in the table phase (thread t owns digits t*DPT ...)
    static_assert(kSortTile % NT == 0 && ITEMS % 2 == 0 && (RADIX % NT == 0 || NT % RADIX == 0), "tile and digit split");
    static_assert(NW * RADIX % (NT * 4) == 0, "the counters are cleared 16 bytes per thread at a time");
    const bool owner = threadIdx.x * DPT < RADIX;     // (NT > RADIX: the upper threads own no digit)
    if (n_dev) n = *n_dev;
    // wave_hist[w][d]: first the running count of digit d inside wave w's chunk of the tile, later the position inside
    // the tile (sorted by digit) of wave w's first key with digit d.
    __shared__ __attribute__((aligned(16))) uint32_t wave_hist[NW][RADIX];
    __shared__ uint32_t gadj[RADIX];             // (global position of the tile's first key of digit d) - (its local position)
    __shared__ __attribute__((aligned(16))) uint2 spair[kSortTile];   // the tile sorted by digit: (key, value)
    __shared__ uint32_t ws[NW + 4];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    // XCD-aware tile order: hardware deals workgroup b to XCD b % 8; give XCD x a contiguous run of tiles.  Consecutive
    // tiles write consecutive runs of every digit's output region (64 bytes on average = half a line), so the two halves
    // of a line meet in ONE L2 instead of being written back separately by two XCDs.
    const uint32_t tile = xcd_contiguous(blockIdx.x, gridDim.x);
    const uint32_t tbase = tile * kSortTile;
    const uint32_t first = tbase + wave * (ITEMS * 64) + lane;   // this lane's first key
    const buf_t bk_in = make_buf(keys_in, n * 4u), bv_in = make_buf(IDENT ? keys_in : vals_in, n * 4u);
    const buf_t bk_out = make_buf(keys_out, n * 4u), bv_out = make_buf(vals_out, n * 4u);

    // the keys: issued before anything else, they arrive behind the prologue (whose barriers wait for LDS only).  Wave w
    // owns keys [tbase + w*ITEMS*64, tbase + (w+1)*ITEMS*64) in rounds of 64 consecutive keys, so (wave, round, lane) order
    // IS input order: stability.
    uint32_t k[ITEMS], v[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        k[i] = __builtin_amdgcn_raw_buffer_load_b32(bk_in, (first + i * 64) * 4u, 0, 0);
        if (!IDENT) v[i] = __builtin_amdgcn_raw_buffer_load_b32(bv_in, (first + i * 64) * 4u, 0, 0);
    }
    {
        uint4* z = reinterpret_cast<uint4*>(&wave_hist[0][0]);
#pragma unroll
        for (uint32_t q = 0; q < NW * RADIX / (NT * 4); q++) z[q * NT + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    }
    // digit bases = exclusive scan of the digit totals; + this tile's position inside every digit's run
    uint32_t gbase[DPT];
    {
        uint32_t tot[DPT], toff[DPT], tsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            tot[j] = owner ? totals[threadIdx.x * DPT + j] : 0u;
            toff[j] = owner ? offs[(size_t)(threadIdx.x * DPT + j) * stride + tile] : 0u;
            tsum += tot[j];
        }
        uint32_t dummy;
        uint32_t digit_base = block_excl_scan_lds<NT>(tsum, ws, &dummy);  // ends with a barrier
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) { gbase[j] = digit_base + toff[j]; digit_base += tot[j]; }
    }

    // this wave's counters, as an LDS-address-space pointer (a generic `volatile` pointer turns every access into a flat_load)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    volatile lds_u32* wh = (volatile lds_u32*)(&wave_hist[wave][0]);
    const bool full = tbase + kSortTile <= n;   // wave-uniform: only the array's last tile can be cut
    uint32_t rank2[ITEMS / 2];   // two 16-bit ranks per register (a rank is < ITEMS * 64 <= 512)
    // FULL tile: no per-key validity logic at all
    auto rank_tile = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        // two rounds at a time: their matches are independent, and interleaved they fill the wait states the hardware
        // needs between a compare that writes an SGPR pair (the ballot) and the bitop3 that reads it
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i += 2) {
            bool valid[2];
            uint32_t d[2], mlo[2], mhi[2];
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
                valid[u] = FULL || first + (i + u) * 64 < n;
                mlo[u] = mhi[u] = 0xFFFFFFFFu;
                if (!FULL) {
                    const uint64_t vb = __ballot(valid[u]);
                    mlo[u] = (uint32_t)vb; mhi[u] = (uint32_t)(vb >> 32);
                }
                d[u] = __builtin_amdgcn_ubfe(k[i + u], shift, BITS);
            }
            if (EXP != 3) {
                // lanes whose digit equals mine: per bit the sign-extended bit (v_bfe_i32), one compare (the ballot) and one
                // v_bitop3 per half: m & ~(ballot ^ sign)
#pragma unroll
                for (uint32_t b = 0; b < BITS; b++) {
                    const uint32_t s0 = (uint32_t)__builtin_amdgcn_sbfe((int)d[0], b, 1u);
                    const uint32_t s1 = (uint32_t)__builtin_amdgcn_sbfe((int)d[1], b, 1u);
                    const uint64_t b0 = __ballot(s0 != 0u), b1 = __ballot(s1 != 0u);
                    mlo[0] = __builtin_amdgcn_bitop3_b32(mlo[0], (uint32_t)b0, s0, 0x90);
                    mhi[0] = __builtin_amdgcn_bitop3_b32(mhi[0], (uint32_t)(b0 >> 32), s0, 0x90);
                    mlo[1] = __builtin_amdgcn_bitop3_b32(mlo[1], (uint32_t)b1, s1, 0x90);
                    mhi[1] = __builtin_amdgcn_bitop3_b32(mhi[1], (uint32_t)(b1 >> 32), s1, 0x90);
                }
            } else {
                mlo[0] = mlo[1] = lane < 32 ? 1u << lane : 0u; mhi[0] = mhi[1] = lane >= 32 ? 1u << (lane - 32) : 0u;
            }
            // all lanes read the running count of their digit, then the group's leader (no lane below it) bumps it;
            // the wave executes in lockstep and its LDS operations are performed in order, so a round sees the
            // updates of the rounds before it
            uint32_t r[2];
#pragma unroll
            for (uint32_t u = 0; u < 2; u++) {
                const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi[u], __builtin_amdgcn_mbcnt_lo(mlo[u], 0u));
                const uint32_t before = wh[d[u]];
                r[u] = before + below;
                if (valid[u] && below == 0u) wh[d[u]] = before + (uint32_t)(__popc(mlo[u]) + __popc(mhi[u]));
            }
            rank2[i / 2] = r[0] | (r[1] << 16);
        }
    };
    if (full) rank_tile(std::true_type{}); else rank_tile(std::false_type{});
    lds_barrier();
    {
        uint32_t c[DPT][NW], csum = 0;
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            const uint32_t d = threadIdx.x * DPT + j;
#pragma unroll
            for (uint32_t w = 0; w < NW; w++) { c[j][w] = owner ? wave_hist[w][d] : 0u; csum += c[j][w]; }
        }
        uint32_t tile_total;
        uint32_t lstart = block_excl_scan_lds<NT>(csum, ws, &tile_total);
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            const uint32_t d = threadIdx.x * DPT + j;
            if (owner) {
                gadj[d] = gbase[j] - lstart;
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) { wave_hist[w][d] = lstart; lstart += c[j][w]; }
            }
        }
    }
    lds_barrier();
    // local scatter: the tile sorted by digit in LDS, then written out in position order so that one store
    // instruction covers contiguous runs (a tile holds 16 keys per 8-bit digit on average: 64-byte runs)
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        if (full || first + i * 64 < n) {
            const uint32_t d = __builtin_amdgcn_ubfe(k[i], shift, BITS);
            const uint32_t lp = wave_hist[wave][d] + ((rank2[i / 2] >> (16 * (i & 1u))) & 0xFFFFu);
            spair[lp] = make_uint2(k[i], IDENT ? first + i * 64 : v[i]);
        }
    }
    lds_barrier();
    const uint32_t nvalid = full ? kSortTile : (tbase < n ? n - tbase : 0u);
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t j = i * NT + threadIdx.x;
        if (full || j < nvalid) {
            const uint2 kv = spair[j];
            uint32_t pos = gadj[__builtin_amdgcn_ubfe(kv.x, shift, BITS)] + j;
            if (EXP == 2) pos = tbase + j;        // linear stores instead of the scatter
            if (EXP == 1) pos = pos == 0xFFFFFFF3u ? 0u : 0x3FFFFFFFu;   // no stores (out of range: dropped)
            if (EXP == 4)   // price of counting the NEXT pass's (digit, destination tile) histogram with one global atomic per key
                __hip_atomic_fetch_add(exp_table + (size_t)__builtin_amdgcn_ubfe(kv.x, (shift + BITS) & 31u, BITS) * stride + (pos >> 12), 1u,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (a position past the end -- impossible unless the tables in memory are corrupt -- is dropped by the
            // descriptor's range check: never a wild store)
            __builtin_amdgcn_raw_buffer_store_b32(kv.x, bk_out, pos * 4u, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(kv.y, bv_out, pos * 4u, 0, 0);
        }
    }
}

SortScratch sort_scratch_layout(uint32_t n)
{
    SortScratch s;
    const size_t tiles = sort_table_stride(sort_num_tiles(n));   // table rows: one word per tile, padded to a multiple of 4
    size_t off = 0;
    s.digit_total = off; off += (size_t)kSortPasses * kRadixMax * 4;   // totals[pass][digit], written by the scan kernel
    s.hist = off;        off += (tiles * kRadixMax * 4 + 255) / 256 * 256;
    s.offs = off;        off += (tiles * kRadixMax * 4 + 255) / 256 * 256;
    s.total = off;
    return s;
}

uint32_t* sort_hist_table(void* sort_scratch, uint32_t n)
{
    return reinterpret_cast<uint32_t*>(static_cast<char*>(sort_scratch) + sort_scratch_layout(n).hist);
}

#ifdef RT_SORT_TUNING
static int tuning_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
#endif

// 3 passes x 10 bits (instead of 4 x 8) for keys of at most 30 bits while the 1024-digit tables stay small
bool sort_three_passes(uint32_t tiles)
{
#ifdef RT_SORT_TUNING
    const int forced = tuning_int("RT_SORT_3PASS", -1);
    if (forced >= 0) return forced != 0;
#endif
    return tiles <= kSort3PassMaxTiles;
}

// threads of the few-tiles kernels: 4 keys per thread in the histogram kernel; the down-sweep's count is an experiment knob
constexpr uint32_t kUpNT = kSortTile / 4;
#ifndef RT_SORT_DS_NT
#define RT_SORT_DS_NT 1024
#endif
constexpr uint32_t kWideNT = RT_SORT_DS_NT;
template <uint32_t BITS>
static void radix_pass(const uint32_t* sk, const uint32_t* sv, uint32_t* dk, uint32_t* dv, uint32_t n, uint32_t shift,
                       uint32_t tiles, uint32_t* hist, uint32_t* offs, uint32_t* dt, hipStream_t st, const uint32_t* n_dev,
                       bool have_hist, bool ident)
{
    const int vec_ok = (reinterpret_cast<uintptr_t>(sk) & 15u) == 0;
    const uint32_t stride = sort_table_stride(tiles);
    if (!have_hist) {
        if (sort_upsweep_quads(tiles)) sort_upsweep_kernel<BITS, 512, 4><<<stride / 4, 512, 0, st>>>(sk, n, shift, tiles, stride, hist, n_dev, vec_ok);
        else sort_upsweep_kernel<BITS, kUpNT, 1><<<tiles, kUpNT, 0, st>>>(sk, n, shift, tiles, stride, hist, n_dev, vec_ok);   // few tiles: 4 keys per thread
    }
    if (tiles <= 1024) sort_scan_kernel<<<(1u << BITS) / 4, 256, 0, st>>>(hist, tiles, stride, offs, dt);
    else sort_scan_wide_kernel<<<1u << BITS, 256, 0, st>>>(hist, tiles, stride, offs, dt);
    // (at most one workgroup per CU: 1024 threads, 4 keys each -- the workgroup's chain is the kernel: 15.8 -> 14.6 us at 1M)
    const bool wide = BITS == 10 && tiles <= 256 * kSortTileScale;
#define RT_DS(NT_, ID_, EX_) sort_downsweep_kernel<BITS, NT_, ID_, EX_><<<tiles, NT_, 0, st>>>(sk, sv, dk, dv, n, shift, stride, offs, dt, n_dev)
#ifdef RT_SORT_TUNING
    const int exper = tuning_int("RT_SORT_EXP", 0);
    if (exper == 1) { RT_DS(512, false, 1); return; }
    if (exper == 2) { RT_DS(512, false, 2); return; }
    if (exper == 3) { RT_DS(512, false, 3); return; }
    if (exper == 4) {
        if (wide) sort_downsweep_kernel<BITS, 1024, false, 4><<<tiles, 1024, 0, st>>>(sk, sv, dk, dv, n, shift, stride, offs, dt, n_dev, hist);
        else sort_downsweep_kernel<BITS, 512, false, 4><<<tiles, 512, 0, st>>>(sk, sv, dk, dv, n, shift, stride, offs, dt, n_dev, hist);
        return;
    }
#endif
    if (wide) {
        if (ident) RT_DS(kWideNT, true, 0); else RT_DS(kWideNT, false, 0);
    } else {
        if (ident) RT_DS(512, true, 0); else RT_DS(512, false, 0);
    }
#undef RT_DS
}

hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st, const uint32_t* n_dev, uint32_t key_bits, bool have_hist0,
                             bool ident0)
{
    if (n == 0) return hipSuccess;
    if (n > kSortMaxCount) return hipErrorInvalidValue;   // n * 4 would wrap the descriptors' 32-bit sizes
    const SortScratch L = sort_scratch_layout(n);
    char* base = static_cast<char*>(sort_scratch);
    uint32_t* digit_total = reinterpret_cast<uint32_t*>(base + L.digit_total);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.offs);
    const uint32_t tiles = sort_num_tiles(n);

    if (key_bits <= 30 && sort_three_passes(tiles)) {
        // Morton keys (30 bits): 3 passes x 10 bits = 60 B/key instead of 80.  An odd number of passes: the input is
        // taken from the temporaries (the Morton kernels write there) so that the result lands in keys / vals.
        uint32_t *sk = tmp_keys, *sv = tmp_vals, *dk = keys, *dv = vals;
        for (uint32_t pass = 0; pass < 3; pass++) {
            radix_pass<10>(sk, sv, dk, dv, n, pass * 10, tiles, hist, offs, digit_total + pass * kRadixMax, st, n_dev,
                           have_hist0 && pass == 0, ident0 && pass == 0);
            uint32_t* x;
            x = sk; sk = dk; dk = x;
            x = sv; sv = dv; dv = x;
        }
        return hipGetLastError();
    }
    uint32_t *sk = keys, *sv = vals, *dk = tmp_keys, *dv = tmp_vals;
    for (uint32_t pass = 0; pass < kSortPasses; pass++) {
        radix_pass<8>(sk, sv, dk, dv, n, pass * 8, tiles, hist, offs, digit_total + pass * kRadixMax, st, n_dev,
                      have_hist0 && pass == 0, ident0 && pass == 0);
        uint32_t* x;
        x = sk; sk = dk; dk = x;
        x = sv; sv = dv; dv = x;
    }
    return hipGetLastError();
}

hipError_t launch_radix_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                             uint32_t n, uint32_t shift, void* sort_scratch, hipStream_t st, const uint32_t* n_dev,
                             uint32_t** digit_total)
{
    const SortScratch L = sort_scratch_layout(n);
    char* base = static_cast<char*>(sort_scratch);
    uint32_t* dt = reinterpret_cast<uint32_t*>(base + L.digit_total);
    if (digit_total) *digit_total = dt;
    if (n == 0) return hipSuccess;
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.offs);
    radix_pass<8>(keys_in, vals_in, keys_out, vals_out, n, shift, sort_num_tiles(n), hist, offs, dt, st, n_dev, false, false);
    return hipGetLastError();
}

}  // namespace rt
