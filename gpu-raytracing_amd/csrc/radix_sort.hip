// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Replaces RadixSort() and its kernels CreateHistogramsLM / PrefixSumExclusive / Distribute
// (RadixSort.cu:47-225).  Same contract: ascending, stable, result in keys/values after ping-ponging
// through the temporaries; 4 passes x 8 bits for arbitrary 32-bit keys (the public entry point), 3 passes
// x 10 bits for the builder's 30-bit Morton keys while the tables stay small.  Different machine mapping:
//   * the keys are cut into TILES of 4096 consecutive keys; a workgroup owns a GROUP of `tpw` consecutive tiles and
//     works through them one after the other (the reference fixes 128 segments and gives each to ONE 32-lane warp
//     that ranks its keys with a 32-step serialised LDS atomic).  The tables (one column per group) shrink by tpw, the
//     per-workgroup prologue (digit bases, the group's offsets) is paid once per group, and the next tile's keys are
//     loaded while the current one is ranked -- the workgroup barriers inside the loop wait for LDS only
//     (`s_waitcnt lgkmcnt(0); s_barrier`), so those loads stay in flight across them;
//   * ranks come from wave64 ballots: BITS ballots give the set of lanes holding the same digit, a
//     popcount below the lane gives the stable rank, and only the group leader touches LDS;
//   * a tile is sorted by digit in LDS and written in position order (64-byte runs per 8-bit digit);
//   * tables live in caller scratch (no malloc/free/memset/sync inside the sort), no global atomics.
// Per pass: upsweep (group digit histogram, 4 B/key read) -> scan (one wave or workgroup per digit)
// -> downsweep (8 B/key read, 8 B/key written).  Algorithmic traffic 20 B/key/pass = 80 B/key.
#include "rt_device.hpp"
#include "rt_launch.hpp"

#ifdef RT_SORT_TUNING
#include <cstdlib>
#endif

namespace rt {

// workgroup barrier that orders LDS traffic only: outstanding global loads (the next tile's keys) and stores (the previous
// tile's output) stay in flight across it.  __syncthreads() would drain them (it is a fence for global memory too).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// block_excl_scan_u32 with LDS-only barriers
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan_lds(uint32_t v, uint32_t* ws, uint32_t* total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int NW = NT / 64;
    const uint32_t incl = wave_incl_scan_u32(v, lane);
    if (lane == 63) ws[wave] = incl;
    lds_barrier();
    if (wave == 0) {
        const uint32_t w = lane < NW ? ws[lane] : 0u;
        const uint32_t wi = wave_incl_scan_u32(w, lane);
        if (lane < NW) ws[lane] = wi - w;
        if (lane == NW - 1) ws[NW] = wi;
    }
    lds_barrier();
    const uint32_t r = ws[wave] + incl - v;
    *total = ws[NW];
    lds_barrier();
    return r;
}

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
template <uint32_t BITS>
__device__ __forceinline__ void hist_add(uint32_t* h, uint32_t d, bool valid)
{
    const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
    if (__popcll(__ballot(valid && d == d0)) >= 8) {
        uint32_t mlo, mhi;
        match_digit_halves<BITS>(d, valid, mlo, mhi);
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
        if (valid && below == 0u) atomicAdd(&h[d], (uint32_t)(__popc(mlo) + __popc(mhi)));
    } else if (valid) {
        atomicAdd(&h[d], 1u);
    }
}

// ---- upsweep: hist[d][group] = number of keys of the group (tpw consecutive tiles) whose digit is d
template <uint32_t BITS, uint32_t NT>
__global__ __launch_bounds__(NT) void sort_upsweep_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                           uint32_t shift, uint32_t num_tiles, uint32_t tpw,
                                                           uint32_t* __restrict__ hist, const uint32_t* n_dev, int vec_ok)
{
    constexpr uint32_t RADIX = 1u << BITS;
    constexpr uint32_t ITEMS = kSortTile / NT;
    if (n_dev) n = *n_dev;   // device-side count (--pairs): tiles past it see no valid key and publish zeros
    __shared__ uint32_t h[RADIX];
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT) h[d] = 0;
    __syncthreads();
    const uint32_t group = blockIdx.x, num_groups = gridDim.x;
    const uint32_t t0 = group * tpw, t1 = min(t0 + tpw, num_tiles);
    for (uint32_t tile = t0; tile < t1; tile++) {
        const uint32_t base = tile * kSortTile;
        if (vec_ok && base + kSortTile <= n) {
            // a full tile: 16-byte loads, 4 consecutive keys per lane (the order inside a tile is irrelevant to a count)
            uint4 q[ITEMS / 4];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS / 4; i++)
                q[i] = reinterpret_cast<const uint4*>(keys + base)[i * NT + threadIdx.x];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS / 4; i++) {
                hist_add<BITS>(h, (q[i].x >> shift) & (RADIX - 1), true);
                hist_add<BITS>(h, (q[i].y >> shift) & (RADIX - 1), true);
                hist_add<BITS>(h, (q[i].z >> shift) & (RADIX - 1), true);
                hist_add<BITS>(h, (q[i].w >> shift) & (RADIX - 1), true);
            }
        } else {
            uint32_t k[ITEMS];
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = base + i * NT + threadIdx.x;
                k[i] = idx < n ? keys[idx] : 0u;
            }
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++)
                hist_add<BITS>(h, (k[i] >> shift) & (RADIX - 1), base + i * NT + threadIdx.x < n);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT)
        hist[(size_t)d * num_groups + group] = h[d];   // no global atomics anywhere in the sort
}

// one WAVE per digit d: offs[d][g] = sum(hist[d][0..g)) (position inside the digit's output run), totals[d] = sum over
// all groups; no barriers.  The digit bases (exclusive scan of totals) are formed by each downsweep workgroup in its
// prologue: one block scan -- cheaper than a launch or RADIX same-address atomics per group.
__global__ __launch_bounds__(256) void sort_scan_kernel(const uint32_t* __restrict__ hist, uint32_t num_groups,
                                                        uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    const int lane = threadIdx.x & 63;
    const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t running = 0;
    const uint32_t* hrow = hist + (size_t)d * num_groups;
    uint32_t* orow = offs + (size_t)d * num_groups;
    for (uint32_t c = 0; c < num_groups; c += 64) {
        const uint32_t t = c + lane;
        const uint32_t v = t < num_groups ? hrow[t] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (t < num_groups) orow[t] = running + incl - v;
        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) totals[d] = running;
}

// the same table, one WORKGROUP per digit (block scans of 256 groups at a time): for many groups, where a single wave per
// digit would run a long serial chain
__global__ __launch_bounds__(256) void sort_scan_wide_kernel(const uint32_t* __restrict__ hist, uint32_t num_groups,
                                                             uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t ws[8];
    const uint32_t d = blockIdx.x;
    uint32_t running = 0;
    const uint32_t* hrow = hist + (size_t)d * num_groups;
    uint32_t* orow = offs + (size_t)d * num_groups;
    for (uint32_t c = 0; c < num_groups; c += 256) {
        const uint32_t t = c + threadIdx.x;
        const uint32_t v = t < num_groups ? hrow[t] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_excl_scan_u32<256>(v, ws, &chunk);
        if (t < num_groups) orow[t] = running + ex;
        running += chunk;
    }
    if (threadIdx.x == 0) totals[d] = running;
}

// ---- downsweep.  NT threads per 4096-key tile: 512 (8 keys per thread), or 1024 (4 per thread) when a pass is at most
// one workgroup per CU -- one workgroup's dependent chain then IS the kernel's duration, and twice the waves halve its
// ranking rounds.  IDENT: the values of this pass are the identity (the builder's first pass without --pairs:
// GenerateMortonCodes writes values[i] = i, BottomUpBuilder.cu:113), so they are not read -- and the Morton kernel
// does not write them.
template <uint32_t BITS, uint32_t NT, bool IDENT, bool PF>
__global__ __launch_bounds__(NT, (NT == 512 && BITS == 8) ? 6 : 4) void sort_downsweep_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, uint32_t n,
                                                            uint32_t shift, uint32_t num_tiles, uint32_t tpw,
                                                            const uint32_t* __restrict__ offs,
                                                            const uint32_t* __restrict__ totals,
                                                            const uint32_t* n_dev)
{
    constexpr uint32_t RADIX = 1u << BITS;
    constexpr uint32_t NW = NT / 64;             // waves
    constexpr uint32_t ITEMS = kSortTile / NT;   // keys per thread and tile
    constexpr uint32_t DPT = (RADIX + NT - 1) / NT;   // digits per thread in the table phase (thread t owns digits t*DPT ...)
    static_assert(kSortTile % NT == 0 && (RADIX % NT == 0 || NT % RADIX == 0), "tile and digit split");
    const bool owner = threadIdx.x * DPT < RADIX;     // (NT > RADIX: the upper threads own no digit)
    if (n_dev) n = *n_dev;
    // wave_hist[w][d]: first the running count of digit d inside wave w's chunk of the tile, later the position inside
    // the tile (sorted by digit) of wave w's first key with digit d.
    __shared__ uint32_t wave_hist[NW][RADIX];
    __shared__ uint32_t gadj[RADIX];             // (global position of the tile's first key of digit d) - (its local position)
    __shared__ uint32_t skey[kSortTile], sval[kSortTile];
    __shared__ uint32_t ws[NW + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware group order: hardware deals workgroup b to XCD b % 8; give XCD x a contiguous run of groups.  Consecutive
    // tiles write consecutive runs of every digit's output region (64 bytes on average = half a line), so the two halves
    // of a line meet in ONE L2 instead of being written back separately by two XCDs.
    const uint32_t num_groups = gridDim.x;
    const uint32_t group = xcd_contiguous(blockIdx.x, num_groups);
    const uint32_t t0 = group * tpw, t1 = min(t0 + tpw, num_tiles);

    // the first tile's keys: issued before anything else, they arrive behind the prologue
    uint32_t k[ITEMS], v[ITEMS];
    {
        const uint32_t wbase = t0 * kSortTile + wave * (ITEMS * 64);
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = wbase + i * 64 + lane;
            k[i] = idx < n ? keys_in[idx] : 0u;
            v[i] = IDENT ? idx : (idx < n ? vals_in[idx] : 0u);
        }
    }
#pragma unroll
    for (uint32_t w = 0; w < NW; w++)
        for (uint32_t d = threadIdx.x; d < RADIX; d += NT) wave_hist[w][d] = 0;
    // digit bases = exclusive scan of the digit totals; + this group's position inside every digit's run
    uint32_t gbase[DPT];
    {
        uint32_t tot[DPT], toff[DPT], tsum = 0;
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            tot[j] = owner ? totals[threadIdx.x * DPT + j] : 0u;
            toff[j] = owner ? offs[(size_t)(threadIdx.x * DPT + j) * num_groups + group] : 0u;
            tsum += tot[j];
        }
        uint32_t dummy;
        uint32_t digit_base = block_excl_scan_lds<NT>(tsum, ws, &dummy);  // ends with a barrier
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) { gbase[j] = digit_base + toff[j]; digit_base += tot[j]; }
    }

    for (uint32_t tile = t0; tile < t1; tile++) {
        // wave w owns keys [base + w*ITEMS*64, base + (w+1)*ITEMS*64) in rounds of 64 consecutive keys, so
        // (wave, round, lane) order IS input order: stability.
        const uint32_t tbase = tile * kSortTile;
        const uint32_t wbase = tbase + wave * (ITEMS * 64);
        // the next tile's keys: in flight during this tile's ranking, table, scatter and write-out
        uint32_t kn[ITEMS], vn[ITEMS];
        if (!PF && tile != t0) {
            // (no prefetch: this tile's keys are loaded here)
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = wbase + i * 64 + lane;
                k[i] = idx < n ? keys_in[idx] : 0u;
                v[i] = IDENT ? idx : (idx < n ? vals_in[idx] : 0u);
            }
        }
        if (PF && tile + 1 < t1) {
            const uint32_t nb = wbase + kSortTile;
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) {
                const uint32_t idx = nb + i * 64 + lane;
                kn[i] = idx < n ? keys_in[idx] : 0u;
                vn[i] = IDENT ? idx : (idx < n ? vals_in[idx] : 0u);
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) { kn[i] = 0u; vn[i] = 0u; }
        }
        uint32_t rank2[ITEMS / 2];   // two 16-bit ranks per register (a rank is < ITEMS * 64 <= 1024)
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = wbase + i * 64 + lane;
            const bool valid = idx < n;
            const uint32_t d = (k[i] >> shift) & (RADIX - 1);
            uint32_t mlo, mhi;
            match_digit_halves<BITS>(d, valid, mlo, mhi);
            const uint32_t below = __builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u));
            // all lanes read the running count, then the group leader bumps it; the wave executes in
            // lockstep and its LDS ops retire in order, so round i+1 sees round i's update.
            volatile uint32_t* wh = &wave_hist[wave][0];
            const uint32_t before = wh[d];
            if (i & 1u) rank2[i / 2] |= (before + below) << 16;
            else rank2[i / 2] = before + below;
            if (valid && below == 0u) wh[d] = before + (uint32_t)(__popc(mlo) + __popc(mhi));
            __builtin_amdgcn_wave_barrier();
        }
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
                    for (uint32_t w = 0; w < NW; w++) { wave_hist[w][d] = lstart; lstart += c[j][w]; gbase[j] += c[j][w]; }
                }
            }
        }
        lds_barrier();
        // local scatter: the tile sorted by digit in LDS, then written out in position order so that one store
        // instruction covers contiguous runs (a tile holds 16 keys per 8-bit digit on average: 64-byte runs)
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t idx = wbase + i * 64 + lane;
            if (idx < n) {
                const uint32_t d = (k[i] >> shift) & (RADIX - 1);
                const uint32_t lp = wave_hist[wave][d] + ((rank2[i / 2] >> (16 * (i & 1u))) & 0xFFFFu);
                skey[lp] = k[i];
                sval[lp] = v[i];
            }
        }
        lds_barrier();
        const uint32_t nvalid = tbase < n ? min(kSortTile, n - tbase) : 0u;
#pragma unroll
        for (uint32_t i = 0; i < ITEMS; i++) {
            const uint32_t j = i * NT + threadIdx.x;
            if (j < nvalid) {
                const uint32_t key = skey[j];
                const uint32_t pos = gadj[(key >> shift) & (RADIX - 1)] + j;
                if (pos < n) {   // (always; the position comes from tables in memory: never a wild store)
                    keys_out[pos] = key;
                    vals_out[pos] = sval[j];
                }
            }
        }
        // the counters of the next tile (nobody reads wave_hist between the scatter and the next ranking)
#pragma unroll
        for (uint32_t w = 0; w < NW; w++)
            for (uint32_t d = threadIdx.x; d < RADIX; d += NT) wave_hist[w][d] = 0;
        if (PF) {
#pragma unroll
            for (uint32_t i = 0; i < ITEMS; i++) { k[i] = kn[i]; v[i] = vn[i]; }
        }
        lds_barrier();
    }
}

SortScratch sort_scratch_layout(uint32_t n)
{
    SortScratch s;
    const size_t tiles = sort_num_tiles(n) ? sort_num_tiles(n) : 1;
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

// tiles per workgroup: one tile each while every workgroup is resident at once (3 workgroups of 512 threads per CU);
// beyond that a workgroup takes a run of consecutive tiles, so that the grid stays one resident generation
uint32_t sort_tiles_per_group(uint32_t tiles)
{
#ifdef RT_SORT_TUNING
    const int forced = tuning_int("RT_SORT_TPW", 0);
    if (forced > 0) return (uint32_t)forced;
#endif
    constexpr uint32_t kResident = 256 * 3;
    return tiles <= kResident ? 1u : (tiles + kResident - 1) / kResident;
}

template <uint32_t BITS>
static void radix_pass(const uint32_t* sk, const uint32_t* sv, uint32_t* dk, uint32_t* dv, uint32_t n, uint32_t shift,
                       uint32_t tiles, uint32_t* hist, uint32_t* offs, uint32_t* dt, hipStream_t st, const uint32_t* n_dev,
                       bool have_hist, bool ident)
{
    const uint32_t tpw = sort_tiles_per_group(tiles);
    const uint32_t groups = (tiles + tpw - 1) / tpw;
    const int vec_ok = (reinterpret_cast<uintptr_t>(sk) & 15u) == 0;
    if (!have_hist) {
        if (groups <= 512) sort_upsweep_kernel<BITS, 1024><<<groups, 1024, 0, st>>>(sk, n, shift, tiles, tpw, hist, n_dev, vec_ok);   // few groups: 4 keys per thread and tile
        else sort_upsweep_kernel<BITS, 512><<<groups, 512, 0, st>>>(sk, n, shift, tiles, tpw, hist, n_dev, vec_ok);
    }
    if (groups <= 512) sort_scan_kernel<<<(1u << BITS) / 4, 256, 0, st>>>(hist, groups, offs, dt);
    else sort_scan_wide_kernel<<<1u << BITS, 256, 0, st>>>(hist, groups, offs, dt);
    // (at most one workgroup per CU: 1024 threads, 4 keys each -- the workgroup's chain is the kernel: 15.8 -> 14.6 us at 1M)
    const bool wide = BITS == 10 && groups <= 256;
    bool pf = tpw > 1;   // one tile per workgroup: nothing to prefetch
#ifdef RT_SORT_TUNING
    pf = pf && tuning_int("RT_SORT_PF", 1) != 0;
#endif
#define RT_DS(NT_, ID_, PF_) sort_downsweep_kernel<BITS, NT_, ID_, PF_><<<groups, NT_, 0, st>>>(sk, sv, dk, dv, n, shift, tiles, tpw, offs, dt, n_dev)
    if (wide) {
        if (ident) RT_DS(1024, true, false); else RT_DS(1024, false, false);
    } else if (pf) {
        if (ident) RT_DS(512, true, true); else RT_DS(512, false, true);
    } else {
        if (ident) RT_DS(512, true, false); else RT_DS(512, false, false);
    }
#undef RT_DS
}

hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st, const uint32_t* n_dev, uint32_t key_bits, bool have_hist0,
                             bool ident0)
{
    if (n == 0) return hipSuccess;
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
