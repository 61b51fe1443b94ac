// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Replaces RadixSort() and its kernels CreateHistogramsLM / PrefixSumExclusive / Distribute
// (RadixSort.cu:47-225).  Same contract: 4 passes x 8 bits, ascending, stable, result back in
// keys/values after ping-ponging through the temporaries.  Different machine mapping:
//   * a workgroup owns a TILE of 4096 consecutive keys (the reference fixes 128 segments and gives
//     each to ONE 32-lane warp that ranks its keys with a 32-step serialised LDS atomic);
//   * ranks come from wave64 ballots: 8 ballots give the set of lanes holding the same digit, a
//     popcount below the lane gives the stable rank, and only the group leader touches LDS;
//   * tables live in caller scratch (no malloc/free/memset/sync inside the sort).
// Per pass: upsweep (tile digit histogram, 4 B/key read) -> scan (256 workgroups, one per digit)
// -> downsweep (8 B/key read, 8 B/key written).  Algorithmic traffic 20 B/key/pass = 80 B/key.
#include "rt_device.hpp"
#include "rt_launch.hpp"

namespace rt {

// lanes (among `valid` ones) whose 8-bit digit equals mine
__device__ __forceinline__ uint64_t match_digit8(uint32_t d, bool valid)
{
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    return m;
}

__global__ __launch_bounds__(256) void sort_upsweep_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                           uint32_t shift, uint32_t num_tiles,
                                                           uint32_t* __restrict__ hist, const uint32_t* n_dev)
{
    if (n_dev) n = *n_dev;   // device-side count (--pairs): tiles past it see no valid key and publish zeros
    __shared__ uint32_t h[kRadix];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * kSortTile;
    const int lane = threadIdx.x & 63;
    uint32_t k[kSortItems];
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = base + i * kSortThreads + threadIdx.x;
        k[i] = idx < n ? keys[idx] : 0u;
    }
    // the tile histogram only needs counts, not ranks.  Spread digits: one LDS atomic per key.  Clustered digits
    // (sorted, flat or constant input -- lanes would queue on one LDS word): group the lanes with 8 ballots and let
    // each group's leader add its size.  The wave chooses by looking at how many lanes share the first lane's digit.
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = base + i * kSortThreads + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t d = (k[i] >> shift) & (kRadix - 1);
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        if (__popcll(__ballot(valid && d == d0)) >= 8) {
            const uint64_t m = match_digit8(d, valid);
            if (valid && lane == __ffsll((unsigned long long)m) - 1) atomicAdd(&h[d], (uint32_t)__popcll(m));
        } else if (valid) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * num_tiles + tile] = h[threadIdx.x];   // no global atomics anywhere in the sort
}

// one workgroup per digit d: offs[d][t] = sum(hist[d][0..t)) (position inside the digit's output run),
// totals[d] = sum over all tiles.  The digit bases (exclusive scan of totals) are formed by each downsweep
// workgroup in its prologue: 256 values, one block scan -- cheaper than a launch or 256 same-address atomics per tile.
__global__ __launch_bounds__(256) void sort_scan_kernel(const uint32_t* __restrict__ hist, uint32_t num_tiles,
                                                        uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t ws[8];
    const uint32_t d = blockIdx.x;
    uint32_t running = 0;
    const uint32_t* hrow = hist + (size_t)d * num_tiles;
    uint32_t* orow = offs + (size_t)d * num_tiles;
    for (uint32_t c = 0; c < num_tiles; c += 256) {
        uint32_t t = c + threadIdx.x;
        uint32_t v = t < num_tiles ? hrow[t] : 0u;
        uint32_t chunk;
        uint32_t ex = block_excl_scan_u32<256>(v, ws, &chunk);
        if (t < num_tiles) orow[t] = running + ex;
        running += chunk;
    }
    if (threadIdx.x == 0) totals[d] = running;
}

__global__ __launch_bounds__(256) void sort_downsweep_kernel(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, uint32_t n,
                                                             uint32_t shift, uint32_t num_tiles,
                                                             const uint32_t* __restrict__ offs,
                                                             const uint32_t* __restrict__ totals,
                                                             const uint32_t* n_dev)
{
    if (n_dev) n = *n_dev;
    // wave_hist[w][d]: first the running count of digit d inside wave w's chunk, later the position inside the
    // tile (sorted by digit) of wave w's first key with digit d.
    __shared__ uint32_t wave_hist[4][kRadix];
    __shared__ uint32_t glob[kRadix];            // global position of local position 0 of digit d's run (mod 2^32)
    __shared__ uint32_t skey[kSortTile], sval[kSortTile];
    __shared__ uint32_t ws[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
#pragma unroll
    for (int w = 0; w < 4; w++) wave_hist[w][threadIdx.x] = 0;
    uint32_t dummy;
    const uint32_t digit_base = block_excl_scan_u32<256>(totals[threadIdx.x], ws, &dummy);  // ends with a barrier

    // wave w owns keys [base + w*1024, base + (w+1)*1024) in rounds of 64 consecutive keys, so
    // (wave, round, lane) order IS input order: stability.
    const uint32_t wbase = tile * kSortTile + wave * (kSortItems * 64);
    uint32_t k[kSortItems], v[kSortItems], rank[kSortItems];
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        k[i] = idx < n ? keys_in[idx] : 0u;
        v[i] = idx < n ? vals_in[idx] : 0u;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (k[i] >> shift) & (kRadix - 1);
        const uint64_t m = match_digit8(d, valid);
        // all lanes read the running count, then the group leader bumps it; the wave executes in
        // lockstep and its LDS ops retire in order, so round i+1 sees round i's update.
        volatile uint32_t* wh = &wave_hist[wave][0];
        const uint32_t before = wh[d];
        rank[i] = before + (uint32_t)__popcll(m & lt_mask);
        if (valid && lane == __ffsll((unsigned long long)m) - 1) wh[d] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {
        const uint32_t d = threadIdx.x;
        const uint32_t c0 = wave_hist[0][d], c1 = wave_hist[1][d], c2 = wave_hist[2][d], c3 = wave_hist[3][d];
        uint32_t tile_total;
        const uint32_t lstart = block_excl_scan_u32<256>(c0 + c1 + c2 + c3, ws, &tile_total);
        wave_hist[0][d] = lstart;
        wave_hist[1][d] = lstart + c0;
        wave_hist[2][d] = lstart + c0 + c1;
        wave_hist[3][d] = lstart + c0 + c1 + c2;
        glob[d] = digit_base + offs[(size_t)d * num_tiles + tile] - lstart;
    }
    __syncthreads();
    // local scatter: the tile sorted by digit in LDS, then written out in position order so that one store
    // instruction covers contiguous runs (a tile holds 16 keys per digit on average: 64-byte runs)
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (k[i] >> shift) & (kRadix - 1);
            const uint32_t lp = wave_hist[wave][d] + rank[i];
            skey[lp] = k[i];
            sval[lp] = v[i];
        }
    }
    __syncthreads();
    const uint32_t tbase = tile * kSortTile;
    const uint32_t nvalid = tbase < n ? min(kSortTile, n - tbase) : 0u;
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        const uint32_t j = i * kSortThreads + threadIdx.x;
        if (j < nvalid) {
            const uint32_t key = skey[j];
            const uint32_t pos = glob[(key >> shift) & (kRadix - 1)] + j;
            keys_out[pos] = key;
            vals_out[pos] = sval[j];
        }
    }
}

SortScratch sort_scratch_layout(uint32_t n)
{
    SortScratch s;
    const size_t tiles = sort_num_tiles(n) ? sort_num_tiles(n) : 1;
    size_t off = 0;
    s.digit_total = off; off += (size_t)kSortPasses * kRadix * 4;   // totals[pass][digit], written by the scan kernel
    s.hist = off;        off += (tiles * kRadix * 4 + 255) / 256 * 256;
    s.offs = off;        off += (tiles * kRadix * 4 + 255) / 256 * 256;
    s.total = off;
    return s;
}

hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st, const uint32_t* n_dev)
{
    if (n == 0) return hipSuccess;
    const SortScratch L = sort_scratch_layout(n);
    char* base = static_cast<char*>(sort_scratch);
    uint32_t* digit_total = reinterpret_cast<uint32_t*>(base + L.digit_total);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.offs);
    const uint32_t tiles = sort_num_tiles(n);

    uint32_t *sk = keys, *sv = vals, *dk = tmp_keys, *dv = tmp_vals;
    for (uint32_t pass = 0; pass < kSortPasses; pass++) {
        const uint32_t shift = pass * kRadixBits;
        uint32_t* dt = digit_total + pass * kRadix;
        sort_upsweep_kernel<<<tiles, kSortThreads, 0, st>>>(sk, n, shift, tiles, hist, n_dev);
        sort_scan_kernel<<<kRadix, 256, 0, st>>>(hist, tiles, offs, dt);
        sort_downsweep_kernel<<<tiles, kSortThreads, 0, st>>>(sk, sv, dk, dv, n, shift, tiles, offs, dt, n_dev);
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
    const uint32_t tiles = sort_num_tiles(n);
    sort_upsweep_kernel<<<tiles, kSortThreads, 0, st>>>(keys_in, n, shift, tiles, hist, n_dev);
    sort_scan_kernel<<<kRadix, 256, 0, st>>>(hist, tiles, offs, dt);
    sort_downsweep_kernel<<<tiles, kSortThreads, 0, st>>>(keys_in, vals_in, keys_out, vals_out, n, shift, tiles, offs, dt, n_dev);
    return hipGetLastError();
}

}  // namespace rt
