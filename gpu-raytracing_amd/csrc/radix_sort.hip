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
                                                           uint32_t* __restrict__ hist,
                                                           uint32_t* __restrict__ digit_total)
{
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
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = base + i * kSortThreads + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t d = (k[i] >> shift) & (kRadix - 1);
        const uint64_t m = match_digit8(d, valid);
        if (valid && lane == __ffsll((unsigned long long)m) - 1) atomicAdd(&h[d], (uint32_t)__popcll(m));
    }
    __syncthreads();
    const uint32_t c = h[threadIdx.x];
    hist[(size_t)threadIdx.x * num_tiles + tile] = c;
    if (c) atomicAdd(&digit_total[threadIdx.x], c);
}

// one workgroup per digit d: offs[d][t] = sum(digit_total[0..d)) + sum(hist[d][0..t))
__global__ __launch_bounds__(256) void sort_scan_kernel(const uint32_t* __restrict__ hist,
                                                        const uint32_t* __restrict__ digit_total,
                                                        uint32_t num_tiles, uint32_t* __restrict__ offs)
{
    __shared__ uint32_t ws[8];
    const uint32_t d = blockIdx.x;
    uint32_t total;
    uint32_t mine = threadIdx.x < d ? digit_total[threadIdx.x] : 0u;
    block_excl_scan_u32<256>(mine, ws, &total);
    uint32_t running = total;
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
}

__global__ __launch_bounds__(256) void sort_downsweep_kernel(const uint32_t* __restrict__ keys_in,
                                                             const uint32_t* __restrict__ vals_in,
                                                             uint32_t* __restrict__ keys_out,
                                                             uint32_t* __restrict__ vals_out, uint32_t n,
                                                             uint32_t shift, uint32_t num_tiles,
                                                             const uint32_t* __restrict__ offs)
{
    // wave_hist[w][d]: first the running count of digit d inside wave w's chunk, later the global
    // output position of wave w's first key with digit d.
    __shared__ uint32_t wave_hist[4][kRadix];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x;
#pragma unroll
    for (int w = 0; w < 4; w++) wave_hist[w][threadIdx.x] = 0;
    __syncthreads();

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
        uint32_t run = offs[(size_t)d * num_tiles + tile];
#pragma unroll
        for (int w = 0; w < 4; w++) {
            uint32_t c = wave_hist[w][d];
            wave_hist[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < (int)kSortItems; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (k[i] >> shift) & (kRadix - 1);
            const uint32_t pos = wave_hist[wave][d] + rank[i];
            keys_out[pos] = k[i];
            vals_out[pos] = v[i];
        }
    }
}

__global__ void sort_zero_totals_kernel(uint32_t* digit_total)
{
    digit_total[blockIdx.x * kRadix + threadIdx.x] = 0;
}

SortScratch sort_scratch_layout(uint32_t n)
{
    SortScratch s;
    const size_t tiles = sort_num_tiles(n) ? sort_num_tiles(n) : 1;
    size_t off = 0;
    s.digit_total = off; off += (size_t)kSortPasses * kRadix * 4;
    s.hist = off;        off += (tiles * kRadix * 4 + 255) / 256 * 256;
    s.offs = off;        off += (tiles * kRadix * 4 + 255) / 256 * 256;
    s.total = off;
    return s;
}

hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st)
{
    if (n == 0) return hipSuccess;
    const SortScratch L = sort_scratch_layout(n);
    char* base = static_cast<char*>(sort_scratch);
    uint32_t* digit_total = reinterpret_cast<uint32_t*>(base + L.digit_total);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.offs);
    const uint32_t tiles = sort_num_tiles(n);

    sort_zero_totals_kernel<<<kSortPasses, kRadix, 0, st>>>(digit_total);
    uint32_t *sk = keys, *sv = vals, *dk = tmp_keys, *dv = tmp_vals;
    for (uint32_t pass = 0; pass < kSortPasses; pass++) {
        const uint32_t shift = pass * kRadixBits;
        uint32_t* dt = digit_total + pass * kRadix;
        sort_upsweep_kernel<<<tiles, kSortThreads, 0, st>>>(sk, n, shift, tiles, hist, dt);
        sort_scan_kernel<<<kRadix, 256, 0, st>>>(hist, dt, tiles, offs);
        sort_downsweep_kernel<<<tiles, kSortThreads, 0, st>>>(sk, sv, dk, dv, n, shift, tiles, offs);
        uint32_t* x;
        x = sk; sk = dk; dk = x;
        x = sv; sv = dv; dv = x;
    }
    return hipGetLastError();
}

}  // namespace rt
