// radix_sort.hip -- stable LSD radix sort of (u32 key, u32 value) pairs for gfx950.
//
// Replaces RadixSort() and its kernels CreateHistogramsLM / PrefixSumExclusive / Distribute
// (RadixSort.cu:47-225).  Same contract: ascending, stable, result in keys/values after ping-ponging
// through the temporaries; 4 passes x 8 bits for arbitrary 32-bit keys (the public entry point), 3 passes
// x 10 bits for the builder's 30-bit Morton keys (60 B/key instead of 80).  Different machine mapping:
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

template <uint32_t BITS, uint32_t NT>
__global__ __launch_bounds__(NT) void sort_upsweep_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                           uint32_t shift, uint32_t num_tiles,
                                                           uint32_t* __restrict__ hist, const uint32_t* n_dev)
{
    constexpr uint32_t RADIX = 1u << BITS;
    if (n_dev) n = *n_dev;   // device-side count (--pairs): tiles past it see no valid key and publish zeros
    __shared__ uint32_t h[RADIX];
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT) h[d] = 0;
    __syncthreads();
    const uint32_t tile = blockIdx.x;
    const uint32_t base = tile * kSortTile;
    const int lane = threadIdx.x & 63;
    constexpr uint32_t ITEMS = kSortTile / NT;
    uint32_t k[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        uint32_t idx = base + i * NT + threadIdx.x;
        k[i] = idx < n ? keys[idx] : 0u;
    }
    // the tile histogram only needs counts, not ranks.  Spread digits: one LDS atomic per key.  Clustered digits
    // (sorted, flat or constant input -- lanes would queue on one LDS word): group the lanes with ballots and let
    // each group's leader add its size.  The wave chooses by looking at how many lanes share the first lane's digit.
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        uint32_t idx = base + i * NT + threadIdx.x;
        const bool valid = idx < n;
        const uint32_t d = (k[i] >> shift) & (RADIX - 1);
        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
        if (__popcll(__ballot(valid && d == d0)) >= 8) {
            const uint64_t m = match_digit<BITS>(d, valid);
            if (valid && lane == __ffsll((unsigned long long)m) - 1) atomicAdd(&h[d], (uint32_t)__popcll(m));
        } else if (valid) {
            atomicAdd(&h[d], 1u);
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT)
        hist[(size_t)d * num_tiles + tile] = h[d];   // no global atomics anywhere in the sort
}

// one WAVE per digit d: offs[d][t] = sum(hist[d][0..t)) (position inside the digit's output run), totals[d] = sum over
// all tiles; no barriers.  The digit bases (exclusive scan of totals) are formed by each downsweep workgroup in its
// prologue: one block scan -- cheaper than a launch or RADIX same-address atomics per tile.
__global__ __launch_bounds__(256) void sort_scan_kernel(const uint32_t* __restrict__ hist, uint32_t num_tiles,
                                                        uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    const int lane = threadIdx.x & 63;
    const uint32_t d = blockIdx.x * 4 + (threadIdx.x >> 6);
    uint32_t running = 0;
    const uint32_t* hrow = hist + (size_t)d * num_tiles;
    uint32_t* orow = offs + (size_t)d * num_tiles;
    for (uint32_t c = 0; c < num_tiles; c += 64) {
        const uint32_t t = c + lane;
        const uint32_t v = t < num_tiles ? hrow[t] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (t < num_tiles) orow[t] = running + incl - v;
        running += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) totals[d] = running;
}

// the same table, one WORKGROUP per digit (block scans of 256 tiles at a time): for many tiles, where a single wave per
// digit would run a long serial chain (2444 tiles: 22 us against 11)
__global__ __launch_bounds__(256) void sort_scan_wide_kernel(const uint32_t* __restrict__ hist, uint32_t num_tiles,
                                                             uint32_t* __restrict__ offs, uint32_t* __restrict__ totals)
{
    __shared__ uint32_t ws[8];
    const uint32_t d = blockIdx.x;
    uint32_t running = 0;
    const uint32_t* hrow = hist + (size_t)d * num_tiles;
    uint32_t* orow = offs + (size_t)d * num_tiles;
    for (uint32_t c = 0; c < num_tiles; c += 256) {
        const uint32_t t = c + threadIdx.x;
        const uint32_t v = t < num_tiles ? hrow[t] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_excl_scan_u32<256>(v, ws, &chunk);
        if (t < num_tiles) orow[t] = running + ex;
        running += chunk;
    }
    if (threadIdx.x == 0) totals[d] = running;
}

// NT threads per 4096-key tile: 256 (16 keys per thread) when there are many tiles, 512 (8 per thread) when there are few --
// a 1M-key pass is 245 workgroups on 256 CUs, i.e. one workgroup's dependent chain (16 ranking rounds, each an LDS
// read-modify-write) IS the kernel's duration; twice the waves halve the rounds.
template <uint32_t BITS, uint32_t NT>
__global__ __launch_bounds__(NT) void sort_downsweep_kernel(const uint32_t* __restrict__ keys_in,
                                                            const uint32_t* __restrict__ vals_in,
                                                            uint32_t* __restrict__ keys_out,
                                                            uint32_t* __restrict__ vals_out, uint32_t n,
                                                            uint32_t shift, uint32_t num_tiles,
                                                            const uint32_t* __restrict__ offs,
                                                            const uint32_t* __restrict__ totals,
                                                            const uint32_t* n_dev)
{
    constexpr uint32_t RADIX = 1u << BITS;
    constexpr uint32_t NW = NT / 64;             // waves
    constexpr uint32_t ITEMS = kSortTile / NT;   // keys per thread
    constexpr uint32_t DPT = (RADIX + NT - 1) / NT;   // digits per thread in the table phases (thread t owns digits t*DPT ...)
    static_assert(kSortTile % NT == 0 && (RADIX % NT == 0 || NT % RADIX == 0), "tile and digit split");
    const bool owner = threadIdx.x * DPT < RADIX;     // (NT > RADIX: the upper threads own no digit)
    if (n_dev) n = *n_dev;
    // wave_hist[w][d]: first the running count of digit d inside wave w's chunk, later the position inside the
    // tile (sorted by digit) of wave w's first key with digit d.
    __shared__ uint32_t wave_hist[NW][RADIX];
    __shared__ uint32_t glob[RADIX];             // global position of local position 0 of digit d's run (mod 2^32)
    __shared__ uint32_t skey[kSortTile], sval[kSortTile];
    __shared__ uint32_t ws[NW + 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware tile order: hardware deals workgroup b to XCD b % 8; give XCD x a contiguous run of tiles.  Consecutive
    // tiles write consecutive runs of every digit's output region (64 bytes on average = half a line), so the two halves
    // of a line now meet in ONE L2 instead of being written back separately by two XCDs.
    const uint32_t tile = xcd_contiguous(blockIdx.x, gridDim.x);
#pragma unroll
    for (uint32_t w = 0; w < NW; w++)
        for (uint32_t d = threadIdx.x; d < RADIX; d += NT) wave_hist[w][d] = 0;
    // digit bases = exclusive scan of the digit totals
    uint32_t tot[DPT], tsum = 0;
#pragma unroll
    for (uint32_t j = 0; j < DPT; j++) { tot[j] = owner ? totals[threadIdx.x * DPT + j] : 0u; tsum += tot[j]; }
    // this tile's position inside every digit's run: a strided gather (one line per digit) that is needed only after the
    // ranking -- issued here, it is in flight behind the scan and the ballots (9.0 -> 8.2 us per workgroup at 1M; hoisting
    // the key loads as well gained nothing: the first barrier then waits for them)
    uint32_t toff[DPT];
#pragma unroll
    for (uint32_t j = 0; j < DPT; j++) toff[j] = owner ? offs[(size_t)(threadIdx.x * DPT + j) * num_tiles + tile] : 0u;
    uint32_t dummy;
    uint32_t digit_base = block_excl_scan_u32<NT>(tsum, ws, &dummy);  // ends with a barrier

    // wave w owns keys [base + w*ITEMS*64, base + (w+1)*ITEMS*64) in rounds of 64 consecutive keys, so
    // (wave, round, lane) order IS input order: stability.
    const uint32_t wbase = tile * kSortTile + wave * (ITEMS * 64);
    uint32_t k[ITEMS], v[ITEMS], rank[ITEMS];
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        k[i] = idx < n ? keys_in[idx] : 0u;
        v[i] = idx < n ? vals_in[idx] : 0u;
    }
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        const bool valid = idx < n;
        const uint32_t d = (k[i] >> shift) & (RADIX - 1);
        const uint64_t m = match_digit<BITS>(d, valid);
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
        uint32_t c[DPT][NW], csum = 0;
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            const uint32_t d = threadIdx.x * DPT + j;
#pragma unroll
            for (uint32_t w = 0; w < NW; w++) { c[j][w] = owner ? wave_hist[w][d] : 0u; csum += c[j][w]; }
        }
        uint32_t tile_total;
        uint32_t lstart = block_excl_scan_u32<NT>(csum, ws, &tile_total);
#pragma unroll
        for (uint32_t j = 0; j < DPT; j++) {
            const uint32_t d = threadIdx.x * DPT + j;
            if (owner) {
                glob[d] = digit_base + toff[j] - lstart;
#pragma unroll
                for (uint32_t w = 0; w < NW; w++) { wave_hist[w][d] = lstart; lstart += c[j][w]; }
                digit_base += tot[j];
            }
        }
    }
    __syncthreads();
    // local scatter: the tile sorted by digit in LDS, then written out in position order so that one store
    // instruction covers contiguous runs (a tile holds 16 keys per 8-bit digit on average: 64-byte runs)
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        uint32_t idx = wbase + i * 64 + lane;
        if (idx < n) {
            const uint32_t d = (k[i] >> shift) & (RADIX - 1);
            const uint32_t lp = wave_hist[wave][d] + rank[i];
            skey[lp] = k[i];
            sval[lp] = v[i];
        }
    }
    __syncthreads();
    const uint32_t tbase = tile * kSortTile;
    const uint32_t nvalid = tbase < n ? min(kSortTile, n - tbase) : 0u;
#pragma unroll
    for (uint32_t i = 0; i < ITEMS; i++) {
        const uint32_t j = i * NT + threadIdx.x;
        if (j < nvalid) {
            const uint32_t key = skey[j];
            const uint32_t pos = glob[(key >> shift) & (RADIX - 1)] + j;
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

template <uint32_t BITS>
static void radix_pass(const uint32_t* sk, const uint32_t* sv, uint32_t* dk, uint32_t* dv, uint32_t n, uint32_t shift,
                       uint32_t tiles, uint32_t* hist, uint32_t* offs, uint32_t* dt, hipStream_t st, const uint32_t* n_dev,
                       bool have_hist = false)
{
    if (!have_hist) {
        if (tiles <= 512) sort_upsweep_kernel<BITS, 1024><<<tiles, 1024, 0, st>>>(sk, n, shift, tiles, hist, n_dev);   // few tiles: 4 keys per thread
        else sort_upsweep_kernel<BITS, kSortThreads><<<tiles, kSortThreads, 0, st>>>(sk, n, shift, tiles, hist, n_dev);
    }
    if (tiles <= 512) sort_scan_kernel<<<(1u << BITS) / 4, 256, 0, st>>>(hist, tiles, offs, dt);
    else sort_scan_wide_kernel<<<1u << BITS, 256, 0, st>>>(hist, tiles, offs, dt);
    // 512 threads per tile (8 keys each): half the ranking rounds of the 256-thread form and fewer registers (more waves
    // per SIMD); the dependent chain of one workgroup is what a pass over few tiles costs
    // (at most one workgroup per CU: 1024 threads, 4 keys each -- the workgroup's chain is the kernel: 15.8 -> 14.6 us at 1M)
    if (BITS == 10 && tiles <= 256) sort_downsweep_kernel<BITS, 1024><<<tiles, 1024, 0, st>>>(sk, sv, dk, dv, n, shift, tiles, offs, dt, n_dev);
    else sort_downsweep_kernel<BITS, 512><<<tiles, 512, 0, st>>>(sk, sv, dk, dv, n, shift, tiles, offs, dt, n_dev);
}

hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st, const uint32_t* n_dev, uint32_t key_bits, bool have_hist0)
{
    if (n == 0) return hipSuccess;
    const SortScratch L = sort_scratch_layout(n);
    char* base = static_cast<char*>(sort_scratch);
    uint32_t* digit_total = reinterpret_cast<uint32_t*>(base + L.digit_total);
    uint32_t* hist = reinterpret_cast<uint32_t*>(base + L.hist);
    uint32_t* offs = reinterpret_cast<uint32_t*>(base + L.offs);
    const uint32_t tiles = sort_num_tiles(n);

    if (key_bits <= 30) {
        // Morton keys (30 bits): 3 passes x 10 bits = 60 B/key instead of 80.  An odd number of passes: the input is
        // taken from the temporaries (the Morton kernels write there) so that the result lands in keys / vals.
        uint32_t *sk = tmp_keys, *sv = tmp_vals, *dk = keys, *dv = vals;
        for (uint32_t pass = 0; pass < 3; pass++) {
            radix_pass<10>(sk, sv, dk, dv, n, pass * 10, tiles, hist, offs, digit_total + pass * kRadixMax, st, n_dev,
                           have_hist0 && pass == 0);
            uint32_t* x;
            x = sk; sk = dk; dk = x;
            x = sv; sv = dv; dv = x;
        }
        return hipGetLastError();
    }
    uint32_t *sk = keys, *sv = vals, *dk = tmp_keys, *dv = tmp_vals;
    for (uint32_t pass = 0; pass < kSortPasses; pass++) {
        radix_pass<8>(sk, sv, dk, dv, n, pass * 8, tiles, hist, offs, digit_total + pass * kRadixMax, st, n_dev,
                      have_hist0 && pass == 0);
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
    radix_pass<8>(keys_in, vals_in, keys_out, vals_out, n, shift, sort_num_tiles(n), hist, offs, dt, st, n_dev);
    return hipGetLastError();
}

}  // namespace rt
