// build_front.hip -- scene bounds + Morton codes (the front of RunBottomUpBuild).
//
// Replaces CalculateSceneAabb (Multiblock.cu:104-114) and GenerateMortonCodes
// (BottomUpBuilder.cu:98-115).  HBM-bound streaming kernels: the 36-byte AoS triangles are read as a
// flat float stream with 16-byte-per-lane loads (the reference reads float3 at a 12-byte stride and
// issues 6 global atomics per triangle on one 24-byte address).
#include "rt_device.hpp"
#include "rt_launch.hpp"
#include "rt_pairing.hpp"

namespace rt {

// ---------------------------------------------------------------------------------------------
// Scene AABB.  Flat element e of the triangle array belongs to axis e % 3.  A thread strides by
// gridDim.x*1024 float4's, and the grid is a multiple of 3 blocks, so (float4 index) % 3 -- the axis
// of the first element of every float4 a thread loads -- is constant per thread (1024 % 3 == 1): min/max are kept in
// that rotated frame and un-rotated once at the end.  Everything is folded in the ordered-int
// domain (DeviceUtils.cuh:3-13), wave-reduced, block-reduced through LDS and finished with 6 integer
// atomics per BLOCK (exact, order independent).
// Build mode (init.status != null): the kernel is the FIRST launch of rt_run_bottom_up_build and also does the build's
// tiny initialisations (status words, the level hand-off counters: BuildWrapper.cu:288-303 does such things with six
// memset / memcpy calls), and every workgroup STORES its partial box at aabb[6 * blockIdx.x] (no atomics, so nothing
// needs resetting first); the Morton kernels fold the gridDim.x partial boxes.
struct BuildInit { uint32_t* status; uint32_t n_tris; uint32_t* arrive; uint32_t arrive_words; };

__global__ __launch_bounds__(1024) void scene_aabb_kernel(const float* __restrict__ f, uint64_t nfloats,
                                                          int* __restrict__ aabb, uint32_t nparts, BuildInit init)
{
    if (init.status && blockIdx.x == 0) {
        if (threadIdx.x < 8) init.status[threadIdx.x] = threadIdx.x == 1 ? init.n_tris : 0u;   // [1] = number of leaves (pairs: overwritten)
        for (uint32_t i = threadIdx.x; i < init.arrive_words; i += 1024) init.arrive[i] = 0u;
    }
    const uint64_t nvec = nfloats >> 2;
    const uint64_t stride = (uint64_t)gridDim.x * 1024;
    uint64_t q = (uint64_t)blockIdx.x * 1024 + threadIdx.x;
    const int r = (int)(q % 3);  // axis of element 0 of every float4 of this thread

    int lo0 = 0x7f7fffff, lo1 = 0x7f7fffff, lo2 = 0x7f7fffff;
    int hi0 = (int)0x80800000, hi1 = (int)0x80800000, hi2 = (int)0x80800000;
    const float4* f4 = reinterpret_cast<const float4*>(f);
    auto fold = [&](const float4& v) {
        const int a = float_to_ordered_int(v.x), b = float_to_ordered_int(v.y);
        const int c = float_to_ordered_int(v.z), d = float_to_ordered_int(v.w);
        lo0 = min(lo0, min(a, d)); hi0 = max(hi0, max(a, d));  // elements 0 and 3 share an axis
        lo1 = min(lo1, b);         hi1 = max(hi1, b);
        lo2 = min(lo2, c);         hi2 = max(hi2, c);
    };
    // 4 independent 16-byte loads in flight per thread
    for (; q + 3 * stride < nvec; q += 4 * stride) {
        const float4 v0 = f4[q], v1 = f4[q + stride], v2 = f4[q + 2 * stride], v3 = f4[q + 3 * stride];
        fold(v0); fold(v1); fold(v2); fold(v3);
    }
    for (; q < nvec; q += stride) fold(f4[q]);
    // un-rotate: frame slot s is axis (r + s) % 3
    int lo[3], hi[3];
    lo[0] = r == 0 ? lo0 : (r == 1 ? lo2 : lo1);
    lo[1] = r == 0 ? lo1 : (r == 1 ? lo0 : lo2);
    lo[2] = r == 0 ? lo2 : (r == 1 ? lo1 : lo0);
    hi[0] = r == 0 ? hi0 : (r == 1 ? hi2 : hi1);
    hi[1] = r == 0 ? hi1 : (r == 1 ? hi0 : hi2);
    hi[2] = r == 0 ? hi2 : (r == 1 ? hi1 : hi0);
    // tail (nfloats % 4 elements) by one thread
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (uint64_t e = nvec << 2; e < nfloats; e++) {
            int v = float_to_ordered_int(f[e]);
            int ax = (int)(e % 3);
            lo[ax] = min(lo[ax], v);
            hi[ax] = max(hi[ax], v);
        }
    }
    __shared__ int red[16][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int l = wave_min_i32(lo[k]), h = wave_max_i32(hi[k]);
        if (lane == 0) { red[wave][k] = l; red[wave][3 + k] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        int v = red[0][k];
        for (int w = 1; w < 16; w++) v = k < 3 ? min(v, red[w][k]) : max(v, red[w][k]);
        // 6 atomics per workgroup into partial box (blockIdx mod nparts): same-address device atomics queue at about
        // 50 ns each, so the build spreads them over kAabbParts copies that the Morton kernel folds
        if (init.status) {
            aabb[6 * blockIdx.x + k] = v;                       // build mode: this workgroup's own slot
        } else {
            int* dst = aabb + 6 * (blockIdx.x % nparts);
            if (k < 3) atomicMin(&dst[k], v); else atomicMax(&dst[k], v);
        }
    }
}

__global__ void reset_aabb_kernel(int* aabb)
{
    // BuildWrapper.cu:288-289: ordered-int "empty" box
    if (threadIdx.x < 6) aabb[threadIdx.x] = threadIdx.x < 3 ? 0x7f7fffff : (int)0x80800000;
}

// ---------------------------------------------------------------------------------------------
// BottomUpBuilder.cu:12-32
__device__ __forceinline__ uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton3d(float x, float y, float z)
{
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits((uint32_t)x) * 4 + expand_bits((uint32_t)y) * 2 + expand_bits((uint32_t)z);
}

// the scene box from `nparts` partial boxes (nparts * 6 <= 256 threads take one word each); every thread gets it
__device__ __forceinline__ void fold_scene_box(const int* __restrict__ parts, uint32_t nparts, int* sbox /* LDS [6] */,
                                               float* mn, float* mx)
{
    if (threadIdx.x < 6) sbox[threadIdx.x] = threadIdx.x < 3 ? 0x7f7fffff : (int)0x80800000;
    __syncthreads();
    // thread t (of the first 6 * (blockDim / 6)) folds word t % 6 of the partial boxes t / 6, t / 6 + blockDim / 6, ... in a
    // register (consecutive threads read consecutive words), then ONE LDS atomic per thread
    const uint32_t per = blockDim.x / 6;
    if (threadIdx.x < per * 6) {
        const uint32_t k = threadIdx.x % 6;
        int acc = k < 3 ? 0x7f7fffff : (int)0x80800000;
        for (uint32_t q = threadIdx.x / 6; q < nparts; q += per) {
            const int v = parts[q * 6 + k];
            acc = k < 3 ? min(acc, v) : max(acc, v);
        }
        if (k < 3) atomicMin(&sbox[k], acc); else atomicMax(&sbox[k], acc);
    }
    __syncthreads();
    for (int k = 0; k < 3; k++) { mn[k] = ordered_int_to_float(sbox[k]); mx[k] = ordered_int_to_float(sbox[3 + k]); }
}

// Morton codes: one block = 256 triangles = 2304 floats staged through LDS with coalesced float4
// loads; each thread then reads its 9 floats at a 9-dword stride (9 is odd: conflict-free).
__global__ __launch_bounds__(256) void morton_kernel(uint32_t* __restrict__ codes, uint32_t* __restrict__ values,
                                                     const float* __restrict__ f, const int* __restrict__ aabb,
                                                     uint32_t n, uint32_t nparts, int* __restrict__ aabb_out)
{
    __shared__ float s[256 * 9];
    __shared__ int sbox[6];
    float bmin[3], bmax[3];
    fold_scene_box(aabb, nparts, sbox, bmin, bmax);
    if (aabb_out && blockIdx.x == 0 && threadIdx.x < 6) aabb_out[threadIdx.x] = sbox[threadIdx.x];   // the folded box, for later readers
    const uint64_t nfloats = (uint64_t)n * 9;
    const uint64_t base = (uint64_t)blockIdx.x * (256 * 9);
    const float4* f4 = reinterpret_cast<const float4*>(f + base);  // 9216-byte block stride: 16-B aligned
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int q = threadIdx.x + k * 256;
        if (q < 576) {
            uint64_t e = base + (uint64_t)q * 4;
            if (e + 4 <= nfloats) {
                float4 v = f4[q];
                s[q * 4 + 0] = v.x; s[q * 4 + 1] = v.y; s[q * 4 + 2] = v.z; s[q * 4 + 3] = v.w;
            } else {
                for (int j = 0; j < 4; j++)
                    if (e + j < nfloats) s[q * 4 + j] = f[e + j];
            }
        }
    }
    __syncthreads();
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    const float* t = &s[threadIdx.x * 9];
    // centre = (v0 + v1 + v2) / 3.0f, left-associated (BottomUpBuilder.cu:104-107)
    float cx = ((t[0] + t[3]) + t[6]) / 3.0f;
    float cy = ((t[1] + t[4]) + t[7]) / 3.0f;
    float cz = ((t[2] + t[5]) + t[8]) / 3.0f;
    const float minx = bmin[0], miny = bmin[1], minz = bmin[2], maxx = bmax[0], maxy = bmax[1], maxz = bmax[2];
    cx = (cx - minx) / (maxx - minx);
    cy = (cy - miny) / (maxy - miny);
    cz = (cz - minz) / (maxz - minz);
    // clamp(c, 0, 1) = fmaxf(0, fminf(c, 1))  (helper_math.h:1161-1164); NaN (flat axis) -> 1
    cx = fmaxf(0.0f, fminf(cx, 1.0f));
    cy = fmaxf(0.0f, fminf(cy, 1.0f));
    cz = fmaxf(0.0f, fminf(cz, 1.0f));
    codes[gid] = morton3d(cx, cy, cz);
    values[gid] = gid;
}

// Morton codes + the tile histograms of the sort's first pass in one launch: a workgroup owns TQ sort tiles
// (4096 triangles each: every lane reads its 36-byte triangle directly -- neighbouring lanes use the rest of every
// line) and counts the low BITS bits of its codes in LDS exactly as sort_upsweep_kernel would (same table layout: rows
// of `stride` words, TQ = 4 publishes 16 bytes per digit).  values == nullptr: the identity values[i] = i
// (BottomUpBuilder.cu:113) is not written; the sort's first pass regenerates it.
template <uint32_t BITS, uint32_t NT, uint32_t TQ>
__global__ __launch_bounds__(NT) void morton_hist_kernel(uint32_t* __restrict__ codes, uint32_t* __restrict__ values,
                                                          const float* __restrict__ f, const int* __restrict__ aabb,
                                                          uint32_t n, uint32_t nparts, int* __restrict__ aabb_out,
                                                          uint32_t* __restrict__ hist, uint32_t stride)
{
    constexpr uint32_t RADIX = 1u << BITS;
    static_assert(TQ == 1 || TQ == 4, "one dword or one uint4 per digit");
    __shared__ uint32_t h[TQ][RADIX];
    __shared__ int sbox[6];
    float bmin[3], bmax[3];
    for (uint32_t d = threadIdx.x; d < RADIX * TQ; d += NT) (&h[0][0])[d] = 0;
    fold_scene_box(aabb, nparts, sbox, bmin, bmax);   // (its barriers also order the zeroing of h)
    if (aabb_out && blockIdx.x == 0 && threadIdx.x < 6) aabb_out[threadIdx.x] = sbox[threadIdx.x];
    const uint32_t tile0 = blockIdx.x * TQ;
    const int lane = threadIdx.x & 63;
    const float minx = bmin[0], miny = bmin[1], minz = bmin[2], maxx = bmax[0], maxy = bmax[1], maxz = bmax[2];
    for (uint32_t t = 0; t < TQ; t++) {
        const uint32_t base = (tile0 + t) * kSortTile;
        if (base >= n) break;
#pragma unroll 4
        for (uint32_t i = 0; i < kSortTile / NT; i++) {
            const uint32_t gid = base + i * NT + threadIdx.x;
            const bool valid = gid < n;
            uint32_t code = 0;
            if (valid) {
                float tr[9];
                load_tri9(f + (size_t)gid * 9, tr);
                float cx = ((tr[0] + tr[3]) + tr[6]) / 3.0f;
                float cy = ((tr[1] + tr[4]) + tr[7]) / 3.0f;
                float cz = ((tr[2] + tr[5]) + tr[8]) / 3.0f;
                cx = (cx - minx) / (maxx - minx);
                cy = (cy - miny) / (maxy - miny);
                cz = (cz - minz) / (maxz - minz);
                cx = fmaxf(0.0f, fminf(cx, 1.0f));
                cy = fmaxf(0.0f, fminf(cy, 1.0f));
                cz = fmaxf(0.0f, fminf(cz, 1.0f));
                code = morton3d(cx, cy, cz);
                codes[gid] = code;
                if (values) values[gid] = gid;
            }
            const uint32_t d = code & (RADIX - 1);
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            if (__popcll(__ballot(valid && d == d0)) >= 8) {
                const uint64_t m = match_digit<BITS>(d, valid);
                if (valid && lane == __ffsll((unsigned long long)m) - 1) atomicAdd(&h[t][d], (uint32_t)__popcll(m));
            } else if (valid) {
                atomicAdd(&h[t][d], 1u);
            }
        }
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < RADIX; d += NT) {
        if (TQ == 4) *reinterpret_cast<uint4*>(hist + (size_t)d * stride + tile0) = make_uint4(h[0][d], h[1 % TQ][d], h[2 % TQ][d], h[3 % TQ][d]);
        else hist[(size_t)d * stride + tile0] = h[0][d];
    }
}

// ---------------------------------------------------------------------------------------------
// --pairs: GenerateMortonCodesPairs (BottomUpBuilder.cu:117-164).  Candidate k = triangles (2k, 2k+1); it yields one
// leaf (a merged quad, or a lone last triangle) or two.  The reference claims leaf slots with atomicAdd (arrival
// order); here slot = exclusive prefix sum of the per-candidate leaf counts in input order:
//   pair_flags_kernel  : merge decision per candidate (1 byte) + leaf count per workgroup
//   pair_scan_kernel   : one workgroup scans the workgroup counts, publishes the total number of leaves L
//   morton_pairs_kernel: workgroup scan + offset -> slot; centres, Morton codes, values (MSB = pair flag)
constexpr uint32_t kPairThreads = 256;

__global__ __launch_bounds__(kPairThreads) void pair_flags_kernel(const float* __restrict__ f, uint32_t n,
                                                                  uint8_t* __restrict__ flags,
                                                                  uint32_t* __restrict__ block_sums)
{
    __shared__ uint32_t ws[8];
    const uint32_t k = blockIdx.x * kPairThreads + threadIdx.x, tid = 2 * k;
    uint32_t valid = 0;
    if (tid < n) {
        bool merge = false;
        if (tid + 1 < n) {
            float A[9], B[9];
            load_tri9(f + (size_t)tid * 9, A);
            load_tri9(f + (size_t)tid * 9 + 9, B);
            merge = pair_merges(A, B);
        }
        flags[k] = merge ? 1 : 0;
        valid = 1 + ((tid + 1 < n && !merge) ? 1u : 0u);
    }
    uint32_t total;
    block_excl_scan_u32<kPairThreads>(valid, ws, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void pair_scan_kernel(uint32_t* __restrict__ block_sums, uint32_t nblocks,
                                                         uint32_t* __restrict__ num_leaves)
{
    __shared__ uint32_t ws[20];
    uint32_t running = 0;
    for (uint32_t c = 0; c < nblocks; c += 1024) {
        const uint32_t i = c + threadIdx.x;
        const uint32_t v = i < nblocks ? block_sums[i] : 0u;
        uint32_t chunk;
        const uint32_t ex = block_excl_scan_u32<1024>(v, ws, &chunk);
        if (i < nblocks) block_sums[i] = running + ex;
        running += chunk;
    }
    if (threadIdx.x == 0) *num_leaves = running;
}

__global__ __launch_bounds__(kPairThreads) void morton_pairs_kernel(uint32_t* __restrict__ codes,
                                                                    uint32_t* __restrict__ values,
                                                                    const float* __restrict__ f,
                                                                    const int* __restrict__ aabb,
                                                                    const uint8_t* __restrict__ flags,
                                                                    const uint32_t* __restrict__ block_offsets,
                                                                    uint32_t n, uint32_t nparts, int* __restrict__ aabb_out)
{
    __shared__ uint32_t ws[8];
    __shared__ int sbox[6];
    float mn[3], mx[3];
    fold_scene_box(aabb, nparts, sbox, mn, mx);
    if (aabb_out && blockIdx.x == 0 && threadIdx.x < 6) aabb_out[threadIdx.x] = sbox[threadIdx.x];
    const uint32_t k = blockIdx.x * kPairThreads + threadIdx.x, tid = 2 * k;
    const bool live = tid < n, second_valid = tid + 1 < n;
    const bool merge = live && flags[k] != 0;
    const uint32_t valid = live ? 1 + ((second_valid && !merge) ? 1u : 0u) : 0u;
    uint32_t total;
    const uint32_t idx = block_offsets[blockIdx.x] + block_excl_scan_u32<kPairThreads>(valid, ws, &total);
    if (!live) return;
    float A[9], B[9];
    load_tri9(f + (size_t)tid * 9, A);
    load_tri9(f + (size_t)(second_valid ? tid + 1 : tid) * 9, B);
    float c1[3], c2[3];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        c1[j] = ((A[j] + A[3 + j]) + A[6 + j]) / 3.0f;           // Triangle::Centre (Common.cuh:240-242)
        c2[j] = ((B[j] + B[3 + j]) + B[6 + j]) / 3.0f;
        if (merge) c1[j] = (c1[j] + c2[j]) * 0.5f;
        c1[j] = fmaxf(0.0f, fminf((c1[j] - mn[j]) / (mx[j] - mn[j]), 1.0f));
        c2[j] = fmaxf(0.0f, fminf((c2[j] - mn[j]) / (mx[j] - mn[j]), 1.0f));
    }
    values[idx] = merge ? (tid | 0x80000000u) : tid;                // MSB marks a pair (BottomUpBuilder.cu:152-153)
    codes[idx] = morton3d(c1[0], c1[1], c1[2]);
    if (second_valid && !merge) {
        values[idx + 1] = tid + 1;
        codes[idx + 1] = morton3d(c2[0], c2[1], c2[2]);
    }
}

hipError_t launch_morton_pairs(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                               uint8_t* flags, uint32_t* block_sums, uint32_t* num_leaves, hipStream_t st, uint32_t nparts,
                               int* aabb_out)
{
    const uint32_t cand = (n + 1) / 2;
    const uint32_t blocks = (cand + kPairThreads - 1) / kPairThreads;
    const float* f = reinterpret_cast<const float*>(tris);
    if (n == 0) { pair_scan_kernel<<<1, 1024, 0, st>>>(block_sums, 0, num_leaves); return hipGetLastError(); }
    pair_flags_kernel<<<blocks, kPairThreads, 0, st>>>(f, n, flags, block_sums);
    pair_scan_kernel<<<1, 1024, 0, st>>>(block_sums, blocks, num_leaves);
    morton_pairs_kernel<<<blocks, kPairThreads, 0, st>>>(codes, values, f, aabb, flags, block_sums, n, nparts, aabb_out);
    return hipGetLastError();
}

hipError_t launch_block_scan(uint32_t* sums, uint32_t count, uint32_t* total, hipStream_t st)
{
    pair_scan_kernel<<<1, 1024, 0, st>>>(sums, count, total);
    return hipGetLastError();
}

hipError_t launch_pair_slots(const rt_triangle* tris, uint32_t n, uint8_t* flags, uint32_t* block_sums,
                             uint32_t* num_leaves, hipStream_t st)
{
    const uint32_t cand = (n + 1) / 2;
    const uint32_t blocks = (cand + kPairThreads - 1) / kPairThreads;
    if (n) pair_flags_kernel<<<blocks, kPairThreads, 0, st>>>(reinterpret_cast<const float*>(tris), n, flags, block_sums);
    pair_scan_kernel<<<1, 1024, 0, st>>>(block_sums, n ? blocks : 0, num_leaves);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
hipError_t launch_reset_aabb(int* aabb, hipStream_t st)
{
    reset_aabb_kernel<<<1, 64, 0, st>>>(aabb);
    return hipGetLastError();
}

static uint32_t scene_aabb_blocks(uint32_t n, bool many)
{
    const uint64_t nfloats = (uint64_t)n * 9;
    // ~4 float4 per thread.  One box (atomics on the same words from every workgroup): at most 255 workgroups of 1024.
    // Separate partial boxes: twice the workgroups pay off.
    uint64_t want = (nfloats / 16 + 1023) / 1024;
    const uint32_t cap = many ? 510u : 255u;
    uint32_t blocks = (uint32_t)(want < 3 ? 3 : (want > cap ? cap : want));
    return blocks / 3 * 3;                                   // multiple of 3 (see kernel comment)
}

hipError_t launch_scene_aabb(const rt_triangle* tris, uint32_t n, int* aabb, hipStream_t st, uint32_t nparts)
{
    if (n == 0) return hipSuccess;
    scene_aabb_kernel<<<scene_aabb_blocks(n, nparts > 1), 1024, 0, st>>>(reinterpret_cast<const float*>(tris), (uint64_t)n * 9, aabb,
                                                                        nparts ? nparts : 1u, BuildInit{nullptr, 0, nullptr, 0});
    return hipGetLastError();
}

// the build's first launch: partial boxes stored at aabb_parts[6 * workgroup] (*nparts_out of them) + the initialisations
hipError_t launch_scene_aabb_build(const rt_triangle* tris, uint32_t n, int* aabb_parts, uint32_t* nparts_out, uint32_t* status,
                                   uint32_t* arrive, uint32_t arrive_words, hipStream_t st)
{
    const uint32_t blocks = scene_aabb_blocks(n, true);
    *nparts_out = blocks;
    scene_aabb_kernel<<<blocks, 1024, 0, st>>>(reinterpret_cast<const float*>(tris), (uint64_t)n * 9, aabb_parts, blocks,
                                               BuildInit{status, n, arrive, arrive_words});
    return hipGetLastError();
}

hipError_t launch_morton(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                         hipStream_t st, uint32_t nparts, int* aabb_out)
{
    if (n == 0) return hipSuccess;
    morton_kernel<<<(n + 255) / 256, 256, 0, st>>>(codes, values, reinterpret_cast<const float*>(tris), aabb, n,
                                                   nparts ? nparts : 1u, aabb_out);
    return hipGetLastError();
}

hipError_t launch_morton_hist(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                              hipStream_t st, uint32_t nparts, int* aabb_out, uint32_t* hist, uint32_t bits)
{
    if (n == 0) return hipSuccess;
    const uint32_t tiles = sort_num_tiles(n), stride = sort_table_stride(tiles);
    const float* f = reinterpret_cast<const float*>(tris);
    const uint32_t np = nparts ? nparts : 1u;
    // few tiles: 1024 threads per tile (4 triangles each) -- a pass over 245 tiles is one workgroup's chain; many: 256
    // threads per tile (four tiles per workgroup, as the sort's own histogram kernel does, was measured slower here: 83 vs
    // 77 us at 10M -- this kernel is bound by the 36-byte triangle reads, not by its table writes)
    if (sort_upsweep_quads(tiles)) {
        if (bits == 10) morton_hist_kernel<10, 256, 1><<<tiles, 256, 0, st>>>(codes, values, f, aabb, n, np, aabb_out, hist, stride);
        else morton_hist_kernel<8, 256, 1><<<tiles, 256, 0, st>>>(codes, values, f, aabb, n, np, aabb_out, hist, stride);
    } else {
        if (bits == 10) morton_hist_kernel<10, 1024, 1><<<tiles, 1024, 0, st>>>(codes, values, f, aabb, n, np, aabb_out, hist, stride);
        else morton_hist_kernel<8, 1024, 1><<<tiles, 1024, 0, st>>>(codes, values, f, aabb, n, np, aabb_out, hist, stride);
    }
    return hipGetLastError();
}

}  // namespace rt
