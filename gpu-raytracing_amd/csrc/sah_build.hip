// sah_build.hip -- the SAH build path for gfx950.
//
// Replaces RunSahBuild (BuildWrapper.cu:140-251) without spatial splits: Setup (Multiblock.cu:139-207),
// GridBlockCounts / GridBlockScan / GridBlockDistribute (:427-546) and both SharedTaskBuild launches
// (SharedTaskBuilder.cu:93-607, 909-967).  Same tree as the reference -- leaves bucketed by centroid into a
// 4 x 4 x 4 grid, one top-down binned-SAH sub-tree per cell (8 bins on the longest centroid axis, leaf threshold 2,
// object-median split when the centroid box has no area), a SAH top tree over the non-empty cells, trace root
// (slot 0, count 1) -- with a different machine mapping and a DETERMINISTIC numbering (the rules are stated above
// ora_build_sah in oracle/rt_oracle.c, which this file matches bit for bit):
//
//   * the reference builds each cell with ONE 512-thread block (64 blocks in total) looping over a task queue;
//     here every task of every cell that is alive in a level is processed in the same launch:
//       sah_bin_kernel       256 positions per workgroup; bins are aggregated in LDS per (task, bin) and flushed with
//                            a handful of integer atomics (ordered-int min / max / add: order independent);
//       sah_split_kernel     one thread per task: plane selection, parent descriptor, child tasks; the whole wave
//                            helps with what scales with a task's size (median child boxes, the per-chunk prefix
//                            of "goes left" counts a task spanning several workgroups needs);
//       sah_partition_kernel stable partition (block scan + that prefix) into the other id buffer;
//     tasks of <= 64 items leave the level loop and are finished by sah_small_kernel, one wave per task
//     (the reference runs its small tasks one thread each: PerInstanceRunTask, SharedTaskBuilder.cu:742-907);
//   * node slots are a function of the split POSITION (slot = bias + 2 * mid), not of an allocation counter, so no
//     kernel needs to agree on an order and the top tree is built in the same launches as the cell trees;
//   * leaf slots = input order (pairs: prefix sums), cell members by one stable 8-bit radix pass (radix_sort.hip).
//
// The depth of the trees is data dependent; the reference reads num_leaves back and loops on the host
// (BuildWrapper.cu:229).  Here the number of LAUNCHES is fixed by n: the level loop runs log2(items per cell / 64) + 3
// levels (kernels of a level nobody reaches return at once), then sah_finish_kernel finishes whatever is still larger
// than 64 items, one workgroup per task, then sah_small_kernel.  No copy, no synchronisation: rt_run_sah_build is a
// sequence of asynchronous launches (hipGraph-capturable, tests/test_gpu_graph.py); its error flags stay in the scratch
// status word like the bottom-up build's.
#include "rt_device.hpp"
#include "rt_launch.hpp"
#include "rt_pairing.hpp"
#ifdef RT_SORT_TUNING
#include <cstdlib>
#endif

namespace rt {

constexpr uint32_t kSahCells = 64;
constexpr uint32_t kSahChunk = 256;       // positions per workgroup in the level kernels
constexpr uint32_t kSahSmall = 64;        // tasks with <= 64 items are finished by one wave
constexpr uint32_t kSahMaxLocal = 32;     // runs of equal task id a chunk can hold (tasks in the loop have >= 65 items)
constexpr uint32_t kInactive = 0xFFFFFFFFu;
constexpr uint32_t kSahMaxLevels = 1024;
constexpr uint32_t kBoundParts = 64, kCellParts = 16;
constexpr int kEmptyLo = 0x7f7fffff, kEmptyHi = (int)0x80800000;   // ordered-int FLT_MAX / -FLT_MAX (BuildWrapper.cu:170-171)
constexpr uint32_t kBinWords = 13;        // p box [6], c box [6], count
constexpr uint32_t kLeafMask = 0x7FFFFFFFu; // item_leaf: TrianglePair index | (two triangles ? 1 << 31 : 0)

enum : uint32_t { kSahErrLocals = 0x100, kSahErrLevels = 0x200 };

struct SahTask { float c[6]; float p[6]; uint32_t start, end, parent_idx, flags; };   // flags bit0: top tree
struct SahSplit { uint32_t kind, plane, mid, left_id, right_id, pad[3]; };            // kind 1 binned, 2 median
struct SahSmall { uint32_t start, end, parent_idx, flags; };   // flags bit0 top tree, bit1 id buffer, bit2 top root

struct SahHeader {
    int gp[6], gc[6];                  // scene primitive / centroid bounds, ordered ints
    uint32_t status[8];                // [0] error flags, [1] number of items L, [2] leaf records, [3] split budget asked
    uint32_t small_count, live_report, pad[2];   // live_report: tasks still alive after the last batch of levels (sah_patch_top_kernel)
    uint32_t cell_count[kSahCells], cell_start[kSahCells], cell_task[kSahCells];
    int cell_p[kSahCells][6], cell_c[kSahCells][6];
    uint32_t level_count[kSahMaxLevels];
    // same-address device atomics from different workgroups queue at about 50 ns each (they are resolved behind the
    // eight L2s), so the bounds are reduced into several partial copies chosen by workgroup index and folded by the
    // next kernel: chains of 16 instead of 1024
    int part_bounds[kBoundParts][12];              // [gp 6, gc 6]
    int part_cell[kCellParts][kSahCells][12];      // [cell_p 6, cell_c 6]
};

struct SahArgs {
    SahHeader* H;
    rt_node* nodes;
    const float* aabbs;                // [B + 64][6]
    const uint32_t* item_leaf;         // [B]
    uint32_t* ids[2];
    uint32_t* task_of[2];
    uint8_t* binof;
    SahTask* tasks[2];
    SahSplit* splits;
    int* bins[2];                      // [task][8][13]
    uint32_t* chunk_hist;              // [chunk][2][8]
    uint32_t* chunk_prefix;            // [chunk]
    SahSmall* small;
    uint32_t n, B, M;                  // triangles; B = upper bound of the item count (n, + n/5 with splits) = first
                                       // top-tree position; positions M = B + 64
};

// ---- min / max on the ordered-int encoding (what the atomics compute)
__device__ __forceinline__ float fmin_ord(float a, float b) { return float_to_ordered_int(a) < float_to_ordered_int(b) ? a : b; }
__device__ __forceinline__ float fmax_ord(float a, float b) { return float_to_ordered_int(a) > float_to_ordered_int(b) ? a : b; }

// float -> int as the reference's device converts it: NaN -> 0, saturating (make_int3 / int() in Multiblock.cu:447, SharedTaskBuilder.cu:222)
__device__ __forceinline__ int cvt_rzi(float f)
{
    if (f != f) return 0;
    return (int)fminf(fmaxf(f, -2147483648.0f), 2147483520.0f);
}

__device__ __forceinline__ float sah_sa(const float* b)   // Common.cuh:293-297
{
    const float lx = b[3] - b[0], ly = b[4] - b[1], lz = b[5] - b[2];
    return 2.0f * (lx * ly + lx * lz + ly * lz);
}

__device__ __forceinline__ void sah_put_node(rt_node* n, const float* b, uint32_t child, uint32_t count, uint32_t type)
{
    uint4* o = reinterpret_cast<uint4*>(n);
    o[0] = make_uint4(__float_as_uint(b[0]), __float_as_uint(b[1]), __float_as_uint(b[2]), count << 29);
    o[1] = make_uint4(__float_as_uint(b[3]), __float_as_uint(b[4]), __float_as_uint(b[5]), (child & kIndexMask) | (type << 29));
}

__device__ __forceinline__ void load_box(const float* aabbs, uint32_t id, float* b)
{
    const float2* q = reinterpret_cast<const float2*>(aabbs + (size_t)id * 6);
    const float2 a = q[0], c = q[1], d = q[2];
    b[0] = a.x; b[1] = a.y; b[2] = c.x; b[3] = c.y; b[4] = d.x; b[5] = d.y;
}

// the leaf descriptor of one item (SharedTaskBuilder.cu:405-446): a triangle leaf, or -- in the top tree -- a cell,
// written with count 0 as a marker and completed by sah_patch_top_kernel once the cell's sub-root exists
__device__ __forceinline__ void sah_leaf_desc(const SahArgs& a, rt_node* out, uint32_t idv)
{
    const uint32_t id = idv;
    float b[6];
    load_box(a.aabbs, id, b);
    if (id < a.B) {
        const uint32_t lv = a.item_leaf[id];
        sah_put_node(out, b, lv & kLeafMask, (lv >> 31) ? 2u : 1u, RT_CHILD_TRI);
    } else {
        sah_put_node(out, b, id - a.B, 0u, RT_CHILD_BOX);
    }
}

__device__ __forceinline__ int sah_axis(const float* c)   // SelectAxis (SharedTaskBuilder.cu:197-204)
{
    const float lx = c[3] - c[0], ly = c[4] - c[1], lz = c[5] - c[2];
    return 2 * (lz > lx && lz > ly) + 1 * (ly > lx && ly >= lz);
}

// ---------------------------------------------------------------------------------------------
__global__ void sah_init_kernel(SahHeader* H, rt_node* nodes, uint32_t n)
{
    const uint32_t t = threadIdx.x;
    if (t < 6) { H->gp[t] = t < 3 ? kEmptyLo : kEmptyHi; H->gc[t] = t < 3 ? kEmptyLo : kEmptyHi; }
    if (t < 8) H->status[t] = (t == 1 || t == 2) ? n : 0u;   // [1] items, [2] leaf records (overwritten by pairs / splits)
    if (t == 0) H->small_count = 0;
    if (t < kSahCells) {
        H->cell_count[t] = 0; H->cell_start[t] = 0; H->cell_task[t] = kInactive;
        for (int k = 0; k < 6; k++) { H->cell_p[t][k] = k < 3 ? kEmptyLo : kEmptyHi; H->cell_c[t][k] = k < 3 ? kEmptyLo : kEmptyHi; }
    }
    for (uint32_t i = t; i < kSahMaxLevels; i += blockDim.x) H->level_count[i] = 0;
    for (uint32_t i = t; i < kBoundParts * 12; i += blockDim.x) (&H->part_bounds[0][0])[i] = (i % 6) < 3 ? kEmptyLo : kEmptyHi;
    for (uint32_t i = t; i < kCellParts * kSahCells * 12; i += blockDim.x) (&H->part_cell[0][0][0])[i] = (i % 6) < 3 ? kEmptyLo : kEmptyHi;
    // the top tree's slots [0, 128): whatever the build does not write is type None
    for (uint32_t i = t; i < 2 * kSahCells * 2; i += blockDim.x) reinterpret_cast<uint4*>(nodes)[i] = make_uint4(0, 0, 0, 0);
}

// Setup (Multiblock.cu:139-207): one thread per candidate (triangles 2k, 2k+1).  Leaf slot = input order.
__global__ __launch_bounds__(256) void sah_setup_kernel(const float* __restrict__ f, uint32_t n,
                                                        rt_triangle_pair* __restrict__ leaves, float* __restrict__ aabbs,
                                                        uint32_t* __restrict__ idsv, uint32_t* __restrict__ item_leaf, SahHeader* H,
                                                        const uint8_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ block_offsets)
{
    __shared__ int sb[12];
    __shared__ uint32_t ws[8];
    if (threadIdx.x < 12) sb[threadIdx.x] = (threadIdx.x % 6) < 3 ? kEmptyLo : kEmptyHi;
    __syncthreads();
    // a bounded number of workgroups walks the tiles: the scene bounds end in 12 same-address global atomics per
    // workgroup, and those serialise across the 8 L2s (about 50 ns each)
    const uint32_t ntiles = ((n + 1) / 2 + 255) / 256;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const uint32_t k = tile * 256 + threadIdx.x, tid = 2 * k;
    const bool live = tid < n, second = tid + 1 < n;
    const bool merge = live && flags && flags[k] != 0;
    const uint32_t valid = live ? 1u + ((second && !merge) ? 1u : 0u) : 0u;
    uint32_t slot = tid;
    if (flags) {
        uint32_t total;
        slot = block_offsets[tile] + block_excl_scan_u32<256>(valid, ws, &total);
    }
    int bnd[12];   // this candidate's contribution to the scene bounds (identity when not live)
#pragma unroll
    for (int j = 0; j < 12; j++) bnd[j] = (j % 6) < 3 ? kEmptyLo : kEmptyHi;
    float A[9], B[9], ab[6], bb[6];
    if (live) {
        load_tri9(f + (size_t)tid * 9, A);
        load_tri9(f + (size_t)(second ? tid + 1 : tid) * 9, B);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            ab[j] = fminf(fminf(A[j], A[3 + j]), A[6 + j]); ab[3 + j] = fmaxf(fmaxf(A[j], A[3 + j]), A[6 + j]);
            bb[j] = fminf(fminf(B[j], B[3 + j]), B[6 + j]); bb[3 + j] = fmaxf(fmaxf(B[j], B[3 + j]), B[6 + j]);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const float ac = (ab[j] + ab[3 + j]) * 0.5f, bc = (bb[j] + bb[3 + j]) * 0.5f;
            bnd[j] = min(float_to_ordered_int(ab[j]), float_to_ordered_int(bb[j]));
            bnd[3 + j] = max(float_to_ordered_int(ab[3 + j]), float_to_ordered_int(bb[3 + j]));
            bnd[6 + j] = min(float_to_ordered_int(ac), float_to_ordered_int(bc));
            bnd[9 + j] = max(float_to_ordered_int(ac), float_to_ordered_int(bc));
        }
    }
    // one LDS atomic per wave and value instead of 64 on the same address
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int r = (j % 6) < 3 ? wave_min_i32(bnd[j]) : wave_max_i32(bnd[j]);
        if ((threadIdx.x & 63) == 0) { if ((j % 6) < 3) atomicMin(&sb[j], r); else atomicMax(&sb[j], r); }
    }
    if (live) {
        uint4* out = reinterpret_cast<uint4*>(leaves + slot);
        float2* bo = reinterpret_cast<float2*>(aabbs + (size_t)slot * 6);
        if (merge) {
            // CreateTrianglePair (Pairing.cuh:60-77), as in the LBVH leaf kernel
            int ra = 0, rb = 0;
            can_form_pair(A, B, ra, rb);
            float r[9], v3[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                r[j] = ra == 1 ? A[6 + j] : (ra == 2 ? A[3 + j] : A[j]);
                r[3 + j] = ra == 1 ? A[j] : (ra == 2 ? A[6 + j] : A[3 + j]);
                r[6 + j] = ra == 1 ? A[3 + j] : (ra == 2 ? A[j] : A[6 + j]);
                v3[j] = rb == 2 ? B[j] : (rb == 1 ? B[3 + j] : B[6 + j]);
            }
            out[0] = make_uint4(__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), tid);
            out[1] = make_uint4(__float_as_uint(r[3]), __float_as_uint(r[4]), __float_as_uint(r[5]), tid + 1);
            out[2] = make_uint4(__float_as_uint(r[6]), __float_as_uint(r[7]), __float_as_uint(r[8]), (uint32_t)ra | ((uint32_t)rb << 16));
            out[3] = make_uint4(__float_as_uint(v3[0]), __float_as_uint(v3[1]), __float_as_uint(v3[2]), 0u);
            float u[6];
#pragma unroll
            for (int j = 0; j < 3; j++) { u[j] = fmin_ord(ab[j], bb[j]); u[3 + j] = fmax_ord(ab[3 + j], bb[3 + j]); }
            bo[0] = make_float2(u[0], u[1]); bo[1] = make_float2(u[2], u[3]); bo[2] = make_float2(u[4], u[5]);
            idsv[slot] = slot;
            item_leaf[slot] = slot | 0x80000000u;
        } else {
            out[0] = make_uint4(__float_as_uint(A[0]), __float_as_uint(A[1]), __float_as_uint(A[2]), tid);
            out[1] = make_uint4(__float_as_uint(A[3]), __float_as_uint(A[4]), __float_as_uint(A[5]), 0u);
            out[2] = make_uint4(__float_as_uint(A[6]), __float_as_uint(A[7]), __float_as_uint(A[8]), 0u);
            out[3] = make_uint4(__float_as_uint(A[6]), __float_as_uint(A[7]), __float_as_uint(A[8]), 0u);
            bo[0] = make_float2(ab[0], ab[1]); bo[1] = make_float2(ab[2], ab[3]); bo[2] = make_float2(ab[4], ab[5]);
            idsv[slot] = slot;
            item_leaf[slot] = slot;
            if (second) {
                out[4] = make_uint4(__float_as_uint(B[0]), __float_as_uint(B[1]), __float_as_uint(B[2]), tid + 1);
                out[5] = make_uint4(__float_as_uint(B[3]), __float_as_uint(B[4]), __float_as_uint(B[5]), 0u);
                out[6] = make_uint4(__float_as_uint(B[6]), __float_as_uint(B[7]), __float_as_uint(B[8]), 0u);
                out[7] = make_uint4(__float_as_uint(B[6]), __float_as_uint(B[7]), __float_as_uint(B[8]), 0u);
                bo[3] = make_float2(bb[0], bb[1]); bo[4] = make_float2(bb[2], bb[3]); bo[5] = make_float2(bb[4], bb[5]);
                idsv[slot + 1] = slot + 1;
                item_leaf[slot + 1] = slot + 1;
            }
        }
    }
    }   // tiles
    __syncthreads();
    if (threadIdx.x < 12) {
        int* g = &H->part_bounds[blockIdx.x % kBoundParts][threadIdx.x];
        if ((threadIdx.x % 6) < 3) atomicMin(g, sb[threadIdx.x]); else atomicMax(g, sb[threadIdx.x]);
    }
}

// ---- spatial splits: SetupSplits / SetupPairSplits (Multiblock.cu:209-425)
// A leaf whose box spans several cells of the 4 x 4 x 4 grid over the SCENE box becomes one reference per cell, its box
// clipped to the cell, while the running total of extra references stays below n / 5.  The reference takes that total
// from an atomic counter (arrival order); here it is the prefix sum in input order, so three passes over the
// candidates (2k, 2k+1): 0 = extra references wanted per workgroup, 1 = split decisions + references per workgroup,
// 2 = write leaf records and references.  Exclusive scans of the workgroup sums in between (pair_scan_kernel).
__device__ __forceinline__ void grid_cell(const float* p, const float* g, int* c)   // CalculateGridcell (Multiblock.cu:86-91)
{
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = min(3, max(0, cvt_rzi(floorf((p[k] - g[k]) * 4.0f / (g[3 + k] - g[k])))));
}
__device__ __forceinline__ void cell_bounds(const int* c, const float* g, float* out)   // CellToBounds (Multiblock.cu:93-102)
{
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float step = (g[3 + k] - g[k]) / 4.0f;
        out[k] = g[k] + (float)c[k] * step;
        out[3 + k] = g[k] + (float)(c[k] + 1) * step;
    }
}
__device__ __forceinline__ void box_intersection(const float* a, const float* b, float* out)   // Common.cuh:269-272
{
#pragma unroll
    for (int k = 0; k < 3; k++) { out[k] = fmax_ord(a[k], b[k]); out[3 + k] = fmin_ord(a[3 + k], b[3 + k]); }
}
__device__ __forceinline__ bool box_valid(const float* b) { return b[3] >= b[0] && b[4] >= b[1] && b[5] >= b[2]; }

struct SplitPassArgs {
    const float* tris;
    uint32_t n, thresh;
    const uint8_t* pair_flags;        // merge decision per candidate, or null (no pairs)
    const uint32_t* pair_offsets;     // leaf-record slot of each workgroup's first candidate (pairs)
    uint8_t* split_flags;             // bit s: leaf s of the candidate is split
    uint32_t* sums_a;                 // pass 0 out / pass 1 in (scanned): extra references wanted
    uint32_t* sums_b;                 // pass 1 out / pass 2 in (scanned): references written
    rt_triangle_pair* leaves;
    float* aabbs;
    uint32_t* idsv;
    uint32_t* item_leaf;
    SahHeader* H;
};

template <int PASS>
__global__ __launch_bounds__(256) void sah_split_pass_kernel(SplitPassArgs a)
{
    __shared__ uint32_t ws[8];
    __shared__ int sb[6];
    if (threadIdx.x < 6) sb[threadIdx.x] = threadIdx.x < 3 ? kEmptyLo : kEmptyHi;
    __syncthreads();
    float grid[6];
#pragma unroll
    for (int k = 0; k < 6; k++) grid[k] = ordered_int_to_float(a.H->gp[k]);
    int cmin[3] = {kEmptyLo, kEmptyLo, kEmptyLo}, cmax[3] = {kEmptyHi, kEmptyHi, kEmptyHi};   // centroid bounds (pass 2)
    const uint32_t ntiles = ((a.n + 1) / 2 + 255) / 256;
    for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint32_t k = tile * 256 + threadIdx.x, tid = 2 * k;
        const bool live = tid < a.n, second = tid + 1 < a.n;
        const bool merge = live && a.pair_flags && a.pair_flags[k] != 0;
        const uint32_t nleaf = live ? (merge ? 1u : 1u + (second ? 1u : 0u)) : 0u;
        float A[9], Bv[9], box[2][6];
        int lo[2][3], hi[2][3];
        uint32_t extra[2] = {0, 0};
        bool want[2] = {false, false};
        if (live) {
            load_tri9(a.tris + (size_t)tid * 9, A);
            load_tri9(a.tris + (size_t)(second ? tid + 1 : tid) * 9, Bv);
#pragma unroll
            for (int j = 0; j < 3; j++) {
                box[0][j] = fminf(fminf(A[j], A[3 + j]), A[6 + j]); box[0][3 + j] = fmaxf(fmaxf(A[j], A[3 + j]), A[6 + j]);
                box[1][j] = fminf(fminf(Bv[j], Bv[3 + j]), Bv[6 + j]); box[1][3 + j] = fmaxf(fmaxf(Bv[j], Bv[3 + j]), Bv[6 + j]);
            }
        }
        float ab[6], bb[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { ab[j] = box[0][j]; bb[j] = box[1][j]; }
        if (merge) {
#pragma unroll
            for (int j = 0; j < 3; j++) { box[0][j] = fmin_ord(ab[j], bb[j]); box[0][3 + j] = fmax_ord(ab[3 + j], bb[3 + j]); }
        }
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if ((uint32_t)s < nleaf) {
                grid_cell(&box[s][0], grid, lo[s]);
                grid_cell(&box[s][3], grid, hi[s]);
                want[s] = lo[s][0] != hi[s][0] || lo[s][1] != hi[s][1] || lo[s][2] != hi[s][2];
                extra[s] = want[s] ? (uint32_t)((hi[s][0] - lo[s][0] + 1) * (hi[s][1] - lo[s][1] + 1) * (hi[s][2] - lo[s][2] + 1) - 1) : 0u;
            }
        }
        uint32_t total;
        if (PASS == 0) {
            block_excl_scan_u32<256>(extra[0] + extra[1], ws, &total);
            if (threadIdx.x == 0) a.sums_a[tile] = total;
            continue;
        }
        // the extra_leaves counter before this candidate: every leaf that WANTS a split adds to it, granted or not
        const uint32_t before = a.sums_a[tile] + block_excl_scan_u32<256>(extra[0] + extra[1], ws, &total);
        bool split[2];
        split[0] = want[0] && before + extra[0] < a.thresh;
        split[1] = want[1] && before + extra[0] + extra[1] < a.thresh;
        uint32_t refs[2] = {0, 0};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if ((uint32_t)s >= nleaf) continue;
            refs[s] = 1;
            if (split[s]) {
                refs[s] = extra[s] + 1;
                if (merge) {   // cells neither triangle's box overlaps are dropped (Multiblock.cu:362-373)
                    refs[s] = 0;
                    int c[3];
                    for (c[2] = lo[s][2]; c[2] <= hi[s][2]; c[2]++)
                        for (c[1] = lo[s][1]; c[1] <= hi[s][1]; c[1]++)
                            for (c[0] = lo[s][0]; c[0] <= hi[s][0]; c[0]++) {
                                float cb[6], ia[6], ib[6];
                                cell_bounds(c, grid, cb);
                                box_intersection(ab, cb, ia);
                                box_intersection(bb, cb, ib);
                                refs[s] += (box_valid(ia) || box_valid(ib)) ? 1u : 0u;
                            }
                }
            }
        }
        if (PASS == 1) {
            if (live) a.split_flags[k] = (uint8_t)((split[0] ? 1 : 0) | (split[1] ? 2 : 0));
            block_excl_scan_u32<256>(refs[0] + refs[1], ws, &total);
            if (threadIdx.x == 0) a.sums_b[tile] = total;
            continue;
        }
        // ---- pass 2: leaf records and references
        uint32_t idx = a.sums_b[tile] + block_excl_scan_u32<256>(refs[0] + refs[1], ws, &total);
        uint32_t rec = tid;   // leaf-record slot: the triangle index, or the pair prefix sum
        if (a.pair_flags) rec = a.pair_offsets[tile] + block_excl_scan_u32<256>(nleaf, ws, &total);
        if (!live) continue;
        uint4* out = reinterpret_cast<uint4*>(a.leaves + rec);
        if (merge) {
            int ra = 0, rb = 0;
            can_form_pair(A, Bv, ra, rb);
            float r[9], v3[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                r[j] = ra == 1 ? A[6 + j] : (ra == 2 ? A[3 + j] : A[j]);
                r[3 + j] = ra == 1 ? A[j] : (ra == 2 ? A[6 + j] : A[3 + j]);
                r[6 + j] = ra == 1 ? A[3 + j] : (ra == 2 ? A[j] : A[6 + j]);
                v3[j] = rb == 2 ? Bv[j] : (rb == 1 ? Bv[3 + j] : Bv[6 + j]);
            }
            out[0] = make_uint4(__float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), tid);
            out[1] = make_uint4(__float_as_uint(r[3]), __float_as_uint(r[4]), __float_as_uint(r[5]), tid + 1);
            out[2] = make_uint4(__float_as_uint(r[6]), __float_as_uint(r[7]), __float_as_uint(r[8]), (uint32_t)ra | ((uint32_t)rb << 16));
            out[3] = make_uint4(__float_as_uint(v3[0]), __float_as_uint(v3[1]), __float_as_uint(v3[2]), 0u);
        } else {
            out[0] = make_uint4(__float_as_uint(A[0]), __float_as_uint(A[1]), __float_as_uint(A[2]), tid);
            out[1] = make_uint4(__float_as_uint(A[3]), __float_as_uint(A[4]), __float_as_uint(A[5]), 0u);
            out[2] = make_uint4(__float_as_uint(A[6]), __float_as_uint(A[7]), __float_as_uint(A[8]), 0u);
            out[3] = make_uint4(__float_as_uint(A[6]), __float_as_uint(A[7]), __float_as_uint(A[8]), 0u);
            if (second) {
                out[4] = make_uint4(__float_as_uint(Bv[0]), __float_as_uint(Bv[1]), __float_as_uint(Bv[2]), tid + 1);
                out[5] = make_uint4(__float_as_uint(Bv[3]), __float_as_uint(Bv[4]), __float_as_uint(Bv[5]), 0u);
                out[6] = make_uint4(__float_as_uint(Bv[6]), __float_as_uint(Bv[7]), __float_as_uint(Bv[8]), 0u);
                out[7] = make_uint4(__float_as_uint(Bv[6]), __float_as_uint(Bv[7]), __float_as_uint(Bv[8]), 0u);
            }
        }
        for (uint32_t s = 0; s < nleaf; s++) {
            const uint32_t leafv = (rec + s) | (merge ? 0x80000000u : 0u);
            const float* bx = s ? box[1] : box[0];
            const int* l3 = s ? lo[1] : lo[0];
            const int* h3 = s ? hi[1] : hi[0];
            const bool sp = s ? split[1] : split[0];
            auto emit = [&](const float* o) {
                float2* bo = reinterpret_cast<float2*>(a.aabbs + (size_t)idx * 6);
                bo[0] = make_float2(o[0], o[1]); bo[1] = make_float2(o[2], o[3]); bo[2] = make_float2(o[4], o[5]);
                a.idsv[idx] = idx;
                a.item_leaf[idx] = leafv;
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    const int ctr = float_to_ordered_int((o[j] + o[3 + j]) * 0.5f);
                    cmin[j] = min(cmin[j], ctr); cmax[j] = max(cmax[j], ctr);
                }
                idx++;
            };
            if (!sp) { emit(bx); continue; }
            int c[3];
            for (c[2] = l3[2]; c[2] <= h3[2]; c[2]++)
                for (c[1] = l3[1]; c[1] <= h3[1]; c[1]++)
                    for (c[0] = l3[0]; c[0] <= h3[0]; c[0]++) {
                        float cb[6], o[6];
                        cell_bounds(c, grid, cb);
                        if (merge) {
                            float ia[6], ib[6];
                            box_intersection(ab, cb, ia);
                            box_intersection(bb, cb, ib);
                            if (!box_valid(ia) && !box_valid(ib)) continue;
#pragma unroll
                            for (int j = 0; j < 3; j++) { o[j] = fmin_ord(ia[j], ib[j]); o[3 + j] = fmax_ord(ia[3 + j], ib[3 + j]); }
                        } else {
                            box_intersection(bx, cb, o);
                        }
                        emit(o);
                    }
        }
    }
    if (PASS == 2) {
        // centroid bounds of the references: registers -> wave -> LDS -> 6 global atomics per workgroup
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const int l = wave_min_i32(cmin[j]), h = wave_max_i32(cmax[j]);
            if ((threadIdx.x & 63) == 0) { atomicMin(&sb[j], l); atomicMax(&sb[3 + j], h); }
        }
        __syncthreads();
        if (threadIdx.x < 6) {
            int* g = &a.H->part_bounds[blockIdx.x % kBoundParts][6 + threadIdx.x];
            if (threadIdx.x < 3) atomicMin(g, sb[threadIdx.x]); else atomicMax(g, sb[threadIdx.x]);
        }
    }
}

// GridBlockCounts (Multiblock.cu:427-468): the grid cell of every leaf (the key of the distribution pass) and the
// per-cell primitive / centroid bounds.  The cell counts come out of the radix pass (its digit totals).
__global__ __launch_bounds__(256) void sah_grid_kernel(const float* __restrict__ aabbs, uint32_t n, const uint32_t* n_dev,
                                                       SahHeader* H, uint32_t* __restrict__ keys, int publish_gp)
{
    // fold the partial scene bounds of the previous kernel (every workgroup does; workgroup 0 publishes them)
    __shared__ int fold[12];
    if (threadIdx.x < 12) fold[threadIdx.x] = (threadIdx.x % 6) < 3 ? kEmptyLo : kEmptyHi;
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < kBoundParts * 12; j += 256) {
        const int v = (&H->part_bounds[0][0])[j];
        if ((j % 6) < 3) atomicMin(&fold[j % 12], v); else atomicMax(&fold[j % 12], v);
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 12) {
        if (threadIdx.x >= 6) H->gc[threadIdx.x - 6] = fold[threadIdx.x];
        else if (publish_gp) H->gp[threadIdx.x] = fold[threadIdx.x];
    }
    const uint32_t L = n_dev ? *n_dev : n;
    // cw[cell]: primitive box [0, 6), centroid box [6, 12), the "max" words kept complemented (max x = ~min ~x on the ordered
    // ints) so that all twelve updates are the same LDS instruction
    __shared__ int cw[kSahCells][12];
    __shared__ uint32_t cnt[kSahCells];
    for (uint32_t j = threadIdx.x; j < kSahCells * 12; j += 256) (&cw[0][0])[j] = kEmptyLo;   // (~kEmptyHi == kEmptyLo)
    if (threadIdx.x < kSahCells) cnt[threadIdx.x] = 0;
    __syncthreads();
    float glo[3], ghi[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { glo[k] = ordered_int_to_float(fold[6 + k]); ghi[k] = ordered_int_to_float(fold[9 + k]); }
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < ((L + 255) & ~255u); i += gridDim.x * 256) {
    int cell = -1;
    int v[12];
#pragma unroll
    for (int k = 0; k < 12; k++) v[k] = kEmptyLo;
    if (i < L) {
        float b[6];
        load_box(aabbs, i, b);
        const float epsilon = 1.1920929e-7f;
        const float gscale = 4 * (1 - epsilon);
        int q[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float ctr = (b[k] + b[3 + k]) * 0.5f;
            q[k] = min(3, max(0, cvt_rzi((ctr - glo[k]) * gscale / (ghi[k] - glo[k]))));
        }
        cell = q[0] + q[1] * 4 + q[2] * 16;
        keys[i] = (uint32_t)cell;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int ctr = float_to_ordered_int((b[k] + b[3 + k]) * 0.5f);
            v[k] = float_to_ordered_int(b[k]); v[3 + k] = ~float_to_ordered_int(b[3 + k]);
            v[6 + k] = ctr; v[9 + k] = ~ctr;
        }
    }
    // consecutive leaves mostly share a cell: when the whole wave agrees, reduce in registers and touch LDS once
    const int c0 = __builtin_amdgcn_readfirstlane(cell);
    if (c0 >= 0 && __builtin_amdgcn_ballot_w64(cell == c0) == ~0ull) {
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const int r = wave_min_i32(v[k]);
            if ((threadIdx.x & 63) == 0) atomicMin(&cw[c0][k], r);
        }
        if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[c0], 64u);
    } else if (cell >= 0) {
        // the wave's lanes fall into a handful of cells (on the bench mesh: four, by height): in a fixed word order every
        // instruction would queue up to 64 lanes on those few words.  Lane i walks the twelve words starting at word i mod 12
        // (the values rotated to match by a four-stage barrel shifter, as in sah_bin_kernel): the queues shrink 12x
        const uint32_t r = threadIdx.x % 12u;
#pragma unroll
        for (int st = 0; st < 4; st++) {
            const bool on = (r >> st) & 1u;
            int t[12];
#pragma unroll
            for (int k = 0; k < 12; k++) t[k] = v[(k + (1 << st)) % 12];
#pragma unroll
            for (int k = 0; k < 12; k++) v[k] = on ? t[k] : v[k];
        }
        uint32_t w = r;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            atomicMin(&cw[cell][w], v[k]);
            w = w == 11u ? 0u : w + 1u;
        }
        atomicAdd(&cnt[cell], 1u);
    }
    }   // tiles
    __syncthreads();
    if (threadIdx.x < kSahCells && cnt[threadIdx.x]) {
        const uint32_t c = threadIdx.x;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            int* g = &H->part_cell[blockIdx.x % kCellParts][c][0];
            atomicMin(&g[k], cw[c][k]); atomicMax(&g[3 + k], ~cw[c][3 + k]);
            atomicMin(&g[6 + k], cw[c][6 + k]); atomicMax(&g[9 + k], ~cw[c][9 + k]);
        }
    }
}

__device__ __forceinline__ void sah_init_bins(int* bins, uint32_t task, uint32_t lane, uint32_t stride)
{
    for (uint32_t j = lane; j < 8 * kBinWords; j += stride) {
        const uint32_t w = j % kBinWords;
        bins[(size_t)task * 8 * kBinWords + j] = w == 12 ? 0 : kEmptyLo;   // max words are kept complemented (~kEmptyHi)
    }
}

// GridBlockScan + the task set-up of SharedTaskBuild::Initialise (Multiblock.cu:470-505, SharedTaskBuilder.cu:93-135):
// cell starts, cell boxes as the top tree's items, the root task of every non-empty cell and of the top tree.
__global__ __launch_bounds__(128) void sah_roots_kernel(SahArgs a, const uint32_t* __restrict__ digit_total, float* aabbs_w)
{
    __shared__ uint32_t ws[4];
    SahHeader* H = a.H;
    const uint32_t t = threadIdx.x;
    const uint32_t cnt = t < kSahCells ? digit_total[t] : 0u;
    uint32_t total, K;
    const uint32_t start = block_excl_scan_u32<128>(cnt, ws, &total);
    const uint32_t k = block_excl_scan_u32<128>(cnt ? 1u : 0u, ws, &K);
    if (t < kSahCells) {
        for (uint32_t q = 0; q < kCellParts; q++)
            for (int j = 0; j < 6; j++) {
                const int pv = H->part_cell[q][t][j], cv = H->part_cell[q][t][6 + j];
                if (j < 3) { H->cell_p[t][j] = min(H->cell_p[t][j], pv); H->cell_c[t][j] = min(H->cell_c[t][j], cv); }
                else { H->cell_p[t][j] = max(H->cell_p[t][j], pv); H->cell_c[t][j] = max(H->cell_c[t][j], cv); }
            }
        H->cell_count[t] = cnt;
        H->cell_start[t] = start;
        float pb[6], cb[6];
        for (int j = 0; j < 6; j++) { pb[j] = ordered_int_to_float(H->cell_p[t][j]); cb[j] = ordered_int_to_float(H->cell_c[t][j]); }
        for (int j = 0; j < 6; j++) aabbs_w[(size_t)(a.B + t) * 6 + j] = pb[j];
        uint32_t task = kInactive;
        if (cnt) {
            a.ids[0][a.B + k] = a.B + t;
            const uint32_t parent = 2 * kSahCells + 2 * start;
            reinterpret_cast<uint4*>(a.nodes + parent + 1)[0] = make_uint4(0, 0, 0, 0);   // the root's sibling slot: None
            reinterpret_cast<uint4*>(a.nodes + parent + 1)[1] = make_uint4(0, 0, 0, 0);
            if (cnt > kSahSmall) {
                task = atomicAdd(&H->level_count[0], 1u);
                SahTask T;
                for (int j = 0; j < 6; j++) { T.c[j] = cb[j]; T.p[j] = pb[j]; }
                T.start = start; T.end = start + cnt; T.parent_idx = parent; T.flags = 0;
                a.tasks[0][task] = T;
                sah_init_bins(a.bins[0], task, 0, 1);
            } else {
                const uint32_t s = atomicAdd(&H->small_count, 1u);
                a.small[s] = SahSmall{start, start + cnt, parent, 0u};
            }
        }
        H->cell_task[t] = task;
    }
    __syncthreads();
    if (t == 64) {   // the top tree: items = the K non-empty cells at positions [n, n + K), root descriptor = slot 0
        uint32_t task = kInactive;
        if (K > kSahSmall) {
            task = atomicAdd(&H->level_count[0], 1u);
            SahTask T;
            for (int j = 0; j < 6; j++) { T.c[j] = ordered_int_to_float(H->gc[j]); T.p[j] = ordered_int_to_float(H->gp[j]); }
            T.start = a.B; T.end = a.B + K; T.parent_idx = 0; T.flags = 1;
            a.tasks[0][task] = T;
            sah_init_bins(a.bins[0], task, 0, 1);
        } else if (K) {
            const uint32_t s = atomicAdd(&H->small_count, 1u);
            a.small[s] = SahSmall{a.B, a.B + K, 0u, 1u | 4u};
        }
        for (uint32_t j = 0; j < kSahCells; j++) a.task_of[0][a.B + j] = j < K ? task : kInactive;
    }
}

// position -> root task of its cell (the sorted keys of the distribution pass are the cell ids)
__global__ __launch_bounds__(256) void sah_assign_kernel(SahArgs a, const uint32_t* n_dev)
{
    const uint32_t L = n_dev ? *n_dev : a.B;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.B) return;
    a.task_of[0][i] = i < L ? a.H->cell_task[a.task_of[0][i] & (kSahCells - 1)] : kInactive;
}

// ---------------------------------------------------------------------------------------------
// runs of equal task id inside a chunk: local index of this position's run, number of runs
// (its barriers order LDS only: the level kernels issue their global loads BEFORE it and let them fly across the scan -- a
// __syncthreads() would wait for them)
__device__ __forceinline__ uint32_t sah_local_runs(uint32_t t, uint32_t t_prev, uint32_t* ws, uint32_t* nloc)
{
    const uint32_t flag = (threadIdx.x > 0 && t != t_prev) ? 1u : 0u;
    uint32_t total;
    const uint32_t ex = block_excl_scan_lds<256>(flag, ws, &total);
    *nloc = total + 1;
    return ex + flag;
}

// BinCentroids (SharedTaskBuilder.cu:206-264) for every task alive in level `lvl`
__global__ __launch_bounds__(256) void sah_bin_kernel(SahArgs a, uint32_t lvl)
{
    if (a.H->level_count[lvl] == 0) return;
    const uint32_t cur = lvl & 1;
    __shared__ int lbins[kSahMaxLocal][8][kBinWords];
    __shared__ uint32_t ltask[kSahMaxLocal], lstate[kSahMaxLocal];
    __shared__ uint32_t ws[8];
    const uint32_t chunk = blockIdx.x, pos = chunk * kSahChunk + threadIdx.x;
    const uint32_t t = pos < a.M ? a.task_of[cur][pos] : kInactive;
    const uint32_t t_prev = (threadIdx.x > 0 && pos - 1 < a.M) ? a.task_of[cur][pos - 1] : kInactive;
    for (uint32_t j = threadIdx.x; j < kSahMaxLocal * 8 * kBinWords; j += 256)
        (&lbins[0][0][0])[j] = (j % kBinWords) == 12 ? 0 : kEmptyLo;   // (~kEmptyHi == kEmptyLo: "max" words are complemented)
    uint32_t nloc;
    const uint32_t local = sah_local_runs(t, t_prev, ws, &nloc);   // ends with a barrier
    if (nloc > kSahMaxLocal) {
        if (threadIdx.x == 0) atomicOr(&a.H->status[0], kSahErrLocals);
        return;
    }
    if (threadIdx.x == 0 || t != t_prev) ltask[local] = t;
    if (t != kInactive) {
        const SahTask* T = &a.tasks[cur][t];
        float c[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c[k] = T->c[k];
        if (!(sah_sa(c) <= 0.0f)) {
            const int axis = sah_axis(c);
            const float epsilon = 1.1920929e-7f;
            const float cmin = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
            const float cmax = axis == 0 ? c[3] : (axis == 1 ? c[4] : c[5]);
            const float k1 = 8 * (1 - epsilon) / (cmax - cmin);
            const uint32_t id = a.ids[cur][pos];
            float b[6];
            load_box(a.aabbs, id, b);
            float ctr[3];
#pragma unroll
            for (int k = 0; k < 3; k++) ctr[k] = (b[k] + b[3 + k]) * 0.5f;
            const float ca = axis == 0 ? ctr[0] : (axis == 1 ? ctr[1] : ctr[2]);
            const int bin = min(7, max(0, cvt_rzi(k1 * (ca - cmin))));   // the reference aborts the build on an out-of-range bin
            a.binof[pos] = (uint8_t)bin;
            int* lb = &lbins[local][bin][0];
            // 12 box words per item.  Lanes that share a (task, bin) would queue on the same LDS word (about 8 cycles
            // per lane); the "max" words are kept complemented (max x = ~min ~x on the ordered ints) so that all 12
            // are the same instruction, and lane i walks them starting at word i mod 12: same-address queues shrink 12x
            int val[12];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                val[k] = float_to_ordered_int(b[k]);
                val[3 + k] = ~float_to_ordered_int(b[3 + k]);
                val[6 + k] = float_to_ordered_int(ctr[k]);
                val[9 + k] = ~float_to_ordered_int(ctr[k]);
            }
            // rotate val[] left by r = lane mod 12 (val[k] <- val[(k + r) mod 12]) with a four-stage barrel shifter -- by 1, 2,
            // 4, 8 on the bits of r: 48 selects instead of the 132 of a select chain per word
            const uint32_t r = threadIdx.x % 12u;
#pragma unroll
            for (int st = 0; st < 4; st++) {
                const bool on = (r >> st) & 1u;
                int t[12];
#pragma unroll
                for (int k = 0; k < 12; k++) t[k] = val[(k + (1 << st)) % 12];
#pragma unroll
                for (int k = 0; k < 12; k++) val[k] = on ? t[k] : val[k];
            }
            uint32_t w = r;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                atomicMin(&lb[w], val[k]);
                w = w == 11u ? 0u : w + 1u;
            }
            // the count: one atomic per (task, bin) group of the wave instead of one per lane
            uint64_t m = __builtin_amdgcn_ballot_w64(true);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const bool bit = (bin >> q) & 1;
                const uint64_t bal = __builtin_amdgcn_ballot_w64(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t l0 = __builtin_amdgcn_readfirstlane(local);
            const uint64_t same0 = __builtin_amdgcn_ballot_w64(local == l0);
            m &= local == l0 ? same0 : ~same0;   // (a wave holds at most a few runs; lanes of other runs pair up the same way)
            bool lead = (threadIdx.x & 63) == (uint32_t)(__ffsll((unsigned long long)m) - 1);
            uint32_t add = (uint32_t)__popcll(m);
            if (local != l0) {   // runs beyond the first two in a wave: fall back to per-lane counting
                const uint32_t l1 = __builtin_amdgcn_readfirstlane(local);
                if (__builtin_amdgcn_ballot_w64(local == l1) != __builtin_amdgcn_ballot_w64(true)) { lead = true; add = 1; }
            }
            if (lead) atomicAdd(&lb[12], (int)add);
        }
    }
    __syncthreads();
    // ---- flush.  Global bins keep the LDS form (max words complemented).  A task that lies inside this chunk owns its
    // bins: plain stores.  A run continuing from the left or to the right shares them with other chunks: atomics --
    // one LANE per word, so a run costs two atomic wave-instructions instead of 13 per wave (a CU issues about one
    // atomic wave-instruction per 50 ns).
    if (threadIdx.x < nloc) {
        const uint32_t tl = ltask[threadIdx.x];
        uint32_t st = 2;   // inactive
        if (tl != kInactive) {
            const SahTask* T = &a.tasks[cur][tl];
            st = (T->start >= chunk * kSahChunk && T->end <= (chunk + 1) * kSahChunk) ? 1u : 0u;
        }
        lstate[threadIdx.x] = st;
    }
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < nloc * 8 * kBinWords; j += 256) {
        const uint32_t l = j / (8 * kBinWords), w = j % (8 * kBinWords);
        if (lstate[l] == 1) a.bins[cur][(size_t)ltask[l] * 8 * kBinWords + w] = (&lbins[l][0][0])[w];
    }
    {
        const uint32_t which = threadIdx.x >> 7, j = threadIdx.x & 127;
        const uint32_t l = which ? nloc - 1 : 0;
        if (!(which && nloc == 1) && j < 8 * kBinWords && lstate[l] == 0) {
            const uint32_t bin = j / kBinWords, w = j % kBinWords;
            if (lbins[l][bin][12] > 0) {
                int* g = a.bins[cur] + (size_t)ltask[l] * 8 * kBinWords + j;
                if (w == 12) atomicAdd(g, lbins[l][bin][12]); else atomicMin(g, lbins[l][bin][w]);
            }
        }
    }
    // the bin histogram of the first and of the last run: what a task spanning several chunks needs for its partition
    if (threadIdx.x < 8) a.chunk_hist[(size_t)chunk * 16 + threadIdx.x] = (uint32_t)lbins[0][threadIdx.x][12];
    else if (threadIdx.x < 16) a.chunk_hist[(size_t)chunk * 16 + threadIdx.x] = (uint32_t)lbins[nloc - 1][threadIdx.x - 8][12];
}

__device__ __forceinline__ void ibox_merge(int* b, const int* o)
{
#pragma unroll
    for (int k = 0; k < 3; k++) { b[k] = min(b[k], o[k]); b[3 + k] = max(b[3 + k], o[3 + k]); }
}
__device__ __forceinline__ void ibox_to_float(const int* b, float* f)
{
#pragma unroll
    for (int k = 0; k < 6; k++) f[k] = ordered_int_to_float(b[k]);
}

// SelectPlane (SharedTaskBuilder.cu:297-350) on the 8 bins of one task (g: [8][kBinWords] ordered ints, max words NOT
// complemented): the binned split with the lowest SAH score (sweep right -> left, strict <, both sides non-empty: the
// highest plane wins ties) -> kind 1, plane, mid and the children's boxes cb[side][p box 6, c box 6]; no such plane:
// kind, plane, mid and cb stay as they were (the caller's object-median split).  One thread; shared by the level loop
// (sah_split_kernel) and the straggler kernel (sah_finish_kernel) so that both pick bit-identical planes.
__device__ __forceinline__ void sah_select_plane(const int* g, uint32_t start, uint32_t& kind, uint32_t& plane, uint32_t& mid, int (*cb)[12])
{
    // prefix left -> right: surface area and count of bins [0, i]
    float sa_l[7];
    uint32_t ln[7];
    {
        int run[6] = {kEmptyLo, kEmptyLo, kEmptyLo, kEmptyHi, kEmptyHi, kEmptyHi};
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            ibox_merge(run, g + i * kBinWords);
            c += (uint32_t)g[i * kBinWords + 12];
            float f[6];
            ibox_to_float(run, f);
            sa_l[i] = sah_sa(f);
            ln[i] = c;
        }
    }
    int run[6];
#pragma unroll
    for (int k = 0; k < 6; k++) run[k] = g[7 * kBinWords + k];
    uint32_t rn = (uint32_t)g[7 * kBinWords + 12];
    float best = 3.402823466e+38f;
    int pl = -1;
#pragma unroll
    for (int i = 6; i >= 0; i--) {
        float f[6];
        ibox_to_float(run, f);
        const float score = sa_l[i] * (float)ln[i] + sah_sa(f) * (float)rn;
        if (score < best && ln[i] && rn) { best = score; pl = i; }
        ibox_merge(run, g + i * kBinWords);
        rn += (uint32_t)g[i * kBinWords + 12];
    }
    if (pl >= 0) {
        kind = 1; plane = (uint32_t)pl; mid = start + ln[pl];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i <= pl) { ibox_merge(&cb[0][0], g + i * kBinWords); ibox_merge(&cb[0][6], g + i * kBinWords + 6); }
            else { ibox_merge(&cb[1][0], g + i * kBinWords); ibox_merge(&cb[1][6], g + i * kBinWords + 6); }
        }
    }
}

// wave-aggregated counter bump: lanes with `want` get consecutive values, one atomic per wave
__device__ __forceinline__ uint32_t wave_alloc(uint32_t* counter, bool want, uint32_t lane)
{
    const uint64_t m = __builtin_amdgcn_ballot_w64(want);
    if (m == 0) return 0;
    uint32_t base = 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    return base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// one THREAD per task: SelectPlane (SharedTaskBuilder.cu:297-350), parent descriptor, children (RunTask after the
// bins are known, :521-606).  The two parts that scale with the size of a task -- the child boxes of an object-median
// split (:465-510) and the per-chunk "goes left" prefix of a task spanning several chunks -- are done by the whole
// wave, one such task at a time.
constexpr uint32_t kSplitWaves = 8;

__global__ __launch_bounds__(kSplitWaves * 64) void sah_split_kernel(SahArgs a, uint32_t lvl)
{
    const uint32_t ntask = a.H->level_count[lvl];
    // tasks per workgroup: 64 when the level is wide; 8 while it is narrow (the first levels, where every task spans
    // many chunks and the per-task wave work below would otherwise queue 8 deep in ONE workgroup)
    const uint32_t tpb = ntask > 2048 ? 64u : 8u;
    if (blockIdx.x * tpb >= ntask) return;
    const uint32_t cur = lvl & 1, nxt = cur ^ 1;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ uint32_t pf_c0[64], pf_c1[64], pf_plane[64];
    __shared__ uint32_t pf_n, init_n;
    __shared__ uint32_t init_ids[128];
    // the bins of this workgroup's 64 tasks are one contiguous 26 KB block: read it coalesced, keep it in LDS with an
    // odd row stride (a thread per task reading its row straight from memory touches 64 cache lines per load)
    constexpr uint32_t kRow = 8 * kBinWords + 1;
    __shared__ int sbins[64 * kRow];
    if (threadIdx.x == 0) { pf_n = 0; init_n = 0; }
    // (the task record is loaded while the bins are staged: both depend on the task index only)
    const uint32_t w = blockIdx.x * tpb + lane;
    const bool valid = wave == 0 && lane < tpb && w < ntask;
    SahTask T = {};
    if (valid) T = a.tasks[cur][w];
    {
        const uint32_t first = blockIdx.x * tpb;
        const uint32_t ntk = min(tpb, ntask - first);
        const int* src = a.bins[cur] + (size_t)first * 8 * kBinWords;
        for (uint32_t j = threadIdx.x; j < ntk * 8 * kBinWords; j += kSplitWaves * 64)
        {
            const uint32_t wd = (j % (8 * kBinWords)) % kBinWords;
            const int v = src[j];
            sbins[(j / (8 * kBinWords)) * kRow + (j % (8 * kBinWords))] = (wd < 12 && (wd % 6) >= 3) ? ~v : v;   // max words are stored complemented
        }
    }
    __syncthreads();
    if (wave == 0) {
    const uint32_t count = T.end - T.start;
    const int bias = (T.flags & 1u) ? -2 * (int)a.B : (int)(2 * kSahCells);

    int cb[2][12];   // child boxes, ordered ints: [side][p box 6, c box 6]
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int k = 0; k < 12; k++) cb[s][k] = (k % 6) < 3 ? kEmptyLo : kEmptyHi;
    uint32_t mid = 0, kind = 2, plane = 0;
    if (valid && !(sah_sa(T.c) <= 0.0f)) sah_select_plane(&sbins[lane * kRow], T.start, kind, plane, mid, cb);
    if (valid && kind == 2) mid = T.start + (count >> 1);
    // object split at the midpoint: child boxes by a wave reduction over the items, one task at a time
    for (uint64_t todo = __builtin_amdgcn_ballot_w64(valid && kind == 2); todo; todo &= todo - 1) {
        const int src = __ffsll((unsigned long long)todo) - 1;
        const uint32_t ts = __shfl(T.start, src, 64), te = __shfl(T.end, src, 64), tm = __shfl(mid, src, 64);
        int v[2][12];
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int k = 0; k < 12; k++) v[s][k] = (k % 6) < 3 ? kEmptyLo : kEmptyHi;
        for (uint32_t i = ts + lane; i < te; i += 64) {
            float b[6];
            load_box(a.aabbs, a.ids[cur][i], b);
            const bool right = i >= tm;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int ctr = float_to_ordered_int((b[3 + k] + b[k]) * 0.5f);
                const int lo = float_to_ordered_int(b[k]), hi = float_to_ordered_int(b[3 + k]);
                if (right) { v[1][k] = min(v[1][k], lo); v[1][3 + k] = max(v[1][3 + k], hi); v[1][6 + k] = min(v[1][6 + k], ctr); v[1][9 + k] = max(v[1][9 + k], ctr); }
                else { v[0][k] = min(v[0][k], lo); v[0][3 + k] = max(v[0][3 + k], hi); v[0][6 + k] = min(v[0][6 + k], ctr); v[0][9 + k] = max(v[0][9 + k], ctr); }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const int r = (k % 6) < 3 ? wave_min_i32(v[s][k]) : wave_max_i32(v[s][k]);
                if ((int)lane == src) cb[s][k] = r;
            }
    }
    const uint32_t child_index = (uint32_t)(bias + 2 * (int)mid);
    uint32_t cid[2] = {kInactive, kInactive};
    {
        // ids of the children: big ones join the next level, small ones the small-task list.  One atomic per counter and
        // wave (issued back to back), ranks from ballots.
        const bool bigL = valid && (mid - T.start) > kSahSmall, bigR = valid && (T.end - mid) > kSahSmall;
        const uint64_t mbl = __builtin_amdgcn_ballot_w64(bigL), mbr = __builtin_amdgcn_ballot_w64(bigR);
        const uint64_t msl = __builtin_amdgcn_ballot_w64(valid && !bigL), msr = __builtin_amdgcn_ballot_w64(valid && !bigR);
        uint32_t base_big = 0, base_small = 0;
        if (lane == 0) {
            const uint32_t nb = (uint32_t)(__popcll(mbl) + __popcll(mbr)), ns = (uint32_t)(__popcll(msl) + __popcll(msr));
            if (nb) base_big = atomicAdd(&a.H->level_count[lvl + 1], nb);
            if (ns) base_small = atomicAdd(&a.H->small_count, ns);
        }
        base_big = __shfl(base_big, 0, 64);
        base_small = __shfl(base_small, 0, 64);
        const uint64_t lt = (1ull << lane) - 1ull;
        const uint32_t idb[2] = {base_big + (uint32_t)__popcll(mbl & lt), base_big + (uint32_t)__popcll(mbl) + (uint32_t)__popcll(mbr & lt)};
        const uint32_t ids[2] = {base_small + (uint32_t)__popcll(msl & lt), base_small + (uint32_t)__popcll(msl) + (uint32_t)__popcll(msr & lt)};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const uint32_t cs = s ? mid : T.start, ce = s ? T.end : mid;
            const bool big = s ? bigR : bigL;
            if (big) {
                SahTask C;
#pragma unroll
                for (int k = 0; k < 6; k++) { C.p[k] = ordered_int_to_float(cb[s][k]); C.c[k] = ordered_int_to_float(cb[s][6 + k]); }
                C.start = cs; C.end = ce; C.parent_idx = child_index + s; C.flags = T.flags;
                a.tasks[nxt][idb[s]] = C;
                cid[s] = idb[s];
                init_ids[atomicAdd(&init_n, 1u)] = idb[s];     // its bins are reset by the whole workgroup below
            } else if (valid) {
                a.small[ids[s]] = SahSmall{cs, ce, child_index + (uint32_t)s, (T.flags & 1u) | (nxt << 1)};
            }
        }
    }
    if (valid) {
        sah_put_node(a.nodes + T.parent_idx, T.p, child_index, 2u, RT_CHILD_BOX);
        SahSplit S;
        S.kind = kind; S.plane = plane; S.mid = mid; S.left_id = cid[0]; S.right_id = cid[1];
        S.pad[0] = S.pad[1] = S.pad[2] = 0;
        a.splits[w] = S;
    }
    // tasks that span several chunks need the per-chunk "goes left" prefix: queued for the whole workgroup
    const uint32_t c0 = T.start / kSahChunk, c1 = valid ? (T.end - 1) / kSahChunk : 0;
    if (valid && kind == 1 && c1 > c0) {
        const uint32_t q = atomicAdd(&pf_n, 1u);
        pf_c0[q] = c0; pf_c1[q] = c1; pf_plane[q] = plane;
    }
    }   // wave 0
    __syncthreads();
    // empty bins for the children that join the next level
    for (uint32_t j = threadIdx.x; j < init_n * 8 * kBinWords; j += kSplitWaves * 64) {
        const uint32_t wd = j % kBinWords;
        a.bins[nxt][(size_t)init_ids[j / (8 * kBinWords)] * 8 * kBinWords + (j % (8 * kBinWords))] = wd == 12 ? 0 : kEmptyLo;
    }
    // "goes left" prefix per chunk (stable partition across workgroups).  Long tasks: one wave per task, a wave scan over
    // its chunks.  Short ones (<= 16 chunks, the bulk at the deeper levels): one thread per (task, chunk) entry, each
    // summing the few chunks before it -- no serial chain of dependent loads per task.
    for (uint32_t q = wave; q < pf_n; q += kSplitWaves) {
        const uint32_t k0 = pf_c0[q], k1 = pf_c1[q], pl = pf_plane[q];
        if (k1 - k0 <= 16) continue;
        uint32_t running = 0;
        for (uint32_t base = k0; base <= k1; base += 64) {
            const uint32_t c = base + lane;
            uint32_t v = 0;
            if (c <= k1) {
                const uint32_t* h = a.chunk_hist + (size_t)c * 16 + (c == k0 ? 8 : 0);
                for (uint32_t b = 0; b <= pl; b++) v += h[b];
            }
            const uint32_t incl = wave_incl_scan_u32(v, (int)lane);
            if (c <= k1 && c > k0) a.chunk_prefix[c] = running + incl - v;
            running += __shfl(incl, 63, 64);
        }
    }
    for (uint32_t j = threadIdx.x; j < pf_n * 16; j += kSplitWaves * 64) {
        const uint32_t q = j >> 4, e = (j & 15) + 1;          // entry e: chunk k0 + e
        const uint32_t k0 = pf_c0[q], k1 = pf_c1[q], pl = pf_plane[q];
        if (k1 - k0 > 16 || k0 + e > k1) continue;
        uint32_t sum = 0;
        for (uint32_t c = k0; c < k0 + e; c++) {
            const uint32_t* h = a.chunk_hist + (size_t)c * 16 + (c == k0 ? 8 : 0);
            for (uint32_t b = 0; b <= pl; b++) sum += h[b];
        }
        a.chunk_prefix[k0 + e] = sum;
    }
}

// PartitionIds (SharedTaskBuilder.cu:352-380), stable: ids and the children's task ids go to the other buffer
__global__ __launch_bounds__(256) void sah_partition_kernel(SahArgs a, uint32_t lvl)
{
    if (a.H->level_count[lvl] == 0) return;
    const uint32_t cur = lvl & 1, nxt = cur ^ 1;
    __shared__ uint32_t lfirst[kSahMaxLocal];
    __shared__ uint32_t ws[8];
    const uint32_t chunk = blockIdx.x, pos = chunk * kSahChunk + threadIdx.x;
    const uint32_t t = pos < a.M ? a.task_of[cur][pos] : kInactive;
    const uint32_t t_prev = (threadIdx.x > 0 && pos - 1 < a.M) ? a.task_of[cur][pos - 1] : kInactive;
    // all of the position's global loads up front (the scans below wait for LDS only)
    const uint32_t id = pos < a.M ? a.ids[cur][pos] : 0u;
    const uint32_t bin = pos < a.M ? a.binof[pos] : 0u;
    const uint32_t cpre = a.chunk_prefix[chunk];
    SahSplit S = {};
    uint32_t start = 0;
    if (t != kInactive) {
        S = a.splits[t];
        start = a.tasks[cur][t].start;
    }
    uint32_t nloc;
    const uint32_t local = sah_local_runs(t, t_prev, ws, &nloc);
    if (nloc > kSahMaxLocal) return;   // reported by sah_bin_kernel
    const bool left = t != kInactive && (S.kind == 1 ? (bin <= S.plane) : (pos < S.mid));
    uint32_t total;
    const uint32_t ex = block_excl_scan_lds<256>((t != kInactive && S.kind == 1 && left) ? 1u : 0u, ws, &total);
    if (threadIdx.x == 0 || t != t_prev) lfirst[local] = ex;
    lds_barrier();
    if (pos >= a.M) return;
    if (t == kInactive) { a.task_of[nxt][pos] = kInactive; return; }
    uint32_t dest = pos;
    if (S.kind == 1) {
        const uint32_t before = start < chunk * kSahChunk ? cpre : 0u;
        const uint32_t lr = before + (ex - lfirst[local]);
        dest = left ? start + lr : S.mid + ((pos - start) - lr);
    }
    a.ids[nxt][dest] = id;
    a.task_of[nxt][dest] = left ? S.left_id : S.right_id;
}

// ---------------------------------------------------------------------------------------------
// one WAVE per small task (<= 64 items): the whole sub-tree, level by level, every sub-task of a level at once.
// (The reference finishes small tasks with one thread each, PerInstanceRunTask, SharedTaskBuilder.cu:742-907.)
// Lane l holds the item at position start + l (id and box in registers); a sub-task is a lane range [s, e).  Per level:
// bins of every sub-task in LDS (integer atomics), plane selection by the sub-task's first lane, child boxes by LDS
// atomics, stable partition with ballots, items moved to their new lanes with ds_permute.
constexpr uint32_t kSmallSegs = 22;   // sub-tasks with >= 3 items alive in one level (<= 64 / 3)

// Word-major LDS tables: the word index of every access is a compile-time constant, so clearing a table is a few stores of
// immediates (the item-major layout of round 2 cleared with `(j % 6) < 3 ? lo : hi` -- an integer modulo per store, ~200 of
// the ~970 vector instructions a level cost; the kernel is bound by instruction issue, profiles/r03_sah_small_experiments.txt)
constexpr uint32_t kSmallCol = 65, kSmallBinRow = kSmallSegs * 8 + 1;   // 64 columns / 176 cells + 1
struct SmallSmem {
    // (row strides odd in banks: the twelve words of one column -- what the lanes of a sub-task update together, each starting
    //  at a different word -- fall into twelve different banks; with strides of 64 and 176 dwords they shared one or two)
    int sbox[2][12][kSmallCol];       // [buffer][p box 6, c box 6][first lane of the sub-task], ordered ints, "max" words complemented
    int bins[7][kSmallBinRow];        // [primitive box 6, count][sub-task * 8 + bin]
    uint32_t ssplit[64];              // per sub-task (at its first lane): kind : 2 | plane : 3 | items of the left child : 8
};
__device__ __forceinline__ void small_box_to_float(const int (*t)[kSmallCol], uint32_t s, float* f)   // words 0..5 of column s
{
#pragma unroll
    for (int k = 0; k < 6; k++) f[k] = ordered_int_to_float(k < 3 ? t[k][s] : ~t[k][s]);
}
// One item's contribution to the boxes of column `col`: primitive box b, centroid box = its centre.  The "max" words are
// kept complemented (max x = ~min ~x on the ordered ints), so all twelve updates are ds_min_i32, and lane i walks the words
// starting at word r = i mod 12 (values rotated to match by a four-stage barrel shifter): the lanes of one sub-task, which
// all update the same column, queue on twelve different words instead of one (the same-address LDS atomics of these
// reductions were half of this kernel's LDS cycles: profiles/r03_sah_small_experiments.txt).
__device__ __forceinline__ void small_box_update(int (*t)[kSmallCol], uint32_t col, const float* b, uint32_t r)
{
    int val[12];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ctr = float_to_ordered_int((b[k] + b[3 + k]) * 0.5f);
        val[k] = float_to_ordered_int(b[k]);
        val[3 + k] = ~float_to_ordered_int(b[3 + k]);
        val[6 + k] = ctr;
        val[9 + k] = ~ctr;
    }
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const bool on = (r >> st) & 1u;
        int u[12];
#pragma unroll
        for (int k = 0; k < 12; k++) u[k] = val[(k + (1 << st)) % 12];
#pragma unroll
        for (int k = 0; k < 12; k++) val[k] = on ? u[k] : val[k];
    }
    uint32_t w = r;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        atomicMin(&t[w][col], val[k]);
        w = w == 11u ? 0u : w + 1u;
    }
}

// LDS traffic between the lanes of ONE wave: its DS operations execute in order, so only the compiler has to be kept
// from moving memory operations across the point
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr uint32_t kSmallWaves = 2;   // independent waves per workgroup.  A wave's tables are 11.2 KB of LDS (9 granules of 1280 B):
                                      // 2-wave workgroups fit 7 to a CU = 14 waves, 4-wave ones 3 = 12 (202 us), 7-wave ones 2 = 14
                                      // but with the tail of the slowest of seven (208 us); 1 wave: 191 us, 2 waves: 189 us

__device__ __forceinline__ void sah_small_task(const SahArgs& a, SmallSmem& S, uint32_t task, uint32_t lane)
{
    const SahSmall R = a.small[task];
    const uint32_t cnt = R.end - R.start, base = R.start;
    const int bias = (R.flags & 1u) ? -2 * (int)a.B : (int)(2 * kSahCells);
    const uint64_t lt_mask = (1ull << lane) - 1ull;

    uint32_t idv = 0;
    float b[6] = {0, 0, 0, 0, 0, 0};
    if (lane < cnt) {
        idv = a.ids[(R.flags >> 1) & 1u][base + lane];
        load_box(a.aabbs, idv, b);
    }
    uint32_t s = 0, e = cnt, parent = R.parent_idx;
    bool active = lane < cnt;
    uint32_t cur = 0;
    const uint32_t rot = lane % 12u, rot6 = lane % 6u;
    // root boxes
#pragma unroll
    for (int w = 0; w < 12; w++) S.sbox[0][w][lane] = kEmptyLo;   // (~kEmptyHi == kEmptyLo)
    wave_lds_sync();
    if (active) small_box_update(S.sbox[0], 0u, b, rot);
    wave_lds_sync();
    if ((R.flags & 4u) && lane < 6) S.sbox[0][6 + lane][0] = lane < 3 ? a.H->gc[lane] : ~a.H->gc[lane];   // the top root's centroid bounds are the scene's
    wave_lds_sync();

    while (__builtin_amdgcn_ballot_w64(active)) {
        const uint32_t nxt = cur ^ 1;
        const uint32_t count = e - s;
        // ---- leaves (SharedTaskBuilder.cu:396-464)
        if (active && count <= 2) {
            if (count == 1) {
                sah_leaf_desc(a, a.nodes + parent, idv);
            } else {
                const uint32_t child = (uint32_t)(bias + 2 * (int)(base + s + 1));
                sah_leaf_desc(a, a.nodes + child + (lane - s), idv);
                if (lane == s) {
                    float pb[6];
                    small_box_to_float(S.sbox[cur], s, pb);
                    sah_put_node(a.nodes + parent, pb, child, 2u, RT_CHILD_BOX);
                }
            }
            active = false;
        }
        // ---- bin
        const bool leader = active && lane == s;
        const uint64_t leaders = __builtin_amdgcn_ballot_w64(leader);
        const uint32_t nseg = (uint32_t)__popcll(leaders);
        const uint32_t seg = (uint32_t)__popcll(leaders & ((1ull << s) - 1ull));
        const uint64_t segmask = active ? (((e >= 64 ? ~0ull : ((1ull << e) - 1ull))) & ~((1ull << s) - 1ull)) : 0ull;
        for (uint32_t j = lane; j < nseg * 8; j += 64) {
#pragma unroll
            for (int w = 0; w < 7; w++) S.bins[w][j] = w < 6 ? kEmptyLo : 0;   // (max words complemented: ~kEmptyHi == kEmptyLo)
        }
#pragma unroll
        for (int w = 0; w < 12; w++) S.sbox[nxt][w][lane] = kEmptyLo;
        float c[6];
#pragma unroll
        for (int k = 0; k < 6; k++) c[k] = active ? ordered_int_to_float(k < 3 ? S.sbox[cur][6 + k][s] : ~S.sbox[cur][6 + k][s]) : 0.0f;
        const bool binned = active && !(sah_sa(c) <= 0.0f);
        int bin = 0;
        if (binned) {
            const int axis = sah_axis(c);
            const float epsilon = 1.1920929e-7f;
            const float cmin = axis == 0 ? c[0] : (axis == 1 ? c[1] : c[2]);
            const float cmax = axis == 0 ? c[3] : (axis == 1 ? c[4] : c[5]);
            const float k1 = 8 * (1 - epsilon) / (cmax - cmin);
            const float ca = axis == 0 ? (b[0] + b[3]) * 0.5f : (axis == 1 ? (b[1] + b[4]) * 0.5f : (b[2] + b[5]) * 0.5f);
            bin = min(7, max(0, cvt_rzi(k1 * (ca - cmin))));
        }
        wave_lds_sync();
        if (binned) {
            const uint32_t cell = seg * 8 + (uint32_t)bin;
            // (as in small_box_update: "max" words complemented, lane i starts at word i mod 6)
            int val[6];
#pragma unroll
            for (int k = 0; k < 3; k++) { val[k] = float_to_ordered_int(b[k]); val[3 + k] = ~float_to_ordered_int(b[3 + k]); }
#pragma unroll
            for (int st = 0; st < 3; st++) {
                const bool on = (rot6 >> st) & 1u;
                int u[6];
#pragma unroll
                for (int k = 0; k < 6; k++) u[k] = val[(k + (1 << st)) % 6];
#pragma unroll
                for (int k = 0; k < 6; k++) val[k] = on ? u[k] : val[k];
            }
            uint32_t w = rot6;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                atomicMin(&S.bins[w][cell], val[k]);
                w = w == 5u ? 0u : w + 1u;
            }
            atomicAdd(&S.bins[6][cell], 1);     // items per bin (round 2: eight 64-bit ballot popcounts per level)
        }
        wave_lds_sync();
        // ---- SelectPlane (SharedTaskBuilder.cu:297-350): the sub-task's first lane sweeps the bins left -> right, its
        // second lane right -> left -- the SAME instructions on two lanes (a sub-task here has >= 3 lanes) -- then the
        // first lane fetches the suffix values and scores the 7 planes (strict <, planes 6 -> 0: the highest wins ties).
        {
            const bool fwd = leader && binned, bwd = active && binned && lane == s + 1;
            float sa_run[7];
            uint32_t n_run[7];
            if (fwd || bwd) {
                // the seven bins of this sweep first -- 49 LDS reads in flight at once instead of a read-wait-merge chain --
                // then the running union, its surface area and the running count, in registers
                int bw[7][6];
                uint32_t bc[7];
#pragma unroll
                for (int i = 0; i < 7; i++) {
                    const uint32_t cell = seg * 8 + (uint32_t)(bwd ? 7 - i : i);
#pragma unroll
                    for (int k = 0; k < 6; k++) bw[i][k] = k < 3 ? S.bins[k][cell] : ~S.bins[k][cell];
                    bc[i] = (uint32_t)S.bins[6][cell];
                }
                int run[6] = {kEmptyLo, kEmptyLo, kEmptyLo, kEmptyHi, kEmptyHi, kEmptyHi};
                uint32_t cc = 0;
#pragma unroll
                for (int i = 0; i < 7; i++) {
#pragma unroll
                    for (int k = 0; k < 3; k++) { run[k] = min(run[k], bw[i][k]); run[3 + k] = max(run[3 + k], bw[i][3 + k]); }
                    cc += bc[i];
                    float f[6];
                    ibox_to_float(run, f);
                    sa_run[i] = sah_sa(f);
                    n_run[i] = cc;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 7; i++) { sa_run[i] = 0.0f; n_run[i] = 0; }
            }
            // suffix j (bins 7 .. 7-j) from the next lane
            float sa_suf[7];
            uint32_t n_suf[7];
#pragma unroll
            for (int i = 0; i < 7; i++) {
                sa_suf[i] = __shfl_down(sa_run[i], 1, 64);
                n_suf[i] = __shfl_down(n_run[i], 1, 64);
            }
            if (leader) {
                uint32_t kind = 2, plane = 0, nl = count >> 1;
                if (binned) {
                    float best = 3.402823466e+38f;
                    int pl = -1;
#pragma unroll
                    for (int i = 6; i >= 0; i--) {
                        const float score = sa_run[i] * (float)n_run[i] + sa_suf[6 - i] * (float)n_suf[6 - i];
                        if (score < best && n_run[i] && n_suf[6 - i]) { best = score; pl = i; nl = n_run[i]; }
                    }
                    if (pl >= 0) { kind = 1; plane = (uint32_t)pl; } else nl = count >> 1;
                }
                S.ssplit[s] = kind | (plane << 2) | (nl << 5);
                // parent descriptor (SharedTaskBuilder.cu:544-558)
                float pb[6];
                small_box_to_float(S.sbox[cur], s, pb);
                sah_put_node(a.nodes + parent, pb, (uint32_t)(bias + 2 * (int)(base + s + nl)), 2u, RT_CHILD_BOX);
            }
        }
        wave_lds_sync();
        // ---- PartitionIds (:352-380), stable; child boxes; move the items
        uint32_t dest = lane;
        if (active) {
            const uint32_t sp = S.ssplit[s], kind = sp & 3u, plane = (sp >> 2) & 7u, nl = sp >> 5;
            bool left;
            if (kind == 1) {
                left = (uint32_t)bin <= plane;
                const uint64_t lm = __builtin_amdgcn_ballot_w64(left) & segmask;   // (all lanes of the segment are here)
                const uint32_t lr = (uint32_t)__popcll(lm & lt_mask);
                dest = left ? s + lr : s + nl + ((lane - s) - lr);
            } else {
                left = lane < s + nl;
            }
            const uint32_t cs = left ? s : s + nl;
            small_box_update(S.sbox[nxt], cs, b, rot);
        }
        // push every item to its new lane (a permutation inside each sub-task; inactive lanes keep theirs)
        idv = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest * 4), (int)idv);
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = __int_as_float(__builtin_amdgcn_ds_permute((int)(dest * 4), __float_as_int(b[k])));
        if (active) {
            const uint32_t nl = S.ssplit[s] >> 5;
            const uint32_t child_index = (uint32_t)(bias + 2 * (int)(base + s + nl));
            if (lane < s + nl) { e = s + nl; parent = child_index; }
            else { s = s + nl; parent = child_index + 1; }
        }
        cur = nxt;
        wave_lds_sync();
    }
}

// ---------------------------------------------------------------------------------------------
// Stragglers.  The level loop is a FIXED number of launches (the host never reads the number of live tasks back, so the
// build is a plain sequence of asynchronous launches like the LBVH and can be captured in a hipGraph); tasks that still
// hold more than kSahSmall items after it -- unbalanced splits: a few tasks on ordinary scenes, deep chains on scenes
// that span many octaves -- are finished here, ONE WORKGROUP PER TASK, the way the reference builds a whole grid cell
// with one block looping over a task queue (SharedTaskBuild's main loop, SharedTaskBuilder.cu:909-967): pop a task, bin
// its items into LDS, select the plane (the same function as the level loop: bit-identical planes), write the parent
// descriptor, partition stably into the other id buffer, push the children that are still large (the smaller one on top:
// the stack stays below log2 of the task size) and queue the small ones for sah_small_kernel, which runs after this
// kernel.  Node slots are a function of the split position, so the order tasks are processed in does not matter.
constexpr uint32_t kFinThreads = 256, kFinStack = 64, kFinGrid = 1024;
// level launches beyond log2(items per cell / kSahSmall).  Measured (tools/sah_loop.py, RT_SAH_BATCH_DELTA on the tuning
// library, profiles/r04_sah_batch_sweep.txt): 1M grid 0.648 / 0.645 / 0.771 / 1.35 ms for margins 3 / 2 / 1 / 0, 10M 5.43 / 5.39 /
// 5.41 / 11.6 -- a level the whole GPU sweeps costs 34 us at 1M; the same tasks through the straggler kernel (one workgroup
// each, plane selection by one thread) cost four times that.  The straggler kernel is the safety net, not the plan.
constexpr uint32_t kSahLevelMargin = 2;

__global__ __launch_bounds__(kFinThreads) void sah_finish_kernel(SahArgs a, uint32_t lvl)
{
    const uint32_t ntask = a.H->level_count[lvl];
    if (blockIdx.x >= ntask) return;
    __shared__ SahTask stack[kFinStack];     // flags: bit 0 top tree, bit 1 the id buffer the task's items are in
    __shared__ uint32_t sp, s_kind, s_plane, s_mid;
    __shared__ int sb[8 * kBinWords];        // the task's bins: the global form (max words complemented) while they fill
    __shared__ int scb[2][12];               // the children's boxes [side][p box 6, c box 6], ordered ints
    __shared__ uint32_t ws[8];
    __shared__ float c_score[8];
    __shared__ uint32_t c_ln[8];
    const uint32_t tid = threadIdx.x;
    for (uint32_t task = blockIdx.x; task < ntask; task += gridDim.x) {
        if (tid == 0) {
            SahTask T = a.tasks[lvl & 1u][task];
            T.flags = (T.flags & 1u) | ((lvl & 1u) << 1);
            stack[0] = T;
            sp = 1;
        }
        __syncthreads();
        while (true) {
            const uint32_t depth = sp;       // (uniform: written before the last barrier)
            if (depth == 0) break;
            const SahTask T = stack[depth - 1];
            const uint32_t cur = (T.flags >> 1) & 1u, nxt = cur ^ 1u;
            const uint32_t count = T.end - T.start;
            for (uint32_t j = tid; j < 8 * kBinWords; j += kFinThreads) sb[j] = (j % kBinWords) == 12 ? 0 : kEmptyLo;
            if (tid < 24) scb[tid / 12][tid % 12] = (tid % 6) < 3 ? kEmptyLo : kEmptyHi;
            __syncthreads();                 // everybody holds T; the tables are clear
            if (tid == 0) { sp = depth - 1; s_kind = 2; s_plane = 0; s_mid = T.start + (count >> 1); }
            // ---- BinCentroids (SharedTaskBuilder.cu:206-264), as sah_bin_kernel
            if (!(sah_sa(T.c) <= 0.0f)) {
                const int axis = sah_axis(T.c);
                const float epsilon = 1.1920929e-7f;
                const float cmin = axis == 0 ? T.c[0] : (axis == 1 ? T.c[1] : T.c[2]);
                const float cmax = axis == 0 ? T.c[3] : (axis == 1 ? T.c[4] : T.c[5]);
                const float k1 = 8 * (1 - epsilon) / (cmax - cmin);
                for (uint32_t i = T.start + tid; i < T.end; i += kFinThreads) {
                    float b[6], ctr[3];
                    load_box(a.aabbs, a.ids[cur][i], b);
#pragma unroll
                    for (int k = 0; k < 3; k++) ctr[k] = (b[k] + b[3 + k]) * 0.5f;
                    const float ca = axis == 0 ? ctr[0] : (axis == 1 ? ctr[1] : ctr[2]);
                    const int bin = min(7, max(0, cvt_rzi(k1 * (ca - cmin))));
                    a.binof[i] = (uint8_t)bin;
                    int* lb = &sb[bin * kBinWords];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        atomicMin(&lb[k], float_to_ordered_int(b[k]));
                        atomicMin(&lb[3 + k], ~float_to_ordered_int(b[3 + k]));
                        atomicMin(&lb[6 + k], float_to_ordered_int(ctr[k]));
                        atomicMin(&lb[9 + k], ~float_to_ordered_int(ctr[k]));
                    }
                    atomicAdd(&lb[12], 1);
                }
                __syncthreads();
                if (tid < 8 * kBinWords) {   // the form SelectPlane reads: max words as they are
                    const uint32_t wd = tid % kBinWords;
                    if (wd < 12 && (wd % 6) >= 3) sb[tid] = ~sb[tid];
                }
                __syncthreads();
                // eight lanes score the seven planes (boxes left / right of plane i are merges of ordered ints: exact and order-free;
                // the score is SelectPlane's own expression), lane 0 picks like the serial sweep: right to left, strict <
                if (tid < 8) {
                    const uint32_t i = tid;
                    float score = 3.402823466e+38f;
                    uint32_t lni = 0;
                    if (i < 7) {
                        int L[6] = {kEmptyLo, kEmptyLo, kEmptyLo, kEmptyHi, kEmptyHi, kEmptyHi}, R[6] = {kEmptyLo, kEmptyLo, kEmptyLo, kEmptyHi, kEmptyHi, kEmptyHi};
                        uint32_t ln = 0, rn = 0;
#pragma unroll
                        for (uint32_t bq = 0; bq < 8; bq++) {
                            if (bq <= i) { ibox_merge(L, sb + bq * kBinWords); ln += (uint32_t)sb[bq * kBinWords + 12]; }
                            else { ibox_merge(R, sb + bq * kBinWords); rn += (uint32_t)sb[bq * kBinWords + 12]; }
                        }
                        float fl[6], fr[6];
                        ibox_to_float(L, fl);
                        ibox_to_float(R, fr);
                        score = sah_sa(fl) * (float)ln + sah_sa(fr) * (float)rn;
                        lni = (ln && rn) ? ln : 0u;
                    }
                    c_score[i] = score; c_ln[i] = lni;
                }
                __syncthreads();
                if (tid == 0) {
                    uint32_t kind = 2, plane = 0, mid = T.start + (count >> 1);
                    float best = 3.402823466e+38f;
                    int pl = -1;
#pragma unroll
                    for (int i = 6; i >= 0; i--)
                        if (c_ln[i] && c_score[i] < best) { best = c_score[i]; pl = i; }
                    if (pl >= 0) {
                        kind = 1; plane = (uint32_t)pl; mid = T.start + c_ln[pl];
                        for (int bq = 0; bq < 8; bq++) {
                            int* dst = &scb[bq <= pl ? 0 : 1][0];
                            ibox_merge(dst, sb + bq * kBinWords);
                            ibox_merge(dst + 6, sb + bq * kBinWords + 6);
                        }
                    }
                    s_kind = kind; s_plane = plane; s_mid = mid;
                }
            }
            __syncthreads();
            const uint32_t kind = s_kind, plane = s_plane, mid = s_mid;
            // ---- the children's boxes of an object-median split (SharedTaskBuilder.cu:465-510) and the stable partition
            // (PartitionIds, :352-380) into the other id buffer
            if (kind == 2) {
                for (uint32_t i = T.start + tid; i < T.end; i += kFinThreads) {
                    const uint32_t id = a.ids[cur][i];
                    float b[6];
                    load_box(a.aabbs, id, b);
                    int* v = &scb[i >= mid ? 1 : 0][0];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const int ctr = float_to_ordered_int((b[3 + k] + b[k]) * 0.5f);
                        atomicMin(&v[k], float_to_ordered_int(b[k]));
                        atomicMax(&v[3 + k], float_to_ordered_int(b[3 + k]));
                        atomicMin(&v[6 + k], ctr);
                        atomicMax(&v[9 + k], ctr);
                    }
                    a.ids[nxt][i] = id;
                }
            } else {
                uint32_t running = 0;        // items that went left so far
                for (uint32_t base = T.start; base < T.end; base += kFinThreads) {
                    const uint32_t i = base + tid;
                    const bool in = i < T.end;
                    const bool left = in && a.binof[i] <= plane;   // (written by this very thread above)
                    uint32_t total;
                    const uint32_t ex = block_excl_scan_u32<kFinThreads>(left ? 1u : 0u, ws, &total);
                    if (in) {
                        const uint32_t lr = running + ex;
                        a.ids[nxt][left ? T.start + lr : mid + ((i - T.start) - lr)] = a.ids[cur][i];
                    }
                    running += total;
                }
            }
            // the ids written above are read by OTHER threads of this workgroup when a child is popped: every wave drains its
            // stores (the vector L1 writes through to this XCD's L2), the workgroup meets, the L1 is invalidated -- the reader is
            // on the same CU, so no L2 write-back (an agent-scope release flushes every dirty line of the XCD's L2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (tid == 0) {
                const int bias = (T.flags & 1u) ? -2 * (int)a.B : (int)(2 * kSahCells);
                const uint32_t child_index = (uint32_t)(bias + 2 * (int)mid);
                sah_put_node(a.nodes + T.parent_idx, T.p, child_index, 2u, RT_CHILD_BOX);
                // larger child first, so that the smaller one is popped next
                const int order0 = (mid - T.start) >= (T.end - mid) ? 0 : 1;
                uint32_t top = depth - 1;
                for (int q = 0; q < 2; q++) {
                    const int sd = q == 0 ? order0 : 1 - order0;
                    const uint32_t cs = sd ? mid : T.start, ce = sd ? T.end : mid;
                    if (ce - cs > kSahSmall) {
                        if (top >= kFinStack) { atomicOr(&a.H->status[0], kSahErrLevels); continue; }   // cannot happen: < 2 * log2(items) entries
                        SahTask C;
#pragma unroll
                        for (int k = 0; k < 6; k++) { C.p[k] = ordered_int_to_float(scb[sd][k]); C.c[k] = ordered_int_to_float(scb[sd][6 + k]); }
                        C.start = cs; C.end = ce; C.parent_idx = child_index + (uint32_t)sd; C.flags = (T.flags & 1u) | (nxt << 1);
                        stack[top++] = C;
                    } else {
                        const uint32_t q2 = atomicAdd(&a.H->small_count, 1u);
                        a.small[q2] = SahSmall{cs, ce, child_index + (uint32_t)sd, (T.flags & 1u) | (nxt << 1)};
                    }
                }
                sp = top;
            }
            __syncthreads();
        }
        __syncthreads();
    }
}

// The small tasks [first, H->small_count): a fixed grid, every wave takes tasks with a grid stride.  The count is read
// on the device: no host round trip.
__global__ __launch_bounds__(kSmallWaves * 64) void sah_small_kernel(SahArgs a, uint32_t first)
{
    __shared__ SmallSmem SS[kSmallWaves];
    const uint32_t nsmall = a.H->small_count;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t task = first + blockIdx.x * kSmallWaves + wave; task < nsmall; task += gridDim.x * kSmallWaves) {
        sah_small_task(a, SS[wave], task, lane);
        wave_lds_sync();
    }
}

// top-tree leaves: copy child / count / type of the cell's sub-root (SharedTaskBuilder.cu:422-446; the reference
// forces type Box, which breaks a cell that holds a single leaf -- the type is copied here)
__global__ void sah_patch_top_kernel(SahArgs a, int records_in_status2, uint32_t lvl)
{
    const uint32_t s = threadIdx.x;
    if (s == 0 && !records_in_status2) a.H->status[2] = a.H->status[1];   // without splits: one item per leaf record
    if (s == 0) a.H->live_report = a.H->level_count[lvl];                 // next to the status words: the host reads both with ONE copy
    if (s >= 2 * kSahCells) return;
    rt_node* nd = a.nodes + s;
    if ((nd->w28 >> 29) != RT_CHILD_BOX || (nd->w12 >> 29) != 0) return;
    const uint32_t cell = nd->w28 & kIndexMask;
    const rt_node* sub = a.nodes + 2 * kSahCells + 2 * a.H->cell_start[cell];
    nd->w12 = sub->w12 & ~kIndexMask;
    nd->w28 = sub->w28;
}

// ---------------------------------------------------------------------------------------------
SahLayout sah_layout(uint32_t n)
{
    SahLayout L;
    const size_t B = (size_t)n + n / 5;          // items with --splits: n leaves + fewer than n/5 extra references
    const size_t M = B + kSahCells;
    const size_t TA = M / (kSahSmall + 1) + 2;
    const size_t chunks = (M + kSahChunk - 1) / kSahChunk;
    auto al = [](size_t v) { return (v + 255) / 256 * 256; };
    size_t off = 0;
    L.header = off;       off = al(off + sizeof(SahHeader));
    L.aabbs = off;        off = al(off + M * 24);
    L.ids0 = off;         off = al(off + M * 4);
    L.ids1 = off;         off = al(off + M * 4);
    L.task_of0 = off;     off = al(off + M * 4);
    L.task_of1 = off;     off = al(off + M * 4);
    L.binof = off;        off = al(off + M);
    L.item_leaf = off;    off = al(off + B * 4 + 4);
    L.tasks0 = off;       off = al(off + TA * sizeof(SahTask));
    L.tasks1 = off;       off = al(off + TA * sizeof(SahTask));
    L.splits = off;       off = al(off + TA * sizeof(SahSplit));
    L.bins0 = off;        off = al(off + TA * 8 * kBinWords * 4);
    L.bins1 = off;        off = al(off + TA * 8 * kBinWords * 4);
    L.chunk_hist = off;   off = al(off + chunks * 16 * 4);
    L.chunk_prefix = off; off = al(off + chunks * 4);
    L.small = off;        off = al(off + M * sizeof(SahSmall));
    L.sort = off;         off = al(off + sort_scratch_layout((uint32_t)B).total);
    L.pair_flags = off;   off = al(off + ((size_t)n + 1) / 2 + 1);
    L.pair_sums = off;    off = al(off + (((size_t)n + 1) / 2 / 256 + 2) * 4);
    L.split_flags = off;  off = al(off + ((size_t)n + 1) / 2 + 1);
    L.split_sums_a = off; off = al(off + (((size_t)n + 1) / 2 / 256 + 2) * 4);
    L.split_sums_b = off; off = al(off + (((size_t)n + 1) / 2 / 256 + 2) * 4);
    L.status = L.header + offsetof(SahHeader, status);
    L.cell_counts = L.header + offsetof(SahHeader, cell_count);
    L.total = off;
    return L;
}

hipError_t launch_sah_build(const rt_triangle* tris, uint32_t n, bool pairs, bool splits, rt_triangle_pair* leaves,
                            rt_node* nodes, void* scratch, hipStream_t st, uint32_t* levels_run, uint32_t* status0)
{
    if (status0) *status0 = 0;
    const SahLayout L = sah_layout(n);
    char* s = static_cast<char*>(scratch);
    SahArgs a;
    a.H = reinterpret_cast<SahHeader*>(s + L.header);
    a.nodes = nodes;
    float* aabbs = reinterpret_cast<float*>(s + L.aabbs);
    uint32_t* item_leaf = reinterpret_cast<uint32_t*>(s + L.item_leaf);
    a.aabbs = aabbs;
    a.item_leaf = item_leaf;
    a.ids[0] = reinterpret_cast<uint32_t*>(s + L.ids0);
    a.ids[1] = reinterpret_cast<uint32_t*>(s + L.ids1);
    a.task_of[0] = reinterpret_cast<uint32_t*>(s + L.task_of0);
    a.task_of[1] = reinterpret_cast<uint32_t*>(s + L.task_of1);
    a.binof = reinterpret_cast<uint8_t*>(s + L.binof);
    a.tasks[0] = reinterpret_cast<SahTask*>(s + L.tasks0);
    a.tasks[1] = reinterpret_cast<SahTask*>(s + L.tasks1);
    a.splits = reinterpret_cast<SahSplit*>(s + L.splits);
    a.bins[0] = reinterpret_cast<int*>(s + L.bins0);
    a.bins[1] = reinterpret_cast<int*>(s + L.bins1);
    a.chunk_hist = reinterpret_cast<uint32_t*>(s + L.chunk_hist);
    a.chunk_prefix = reinterpret_cast<uint32_t*>(s + L.chunk_prefix);
    a.small = reinterpret_cast<SahSmall*>(s + L.small);
    a.n = n;
    a.B = n + (splits ? n / 5 : 0);      // host-known upper bound of the item count; the top tree's items start here
    a.M = a.B + kSahCells;
    uint32_t* num_items = &a.H->status[1];
    uint32_t* num_records = &a.H->status[2];
    const uint32_t* n_dev = (pairs || splits) ? num_items : nullptr;
    if (levels_run) *levels_run = 0;

    sah_init_kernel<<<1, 256, 0, st>>>(a.H, nodes, n);
    if (n == 0) return hipGetLastError();
    const uint32_t cand = (n + 1) / 2, cblocks = (cand + 255) / 256;
    const uint32_t pblocks = cblocks < 1024 ? cblocks : 1024;
    uint8_t* pflags = nullptr;
    uint32_t* psums = nullptr;
    hipError_t e = hipSuccess;
    if (pairs) {
        pflags = reinterpret_cast<uint8_t*>(s + L.pair_flags);
        psums = reinterpret_cast<uint32_t*>(s + L.pair_sums);
        e = launch_pair_slots(tris, n, pflags, psums, splits ? num_records : num_items, st);
        if (e != hipSuccess) return e;
    }
    if (splits) {
        // SetupSplits / SetupPairSplits after CalculateSceneAabb (BuildWrapper.cu:188-210)
        e = launch_scene_aabb(tris, n, a.H->gp, st);
        if (e != hipSuccess) return e;
        SplitPassArgs sp;
        sp.tris = reinterpret_cast<const float*>(tris);
        sp.n = n; sp.thresh = n / 5;
        sp.pair_flags = pflags; sp.pair_offsets = psums;
        sp.split_flags = reinterpret_cast<uint8_t*>(s + L.split_flags);
        sp.sums_a = reinterpret_cast<uint32_t*>(s + L.split_sums_a);
        sp.sums_b = reinterpret_cast<uint32_t*>(s + L.split_sums_b);
        sp.leaves = leaves; sp.aabbs = aabbs; sp.idsv = a.ids[1]; sp.item_leaf = item_leaf; sp.H = a.H;
        sah_split_pass_kernel<0><<<pblocks, 256, 0, st>>>(sp);
        e = launch_block_scan(sp.sums_a, cblocks, &a.H->status[3], st);
        if (e != hipSuccess) return e;
        sah_split_pass_kernel<1><<<pblocks, 256, 0, st>>>(sp);
        e = launch_block_scan(sp.sums_b, cblocks, num_items, st);
        if (e != hipSuccess) return e;
        sah_split_pass_kernel<2><<<pblocks, 256, 0, st>>>(sp);
    } else {
        sah_setup_kernel<<<pblocks, 256, 0, st>>>(reinterpret_cast<const float*>(tris), n, leaves, aabbs, a.ids[1], item_leaf, a.H, pflags, psums);
    }
    const uint32_t iblocks = (a.B + 255) / 256;
    sah_grid_kernel<<<iblocks < 1024 ? iblocks : 1024, 256, 0, st>>>(aabbs, a.B, n_dev, a.H, a.task_of[1], splits ? 0 : 1);
    // GridBlockDistribute: cell members in ascending leaf index = one stable radix pass on the cell id
    uint32_t* digit_total = nullptr;
    e = launch_radix_pass(a.task_of[1], a.ids[1], a.task_of[0], a.ids[0], a.B, 0, s + L.sort, st, n_dev, &digit_total);
    if (e != hipSuccess) return e;
    sah_roots_kernel<<<1, 128, 0, st>>>(a, digit_total, aabbs);
    sah_assign_kernel<<<iblocks, 256, 0, st>>>(a, n_dev);

    const uint32_t chunks = (a.M + kSahChunk - 1) / kSahChunk;
    const uint32_t TA = a.M / (kSahSmall + 1) + 2;
    const uint32_t split_blocks = (TA + 63) / 64 > 2048 / 8 ? (TA + 63) / 64 : (TA < 2048 ? (TA + 7) / 8 : 2048 / 8);
    uint32_t lvl = 0;
    // levels until every task has <= kSahSmall items: about log2(items per cell / kSahSmall) when the splits are
    // balanced, plus a margin.  The number of launches is FIXED by n: kernels of a level nobody reaches return at once
    // (level_count[lvl] == 0), and whatever is still alive after the last level is finished by sah_finish_kernel, one
    // workgroup per task.  Nothing is read back: the build is a sequence of asynchronous launches.
    uint32_t batch = kSahLevelMargin;
    for (uint32_t per_cell = a.B / kSahCells; per_cell > kSahSmall; per_cell >>= 1) batch++;
#ifdef RT_SORT_TUNING
    if (const char* e = getenv("RT_SAH_BATCH_DELTA")) { const int b = (int)batch + atoi(e); batch = b < 0 ? 0u : (uint32_t)b; }   // tools/sah_loop.py sweeps
#endif
    constexpr uint32_t kSmallGrid = 32768 / kSmallWaves;
    for (uint32_t i = 0; i < batch && lvl + 1 < kSahMaxLevels; i++, lvl++) {
        sah_bin_kernel<<<chunks, 256, 0, st>>>(a, lvl);
        sah_split_kernel<<<split_blocks, kSplitWaves * 64, 0, st>>>(a, lvl);
        sah_partition_kernel<<<chunks, 256, 0, st>>>(a, lvl);
    }
    sah_finish_kernel<<<kFinGrid, kFinThreads, 0, st>>>(a, lvl);
    sah_small_kernel<<<kSmallGrid, kSmallWaves * 64, 0, st>>>(a, 0u);
    // The patch copies every cell's sub-root descriptor (w12 / w28 of the cell tree's root slot) into its top-tree leaf.
    // Invariant: a cell's root is a level-0 task, a straggler's ancestor or a small task queued by sah_roots_kernel -- all
    // of them have written their descriptor when the three kernels above are done.
    sah_patch_top_kernel<<<1, 128, 0, st>>>(a, splits ? 1 : 0, lvl);
    if (levels_run) *levels_run = lvl;
    return hipGetLastError();
}

}  // namespace rt
