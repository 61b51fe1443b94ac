// rt_launch.hpp -- host-side launch functions of each kernel group and the scratch layout.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "rt_abi.h"

namespace rt {

// hipFuncSetAttribute is a per-DEVICE setting: run `fn` the first time a launch is made on each device of the process
// (a single-process multi-GPU caller, main.cu's ncclCommInitAll shape, builds on every device).  `fn` is idempotent,
// so two threads racing on the same device only repeat it.
constexpr int kMaxDevices = 64;
struct PerDeviceOnce {
    std::atomic<unsigned char> done[kMaxDevices] = {};
    template <class F> hipError_t operator()(F fn)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= kMaxDevices) return fn();
        if (done[dev].load(std::memory_order_acquire)) return hipSuccess;
        e = fn();
        if (e == hipSuccess) done[dev].store(1, std::memory_order_release);
        return e;
    }
};

// ---- radix sort geometry (radix_sort.hip)
// keys per workgroup (tile).  4096 shipped; -DRT_SORT_TILE=2048 / 1024 and -DRT_SORT_DS_NT=<threads of the few-tiles down-sweep>
// are experiment arms (profiles/r04_sort_experiments.txt (b)): every table and threshold below scales with the tile
#ifndef RT_SORT_TILE
#define RT_SORT_TILE 4096
#endif
constexpr uint32_t kSortTile = RT_SORT_TILE;
constexpr uint32_t kSortTileScale = 4096 / kSortTile;      // tiles per 4096 keys
static_assert(kSortTile == 4096 || kSortTile == 2048 || kSortTile == 1024, "tile sizes the kernels are written for");
constexpr uint32_t kRadixBits = 8;
constexpr uint32_t kRadix = 1u << kRadixBits;
constexpr uint32_t kSortPasses = 4;
constexpr uint32_t kRadixMax = 1024;   // the 10-bit passes of the Morton-key sort; tables are sized for it
// The down-sweep addresses keys and values through buffer descriptors of n * 4 bytes with 32-bit byte offsets: n * 4 must
// not wrap.  The builder stays far below (29-bit indices: n <= 2^28); the public sort entry points refuse larger counts.
constexpr uint32_t kSortMaxCount = 0x3FFFFFFFu;

inline uint32_t sort_num_tiles(uint32_t n) { return (n + kSortTile - 1) / kSortTile; }
// the sort's tables hold one word per (digit, tile); a row is padded to a multiple of 4 words (one aligned 16-byte access
// covers four tiles)
inline uint32_t sort_table_stride(uint32_t tiles) { return tiles ? (tiles + 3u) & ~3u : 4u; }
// many tiles: the histogram kernels take four tiles per workgroup and publish 16 bytes per digit
inline bool sort_upsweep_quads(uint32_t tiles) { return tiles > 512 * kSortTileScale; }

struct SortScratch {
    size_t digit_total;  // uint32[kSortPasses][kRadixMax]
    size_t hist;         // uint32[radix][num_tiles]
    size_t offs;         // uint32[radix][num_tiles]
    size_t total;
};
SortScratch sort_scratch_layout(uint32_t n);

// ---- LBVH level geometry (lbvh_levels.hip)
constexpr uint32_t kLeafCap = 512;    // leaves per workgroup at the leaf level
constexpr uint32_t kLeafThreads = 512;
constexpr uint32_t kUpperCap = 2048;  // open roots one upper-level pass holds in LDS
constexpr uint32_t kUpperFan = 64;    // previous-level blocks folded by one upper-level block (when their open roots fit one pass)
constexpr uint32_t kSubFan = 16;      // ... else in sub-passes of this many blocks (kSubFan * kMaxOpen always fits)
constexpr uint32_t kMaxOpen = 128;    // open sub-tree roots a block can emit (>= 2 * max tree depth 62)
constexpr uint32_t kRecDwords = 12;   // segment record: f, l, desc, cc, min[3], max[3], delta at its left end, delta at its right end
constexpr uint32_t kMaxLevels = 6;    // 2^28 leaves -> 2^19 blocks -> 8192 -> 128 -> 2 -> 1

struct LevelPlan {
    uint32_t num_levels;
    uint32_t fan;                    // previous-level blocks folded by one upper-level block (<= kUpperFan)
    uint32_t blocks[kMaxLevels];
    size_t cnt_off[kMaxLevels];      // uint32[blocks]
    size_t rec_off[kMaxLevels];      // uint32[blocks][kMaxOpen][kRecDwords]
    size_t arrive_lvl[kMaxLevels];   // uint32[blocks] tickets taken at this level's blocks (inside the arrive region)
    size_t sub_cnt_off[kMaxLevels];  // fallback scratch of the upper levels
    size_t sub_rec_off[kMaxLevels];
    size_t arrive_off, arrive_bytes; // all arrival counters, contiguous: must be zero when the build kernel starts
    size_t sink_off;                 // 64 bytes nobody reads: where the upper passes' predicated-off stores land
    size_t total;
};
LevelPlan lbvh_level_plan(uint32_t n);

// ---- whole-build scratch layout
struct BuLayout {
    size_t p_aabb;          // int32[6]
    size_t status;          // uint32[8]: [0] error flags, [1] number of leaves L of the last build
    size_t morton;          // uint32[n]
    size_t sorted_indices;  // uint32[n]
    size_t tmp_keys;        // uint32[n]
    size_t tmp_vals;        // uint32[n]
    size_t sort;            // SortScratch
    size_t levels;          // LevelPlan
    size_t hybrid;          // hybrid top-tree work area
    size_t pair_flags;      // uint8[(n+1)/2] merge decision per candidate (--pairs)
    size_t pair_sums;       // uint32[ceil((n+1)/2 / 256)] leaf counts / offsets per workgroup (--pairs)
    size_t aabb_parts;      // int32[kAabbParts][6] partial scene boxes (one per workgroup of the scene-box kernel)
    size_t total;
};
BuLayout bu_layout(uint32_t n);

// ---- SAH build scratch (sah_build.hip)
struct SahLayout {
    size_t header, aabbs, ids0, ids1, task_of0, task_of1, binof, tasks0, tasks1, splits, bins0, bins1, chunk_hist,
        chunk_prefix, small, sort, pair_flags, pair_sums, item_leaf, split_flags, split_sums_a, split_sums_b;
    size_t status;       // uint32[8] inside the header: [0] error flags, [1] number of leaves L
    size_t cell_counts;  // uint32[64] inside the header
    size_t total;
};
SahLayout sah_layout(uint32_t n);

// ---- launches
hipError_t launch_reset_aabb(int* aabb, hipStream_t st);
// scene box: `aabb` holds nparts ordered-int boxes (6 ints each, reset to empty by the caller); workgroup b folds into box
// b mod nparts.  The Morton kernels fold the nparts boxes and (aabb_out != null) publish the result.
constexpr uint32_t kAabbParts = 510;   // at most this many partial boxes (one per workgroup of the build's scene-box kernel)
hipError_t launch_scene_aabb(const rt_triangle* tris, uint32_t n, int* aabb, hipStream_t st, uint32_t nparts = 1);
// the build's first launch (n > 0): partial boxes by plain stores, *nparts_out of them, + status words and the LBVH level
// hand-off counters zeroed
hipError_t launch_scene_aabb_build(const rt_triangle* tris, uint32_t n, int* aabb_parts, uint32_t* nparts_out, uint32_t* status,
                                   uint32_t* arrive, uint32_t arrive_words, hipStream_t st);
hipError_t launch_morton(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                         hipStream_t st, uint32_t nparts = 1, int* aabb_out = nullptr);
// the same codes / values plus the first sort pass's tile histograms (digit = low `bits` bits, bits = 8 or 10) in one
// launch: one workgroup per sort tile
// values == nullptr: they are not written (the sort's first pass then takes them as the identity)
hipError_t launch_morton_hist(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                              hipStream_t st, uint32_t nparts, int* aabb_out, uint32_t* hist, uint32_t bits);
hipError_t launch_morton_pairs(uint32_t* codes, uint32_t* values, const rt_triangle* tris, const int* aabb, uint32_t n,
                               uint8_t* flags, uint32_t* block_sums, uint32_t* num_leaves, hipStream_t st,
                               uint32_t nparts = 1, int* aabb_out = nullptr);
// n_dev (may be null): device word holding the real element count (<= n); n then only sizes the grids.
// key_bits <= 30 (Morton keys): 3 passes of 10 bits, and the INPUT is taken from (tmp_keys, tmp_vals); the sorted result
// is in (keys, vals) either way.
// have_hist0: the first pass's tile histograms are already in the sort scratch (launch_morton_hist wrote them).
// ident0: the input values are the identity (values[i] = i) and are not read.
hipError_t launch_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n,
                             void* sort_scratch, hipStream_t st, const uint32_t* n_dev = nullptr, uint32_t key_bits = 32,
                             bool have_hist0 = false, bool ident0 = false);
bool sort_three_passes(uint32_t tiles);          // keys of <= 30 bits: 3 x 10-bit passes (else 4 x 8)
// the builder's choice: tiles of the Morton-key sort up to which 3 x 10-bit passes beat 4 x 8-bit (measured: 85 vs 93 us at
// 245 tiles, 420 vs 300 us at 2444 -- the 1024-digit tables and 16-byte runs cost more than the saved pass)
constexpr uint32_t kSort3PassMaxTiles = 512 * kSortTileScale;
uint32_t* sort_hist_table(void* sort_scratch, uint32_t n);   // hist[radix][tiles] of the sort scratch
hipError_t launch_lbvh_levels(const rt_triangle* tris, const uint32_t* codes, const uint32_t* sorted_indices,
                              uint32_t n, rt_triangle_pair* leaves, rt_node* nodes, void* level_scratch,
                              uint32_t* status, hipStream_t st, const uint32_t* n_dev = nullptr);

// one stable 8-bit pass (keys_in, vals_in) -> (keys_out, vals_out) on bits [shift, shift+8); *digit_total (device,
// uint32[256], valid after the pass) = number of keys per digit
hipError_t launch_radix_pass(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                             uint32_t n, uint32_t shift, void* sort_scratch, hipStream_t st, const uint32_t* n_dev,
                             uint32_t** digit_total);
// --pairs leaf slots: merge flag per candidate (2k, 2k+1), per-workgroup slot offsets, *num_leaves = L
hipError_t launch_pair_slots(const rt_triangle* tris, uint32_t n, uint8_t* flags, uint32_t* block_sums,
                             uint32_t* num_leaves, hipStream_t st);
// RunSahBuild.  Asynchronous (a fixed number of launches; nothing is read back).  *levels_run (may be null) = level launches
// enqueued; *status0 (may be null) is set to 0 -- the error flags live in the scratch status word.
hipError_t launch_sah_build(const rt_triangle* tris, uint32_t n, bool pairs, bool splits, rt_triangle_pair* leaves,
                            rt_node* nodes, void* scratch, hipStream_t st, uint32_t* levels_run, uint32_t* status0 = nullptr);
// in-place exclusive scan of per-workgroup sums (one workgroup); *total = their sum
hipError_t launch_block_scan(uint32_t* sums, uint32_t count, uint32_t* total, hipStream_t st);

hipError_t launch_hybrid_top(rt_node* nodes, const int* aabb_ordered, uint32_t n, hipStream_t st,
                             const uint32_t* n_dev = nullptr);

struct TraceLaunch {
    rt_accel as;
    rt_scene scene;
    uint64_t* counters;
    int render_type;
    uint8_t* rgba8;
    uint32_t w, h, y0, y1, spp;
    uint32_t strip_rows = 0, strip_first = 0, strip_stride = 1;   // strip_rows > 0: interleaved strips, compact output
};
hipError_t launch_trace(const TraceLaunch& t, hipStream_t st);

}  // namespace rt
