// lbvh_levels.hip -- LBVH topology + leaf repack + bounding boxes in one bottom-up sweep.
//
// Replaces GenerateHierarchy (BottomUpBuilder.cu:167-215), GenerateTriangles (:287-312) and
// GenerateAABBs (:247-285) with the same OUTPUT (Karras node numbering, Node/TrianglePair bytes) and a
// different machine mapping.  The reference finds every internal node's range by binary searches and
// then refits boxes by walking leaf->root with one GLOBAL atomic per level on `volatile` memory.  On
// MI355X a device-scope atomic + cross-XCD visibility costs microseconds per hop, so here the tree is
// agglomerated bottom-up INSIDE LDS:
//
//   * the radix tree over the sorted keys (code_i, i) is unique, so it can be built bottom-up: a
//     finished sub-tree covering leaves [f,l] is the LEFT child of its parent iff
//     delta(l,l+1) > delta(f-1,f)   (delta = common-prefix length `cpl`, BottomUpBuilder.cu:34-38),
//     and then the parent's split is l, otherwise the split is f-1;
//   * Karras numbers an internal node by the end of its range that touches its sibling: a left child
//     has index = its `last`, a right child index = its `first`, the root is 0
//     (BottomUpBuilder.cu:188-194: children of `idx` are `split` and `split+1`).  So the index of a
//     node is known the moment its range and its own merge direction are known -- no searches;
//   * the two children of a parent rendezvous on the boundary between them with ONE LDS exchange:
//     the first arriver leaves its state (range, box, descriptor) in LDS and retires, the second
//     emits the parent's two Node slots (each 32-byte slot = box + descriptor of one child) and
//     climbs on.  A workgroup owns 512 consecutive leaves; sub-trees that cannot finish inside it
//     (their sibling lies in another workgroup) are emitted as "open roots" -- at most 2 x depth <= 124
//     per workgroup, 9 on average on the bench mesh -- and the next level (64x fewer blocks) treats
//     those as its leaves;
//   * at the leaf level every completed node lies in the block's own index range, so the block's node
//     pairs are assembled in LDS and written in one sweep of 64-byte pieces (the scattered per-merge
//     stores made the kernel vector-memory-issue bound: 70 store instructions per wave).  The staged
//     pair of a node doubles as the waiting place of its children (leaf_pass): LDS per leaf decides how
//     many workgroups share a CU, and those are what hides one workgroup's climb;
//   * two launches build the whole hierarchy: the leaf level, then ALL upper levels: when a workgroup
//     has written its block's open roots (write-through stores) it takes a ticket on the counter of
//     the next level's block (one relaxed device atomic); the workgroup that takes the LAST ticket of a
//     block acquires and processes that block, and so on up to the root.  Nobody waits for anybody:
//     there is no spin and no dependence on dispatch order, a workgroup either continues upward or
//     exits.  (Chaining the leaf level into the same launch was measured slower: the leaf pass then
//     shares its register budget with the upper passes.)
//   * an upper pass over at most 1023 open roots does not climb at all (table_pass): the node that splits
//     at a boundary reaches on either side to the nearest boundary with a smaller delta, so one thread per
//     boundary finds its node's range by binary descents over a sparse table of delta minima, takes its
//     two children's boxes as range unions from sparse tables of boxes, and derives its Karras index, its
//     parent and its side from the range -- no node waits for another.  A climb lasts as long as the
//     deepest path of its tree (16 - 22 dependent merges of about 0.85 us each in these passes); the table
//     pass takes 7 - 12 us whatever the tree looks like.  The climb is what the leaf level runs (leaf_pass)
//     -- there the data movement, not the climb, is most of the time -- and the fallback for larger upper
//     passes (level_pass).
//
// Node words: w28 = child:29|type:3 and the box of a slot are written by the workgroup that completes
// the OWNING node; w12 = parent:29|count:3 of a pair is written by whoever completes the pair's parent
// (it knows the parent slot; the counts travel with the segment as 2 bits).  Every dword of every
// slot is written exactly once, so no write ordering between threads is needed.
#include "rt_device.hpp"
#include "rt_launch.hpp"
#include "rt_pairing.hpp"

namespace rt {

#ifdef RT_LBVH_TIMING
// librt_amd_timing.so (see the Makefile): thread 0 of every workgroup leaves 100 MHz timestamps at the phase boundaries
// (tools/lbvh_phases.py)
constexpr uint32_t kStampBlocks = 32768, kStampSlots = 24;
__device__ unsigned long long g_stamp[2][kStampBlocks * kStampSlots];
#define RT_STAMP(arr, k) do { if (threadIdx.x == 0 && blockIdx.x < kStampBlocks) g_stamp[arr][blockIdx.x * kStampSlots + (k)] = wall_clock64(); } while (0)
#endif
#ifndef RT_LBVH_TIMING
#define RT_STAMP(arr, k) do { } while (0)
#endif

// Error flags of the build's status word (rt_bu_scratch_layout.status[0]).  Every global address this file forms from a
// word it has read from memory -- the sorted triangle index, a hand-off record's leaf range and node descriptor -- is
// range-checked first: a corrupt word (an upstream stage that failed, stale scratch) raises a flag and is replaced by a
// harmless value instead of becoming a wild address (a GPU memory-access fault takes the process down).
constexpr uint32_t kErrOpenOverflow = 1u;   // more than kMaxOpen open roots in one block (impossible: depth <= 62)
constexpr uint32_t kErrSortedIndex = 2u;    // sorted_idx[i] does not name a triangle of the input
constexpr uint32_t kErrRecord = 4u;         // a hand-off record of the level below is not a sub-tree of this build

// a hand-off record (f, l, desc, cc): leaves [f, l] of n, desc = a leaf (Tri, index < n) or a node pair (Box, even slot
// index <= 2(n-2))
__device__ __forceinline__ bool record_ok(uint32_t f, uint32_t l, uint32_t desc, uint32_t n)
{
    const uint32_t ix = desc & kIndexMask, ty = desc >> 29;
    const bool d_ok = ty == RT_CHILD_TRI ? ix < n : (ty == RT_CHILD_BOX && (ix & 1u) == 0u && ix + 4 <= 2 * n);
    return f <= l && l < n && d_ok;
}

constexpr uint32_t kLockEmpty = 0xFFFFFFFFu;
constexpr uint32_t kLockDone = 0xFFFFFFFEu;
constexpr uint32_t kCap = kUpperCap;  // segments an upper pass handles in LDS (the leaf pass: kLeafCap leaves)
// open roots of 64 source blocks an upper block folds in ONE pass (else kSubFan at a time).  Real scenes give about 10
// per block (2 x depth 62 = 124 is the bound), so the sub-pass path only runs in the test variant of the library that
// the Makefile builds with a tiny value here (librt_amd_smallcap.so, tests/test_gpu_parity.py).
#ifndef RT_LBVH_FAST_CAP
#define RT_LBVH_FAST_CAP kCap
#endif
static_assert(RT_LBVH_FAST_CAP <= kCap, "one pass holds kCap segments");
static_assert(kLeafCap <= kCap && kSubFan * kMaxOpen <= kCap && kUpperFan % kSubFan == 0 && kUpperFan / kSubFan * kMaxOpen <= kCap,
              "the fallback's sub-passes and its merging pass always fit one pass");
constexpr uint32_t kPrefSlots = 64;   // pref[]: filled by one wave, searched in 6 steps
static_assert(kUpperFan <= kPrefSlots, "one prefix entry per source block");

struct LevelArgs {
    const float* tris;           // 9 floats per triangle
    const uint32_t* codes;       // sorted Morton codes
    const uint32_t* sorted_idx;  // original triangle per sorted position
    uint32_t n;                  // number of leaves; superseded by *n_dev when that is set (--pairs: L is a device value)
    const uint32_t* n_dev;
    rt_triangle_pair* leaves;
    rt_node* nodes;
    uint32_t* status;
    uint32_t* sink;              // 16 dwords nobody reads (see the merge step)
    // per level k: blocks[k] blocks; cnt[k][block] open roots, rec[k][block][kMaxOpen][kRecDwords] their records;
    // arrive[k][block] (k >= 1) tickets taken by the level k-1 blocks that feed it (zero before the launch);
    // sub_cnt / sub_rec [k][block][kUpperFan / kSubFan] scratch of the fallback path
    uint32_t num_levels;
    uint32_t fan;                // previous-level blocks one upper-level block folds (<= kUpperFan)
    uint32_t blocks[kMaxLevels];
    uint32_t* cnt[kMaxLevels];
    uint32_t* rec[kMaxLevels];
    uint32_t* arrive[kMaxLevels];
    uint32_t* sub_cnt[kMaxLevels];
    uint32_t* sub_rec[kMaxLevels];
};

// LDS carve of the climbing pass (dwords)
struct UpperCfg {
    static constexpr uint32_t CAP = kCap;
    static constexpr uint32_t NT = 1024;                      // threads of the workgroup
    static constexpr uint32_t PER = (CAP + 1 + NT - 1) / NT;  // boundaries per thread in the final compaction
    static constexpr uint32_t oDl = 0;                 // int   [CAP+1]  delta at boundary b
    static constexpr uint32_t oBnd = oDl + CAP + 4;    // int   [CAP+1]  last leaf left of boundary b
    static constexpr uint32_t oLock = oBnd + CAP + 4;  // u32   [CAP+1]
    static constexpr uint32_t oRange = oLock + CAP + 4;  // u32 [CAP]  sf:11 | sl:11 | cc:2 | (delta at the far end + 1):7
    static constexpr uint32_t oDesc = oRange + CAP;
    static constexpr uint32_t oBox = oDesc + CAP;      // float [6][CAP]
    static constexpr uint32_t oAbs = oBox + 6 * CAP;   // u32 [CAP]  the leaf index at the segment's far end
    static constexpr uint32_t oWs = oAbs + CAP;        // scan workspace [0, 32), hand-off flag [32], prefix table [40, 40 + 65)
    static constexpr uint32_t kDwords = oWs + 40 + kPrefSlots + 8;
    static constexpr size_t kBytes = (size_t)kDwords * 4;   // 90 KB: the upper levels run a handful of workgroups
};
static_assert(kCap <= 2048, "a deposited range packs two 11-bit segment indices");

// Hand-off stores: write-through at agent scope (`sc1`), so that the records a workgroup leaves for the next level are
// in memory once its `s_waitcnt vmcnt(0)` returns -- without an L2 write-back per workgroup (an agent-scope release
// fence = buffer_wbl2 flushes every dirty line of the XCD's L2, i.e. the leaves and nodes everybody is streaming out:
// measured 2.3x on the 10M build).  MI355X_MICROARCH.md, inter-workgroup visibility: sc1 stores, drained, then the
// ticket; the last arriver acquires before its workgroup reads.
// They are relaxed agent-scope atomic stores (the compiler emits global_store_dwordx2 / dword ... sc1).  The same stores
// written as inline asm (one global_store_dwordx4 ... sc1 per 16 bytes) were corrupted under load: the compiler reused
// the data registers of an asm store for the next record word right after it, and one word of a record then carried the
// value of the word stored next from the same register (z of the box = the delta stored 16 bytes later; found by
// tests/test_gpu_parity.py::test_lbvh_handoff_under_concurrent_load after a change of register allocation: 103 of
// 1500 builds wrong with the asm stores, 0 of 3000 with these).  The compiler handles the hazards of its own stores.
__device__ __forceinline__ void store_sc1(uint4* p, uint32_t x, uint32_t y, uint32_t z, uint32_t w)
{
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    __hip_atomic_store(q, (unsigned long long)x | ((unsigned long long)y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)z | ((unsigned long long)w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_sc1(uint32_t* p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// streaming store (global_store_dwordx4 ... nt) of 16 bytes
__device__ __forceinline__ void store_stream(uint4* p, const uint4& q)
{
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(u4v{q.x, q.y, q.z, q.w}, reinterpret_cast<u4v*>(p));
}

__device__ __forceinline__ int delta_adjacent(const uint32_t* __restrict__ codes, int g, uint32_t n)
{
    // cpl(g, g+1) (BottomUpBuilder.cu:34-38); -1 outside [0, n-1] like Karras' delta
    if (g < 0 || (uint32_t)(g + 1) >= n) return -1;
    const uint32_t c0 = codes[g], c1 = codes[g + 1];
    return c0 == c1 ? 32 + __clz((uint32_t)g ^ (uint32_t)(g + 1)) : __clz(c0 ^ c1);
}

// One pass of an upper level by climbing (the fallback of table_pass): agglomerate the S open roots of the source blocks
// whose records start at src_rec (pref[] in LDS = prefix of their counts) inside LDS, write every node that completes
// straight to memory and emit what stays open to (out_cnt, out_rec).  All 1024 threads of the workgroup call it together.
__device__ __forceinline__ void level_pass(const LevelArgs& a, uint32_t* smem, uint32_t n, uint32_t S,
                                           const uint32_t* src_rec, uint32_t* out_cnt, uint32_t* out_rec, uint32_t so = 0)
{
    using C = UpperCfg;
    (void)so;
    RT_STAMP(1, so);
    constexpr uint32_t NT = C::NT;
    int* dl = reinterpret_cast<int*>(smem + C::oDl);
    int* bnd = reinterpret_cast<int*>(smem + C::oBnd);
    uint32_t* lock = smem + C::oLock;
    uint32_t* s_range = smem + C::oRange;
    uint32_t* s_desc = smem + C::oDesc;
    float* s_box = reinterpret_cast<float*>(smem + C::oBox);
    uint32_t* s_abs = smem + C::oAbs;
    uint32_t* ws = smem + C::oWs;            // [0..17) scan scratch
    const uint32_t* pref = smem + C::oWs + 40;  // [0..64] prefix of the source blocks' counts
    uint32_t* sink = a.sink;
    const uint32_t tid = threadIdx.x;

    // record of local segment s: source block = the last one whose prefix is <= s
    auto rec_ptr = [&](uint32_t s) -> const uint32_t* {
        uint32_t pb = 0;
#pragma unroll
        for (uint32_t step = kPrefSlots / 2; step; step >>= 1) pb += (pref[pb + step] <= s) ? step : 0u;
        return src_rec + ((size_t)pb * kMaxOpen + (s - pref[pb])) * kRecDwords;
    };

    {
        // node addresses of this pass are formed from the records: check them all first (the extra sweep over the records
        // costs nothing that matters here)
        bool bad = false;
        for (uint32_t s0 = tid; s0 < S; s0 += NT) {
            const uint32_t* r = rec_ptr(s0);
            bad |= !record_ok(r[0], r[1], r[2], n);
        }
        if (__syncthreads_or(bad)) {
            if (tid == 0) {
                atomicOr(a.status, kErrRecord);
                store_sc1(out_cnt, 0u);
            }
            __syncthreads();
            return;
        }
    }
    for (uint32_t b = tid; b <= S; b += NT) {
        lock[b] = kLockEmpty;
        if (S == 0) {
            dl[b] = -1;
            bnd[b] = -1;
        } else {
            // the deltas at a segment's two ends travel in its record (they were computed one level below):
            // boundary b is the left end of segment b, the last boundary the right end of segment S-1
            const uint32_t* r = rec_ptr(b < S ? b : S - 1);
            dl[b] = (int)(b < S ? r[10] : r[11]);
            bnd[b] = b < S ? (int)r[0] - 1 : (int)r[1];
        }
    }
    __syncthreads();
    RT_STAMP(1, so + 1);

    for (uint32_t s0 = tid; s0 < S; s0 += NT) {
        uint32_t sf = s0, sl = s0, desc, cc;
        uint32_t fabs, labs;   // first / last leaf under this segment
        float bx[6];
        {
            const uint32_t* r = rec_ptr(s0);
            const uint4 r0 = reinterpret_cast<const uint4*>(r)[0];
            const uint4 r1 = reinterpret_cast<const uint4*>(r)[1];
            const uint4 r2 = reinterpret_cast<const uint4*>(r)[2];
            fabs = r0.x;
            labs = r0.y;
            desc = r0.z;
            cc = r0.w;
            bx[0] = __uint_as_float(r1.x); bx[1] = __uint_as_float(r1.y); bx[2] = __uint_as_float(r1.z);
            bx[3] = __uint_as_float(r1.w); bx[4] = __uint_as_float(r2.x); bx[5] = __uint_as_float(r2.y);
        }

        // The climb is a chain of dependent LDS round trips (one workgroup's whole pass lasts as long as its deepest
        // path), so a step makes only two: deposit + exchange, then the sibling's state.  Everything else a step needs
        // travels in registers or in the deposit: the deltas at the two ends of the range (the far one is part of the
        // deposit) and the leaf indices at the two ends.
        int ldl = dl[sf], rdl = dl[sl + 1];
        while (true) {
            if (ldl < 0 && rdl < 0) break;  // covers every leaf: the finished root
            // go_right (ldl < rdl): I am the LEFT child of my parent.  As a mask, so that what depends on it is bit
            // selects and not the compiler's if / else (deltas lie in [-1, 63]: the difference cannot overflow)
            const uint32_t gm = (uint32_t)((ldl - rdl) >> 31);
            auto pick = [gm](uint32_t right, uint32_t left) { return left ^ ((right ^ left) & gm); };  // go_right ? right : left
            const uint32_t b = pick(sl + 1, sf);
            const int far = min(ldl, rdl);

            s_range[s0] = sf | (sl << 11) | (cc << 22) | ((uint32_t)(far + 1) << 24);
            s_desc[s0] = desc;
#pragma unroll
            for (int k = 0; k < 6; k++) s_box[k * C::CAP + s0] = bx[k];
            s_abs[s0] = pick(fabs, labs);
            // my state is in LDS before the exchange makes me findable: the DS operations of one wave are performed in
            // issue order, so the compiler must keep the order and the hardware does
            asm volatile("" ::: "memory");
            const uint32_t other = atomicExch(&lock[b], s0);
            asm volatile("" ::: "memory");
            if (other == kLockEmpty) break;  // first at the rendezvous: the sibling will carry on
            lock[b] = kLockDone;

            const uint32_t orange = s_range[other];
            const uint32_t odesc = s_desc[other], occ = (orange >> 22) & 3u;
            const uint32_t oabs = s_abs[other];
            const int ofar = (int)(orange >> 24) - 1;
            float ob[6];
#pragma unroll
            for (int k = 0; k < 6; k++) ob[k] = s_box[k * C::CAP + other];

            // The new node's pair = [left child: box, descriptor][right child: box, descriptor].  I am the left child iff
            // go_right, so "mine" and "the sibling's" go to slot my / slot 1 - my: selects on addresses, not on the
            // fourteen values (the climb is bound by the instructions on its dependent chain).
            const uint32_t osf = orange & 0x7FFu, osl = (orange >> 11) & 0x7FFu;
            const uint32_t my = gm + 1u;
            const bool is_root = (far & ofar) < 0;       // the deltas at both ends of the merged range are -1: node 0
            // the merged range's end deltas are (far, ofar) in my direction's order (never equal below the root: the
            // codes inside the range agree on more bits than either delta); a LEFT child iff the left one is smaller
            const uint32_t lm = (uint32_t)((far - ofar) >> 31) ^ ~gm;   // all ones: left child -> index = last leaf
            const uint32_t nsf = pick(sf, osf), nsl = pick(osl, sl);
            const uint32_t fP = pick(fabs, oabs);
            const uint32_t lP = pick(oabs, labs);
            const uint32_t idx = is_root ? 0u : (fP ^ ((lP ^ fP) & lm));  // Karras index of the new node

            const bool mbox = (desc >> 29) == RT_CHILD_BOX, obox = (odesc >> 29) == RT_CHILD_BOX;
            // the new node's pair, then parent:29|count:3 of the children's own pairs (BottomUpBuilder.cu:204-213, :265,
            // :282), straight to memory (few nodes, indices anywhere)
            uint32_t* nw = reinterpret_cast<uint32_t*>(a.nodes + (size_t)idx * 2);
            uint32_t* nm = nw + my * 8;
            uint32_t* no = nw + 8 - my * 8;
            nm[0] = __float_as_uint(bx[0]); nm[1] = __float_as_uint(bx[1]); nm[2] = __float_as_uint(bx[2]);
            *reinterpret_cast<uint4*>(nm + 4) = make_uint4(__float_as_uint(bx[3]), __float_as_uint(bx[4]), __float_as_uint(bx[5]), desc);
            no[0] = __float_as_uint(ob[0]); no[1] = __float_as_uint(ob[1]); no[2] = __float_as_uint(ob[2]);
            *reinterpret_cast<uint4*>(no + 4) = make_uint4(__float_as_uint(ob[3]), __float_as_uint(ob[4]), __float_as_uint(ob[5]), odesc);
            // The step is straight-line code: with a few lanes of one wave on the critical path every taken branch is
            // an instruction-fetch round trip nobody hides, so the stores that only a box child needs go to a sink
            // when the child is a leaf, and the root's own parent words are written after the loop.
            {
                uint32_t* c = mbox ? reinterpret_cast<uint32_t*>(a.nodes + (desc & kIndexMask)) : sink;
                c[3] = (idx * 2 + my) | (((cc & 1u) ? 2u : 1u) << 29);
                c[11] = (idx * 2 + my) | (((cc & 2u) ? 2u : 1u) << 29);
            }
            {
                uint32_t* c = obox ? reinterpret_cast<uint32_t*>(a.nodes + (odesc & kIndexMask)) : sink;
                c[3] = (idx * 2 + 1 - my) | (((occ & 1u) ? 2u : 1u) << 29);
                c[11] = (idx * 2 + 1 - my) | (((occ & 2u) ? 2u : 1u) << 29);
            }

            sf = nsf;
            sl = nsl;
            ldl = (int)pick((uint32_t)ldl, (uint32_t)ofar);
            rdl = (int)pick((uint32_t)ofar, (uint32_t)rdl);
            fabs = fP;
            labs = lP;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bx[k] = fminf(bx[k], ob[k]);
                bx[3 + k] = fmaxf(bx[3 + k], ob[3 + k]);
            }
            desc = ((idx * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29);
            cc = (mbox ? 1u << my : 0u) | (obox ? 2u >> my : 0u);
        }
        if ((ldl & rdl) < 0 && (desc >> 29) == RT_CHILD_BOX) {
            // this thread completed the root.  Q3: the reference leaves the root pair's parent undefined; defined as 0
            uint32_t* nw = reinterpret_cast<uint32_t*>(a.nodes);
            nw[3] = ((cc & 1u) ? 2u : 1u) << 29;
            nw[11] = ((cc & 2u) ? 2u : 1u) << 29;
        }
    }
    __syncthreads();
    RT_STAMP(1, so + 2);

    RT_STAMP(1, so + 3);
    // open roots = rendezvous points where only one child ever arrived, in boundary (= leaf) order
    uint32_t ids[C::PER];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < C::PER; j++) {
        const uint32_t b = tid * C::PER + j;
        uint32_t v = (b <= S) ? lock[b] : kLockEmpty;
        ids[j] = v;
        mine += (v < kLockDone) ? 1u : 0u;
    }
    uint32_t total;
    uint32_t pos = block_excl_scan_u32<NT>(mine, ws, &total);
#pragma unroll
    for (uint32_t j = 0; j < C::PER; j++) {
        const uint32_t id = ids[j];
        if (id < kLockDone) {
            if (pos < kMaxOpen) {
                const uint32_t rg = s_range[id];
                const uint32_t osf = rg & 0x7FFu, osl = (rg >> 11) & 0x7FFu;
                const uint32_t f = (uint32_t)(bnd[osf] + 1);
                const uint32_t l = (uint32_t)bnd[osl + 1];
                uint4* o = reinterpret_cast<uint4*>(out_rec + (size_t)pos * kRecDwords);
                store_sc1(o + 0, f, l, s_desc[id], (rg >> 22) & 3u);
                store_sc1(o + 1, __float_as_uint(s_box[0 * C::CAP + id]), __float_as_uint(s_box[1 * C::CAP + id]),
                          __float_as_uint(s_box[2 * C::CAP + id]), __float_as_uint(s_box[3 * C::CAP + id]));
                store_sc1(o + 2, __float_as_uint(s_box[4 * C::CAP + id]), __float_as_uint(s_box[5 * C::CAP + id]),
                          (uint32_t)dl[osf], (uint32_t)dl[osl + 1]);
            }
            pos++;
        }
    }
    if (tid == 0) {
        store_sc1(out_cnt, min(total, kMaxOpen));
        if (total > kMaxOpen) atomicOr(a.status, kErrOpenOverflow);  // cannot happen: <= 2 * depth(62) open roots
    }
    __syncthreads();   // LDS is reused by the next pass of this workgroup
    RT_STAMP(1, so + 4);
}

// ---- the leaf level's pass.  Same agglomeration as level_pass, laid out for occupancy: the leaf level is where the
// bytes are (44 bytes loaded and 128 bytes stored per leaf) and a workgroup alternates between memory phases and the
// climb, so what hides the climb is other workgroups of the same CU.  LDS decides how many there are.  Here a segment's
// waiting state IS its half of its parent's node pair: the node that splits at boundary b is staged at stage[b], the first
// child to arrive writes its box and descriptor straight into its slot there and exchanges (range | cc | far delta, own
// split boundary) through the 64-bit lock word of b; the second arriver gets that word back from the exchange (no second
// LDS round trip for the range), reads the sibling's slot, writes its own, and leaves (done, Karras index) in the lock
// word.  No separate box / descriptor / range arrays: 76 bytes of LDS per leaf instead of 104, FOUR 512-thread
// workgroups per CU (every wave slot of the CU) instead of three, and seven LDS stores fewer on a merge's dependent chain.
// The final sweep sends pair stage[b] to nodes[2 * index(b)]: 64-byte pieces, all inside the block's own node range.
struct LeafCfg {
    static constexpr uint32_t CAP = kLeafCap, NT = kLeafThreads;
    static constexpr uint32_t PER = (CAP + 1 + NT - 1) / NT;
    static constexpr uint32_t oDl = 0;                        // int [CAP+1] delta at boundary b
    static constexpr uint32_t oLock = oDl + CAP + 4;          // u64 [CAP+1] exchange word of boundary b
    static constexpr uint32_t oStage = oLock + 2 * CAP + 4;   // u32 [CAP+1][16] the pair of the node that splits at b
    static constexpr uint32_t oWs = oStage + 16 * (CAP + 1);  // scan workspace
    static constexpr uint32_t kDwords = oWs + 40;
    static constexpr size_t kBytes = (size_t)kDwords * 4;
};
// LDS is handed out in 1280-byte granules (160 KB / 128)
static_assert((LeafCfg::kBytes + 1279) / 1280 * 4 <= 128, "four leaf workgroups per CU");
static_assert(LeafCfg::oLock % 2 == 0 && LeafCfg::oStage % 4 == 0, "64-bit lock words, 16-byte stage chunks");
static_assert(kLeafCap <= 512, "a lock word packs two 9-bit leaf positions");
constexpr unsigned long long kLock64Empty = ~0ull;

// What a thread fetches for its leaf of a block, kept in registers until the block's turn: the sorted index, the triangle
// it names and the Morton codes on both sides of the boundaries it owns (all issued before anything waits).  Separate from
// the pass so that a workgroup can fetch for a second block while it works on the first (RT_LEAF_BLOCKS=2, see the kernel).
struct LeafFetch {
    uint32_t sv, src;
    float v[9];
    uint32_t c0, c1, e0, e1;   // codes[g], codes[g + 1] of boundary tid; thread 0: also of boundary kLeafCap
};
// step 1: the sorted index and the codes (independent loads)
__device__ __forceinline__ void leaf_fetch_index(const LevelArgs& a, uint32_t n, uint32_t B0, uint32_t S, LeafFetch& f)
{
    const uint32_t tid = threadIdx.x;
    f.sv = f.src = 0u;
    f.c0 = f.c1 = f.e0 = f.e1 = 0u;
    if (tid < S) f.sv = a.sorted_idx[B0 + tid];
    if (tid <= S) {
        const int g = (int)B0 + (int)tid - 1;
        if (g >= 0 && (uint32_t)(g + 1) < n) { f.c0 = a.codes[g]; f.c1 = a.codes[g + 1]; }
    }
    if (tid == 0 && S == LeafCfg::CAP) {
        const int g = (int)B0 + (int)LeafCfg::CAP - 1;
        if ((uint32_t)(g + 1) < n) { f.e0 = a.codes[g]; f.e1 = a.codes[g + 1]; }
    }
}
// step 2: the triangle the index names (GenerateTriangles, BottomUpBuilder.cu:287-312, fused)
__device__ __forceinline__ void leaf_fetch_triangle(const LevelArgs& a, uint32_t S, LeafFetch& f)
{
#pragma unroll
    for (int k = 0; k < 9; k++) f.v[k] = 0.0f;
    if (threadIdx.x < S) {
        f.src = f.sv & 0x7FFFFFFFu;
        // the gather address comes from memory: never past the triangle array (a quad leaf also reads triangle src + 1)
        if (f.src + (f.sv >> 31) >= a.n) {
            atomicOr(a.status, kErrSortedIndex);
            f.sv = f.src = 0u;
        }
        load_tri9(a.tris + (size_t)f.src * 9, f.v);   // 36 bytes at a 4-byte-aligned address: 2 x 16-byte loads + 1 dword
    }
}
// cpl of the adjacent keys g, g + 1 from their codes (delta_adjacent with the loads done by leaf_fetch_index)
__device__ __forceinline__ int delta_from_codes(uint32_t c0, uint32_t c1, int g, uint32_t n)
{
    if (g < 0 || (uint32_t)(g + 1) >= n) return -1;
    return c0 == c1 ? 32 + __clz((uint32_t)g ^ (uint32_t)(g + 1)) : __clz(c0 ^ c1);
}

// Every barrier of the pass orders LDS only (lds_barrier: threads meet through LDS; what goes to global memory is
// write-only here): a __syncthreads() would also drain the vector-memory counter, i.e. wait for the OTHER block's gather.
__device__ __forceinline__ void leaf_pass(const LevelArgs& a, uint32_t* smem, uint32_t n, uint32_t B0, uint32_t S,
                                          uint32_t* out_cnt, uint32_t* out_rec, const LeafFetch& F)
{
    using C = LeafCfg;
    constexpr uint32_t NT = C::NT;
    RT_STAMP(0, 0);
    int* dl = reinterpret_cast<int*>(smem + C::oDl);
    unsigned long long* lock = reinterpret_cast<unsigned long long*>(smem + C::oLock);
    uint32_t* stage = smem + C::oStage;
    uint32_t* ws = smem + C::oWs;
    // stores that only a box child needs go here when the child is a leaf (dwords 3 and 11 are written): slot 0 of stage[0]
    // is nobody's (the node that splits at the block's left edge has its left child in the previous block) and dword 11 is
    // the parent word of its slot 1, which an upper level writes to memory, never this block
    uint32_t* sink = stage;
    const uint32_t tid = threadIdx.x;

    static_assert(C::CAP == NT, "one leaf per thread");
    const uint32_t s0 = tid, i = B0 + s0;
    const bool act = s0 < S;
    const uint32_t sv = F.sv, src = F.src;
    float v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = F.v[k];
    if (tid <= S) {
        lock[tid] = kLock64Empty;
        dl[tid] = delta_from_codes(F.c0, F.c1, (int)B0 + (int)tid - 1, n);
    }
    if (tid == 0 && S == C::CAP) {
        lock[C::CAP] = kLock64Empty;
        dl[C::CAP] = delta_from_codes(F.e0, F.e1, (int)B0 + (int)C::CAP - 1, n);
    }
    RT_STAMP(0, 1);

    float bx[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (act) {
        // the leaf goes to LDS first (the staging area is free until the climb starts) and leaves in the sweep below
        uint4* out = reinterpret_cast<uint4*>(stage + s0 * 16);
        float v3[3] = {v[6], v[7], v[8]};
        if (sv >> 31) {
            // a quad leaf (--pairs): CreateTrianglePair (Pairing.cuh:60-77): A rotated so the shared edge is
            // (v1, v2), v3 = B's vertex off that edge; ids = (src, src+1); rotations = (rot_a, rot_b)
            float B[9];
            load_tri9(a.tris + (size_t)src * 9 + 9, B);
            int ra = 0, rb = 0;
            can_form_pair(v, B, ra, rb);
            float r[9];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                r[k] = ra == 1 ? v[6 + k] : (ra == 2 ? v[3 + k] : v[k]);
                r[3 + k] = ra == 1 ? v[k] : (ra == 2 ? v[6 + k] : v[3 + k]);
                r[6 + k] = ra == 1 ? v[3 + k] : (ra == 2 ? v[k] : v[6 + k]);
                v3[k] = rb == 2 ? B[k] : (rb == 1 ? B[3 + k] : B[6 + k]);
            }
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] = r[k];
            out[0] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), src);
            out[1] = make_uint4(__float_as_uint(v[3]), __float_as_uint(v[4]), __float_as_uint(v[5]), src + 1);
            out[2] = make_uint4(__float_as_uint(v[6]), __float_as_uint(v[7]), __float_as_uint(v[8]), (uint32_t)ra | ((uint32_t)rb << 16));
            out[3] = make_uint4(__float_as_uint(v3[0]), __float_as_uint(v3[1]), __float_as_uint(v3[2]), 0u);
        } else {
            out[0] = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), src);
            out[1] = make_uint4(__float_as_uint(v[3]), __float_as_uint(v[4]), __float_as_uint(v[5]), 0u);
            out[2] = make_uint4(__float_as_uint(v[6]), __float_as_uint(v[7]), __float_as_uint(v[8]), 0u);
            out[3] = make_uint4(__float_as_uint(v[6]), __float_as_uint(v[7]), __float_as_uint(v[8]), 0u);
        }
        // GenerateAABBs leaf box (BottomUpBuilder.cu:259-267; v3 only widens it for a quad, it equals v2 otherwise)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            bx[k] = fminf(fminf(fminf(v[k], v[3 + k]), v[6 + k]), v3[k]);
            bx[3 + k] = fmaxf(fmaxf(fmaxf(v[k], v[3 + k]), v[6 + k]), v3[k]);
        }
    }
    {
        // The block's 32 KB of leaves leave as whole 128-byte lines, 16 bytes per lane on consecutive addresses.  Written by
        // their own threads (four 16-byte stores per lane at a 64-byte stride) every store instruction touched 32 lines and
        // completed none: 565 -> 477 us for this kernel on the 10M mesh.  Streaming stores (nt): nobody reads leaves or nodes
        // before the build is over, and what a default-policy store leaves behind in the Infinity Cache is written back
        // during the NEXT kernel that streams (scene_aabb of the following build: 100 -> 60 us at 10M).  nt on the strided
        // per-thread stores was the opposite: 838 us (profiles/r03_leaf_store_experiments.txt).
        lds_barrier();
        RT_STAMP(0, 5);
        const uint4* st4 = reinterpret_cast<const uint4*>(stage);
        uint4* dst = reinterpret_cast<uint4*>(a.leaves + B0);
#pragma unroll
        for (uint32_t k = 0; k < 4 * C::CAP / NT; k++) {
            const uint32_t c = tid + k * NT;
            if ((c >> 2) < S) store_stream(dst + c, st4[c]);
        }
        lds_barrier();   // the climb reuses the staging area
        RT_STAMP(0, 6);
    }
    if (act) {
        uint32_t sf = s0, sl = s0, cc = 0, myb = 0;   // myb: the boundary my own node splits at (a box segment)
        uint32_t desc = (i & kIndexMask) | ((uint32_t)RT_CHILD_TRI << 29);

        // The climb: a chain of dependent LDS round trips (a workgroup's pass lasts as long as its deepest path), two per
        // step: slot + exchange, then the sibling's slot.
        int ldl = dl[sf], rdl = dl[sl + 1];
        while (true) {
            if (ldl < 0 && rdl < 0) break;  // covers every leaf: the finished root
            // go_right (ldl < rdl): I am the LEFT child of my parent.  As a mask, so that what depends on it is bit selects
            // (deltas lie in [-1, 63]: the difference cannot overflow)
            const uint32_t gm = (uint32_t)((ldl - rdl) >> 31);
            auto pick = [gm](uint32_t right, uint32_t left) { return left ^ ((right ^ left) & gm); };  // go_right ? right : left
            const uint32_t b = pick(sl + 1, sf);
            const int far = min(ldl, rdl);
            const uint32_t my = gm + 1u;   // my slot of the parent's pair: 0 = left child

            uint32_t* nw = stage + b * 16;
            uint32_t* nm = nw + my * 8;
            nm[0] = __float_as_uint(bx[0]); nm[1] = __float_as_uint(bx[1]); nm[2] = __float_as_uint(bx[2]);
            *reinterpret_cast<uint4*>(nm + 4) = make_uint4(__float_as_uint(bx[3]), __float_as_uint(bx[4]), __float_as_uint(bx[5]), desc);
            const unsigned long long word = (unsigned long long)(sf | (sl << 9) | (cc << 18) | ((uint32_t)(far + 1) << 20)) |
                                            ((unsigned long long)myb << 32);
            // my slot is in LDS before the exchange makes me findable: the DS operations of one wave are performed in issue
            // order, so the compiler must keep the order and the hardware does
            asm volatile("" ::: "memory");
            const unsigned long long other = atomicExch(&lock[b], word);
            asm volatile("" ::: "memory");
            if (other == kLock64Empty) break;  // first at the rendezvous: the sibling will carry on

            const uint32_t* no = nw + 8 - my * 8;
            float ob[6];
            ob[0] = __uint_as_float(no[0]); ob[1] = __uint_as_float(no[1]); ob[2] = __uint_as_float(no[2]);
            const uint4 o4 = *reinterpret_cast<const uint4*>(no + 4);
            ob[3] = __uint_as_float(o4.x); ob[4] = __uint_as_float(o4.y); ob[5] = __uint_as_float(o4.z);
            const uint32_t odesc = o4.w;
            const uint32_t orange = (uint32_t)other, omyb = (uint32_t)(other >> 32);
            const uint32_t osf = orange & 0x1FFu, osl = (orange >> 9) & 0x1FFu, occ = (orange >> 18) & 3u;
            const int ofar = (int)(orange >> 20) - 1;

            const bool is_root = (far & ofar) < 0;       // the deltas at both ends of the merged range are -1: node 0
            // the merged range's end deltas are (far, ofar) in my direction's order (never equal below the root: the
            // codes inside the range agree on more bits than either delta); a LEFT child iff the left one is smaller
            const uint32_t lm = (uint32_t)((far - ofar) >> 31) ^ ~gm;   // all ones: left child -> index = last leaf
            const uint32_t nsf = pick(sf, osf), nsl = pick(osl, sl);
            const uint32_t fP = B0 + nsf, lP = B0 + nsl;
            const uint32_t idx = is_root ? 0u : (fP ^ ((lP ^ fP) & lm));  // Karras index of the new node
            lock[b] = (unsigned long long)kLockDone | ((unsigned long long)idx << 32);

            // parent:29|count:3 of the children's own pairs (BottomUpBuilder.cu:204-213, :265, :282); straight-line: a
            // leaf child has no pair, its two stores go to the sink
            const bool mbox = (desc >> 29) == RT_CHILD_BOX, obox = (odesc >> 29) == RT_CHILD_BOX;
            {
                uint32_t* c = mbox ? stage + myb * 16 : sink;
                c[3] = (idx * 2 + my) | (((cc & 1u) ? 2u : 1u) << 29);
                c[11] = (idx * 2 + my) | (((cc & 2u) ? 2u : 1u) << 29);
            }
            {
                uint32_t* c = obox ? stage + omyb * 16 : sink;
                c[3] = (idx * 2 + 1 - my) | (((occ & 1u) ? 2u : 1u) << 29);
                c[11] = (idx * 2 + 1 - my) | (((occ & 2u) ? 2u : 1u) << 29);
            }

            sf = nsf;
            sl = nsl;
            ldl = (int)pick((uint32_t)ldl, (uint32_t)ofar);
            rdl = (int)pick((uint32_t)ofar, (uint32_t)rdl);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bx[k] = fminf(bx[k], ob[k]);
                bx[3 + k] = fmaxf(bx[3 + k], ob[3 + k]);
            }
            desc = ((idx * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29);
            cc = (mbox ? 1u << my : 0u) | (obox ? 2u >> my : 0u);
            myb = b;
        }
        if ((ldl & rdl) < 0 && (desc >> 29) == RT_CHILD_BOX) {
            // this thread completed the root.  Q3: the reference leaves the root pair's parent undefined; defined as 0
            uint32_t* nw = stage + myb * 16;
            nw[3] = ((cc & 1u) ? 2u : 1u) << 29;
            nw[11] = ((cc & 2u) ? 2u : 1u) << 29;
        }
    }
    lds_barrier();
    RT_STAMP(0, 2);

    {
        // the pairs completed here, 16 bytes per lane, four lanes per pair (sweeping in destination order instead -- whole
        // 128-byte lines through an index -> boundary table -- measured the same: 467 vs 470 us at 10M).  A pair whose parent is completed at an upper
        // level leaves with undefined w12 words (dwords 3 and 11); the upper-level kernel, which runs after this one,
        // writes them -- as it writes every word of the pairs not completed here.
        const uint4* st4 = reinterpret_cast<const uint4*>(stage);
        uint4* dst = reinterpret_cast<uint4*>(a.nodes);
#pragma unroll
        for (uint32_t k = 0; k < 4 * C::CAP / NT; k++) {
            const uint32_t c = tid + k * NT;          // 16-byte chunk of the pair that splits at boundary c / 4
            const unsigned long long L = lock[c >> 2];
            if ((c >> 2) < S && (uint32_t)L == kLockDone) store_stream(dst + (size_t)(uint32_t)(L >> 32) * 4 + (c & 3u), st4[c]);
        }
    }

    RT_STAMP(0, 3);
    // open roots = rendezvous points where only one child ever arrived, in boundary (= leaf) order
    unsigned long long words[C::PER];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t j = 0; j < C::PER; j++) {
        const uint32_t b = tid * C::PER + j;
        const unsigned long long L = (b <= S) ? lock[b] : kLock64Empty;
        words[j] = L;
        mine += ((uint32_t)L < kLockDone) ? 1u : 0u;
    }
    uint32_t total;
    uint32_t pos = block_excl_scan_lds<NT>(mine, ws, &total);
#pragma unroll
    for (uint32_t j = 0; j < C::PER; j++) {
        const uint32_t rg = (uint32_t)words[j];
        if (rg < kLockDone) {
            if (pos < kMaxOpen) {
                const uint32_t b = tid * C::PER + j;
                const uint32_t osf = rg & 0x1FFu, osl = (rg >> 9) & 0x1FFu;
                const uint32_t* h = stage + b * 16 + (osl + 1 == b ? 0u : 8u);   // a left child waits at its right end
                uint4* o = reinterpret_cast<uint4*>(out_rec + (size_t)pos * kRecDwords);
                store_sc1(o + 0, B0 + osf, B0 + osl, h[7], (rg >> 18) & 3u);
                store_sc1(o + 1, h[0], h[1], h[2], h[4]);
                store_sc1(o + 2, h[5], h[6], (uint32_t)dl[osf], (uint32_t)dl[osl + 1]);
            }
            pos++;
        }
    }
    if (tid == 0) {
        store_sc1(out_cnt, min(total, kMaxOpen));
        if (total > kMaxOpen) atomicOr(a.status, kErrOpenOverflow);  // cannot happen: <= 2 * depth(62) open roots
    }
    lds_barrier();   // the workgroup's second block reuses the LDS
    RT_STAMP(0, 4);
}

// prefix table of the open-root counts of `nb` (<= 64) source blocks -> pref[0..64] in LDS; returns their sum
__device__ __forceinline__ uint32_t load_prefix(uint32_t* smem, const uint32_t* src_cnt, uint32_t nb)
{
    uint32_t* pref = smem + UpperCfg::oWs + 40;
    const uint32_t tid = threadIdx.x;
    if (tid < 64) {
        uint32_t c = tid < nb ? src_cnt[tid] : 0u;
        c = min(c, kMaxOpen);
        const uint32_t incl = wave_incl_scan_u32(c, (int)tid);
        pref[tid + 1] = incl;
        if (tid == 0) pref[0] = 0;
    }
    __syncthreads();
    return pref[kPrefSlots];
}

// ---- an upper pass by range searches instead of a climb (table_pass).
//
// The climb of a pass is a chain of dependent merges as long as the deepest path of its tree (16 - 22 for the 300 - 650
// open roots the upper passes of the 1M build fold, at about 0.85 us each).  A node of a radix tree is determined by its
// split alone: the node that splits at boundary b reaches, on either side, to the nearest boundary with a SMALLER delta
// (inside a node every other boundary has a larger delta than its split, and the two ends have smaller ones).  So one
// thread per boundary finds its node's range with two binary descents over a sparse table of delta minima; the boxes of
// its two children are two range unions from sparse tables of boxes (min / max are exact and idempotent: two
// overlapping power-of-two windows); the Karras index, the parent (the node that splits at the end with the larger
// delta) and the side follow from the range.  No dependence between nodes: a few barrier-separated table steps and two
// short phases.  Same Node words and the same open-root records as the climb (tests: every upper pass of 1 .. 1023 open
// roots takes this path; the climb keeps the larger ones and the test variant of the library).
// Two sizes: up to 511 open roots all six box planes are tabled at once (9 levels); up to 1023 the box table holds one
// axis (its min and its max plane, 10 levels) and is built three times -- the LDS of a CU holds no more.
template <uint32_t P_, uint32_t LEVELS_, uint32_t NPASS_>
struct TableCfgT {
    static constexpr uint32_t P = P_;                 // boundaries 0 .. P-1, at most P-1 open roots
    static constexpr uint32_t LEVELS = LEVELS_;       // windows of 1 .. 2^(LEVELS-1) entries
    static constexpr uint32_t NPASS = NPASS_;         // builds of the box table
    static constexpr uint32_t PLANES = 6 / NPASS_;    // box planes per build: axes [pass * PLANES/2, ...) min and max
    static constexpr uint32_t oTab = 0;                                   // float [LEVELS-1][PLANES][P]: levels 1 .. LEVELS-1
    static constexpr uint32_t oBox0 = oTab + (LEVELS - 1) * PLANES * P;   // float [6][P]: the open roots' boxes (level 0)
    static constexpr uint32_t oDelta = oBox0 + 6 * P;                     // int   [LEVELS][P + 4]: minima over boundaries
    static constexpr uint32_t oF = oDelta + LEVELS * (P + 4);             // u32 [P] first leaf of segment
    static constexpr uint32_t oL = oF + P;                                // u32 [P] last leaf
    static constexpr uint32_t oDesc = oL + P;
    static constexpr uint32_t oCc = oDesc + P;
    static constexpr uint32_t oIdx = oCc + P;                             // u32 [P] Karras index of the node that splits at boundary b
    static constexpr uint32_t oWs = oIdx + P;                             // 24 dwords of scan scratch
    static constexpr uint32_t kDwords = oWs + 24;
    static constexpr size_t kBytes = (size_t)kDwords * 4;
};
typedef TableCfgT<512, 9, 1> TableSmall;
typedef TableCfgT<1024, 10, 3> TableBig;
constexpr uint32_t kTopCap = TableSmall::P - 1, kTopCapBig = TableBig::P - 1;
static_assert(TableSmall::kBytes <= 160 * 1024 && TableBig::kBytes <= 160 * 1024, "the tables of a pass fit the LDS of a CU");
constexpr uint32_t kNoNode = 0xFFFFFFFFu;   // nd_idx: no complete node splits at this boundary

// FINAL: the last level (every node complete, nothing left open).  Otherwise a node is complete iff a smaller delta
// exists on both sides inside the block (its outer boundaries included); complete nodes whose parent is not complete
// here, and segments in the same situation, are this block's open roots: records for the next level, in leaf order.
template <bool FINAL, typename T>
__device__ __forceinline__ void table_pass(const LevelArgs& a, uint32_t* smem, uint32_t n, uint32_t S, const uint32_t* src_rec,
                                           uint32_t* out_cnt, uint32_t* out_rec, uint32_t so = 0)
{
    (void)so;
    RT_STAMP(1, so);
    constexpr uint32_t P = T::P, LEVELS = T::LEVELS, NPASS = T::NPASS, PLANES = T::PLANES, HALF = T::PLANES / 2;
    float* tab = reinterpret_cast<float*>(smem + T::oTab);
    float* box0 = reinterpret_cast<float*>(smem + T::oBox0);
    int* td = reinterpret_cast<int*>(smem + T::oDelta);
    uint32_t* sg_f = smem + T::oF;
    uint32_t* sg_l = smem + T::oL;
    uint32_t* sg_desc = smem + T::oDesc;
    uint32_t* sg_cc = smem + T::oCc;
    uint32_t* nd_idx = smem + T::oIdx;
    const uint32_t tid = threadIdx.x;
    // plane p of the current build (p < HALF: min of axis pass * HALF + p; else max), window 2^k at i
    auto TAB = [&](uint32_t k, uint32_t p, uint32_t i) -> float& { return tab[((k - 1) * PLANES + p) * P + i]; };
    auto B0 = [&](uint32_t c, uint32_t i) -> float& { return box0[c * P + i]; };   // c: min x y z, max x y z
    auto TD = [&](uint32_t k, uint32_t i) -> int& { return td[k * (P + 4) + i]; };

    // the records (the prefix table that locates them lives in LDS this pass is about to overwrite: registers first)
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0;
    if (tid < S) {
        const uint32_t* pref = smem + UpperCfg::oWs + 40;
        uint32_t pb = 0;
#pragma unroll
        for (uint32_t step = kPrefSlots / 2; step; step >>= 1) pb += (pref[pb + step] <= tid) ? step : 0u;
        const uint4* r = reinterpret_cast<const uint4*>(src_rec + ((size_t)pb * kMaxOpen + (tid - pref[pb])) * kRecDwords);
        r0 = r[0]; r1 = r[1]; r2 = r[2];
    }
    // node addresses of this pass are formed from r0: refuse records that are not sub-trees of this build
    if (__syncthreads_or(tid < S && !record_ok(r0.x, r0.y, r0.z, n))) {
        if (tid == 0) {
            atomicOr(a.status, kErrRecord);
            if (!FINAL) store_sc1(out_cnt, 0u);
        }
        __syncthreads();
        return;
    }
    if (tid < S) {
        sg_f[tid] = r0.x; sg_l[tid] = r0.y; sg_desc[tid] = r0.z; sg_cc[tid] = r0.w;
        B0(0, tid) = __uint_as_float(r1.x); B0(1, tid) = __uint_as_float(r1.y); B0(2, tid) = __uint_as_float(r1.z);
        B0(3, tid) = __uint_as_float(r1.w); B0(4, tid) = __uint_as_float(r2.x); B0(5, tid) = __uint_as_float(r2.y);
        TD(0, tid) = (int)r2.z;                       // boundary t = the left end of segment t
        if (tid == S - 1) TD(0, S) = (int)r2.w;       // the last boundary = the right end of the last segment
    }
    __syncthreads();

    // ---- per build of the box table: its levels (with the delta minima in the first build), then -- in the first
    // build -- the node that splits at boundary b (1 <= b <= S-1): range and Karras index; then the two children's boxes
    // in this build's planes
    const uint32_t b = tid;
    const bool node = b >= 1 && b < S;
    bool complete = false;
    uint32_t sf = 0, sl = 0, idx = 0, Lj = 0, Rj = 0;
    int pl = -1, pr = -1;
    float bo[6] = {0, 0, 0, 0, 0, 0}, bp[6] = {0, 0, 0, 0, 0, 0};   // boxes of the children [sf, b-1] and [b, sl]
#pragma unroll 1
    for (uint32_t pass = 0; pass < NPASS; pass++) {
        const uint32_t ax0 = pass * HALF;   // first axis of this build
        // two levels per barrier: level k from two windows of level k-1, level k+1 from four (unrolled: the steps are
        // bound by instruction issue -- every thread of the workgroup takes part -- and k decides addresses)
#pragma unroll
        for (uint32_t k = 1; k < LEVELS; k += 2) {
            const uint32_t h = 1u << (k - 1);
            const bool two = k + 1 < LEVELS;
            const bool in1 = tid + 2 * h <= S, in2 = two && tid + 4 * h <= S;
            if (in1) {
#pragma unroll
                for (uint32_t c = 0; c < HALF; c++) {
                    auto lo = [&](uint32_t i) { return k == 1 ? B0(ax0 + c, i) : TAB(k - 1, c, i); };
                    auto hi = [&](uint32_t i) { return k == 1 ? B0(3 + ax0 + c, i) : TAB(k - 1, HALF + c, i); };
                    const float l01 = fminf(lo(tid), lo(tid + h)), h01 = fmaxf(hi(tid), hi(tid + h));
                    TAB(k, c, tid) = l01;
                    TAB(k, HALF + c, tid) = h01;
                    if (in2) {
                        TAB(k + 1, c, tid) = fminf(l01, fminf(lo(tid + 2 * h), lo(tid + 3 * h)));
                        TAB(k + 1, HALF + c, tid) = fmaxf(h01, fmaxf(hi(tid + 2 * h), hi(tid + 3 * h)));
                    }
                }
            }
            if (pass == 0) {
                const bool d1 = tid + 2 * h <= S + 1, d2 = two && tid + 4 * h <= S + 1;
                if (d1) {
                    const int m01 = min(TD(k - 1, tid), TD(k - 1, tid + h));
                    TD(k, tid) = m01;
                    if (d2) TD(k + 1, tid) = min(m01, min(TD(k - 1, tid + 2 * h), TD(k - 1, tid + 3 * h)));
                }
            }
            __syncthreads();
        }
        if (pass == 0 && node) {
            const int v = TD(0, b);
            uint32_t pos = b;            // every boundary in [pos, b-1] has a larger delta than v
#pragma unroll
            for (int k = (int)LEVELS - 1; k >= 0; k--) {
                const uint32_t w = 1u << k;
                if (pos >= w && TD((uint32_t)k, pos - w) > v) pos -= w;
            }
            uint32_t q = b + 1;          // every boundary in [b+1, q-1] has a larger delta than v
#pragma unroll
            for (int k = (int)LEVELS - 1; k >= 0; k--) {
                const uint32_t w = 1u << k;
                if (q + w <= S + 1 && TD((uint32_t)k, q) > v) q += w;
            }
            complete = pos > 0 && q <= S;
            if (FINAL && !complete) {
                atomicOr(a.status, kErrOpenOverflow);   // cannot happen at the last level: the outermost deltas are -1
                pos = 1; q = S; complete = true;
            }
            if (complete) {
                Lj = pos - 1; Rj = q;
                sf = Lj; sl = Rj - 1;
                pl = TD(0, Lj); pr = TD(0, Rj);
                const bool is_root = (pl & pr) < 0;
                idx = is_root ? 0u : (pl < pr ? sg_l[sl] : sg_f[sf]);   // a left child is numbered by its last leaf, a right child by its first
            }
        }
        if (complete) {
            // union over [s0, s1] of this build's planes: two overlapping windows of 2^k entries (min / max: exact)
            auto range_planes = [&](uint32_t s0, uint32_t s1, float* o) {
                const uint32_t k = 31u - (uint32_t)__clz(s1 - s0 + 1);
                const uint32_t t1 = s1 + 1 - (1u << k);
#pragma unroll
                for (uint32_t c = 0; c < HALF; c++) {
                    const float lo0 = k == 0 ? B0(ax0 + c, s0) : TAB(k, c, s0), lo1 = k == 0 ? B0(ax0 + c, t1) : TAB(k, c, t1);
                    const float hi0 = k == 0 ? B0(3 + ax0 + c, s0) : TAB(k, HALF + c, s0), hi1 = k == 0 ? B0(3 + ax0 + c, t1) : TAB(k, HALF + c, t1);
                    o[ax0 + c] = fminf(lo0, lo1);
                    o[3 + ax0 + c] = fmaxf(hi0, hi1);
                }
            };
            range_planes(sf, b - 1, bo);
            range_planes(b, sl, bp);
        }
        if (pass + 1 < NPASS) __syncthreads();   // the table is rebuilt
    }

    RT_STAMP(1, so + 1);   // tables built, ranges and child boxes known
    if (tid <= S) nd_idx[tid] = complete ? idx : kNoNode;   // (boundaries 0 and S split no node of this block)
    __syncthreads();
    // is the node that splits at boundary pb complete here?  (the parent of whatever ends at pb with the larger delta)
    auto parent_of = [&](int dleft, int dright, uint32_t bl, uint32_t br, uint32_t& side) -> uint32_t {
        side = dleft < dright ? 0u : 1u;                     // my sibling is on the right: I am the left child
        const uint32_t pb = dleft < dright ? br : bl;
        return (pb >= 1 && pb < S) ? nd_idx[pb] : kNoNode;
    };
    bool node_open = false;
    uint32_t cc_node = 0;
    if (complete) {
        // children: segments [sf, b-1] and [b, sl]; a single segment is the open root itself, else an internal node of this
        // pass (left children are numbered by their last leaf, right children by their first)
        const bool singleL = sf == b - 1, singleR = b == sl;
        const uint32_t dO = singleL ? sg_desc[sf] : (((sg_l[b - 1] * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29));
        const uint32_t dP = singleR ? sg_desc[sl] : (((sg_f[b] * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29));
        const bool boxO = (dO >> 29) == RT_CHILD_BOX, boxP = (dP >> 29) == RT_CHILD_BOX;
        const bool is_root = (pl & pr) < 0;
        uint32_t* nw = reinterpret_cast<uint32_t*>(a.nodes + (size_t)idx * 2);
        // my own parent words: my parent splits at the end of my range that has the larger delta.  If it is not complete
        // here, the level that completes it writes them -- and this one must not touch them (two writers of the same
        // bytes from different XCDs inside one launch)
        uint32_t side;
        const uint32_t pidx = is_root ? 0u : parent_of(pl, pr, Lj, Rj, side);
        cc_node = (boxO ? 1u : 0u) | (boxP ? 2u : 0u);
        node_open = !is_root && pidx == kNoNode;
        if (!node_open) {
            const uint32_t pslot = is_root ? 0u : pidx * 2 + side;
            nw[3] = pslot | ((boxO ? 2u : 1u) << 29);
            nw[11] = pslot | ((boxP ? 2u : 1u) << 29);
        }
        nw[0] = __float_as_uint(bo[0]); nw[1] = __float_as_uint(bo[1]); nw[2] = __float_as_uint(bo[2]);
        *reinterpret_cast<uint4*>(nw + 4) = make_uint4(__float_as_uint(bo[3]), __float_as_uint(bo[4]), __float_as_uint(bo[5]), dO);
        nw[8] = __float_as_uint(bp[0]); nw[9] = __float_as_uint(bp[1]); nw[10] = __float_as_uint(bp[2]);
        *reinterpret_cast<uint4*>(nw + 12) = make_uint4(__float_as_uint(bp[3]), __float_as_uint(bp[4]), __float_as_uint(bp[5]), dP);
        // a child that is an open root of the level below: its pair exists already, its parent words are mine to write
        if (singleL && boxO) {
            uint32_t* c = reinterpret_cast<uint32_t*>(a.nodes + (dO & kIndexMask));
            const uint32_t cc = sg_cc[sf];
            c[3] = (idx * 2) | (((cc & 1u) ? 2u : 1u) << 29);
            c[11] = (idx * 2) | (((cc & 2u) ? 2u : 1u) << 29);
        }
        if (singleR && boxP) {
            uint32_t* c = reinterpret_cast<uint32_t*>(a.nodes + (dP & kIndexMask));
            const uint32_t cc = sg_cc[sl];
            c[3] = (idx * 2 + 1) | (((cc & 1u) ? 2u : 1u) << 29);
            c[11] = (idx * 2 + 1) | (((cc & 2u) ? 2u : 1u) << 29);
        }
    }
    RT_STAMP(1, so + 2);
    if (!FINAL) {
        // ---- open roots, in leaf order: an open root is identified by the segment it starts at
        bool seg_open = false;
        if (tid < S) {
            uint32_t side;
            seg_open = parent_of(TD(0, tid), TD(0, tid + 1), tid, tid + 1, side) == kNoNode;
        }
        __syncthreads();                       // nd_idx has been read by everybody: it becomes the flag / rank array
        nd_idx[tid < P ? tid : 0] = 0u;        // (1024 threads, P <= 1024 entries)
        __syncthreads();
        if (seg_open) nd_idx[tid] = 1u;
        if (node_open) nd_idx[sf] = 1u;        // (a segment inside a complete node is not open: no clash)
        __syncthreads();
        uint32_t total;
        uint32_t* ws = smem + T::oWs;
        const uint32_t rank = block_excl_scan_u32<1024>(tid < S ? nd_idx[tid] : 0u, ws, &total);
        if (tid < P) nd_idx[tid] = rank;
        __syncthreads();
        if (seg_open) {
            const uint32_t j = nd_idx[tid];
            if (j < kMaxOpen) {
                uint4* o = reinterpret_cast<uint4*>(out_rec + (size_t)j * kRecDwords);
                store_sc1(o + 0, sg_f[tid], sg_l[tid], sg_desc[tid], sg_cc[tid]);
                store_sc1(o + 1, __float_as_uint(B0(0, tid)), __float_as_uint(B0(1, tid)), __float_as_uint(B0(2, tid)), __float_as_uint(B0(3, tid)));
                store_sc1(o + 2, __float_as_uint(B0(4, tid)), __float_as_uint(B0(5, tid)), (uint32_t)TD(0, tid), (uint32_t)TD(0, tid + 1));
            }
        }
        if (node_open) {
            const uint32_t j = nd_idx[sf];
            if (j < kMaxOpen) {
                float u[6];
#pragma unroll
                for (int c = 0; c < 3; c++) { u[c] = fminf(bo[c], bp[c]); u[3 + c] = fmaxf(bo[3 + c], bp[3 + c]); }
                uint4* o = reinterpret_cast<uint4*>(out_rec + (size_t)j * kRecDwords);
                store_sc1(o + 0, sg_f[sf], sg_l[sl], ((idx * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29), cc_node);
                store_sc1(o + 1, __float_as_uint(u[0]), __float_as_uint(u[1]), __float_as_uint(u[2]), __float_as_uint(u[3]));
                store_sc1(o + 2, __float_as_uint(u[4]), __float_as_uint(u[5]), (uint32_t)pl, (uint32_t)pr);
            }
        }
        if (tid == 0) {
            store_sc1(out_cnt, min(total, kMaxOpen));
            if (total > kMaxOpen) atomicOr(a.status, kErrOpenOverflow);  // cannot happen: <= 2 * depth(62) open roots
        }
    }
    __syncthreads();   // LDS is reused
    RT_STAMP(1, so + 3);
    RT_STAMP(1, so + 4);
}

// ---- level 0: one 512-thread workgroup per block of 512 leaves (grids are sized for the largest possible n).
// kLeafBlocksPerWg = 2 (a workgroup takes two consecutive blocks and issues the second block's gather before it works on the
// first: one generation of 980 workgroups at 1M triangles instead of two uneven ones of 1959) was built and measured in
// round 4: 52.0 vs 47.5 us at 1M, 491 vs 470 us at 10M -- the CU works on four blocks at a time either way, and
// independent workgroups drift apart (one climbs in LDS while its neighbour waits for memory) where paired blocks march
// in step.  Kept as a compile-time arm (-DRT_LEAF_BLOCKS=2).
#ifndef RT_LEAF_BLOCKS
#define RT_LEAF_BLOCKS 1
#endif
constexpr uint32_t kLeafBlocksPerWg = RT_LEAF_BLOCKS;
__global__ __launch_bounds__(kLeafThreads, 4) void lbvh_leaf_kernel(LevelArgs a)   // four workgroups per CU: 32 waves
{
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t blkA = blockIdx.x * kLeafBlocksPerWg, blkB = blkA + 1;
    const uint32_t A0 = blkA * kLeafCap, B0 = blkB * kLeafCap;
    const uint32_t SA = A0 < n ? min(kLeafCap, n - A0) : 0u;
    LeafFetch FA;
    leaf_fetch_index(a, n, A0, SA, FA);
    if (kLeafBlocksPerWg == 2) {
        const uint32_t SB = (blkB < a.blocks[0] && B0 < n) ? min(kLeafCap, n - B0) : 0u;
        LeafFetch FB;
        leaf_fetch_index(a, n, B0, SB, FB);
        leaf_fetch_triangle(a, SA, FA);
        leaf_fetch_triangle(a, SB, FB);
        leaf_pass(a, smem, n, A0, SA, a.cnt[0] + blkA, a.rec[0] + (size_t)blkA * kMaxOpen * kRecDwords, FA);
        if (blkB < a.blocks[0])
            leaf_pass(a, smem, n, B0, SB, a.cnt[0] + blkB, a.rec[0] + (size_t)blkB * kMaxOpen * kRecDwords, FB);
    } else {
        leaf_fetch_triangle(a, SA, FA);
        leaf_pass(a, smem, n, A0, SA, a.cnt[0] + blkA, a.rec[0] + (size_t)blkA * kMaxOpen * kRecDwords, FA);
    }
}

// ---- all upper levels in one launch: one workgroup per level-1 block; the workgroup that completes the inputs of a
// block of the next level carries on with that block
__global__ __launch_bounds__(1024) void lbvh_upper_kernel(LevelArgs a)
{
    using C = UpperCfg;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* flag = smem + C::oWs + 32;
    const uint32_t tid = threadIdx.x;
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    uint32_t blk = blockIdx.x;
    RT_STAMP(1, 0);

    for (uint32_t lvl = 1; lvl < a.num_levels; lvl++) {
        if (lvl > 1) {
            // hand-off of this workgroup's records (sc1 write-through stores): every storing wave drains its stores,
            // the workgroup meets, one lane takes a ticket; the last ticket acquires (invalidates this CU's L1) before
            // the barrier lets the other waves read the group's records
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint32_t grp = blk / a.fan;
            const uint32_t nb_in = min(a.fan, a.blocks[lvl - 1] - grp * a.fan);
            if (tid == 0) {
                const uint32_t ticket = __hip_atomic_fetch_add(a.arrive[lvl] + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool last = ticket == nb_in - 1;
                if (last) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *flag = last ? 1u : 0u;
            }
            __syncthreads();
            RT_STAMP(1, 1 + (lvl - 1) * 7 - 1);
            if (*flag == 0) return;
            blk = grp;
        }
        const uint32_t first = blk * a.fan;
        const uint32_t nb = min(a.fan, a.blocks[lvl - 1] - first);
        const uint32_t* src_cnt = a.cnt[lvl - 1] + first;
        const uint32_t* src_rec = a.rec[lvl - 1] + (size_t)first * kMaxOpen * kRecDwords;
        uint32_t* out_cnt = a.cnt[lvl] + blk;
        uint32_t* out_rec = a.rec[lvl] + (size_t)blk * kMaxOpen * kRecDwords;
        const uint32_t S = load_prefix(smem, src_cnt, nb);
        RT_STAMP(1, 1 + (lvl - 1) * 7);
        // (the test variant of the library keeps the climb at the last level too: it exists to exercise the sub-pass path)
        const bool last = lvl + 1 == a.num_levels;
        if (RT_LBVH_FAST_CAP == kCap && S >= (last ? 2u : 1u) && S <= kTopCapBig) {
            const uint32_t so = 2 + (lvl - 1) * 7;
            if (S <= kTopCap) {
                if (last) table_pass<true, TableSmall>(a, smem, n, S, src_rec, out_cnt, out_rec, so);
                else table_pass<false, TableSmall>(a, smem, n, S, src_rec, out_cnt, out_rec, so);
            } else {
                if (last) table_pass<true, TableBig>(a, smem, n, S, src_rec, out_cnt, out_rec, so);
                else table_pass<false, TableBig>(a, smem, n, S, src_rec, out_cnt, out_rec, so);
            }
        } else if (S <= RT_LBVH_FAST_CAP) {
            level_pass(a, smem, n, S, src_rec, out_cnt, out_rec, 2 + (lvl - 1) * 7);
        } else {
            // more open roots than one pass holds (deep trees: long runs of equal codes): kSubFan source blocks at a
            // time (always fit: kSubFan * kMaxOpen <= CAP) into this block's scratch, then one pass over those results
            constexpr uint32_t kSubs = kUpperFan / kSubFan;
            uint32_t* sc = a.sub_cnt[lvl] + (size_t)blk * kSubs;
            uint32_t* sr = a.sub_rec[lvl] + (size_t)blk * kSubs * kMaxOpen * kRecDwords;
            for (uint32_t j = 0; j < kSubs; j++) {
                const uint32_t sfirst = j * kSubFan;
                const uint32_t snb = sfirst < nb ? min(kSubFan, nb - sfirst) : 0u;
                const uint32_t SS = load_prefix(smem, src_cnt + sfirst, snb);
                level_pass(a, smem, n, SS, src_rec + (size_t)sfirst * kMaxOpen * kRecDwords, sc + j,
                                  sr + (size_t)j * kMaxOpen * kRecDwords);
            }
            // this workgroup's own stores, read back by its other waves: drain, make them visible, meet
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const uint32_t S2 = load_prefix(smem, sc, kSubs);
            level_pass(a, smem, n, S2, sr, out_cnt, out_rec);
        }
    }
}

// n < 2: n == 1 -> slot 0 describes the single leaf, slot 1 is type None; n == 0 -> both None.
__global__ void lbvh_tiny_kernel(const rt_triangle_pair* leaves, rt_node* nodes, uint32_t n, const uint32_t* n_dev)
{
    if (n_dev) n = *n_dev;
    if (threadIdx.x != 0 || n >= 2) return;
    uint32_t* nw = reinterpret_cast<uint32_t*>(nodes);
    for (int k = 0; k < 16; k++) nw[k] = 0;
    if (n == 1) {
        const rt_triangle_pair t = leaves[0];
        nodes[0].min.x = fminf(fminf(fminf(t.v0.x, t.v1.x), t.v2.x), t.v3.x);
        nodes[0].min.y = fminf(fminf(fminf(t.v0.y, t.v1.y), t.v2.y), t.v3.y);
        nodes[0].min.z = fminf(fminf(fminf(t.v0.z, t.v1.z), t.v2.z), t.v3.z);
        nodes[0].max.x = fmaxf(fmaxf(fmaxf(t.v0.x, t.v1.x), t.v2.x), t.v3.x);
        nodes[0].max.y = fmaxf(fmaxf(fmaxf(t.v0.y, t.v1.y), t.v2.y), t.v3.y);
        nodes[0].max.z = fmaxf(fmaxf(fmaxf(t.v0.z, t.v1.z), t.v2.z), t.v3.z);
        nodes[0].w12 = 1u << 29;
        nodes[0].w28 = 0u | ((uint32_t)RT_CHILD_TRI << 29);
    }
}

#ifdef RT_SORT_TUNING
#include <cstdlib>
#endif
// Blocks of the previous level one upper-level block folds: 48 while that needs no more levels than 64 does (about 10 open
// roots per leaf block on real scenes: 48 blocks keep a pass inside the small range-search tables, <= 511 roots).  Measured
// (tools/lbvh_fan_sweep.sh, upper kernel at 1M / 10M triangles): fan 16: 25.9 / 104 us, 32: 27.4 / 61.0, 48: 24.9 / 55.0,
// 64: 25.2 / 60.2.
static uint32_t lbvh_upper_fan(uint32_t leaf_blocks)
{
    auto levels = [](uint32_t b, uint32_t fan) { uint32_t l = 1; while (b > 1 && l < kMaxLevels) { b = (b + fan - 1) / fan; l++; } return l; };
#ifdef RT_SORT_TUNING
    if (const char* e = getenv("RT_LBVH_FAN")) { const int f = atoi(e); if (f >= 2 && f <= (int)kUpperFan) return (uint32_t)f; }
#endif
    const uint32_t want = levels(leaf_blocks, kUpperFan);
    return levels(leaf_blocks, 48u) == want ? 48u : kUpperFan;
}

LevelPlan lbvh_level_plan(uint32_t n)
{
    LevelPlan p;
    p.num_levels = 0;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    uint32_t blocks = (n + kLeafCap - 1) / kLeafCap;
    if (blocks == 0) blocks = 1;
    p.fan = lbvh_upper_fan(blocks);
    // the arrival counters of all levels first, contiguous: the build's init kernel zeroes [arrive_off, +arrive_bytes)
    p.sink_off = take(64);
    p.arrive_off = off;
    {
        uint32_t b = blocks;
        size_t words = 0;
        for (uint32_t k = 0; k < kMaxLevels; k++) { words += b; if (b == 1) break; b = (b + p.fan - 1) / p.fan; }
        p.arrive_bytes = words * 4;
        off += (p.arrive_bytes + 255) / 256 * 256;
    }
    size_t arrive = p.arrive_off;
    constexpr uint32_t kSubs = kUpperFan / kSubFan;
    while (true) {
        const uint32_t k = p.num_levels++;
        p.blocks[k] = blocks;
        p.arrive_lvl[k] = arrive;
        arrive += (size_t)blocks * 4;
        p.cnt_off[k] = take((size_t)blocks * 4);
        p.rec_off[k] = take((size_t)blocks * kMaxOpen * kRecDwords * 4);
        p.sub_cnt_off[k] = k ? take((size_t)blocks * kSubs * 4) : 0;
        p.sub_rec_off[k] = k ? take((size_t)blocks * kSubs * kMaxOpen * kRecDwords * 4) : 0;
        if (blocks == 1 || p.num_levels == kMaxLevels) break;
        blocks = (blocks + p.fan - 1) / p.fan;
    }
    p.total = off;
    return p;
}

constexpr size_t kUpperLds = TableBig::kBytes > UpperCfg::kBytes ? TableBig::kBytes : UpperCfg::kBytes;
static_assert(TableSmall::kBytes <= kUpperLds, "one LDS size for every pass of the upper kernel");

hipError_t launch_lbvh_levels(const rt_triangle* tris, const uint32_t* codes, const uint32_t* sorted_indices,
                              uint32_t n, rt_triangle_pair* leaves, rt_node* nodes, void* level_scratch,
                              uint32_t* status, hipStream_t st, const uint32_t* n_dev)
{
    static PerDeviceOnce once;
    const hipError_t attr_err = once([] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lbvh_leaf_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LeafCfg::kBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lbvh_upper_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kUpperLds);
        return e;
    });
    if (attr_err != hipSuccess) return attr_err;

    if (n >= 1) {
        const LevelPlan p = lbvh_level_plan(n);
        char* base = static_cast<char*>(level_scratch);
        LevelArgs a;
        a.tris = reinterpret_cast<const float*>(tris);
        a.codes = codes;
        a.sorted_idx = sorted_indices;
        a.n = n;
        a.n_dev = n_dev;
        a.leaves = leaves;
        a.nodes = nodes;
        a.status = status;
        a.sink = reinterpret_cast<uint32_t*>(base + p.sink_off);
        a.num_levels = p.num_levels;
        a.fan = p.fan;
        for (uint32_t k = 0; k < kMaxLevels; k++) {
            const bool on = k < p.num_levels;
            a.blocks[k] = on ? p.blocks[k] : 0;
            a.cnt[k] = on ? reinterpret_cast<uint32_t*>(base + p.cnt_off[k]) : nullptr;
            a.rec[k] = on ? reinterpret_cast<uint32_t*>(base + p.rec_off[k]) : nullptr;
            a.arrive[k] = on ? reinterpret_cast<uint32_t*>(base + p.arrive_lvl[k]) : nullptr;
            a.sub_cnt[k] = on && k ? reinterpret_cast<uint32_t*>(base + p.sub_cnt_off[k]) : nullptr;
            a.sub_rec[k] = on && k ? reinterpret_cast<uint32_t*>(base + p.sub_rec_off[k]) : nullptr;
        }
        // the arrival counters must be zero: the caller's init kernel clears them (rt_run_bottom_up_build), see
        // lbvh_arrive_region()
        lbvh_leaf_kernel<<<(p.blocks[0] + kLeafBlocksPerWg - 1) / kLeafBlocksPerWg, kLeafThreads, LeafCfg::kBytes, st>>>(a);
        if (p.num_levels > 1) lbvh_upper_kernel<<<p.blocks[1], 1024, kUpperLds, st>>>(a);
    }
    if (n < 2 || n_dev) lbvh_tiny_kernel<<<1, 64, 0, st>>>(leaves, nodes, n, n_dev);
    return hipGetLastError();
}

}  // namespace rt

#ifdef RT_LBVH_TIMING
extern "C" __attribute__((visibility("default"))) int rt_debug_lbvh_stamps(unsigned long long* out, int arr)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rt::g_stamp), sizeof(unsigned long long) * rt::kStampBlocks * rt::kStampSlots,
                                    (size_t)arr * sizeof(unsigned long long) * rt::kStampBlocks * rt::kStampSlots);
}
extern "C" __attribute__((visibility("default"))) int rt_debug_lbvh_stamps_clear()
{
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt::g_stamp));
    if (e != hipSuccess) return (int)e;
    return (int)hipMemset(p, 0, sizeof(unsigned long long) * 2 * rt::kStampBlocks * rt::kStampSlots);
}
#endif
