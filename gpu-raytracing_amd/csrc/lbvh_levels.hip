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
//     climbs on.  A workgroup owns 1024 consecutive leaves; sub-trees that cannot finish inside it
//     (their sibling lies in another workgroup) are emitted as "open roots" -- at most 2 x depth <= 124
//     per workgroup -- and the next, 16x smaller level treats those as its leaves.  4 launches build
//     1M triangles; nothing is exchanged between workgroups inside a launch, so there is no
//     inter-workgroup protocol, no global atomic and no spin.
//
// Node words: w28 = child:29|type:3 and the box of a slot are written by the workgroup that completes
// the OWNING node; w12 = parent:29|count:3 of a pair is written by whoever completes the pair's parent
// (it knows the parent slot; the counts travel with the segment as 2 bits).  Every dword of every
// slot is written exactly once, so no write ordering between threads is needed.
#include <mutex>

#include "rt_device.hpp"
#include "rt_launch.hpp"
#include "rt_pairing.hpp"

namespace rt {

constexpr uint32_t kLockEmpty = 0xFFFFFFFFu;
constexpr uint32_t kLockDone = 0xFFFFFFFEu;

struct LevelArgs {
    const float* tris;           // 9 floats per triangle
    const uint32_t* codes;       // sorted Morton codes
    const uint32_t* sorted_idx;  // original triangle per sorted position
    uint32_t n;                  // number of leaves; superseded by *n_dev when that is set (--pairs: L is a device value)
    const uint32_t* n_dev;
    rt_triangle_pair* leaves;
    rt_node* nodes;
    const uint32_t* prev_cnt;    // upper levels: open-root counts of the previous level's workgroups
    const uint32_t* prev_rec;    // upper levels: their records [blocks][kMaxOpen][kRecDwords]
    uint32_t prev_blocks;
    uint32_t* out_cnt;
    uint32_t* out_rec;
    uint32_t* status;
};

template <bool LEAF>
struct LevelCfg {
    static constexpr uint32_t CAP = LEAF ? kLeafCap : kUpperCap;
    static constexpr uint32_t PER = (CAP + 1 + 1023) / 1024;  // boundaries per thread in the final compaction
    // LDS carve (dwords)
    static constexpr uint32_t oDl = 0;                 // int   [CAP+1]  delta at boundary b
    static constexpr uint32_t oBnd = oDl + CAP + 4;    // int   [CAP+1]  last leaf left of boundary b
    static constexpr uint32_t oLock = oBnd + CAP + 4;  // u32   [CAP+1]
    static constexpr uint32_t oRange = oLock + CAP + 4;  // u32 [CAP]  sf | sl << 16
    static constexpr uint32_t oDesc = oRange + CAP;
    static constexpr uint32_t oCc = oDesc + CAP;
    static constexpr uint32_t oBox = oCc + CAP;        // float [6][CAP]
    static constexpr uint32_t oWs = oBox + 6 * CAP;    // scan workspace + prefix table
    static constexpr uint32_t kDwords = oWs + 64;
    static constexpr size_t kBytes = (size_t)kDwords * 4;
};

__device__ __forceinline__ int delta_adjacent(const uint32_t* __restrict__ codes, int g, uint32_t n)
{
    // cpl(g, g+1) (BottomUpBuilder.cu:34-38); -1 outside [0, n-1] like Karras' delta
    if (g < 0 || (uint32_t)(g + 1) >= n) return -1;
    const uint32_t c0 = codes[g], c1 = codes[g + 1];
    return c0 == c1 ? 32 + __clz((uint32_t)g ^ (uint32_t)(g + 1)) : __clz(c0 ^ c1);
}

template <bool LEAF>
__global__ __launch_bounds__(1024) void lbvh_level_kernel(LevelArgs a)
{
    using C = LevelCfg<LEAF>;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    int* dl = reinterpret_cast<int*>(smem + C::oDl);
    int* bnd = reinterpret_cast<int*>(smem + C::oBnd);
    uint32_t* lock = smem + C::oLock;
    uint32_t* s_range = smem + C::oRange;
    uint32_t* s_desc = smem + C::oDesc;
    uint32_t* s_cc = smem + C::oCc;
    float* s_box = reinterpret_cast<float*>(smem + C::oBox);
    uint32_t* ws = smem + C::oWs;      // [0..17) scan scratch
    uint32_t* pref = smem + C::oWs + 32;  // [0..17) prefix of previous-level counts

    const uint32_t tid = threadIdx.x;
    const uint32_t n = a.n_dev ? *a.n_dev : a.n;
    const uint32_t B0 = blockIdx.x * kLeafCap;  // leaf level only
    uint32_t S;

    if (LEAF) {
        S = B0 < n ? min(kLeafCap, n - B0) : 0u;   // grids are sized for the largest possible n
    } else {
        if (tid < 64) {
            const uint32_t pb = blockIdx.x * kUpperFan + tid;
            uint32_t c = (tid < kUpperFan && pb < a.prev_blocks) ? a.prev_cnt[pb] : 0u;
            c = min(c, kMaxOpen);
            uint32_t incl = wave_incl_scan_u32(c, (int)tid);
            if (tid < kUpperFan) pref[tid + 1] = incl;
            if (tid == 0) pref[0] = 0;
        }
        __syncthreads();
        S = pref[kUpperFan];
    }

    // record of local segment s at an upper level
    auto rec_ptr = [&](uint32_t s) -> const uint32_t* {
        uint32_t pb = 0;
#pragma unroll
        for (uint32_t k = 1; k < kUpperFan; k++) pb += (pref[k] <= s) ? 1u : 0u;
        return a.prev_rec + ((size_t)(blockIdx.x * kUpperFan + pb) * kMaxOpen + (s - pref[pb])) * kRecDwords;
    };

    for (uint32_t b = tid; b <= S; b += 1024) {
        lock[b] = kLockEmpty;
        int g;
        if (LEAF) {
            g = (int)B0 + (int)b - 1;
        } else {
            g = (S == 0) ? -1 : ((b == 0) ? (int)rec_ptr(0)[0] - 1 : (int)rec_ptr(b - 1)[1]);
            bnd[b] = g;
        }
        dl[b] = S ? delta_adjacent(a.codes, g, n) : -1;
    }
    __syncthreads();

    for (uint32_t s0 = tid; s0 < S; s0 += 1024) {
        uint32_t sf = s0, sl = s0, desc, cc;
        float bx[6];
        if (LEAF) {
            // GenerateTriangles (BottomUpBuilder.cu:287-312) fused: gather the triangle, emit the
            // 64-byte leaf in sorted order (ids defined), keep its box in registers.
            const uint32_t i = B0 + s0;
            const uint32_t sv = a.sorted_idx[i];
            const uint32_t src = sv & 0x7FFFFFFFu;
            float v[9];
            load_tri9(a.tris + (size_t)src * 9, v);   // 36 bytes at a 4-byte-aligned address: 2 x 16-byte loads + 1 dword
            uint4* out = reinterpret_cast<uint4*>(a.leaves + i);
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
                // GenerateTriangles (BottomUpBuilder.cu:287-312) fused: the 64-byte leaf in sorted order (ids defined)
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
            desc = (i & kIndexMask) | ((uint32_t)RT_CHILD_TRI << 29);
            cc = 0;
        } else {
            const uint32_t* r = rec_ptr(s0);
            const uint4 r0 = reinterpret_cast<const uint4*>(r)[0];
            const uint4 r1 = reinterpret_cast<const uint4*>(r)[1];
            const uint4 r2 = reinterpret_cast<const uint4*>(r)[2];
            desc = r0.z;
            cc = r0.w;
            bx[0] = __uint_as_float(r1.x); bx[1] = __uint_as_float(r1.y); bx[2] = __uint_as_float(r1.z);
            bx[3] = __uint_as_float(r1.w); bx[4] = __uint_as_float(r2.x); bx[5] = __uint_as_float(r2.y);
        }

        while (true) {
            const int ldl = dl[sf], rdl = dl[sl + 1];
            if (ldl < 0 && rdl < 0) break;  // covers every leaf: the finished root
            const bool go_right = ldl < rdl;  // I am the LEFT child of my parent
            const uint32_t b = go_right ? sl + 1 : sf;

            s_range[s0] = sf | (sl << 16);
            s_desc[s0] = desc;
            s_cc[s0] = cc;
#pragma unroll
            for (int k = 0; k < 6; k++) s_box[k * C::CAP + s0] = bx[k];
            // my state is in LDS before the exchange makes me findable (one wave's DS ops retire in order)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint32_t other = atomicExch(&lock[b], s0);
            asm volatile("" ::: "memory");
            if (other == kLockEmpty) break;  // first at the rendezvous: the sibling will carry on
            lock[b] = kLockDone;

            const uint32_t orange = s_range[other];
            const uint32_t odesc = s_desc[other], occ = s_cc[other];
            float ob[6];
#pragma unroll
            for (int k = 0; k < 6; k++) ob[k] = s_box[k * C::CAP + other];

            // left / right child of the new node
            uint32_t Lsf, Rsl, Ldesc, Rdesc, Lcc, Rcc;
            float Lb[6], Rb[6];
            if (go_right) {
                Lsf = sf; Rsl = orange >> 16; Ldesc = desc; Rdesc = odesc; Lcc = cc; Rcc = occ;
#pragma unroll
                for (int k = 0; k < 6; k++) { Lb[k] = bx[k]; Rb[k] = ob[k]; }
            } else {
                Lsf = orange & 0xFFFFu; Rsl = sl; Ldesc = odesc; Rdesc = desc; Lcc = occ; Rcc = cc;
#pragma unroll
                for (int k = 0; k < 6; k++) { Lb[k] = ob[k]; Rb[k] = bx[k]; }
            }
            const int pl = dl[Lsf], pr = dl[Rsl + 1];
            const bool is_root = pl < 0 && pr < 0;
            const uint32_t fP = LEAF ? B0 + Lsf : (uint32_t)(bnd[Lsf] + 1);
            const uint32_t lP = LEAF ? B0 + Rsl : (uint32_t)bnd[Rsl + 1];
            const uint32_t idx = is_root ? 0u : (pl < pr ? lP : fP);  // Karras index of the new node

            const bool Lbox = (Ldesc >> 29) == RT_CHILD_BOX, Rbox = (Rdesc >> 29) == RT_CHILD_BOX;
            uint32_t* nw = reinterpret_cast<uint32_t*>(a.nodes + (size_t)idx * 2);
            nw[0] = __float_as_uint(Lb[0]); nw[1] = __float_as_uint(Lb[1]); nw[2] = __float_as_uint(Lb[2]);
            *reinterpret_cast<uint4*>(nw + 4) =
                make_uint4(__float_as_uint(Lb[3]), __float_as_uint(Lb[4]), __float_as_uint(Lb[5]), Ldesc);
            nw[8] = __float_as_uint(Rb[0]); nw[9] = __float_as_uint(Rb[1]); nw[10] = __float_as_uint(Rb[2]);
            *reinterpret_cast<uint4*>(nw + 12) =
                make_uint4(__float_as_uint(Rb[3]), __float_as_uint(Rb[4]), __float_as_uint(Rb[5]), Rdesc);
            if (is_root) {  // Q3: the reference leaves the root pair's parent undefined; defined as 0
                nw[3] = (Lbox ? 2u : 1u) << 29;
                nw[11] = (Rbox ? 2u : 1u) << 29;
            }
            // parent:29|count:3 of the children's own pairs (BottomUpBuilder.cu:204-213, :265, :282)
            if (Lbox) {
                uint32_t* cw = reinterpret_cast<uint32_t*>(a.nodes + (Ldesc & kIndexMask));
                cw[3] = (idx * 2) | (((Lcc & 1u) ? 2u : 1u) << 29);
                cw[11] = (idx * 2) | (((Lcc & 2u) ? 2u : 1u) << 29);
            }
            if (Rbox) {
                uint32_t* cw = reinterpret_cast<uint32_t*>(a.nodes + (Rdesc & kIndexMask));
                cw[3] = (idx * 2 + 1) | (((Rcc & 1u) ? 2u : 1u) << 29);
                cw[11] = (idx * 2 + 1) | (((Rcc & 2u) ? 2u : 1u) << 29);
            }

            sf = Lsf;
            sl = Rsl;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                bx[k] = fminf(Lb[k], Rb[k]);
                bx[3 + k] = fmaxf(Lb[3 + k], Rb[3 + k]);
            }
            desc = ((idx * 2) & kIndexMask) | ((uint32_t)RT_CHILD_BOX << 29);
            cc = (Lbox ? 1u : 0u) | (Rbox ? 2u : 0u);
        }
    }
    __syncthreads();

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
    uint32_t pos = block_excl_scan_u32<1024>(mine, ws, &total);
#pragma unroll
    for (uint32_t j = 0; j < C::PER; j++) {
        const uint32_t id = ids[j];
        if (id < kLockDone) {
            if (pos < kMaxOpen) {
                const uint32_t rg = s_range[id];
                const uint32_t osf = rg & 0xFFFFu, osl = rg >> 16;
                const uint32_t f = LEAF ? B0 + osf : (uint32_t)(bnd[osf] + 1);
                const uint32_t l = LEAF ? B0 + osl : (uint32_t)bnd[osl + 1];
                uint4* o = reinterpret_cast<uint4*>(a.out_rec + ((size_t)blockIdx.x * kMaxOpen + pos) * kRecDwords);
                o[0] = make_uint4(f, l, s_desc[id], s_cc[id]);
                o[1] = make_uint4(__float_as_uint(s_box[0 * C::CAP + id]), __float_as_uint(s_box[1 * C::CAP + id]),
                                  __float_as_uint(s_box[2 * C::CAP + id]), __float_as_uint(s_box[3 * C::CAP + id]));
                o[2] = make_uint4(__float_as_uint(s_box[4 * C::CAP + id]), __float_as_uint(s_box[5 * C::CAP + id]), 0u, 0u);
            }
            pos++;
        }
    }
    if (tid == 0) {
        a.out_cnt[blockIdx.x] = min(total, kMaxOpen);
        if (total > kMaxOpen) atomicOr(a.status, 1u);  // cannot happen: <= 2 * depth(62) open roots
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

LevelPlan lbvh_level_plan(uint32_t n)
{
    LevelPlan p;
    p.num_levels = 0;
    size_t off = 0;
    uint32_t blocks = (n + kLeafCap - 1) / kLeafCap;
    if (blocks == 0) blocks = 1;
    while (true) {
        const uint32_t k = p.num_levels++;
        p.blocks[k] = blocks;
        p.cnt_off[k] = off;
        off += ((size_t)blocks * 4 + 255) / 256 * 256;
        p.rec_off[k] = off;
        off += (size_t)blocks * kMaxOpen * kRecDwords * 4;
        if (blocks == 1 || p.num_levels == kMaxLevels) break;
        blocks = (blocks + kUpperFan - 1) / kUpperFan;
    }
    p.total = off;
    return p;
}

hipError_t launch_lbvh_levels(const rt_triangle* tris, const uint32_t* codes, const uint32_t* sorted_indices,
                              uint32_t n, rt_triangle_pair* leaves, rt_node* nodes, void* level_scratch,
                              uint32_t* status, hipStream_t st, const uint32_t* n_dev)
{
    static PerDeviceOnce once;
    const hipError_t attr_err = once([] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lbvh_level_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)LevelCfg<false>::kBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lbvh_level_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)LevelCfg<true>::kBytes);
        return e;
    });
    if (attr_err != hipSuccess) return attr_err;

    if (n >= 2 || n == 1) {
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
        for (uint32_t k = 0; k < p.num_levels; k++) {
            a.prev_cnt = k ? reinterpret_cast<const uint32_t*>(base + p.cnt_off[k - 1]) : nullptr;
            a.prev_rec = k ? reinterpret_cast<const uint32_t*>(base + p.rec_off[k - 1]) : nullptr;
            a.prev_blocks = k ? p.blocks[k - 1] : 0;
            a.out_cnt = reinterpret_cast<uint32_t*>(base + p.cnt_off[k]);
            a.out_rec = reinterpret_cast<uint32_t*>(base + p.rec_off[k]);
            if (k == 0)
                lbvh_level_kernel<true><<<p.blocks[k], 1024, LevelCfg<true>::kBytes, st>>>(a);
            else
                lbvh_level_kernel<false><<<p.blocks[k], 1024, LevelCfg<false>::kBytes, st>>>(a);
        }
    }
    if (n < 2 || n_dev) lbvh_tiny_kernel<<<1, 64, 0, st>>>(leaves, nodes, n, n_dev);
    return hipGetLastError();
}

}  // namespace rt
