// trace_kernel.hip -- primary-ray generation, BVH traversal, intersection, shading, RGBA8 store.
//
// Replaces TraceRays / TraceRay / IntersectRayAabb / IntersectRayTriangle(Pair) / AmbientShader
// (Tracer.cu:187-200, 256-374, 376-469, 471-595).  Per-ray semantics -- and therefore the per-ray box /
// triangle test counts, which are OUTPUTS of the kBoxtests / kTriangleTests modes -- follow the
// reference exactly: children visited in slot order, a leaf hit is intersected before the next slot's
// box is compared against tmax, the nearest Box child is continued first (ties: larger child index),
// the others are pushed in encounter order, popped entries are not re-culled.
//
// Machine mapping for wave64 (what differs from the reference's one-thread-one-ray loop):
//   * one wave = one 8x8 pixel tile (Morton order inside the tile): the 64 rays walk the same upper
//     tree, their 64-byte sibling-pair loads hit the same lines (L1 hit rate 97 % on the bench scene);
//   * a sibling pair (2 x 32-byte slots, 64-byte aligned) is fetched with four 16-byte loads issued
//     together; front/back of BOTH boxes are computed at once (they do not depend on tmax), the
//     tmax/tmin comparisons are then applied in slot order;
//   * WAVE-LEVEL TWO-PHASE SCHEDULE: a leaf hit is rare per ray (about 1 in 70 steps) but almost
//     certain per 64-lane step, and the triangle test is the long divergent path.  A lane that needs a
//     leaf test parks (phase LEAF0 / LEAF1, keeping the second slot's front/back in registers) and the
//     wave keeps stepping boxes for the others; when parked lanes outnumber stepping lanes
//     (one __builtin_amdgcn_ballot_w64 pair per step) the wave runs ONE leaf phase for all of them.
//     Only the interleaving ACROSS lanes changes; each ray's own sequence of tests is untouched;
//   * the traversal stack is a lane-interleaved LDS column addressed through an address_space(3)
//     pointer (ds_read/ds_write; conflict-free: bank = lane % 32 in both 32-lane halves); an entry is
//     one packed dword child:29|count:3; entries beyond the LDS depth spill to private memory;
//   * the reference's push-then-pop of the nearest child is kept in a register instead;
//   * 1/direction is computed once per ray (bit-identical to recomputing it per box: IEEE division);
//   * test counters are wave-reduced and added with ONE 64-bit atomic pair per wave (reference: one
//     atomic per ray, Tracer.cu:503);
//   * workgroups are dealt to XCDs round-robin by the hardware; the tile order is remapped so that an XCD
//     takes runs of 8 consecutive workgroups (256 x 8 pixels) spread over the whole frame: neighbouring
//     rays share an L2, and every XCD gets its share of the expensive parts of the image.
// Compiled with -ffp-contract=off: results are bit-identical to the C oracle.
#include <cstdio>
#include <cstdlib>

#include "rt_device.hpp"
#include "rt_launch.hpp"
#include "rt_math.h"

namespace rt {

constexpr int kStackLds = 16;    // LDS-resident stack entries per lane (bench scenes peak at 10)
constexpr int kStackMax = 64;    // reference stack size (Tracer.cu:314)
constexpr uint32_t kPrefetchMinPrims = 8u << 20;   // scenes from here on take the pair-prefetch instantiation (see trace_kernel)
#ifndef RT_TRACE_WAVES
#define RT_TRACE_WAVES 4
#endif
constexpr int kTraceWaves = RT_TRACE_WAVES;   // waves (8x8 tiles) per workgroup
#ifndef RT_TRACE_MIN_WAVES
#define RT_TRACE_MIN_WAVES 7   // waves per SIMD the register allocator must fit (72 VGPRs: no spills; 8 -> 64 VGPRs spills)
#endif
// Tuning build only (make TUNING=1 -> -DRT_TRACE_TUNING; never in the shipped library): render type 100 writes the raw
// u32 box-test count per pixel (tools/divergence_stats.py) and RT_TRACE_PARK="num,den" overrides the park threshold.
#ifdef RT_TRACE_TUNING
constexpr int kRenderDebugBoxCount = 100;
#else
constexpr int kRenderDebugBoxCount = -1;   // matches no render type: every use below folds away
#endif
// the wave runs a box step while  stepping * park_den >= parked * park_num  (else one leaf phase)
constexpr int kParkNum = 8, kParkDen = 1;   // (round-2 sweep under the chunked XCD order, tools/sweep_park.sh: 4..8 equal on the 1080p LBVH frame; 4 is +5 % on the SAH tree but -4 % on the 4K x 16 spp frame)

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// ---- publication of the test counters (see the end of trace_kernel).  The caller's four counters are ONE 32-byte target
// for every workgroup of a frame, and same-address device atomics queue at the memory side at ~18 ns each: 8,100
// workgroups = 0.15 ms behind which a sparse frame waits (camera B of the bench: 2,690 instead of 3,330 Mrays/s serial).
// So a launch borrows one of kCtrSlots slots of module-static device memory: workgroup b adds its sums into row b mod
// kCtrSub of the slot (16 queues instead of one) and takes a ticket on that row; the last ticket of a row takes a ticket on the
// slot; the last of those folds the 16 rows into the caller's counters and leaves the slot zeroed (nobody waits; the
// "last arriver continues" hand-off of lbvh_levels.hip).  Slots are handed out round-robin by launch_trace: two launches
// share one only if more than kCtrSlots launches with counters are in flight at once.
constexpr uint32_t kCtrSlots = 256, kCtrSub = 16;
struct alignas(64) CtrSlot {
    // one 64-byte line per row: [box, tri, box-phase steps, leaf-phase steps, the row's tickets, pad x 3] -- the ticket lives
    // in its row's line: sixteen tickets in ONE line would queue exactly like the single target this replaces
    unsigned long long part[kCtrSub][8];
    unsigned long long top, pad[7];
};
__device__ CtrSlot g_ctr[kCtrSlots];

struct TraceParams {
    const rt_node* nodes;
    const rt_triangle_pair* leaves;
    const rt_attributes* attributes;
    const rt_material* materials;
    const rt_texture* textures;
    const rt_camera* camera;
    float light[3];
    uint32_t root, count, num_materials, num_textures;
    uint8_t* rgba8;
    uint32_t w, h, y0, y1, spp;
    unsigned long long* counters;
    uint32_t tiles_x, num_tiles;
    // interleaved strips (rt_trace_strips): strip_tiles > 0 tile rows per strip; tile row t of the launch is row
    // (t % strip_tiles) of strip  strip_first + (t / strip_tiles) * strip_stride  and is stored at tile row t (compactly)
    uint32_t strip_tiles, strip_first, strip_stride;
    int park_num, park_den;
    uint32_t ctr_slot;   // counters != null: this launch's slot of g_ctr
};

struct Ray {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz, tmin, tmax;
};
struct Hit {
    uint32_t primitive_id, tri_id;
    float bu, bv;
};

// Tracer.cu:256-291
__device__ __forceinline__ bool intersect_tri(float v0x, float v0y, float v0z, float v1x, float v1y, float v1z,
                                              float v2x, float v2y, float v2z, Ray& r, Hit& h, uint32_t tri_id,
                                              uint32_t prim_id)
{
    const float epsilon = 0.000000001f;
    const float e1x = v1x - v0x, e1y = v1y - v0y, e1z = v1z - v0z;
    const float e2x = v2x - v0x, e2y = v2y - v0y, e2z = v2z - v0z;
    const float hx = r.dy * e2z - r.dz * e2y, hy = r.dz * e2x - r.dx * e2z, hz = r.dx * e2y - r.dy * e2x;
    const float a = e1x * hx + e1y * hy + e1z * hz;
    if (a > -epsilon && a < epsilon) return false;
    const float f = 1.0f / a;
    const float sx = r.ox - v0x, sy = r.oy - v0y, sz = r.oz - v0z;
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if (u < 0.0f || u > 1.0f) return false;
    const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
    const float v = f * (r.dx * qx + r.dy * qy + r.dz * qz);
    if (v < 0.0f || (u + v) > 1.0f) return false;
    const float t = f * (e2x * qx + e2y * qy + e2z * qz);
    if (t < r.tmin || t > r.tmax) return false;
    r.tmax = t;
    h.primitive_id = prim_id;
    h.tri_id = tri_id;
    h.bu = u;
    h.bv = v;
    return true;
}

// per-lane traversal state
enum : uint32_t { PH_STEP = 0, PH_LEAF0 = 1, PH_LEAF1 = 2, PH_DONE = 3 };
constexpr uint32_t kNoNear = 0xFFFFFFFFu;  // "no Box child hit yet" (child 2^29-1, count 7: not a real entry)

// The private spill array is NOT a member: a dynamically indexed member would pin the whole struct in scratch.
typedef uint32_t SpillArray[kStackMax - kStackLds];

struct Trav {
    lds_u32* lds;  // this lane's stack column: entry k at lds[k * 64]
    uint32_t* spill;
    int sp;
    uint32_t cur;       // slots still to visit of the current node: first slot : 29 | slot count : 3
    uint32_t near_e;    // nearest Box child so far (packed like cur) or kNoNear
    float near_d;       // its front distance (+inf while kNoNear)
    uint32_t phase;
    uint32_t leaf;      // pending leaf: index : 29 | count : 3
    // second slot of the current pair, kept while parked in PH_LEAF0
    float f1, k1;       // front / back
    uint32_t e1;        // child : 29 | count : 3
    uint32_t t1;        // type, 0 = none / absent
    uint32_t box_tests, tri_tests;
    // PF instantiations only (scenes whose tree does not fit the caches, see launch_trace): the next pair's four 16-byte
    // loads, issued as soon as advance() has picked it -- before the wave's next vote -- and carried in registers
    uint4 pf0, pf1, pf2, pf3;

    __device__ __forceinline__ void push(uint32_t e)
    {
        if (sp < kStackLds) lds[sp * 64] = e;
        else if (sp < kStackMax) spill[sp - kStackLds] = e;
        sp = min(sp + 1, kStackMax);  // a push onto a full stack is dropped (the reference overruns its array, Tracer.cu:353-369)
    }
    // A Box child (Tracer.cu:338-363), predicated on `in`: the first hit becomes `near`; a closer one (ties:
    // larger child index) displaces `near` onto the stack; otherwise it is pushed itself.  Bitwise logic on
    // purpose (no short-circuit branches); with no near yet near_d = +inf makes every hit "closer".
    __device__ __forceinline__ void inner_hit(bool in, uint32_t e, float front)
    {
        const bool closer = (front < near_d) | ((front == near_d) & ((e & kIndexMask) > (near_e & kIndexMask)));
        if (in & (near_e != kNoNear)) push(closer ? near_e : e);
        const bool take = in & closer;
        near_e = take ? e : near_e;
        near_d = take ? front : near_d;
    }
    // the current pair is finished: remaining slots of the same node (count > 2 only, never in an LBVH), else
    // the nearest child (the reference pushes it last and pops it first -- so on a FULL stack that push is dropped
    // like any other and the entry below it is popped instead: same rule as the oracle), else a popped entry, else done
    __device__ __forceinline__ void advance()
    {
        const uint32_t cnt = cur >> 29;
        if (cnt > 2) { cur = ((cur & kIndexMask) + 2) | ((cnt - 2) << 29); return; }
        const bool keep = (near_e != kNoNear) & (sp < kStackMax);
        if (keep) cur = near_e;
        else if (sp == 0) phase = PH_DONE;
        else { --sp; cur = sp < kStackLds ? lds[sp * 64] : spill[sp - kStackLds]; }
        near_e = kNoNear;
        near_d = __builtin_inff();
    }
    // second slot of the pair, evaluated with the CURRENT tmax (after any leaf hit of the first slot)
    __device__ __forceinline__ void second_slot(float tmin, float tmax)
    {
        const bool valid = t1 != RT_CHILD_NONE;
        const bool hit = valid & (k1 >= f1) & (f1 <= tmax) & (k1 >= tmin);
        box_tests += valid ? 1u : 0u;
        const bool is_leaf = hit & (t1 == RT_CHILD_TRI);
        inner_hit(hit & !is_leaf, e1, f1);
        if (is_leaf) { leaf = e1; phase = PH_LEAF1; }
    }
};

// IntersectRayAabb without the tmax/tmin comparisons (Tracer.cu:187-197): front/back of one slot
__device__ __forceinline__ void slab(const uint4& a, const uint4& b, const Ray& r, float& front, float& back)
{
    // x and y go through v_pk_add_f32 / v_pk_mul_f32 (two IEEE f32 ops per instruction, same rounding)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f o2 = {r.ox, r.oy}, i2 = {r.ix, r.iy};
    const v2f lo = {__uint_as_float(a.x), __uint_as_float(a.y)}, hi = {__uint_as_float(b.x), __uint_as_float(b.y)};
    const v2f t1 = (lo - o2) * i2, t2 = (hi - o2) * i2;
    const float t1z = (__uint_as_float(a.z) - r.oz) * r.iz, t2z = (__uint_as_float(b.z) - r.oz) * r.iz;
    front = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1z, t2z));
    back = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1z, t2z));
}

template <bool PF>
__device__ __forceinline__ void prefetch_pair(const TraceParams& p, Trav& t)
{
    if constexpr (PF) {
        if (t.phase == PH_STEP) {
            const uint4* np = reinterpret_cast<const uint4*>(p.nodes + (t.cur & kIndexMask));
            const int o1 = (t.cur >> 29) > 1 ? 2 : 0;
            t.pf0 = np[0]; t.pf1 = np[1]; t.pf2 = np[o1]; t.pf3 = np[o1 + 1];
        }
    }
}

// One box step of a lane (Tracer.cu:323-352 for one pair): both slots of the current pair are loaded and both slabs
// computed before the ordered tmax compares; a leaf in the first slot parks the lane with the second slot's slab kept.
template <bool PF>
__device__ __forceinline__ void box_step(const TraceParams& p, const Ray& r, Trav& t)
{
    const uint32_t cnt = t.cur >> 29;
    const uint4* np = reinterpret_cast<const uint4*>(p.nodes + (t.cur & kIndexMask));
#ifdef RT_EXP_BOX_PAD   // experiment arm (csrc/Makefile librt_amd_exp.so): N extra VALU instructions per box step -- which pipe bounds the kernel?
#pragma unroll
    for (int q = 0; q < RT_EXP_BOX_PAD; q++) asm volatile("v_mov_b32 %0, %0" : "+v"(t.box_tests));
#endif
    const bool two = cnt > 1;
    const int o1 = two ? 2 : 0;  // all four loads issue together; a lone slot is simply read twice
    uint4 a0, b0, a1, b1;
    if constexpr (PF) { a0 = t.pf0; b0 = t.pf1; a1 = t.pf2; b1 = t.pf3; }
    else { a0 = np[0]; b0 = np[1]; a1 = np[o1]; b1 = np[o1 + 1]; }
    float f0, k0;
    slab(a0, b0, r, f0, k0);
    slab(a1, b1, r, t.f1, t.k1);
    t.e1 = (b1.w & kIndexMask) | (a1.w & ~kIndexMask);
    t.t1 = two ? (b1.w >> 29) : (uint32_t)RT_CHILD_NONE;
    const uint32_t type0 = b0.w >> 29;
    const uint32_t e0 = (b0.w & kIndexMask) | (a0.w & ~kIndexMask);
    const bool valid0 = type0 != RT_CHILD_NONE;
    const bool hit0 = valid0 & (k0 >= f0) & (f0 <= r.tmax) & (k0 >= r.tmin);
    t.box_tests += valid0 ? 1u : 0u;
    const bool leaf0 = hit0 & (type0 == RT_CHILD_TRI);
    t.inner_hit(hit0 & !leaf0, e0, f0);
    if (leaf0) { t.leaf = e0; t.phase = PH_LEAF0; }
    else {
        t.second_slot(r.tmin, r.tmax);
        if (t.phase == PH_STEP) { t.advance(); prefetch_pair<PF>(p, t); }
    }
}

// Tracer.cu:308-374, restructured as described in the file header.  Returns tri_hit.
// steps[0] / steps[1] count the wave's box-phase / leaf-phase iterations (profiling aid).
template <bool PF>
__device__ __forceinline__ bool trace_ray(const TraceParams& p, Ray& r, Hit& h, Trav& t, bool active, uint32_t* steps)
{
    t.sp = 0;
    t.cur = (p.root & kIndexMask) | (p.count << 29);
    t.near_e = kNoNear;
    t.near_d = __builtin_inff();
    t.phase = (active && p.count > 0) ? PH_STEP : PH_DONE;
    t.box_tests = 0;
    t.tri_tests = 0;
    t.t1 = 0;
    t.e1 = 0;
    t.f1 = t.k1 = 0.0f;
    t.leaf = 0;
    prefetch_pair<PF>(p, t);
    bool tri_hit = false;
    uint32_t nbox = 0, nleaf = 0;
#ifdef RT_EXP_UNIFORM_STATS
    uint32_t ustat[4] = {0, 0, 0, 0};
#endif

    while (true) {
        // ---------------------------------------------------- box phase: step while enough lanes want to
        uint64_t stepping, parked;
        while (true) {
            stepping = __builtin_amdgcn_ballot_w64(t.phase == PH_STEP);
            parked = __builtin_amdgcn_ballot_w64((t.phase - 1u) < 2u);
            if (stepping == 0 || __popcll(stepping) * p.park_den < __popcll(parked) * p.park_num) break;
            nbox += 2;   // two box steps per vote (below)
#ifdef RT_EXP_UNIFORM_STATS   // experiment: how often do the stepping lanes of a wave sit on <= 1 / 2 / 4 distinct pairs?  (steps[1] = histogram packed 16 bits each)
            {
                uint64_t rest = stepping;
                int distinct = 0;
                while (rest && distinct < 5) {
                    const int l0 = __ffsll((unsigned long long)rest) - 1;
                    const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)t.cur, l0);
                    rest &= ~__builtin_amdgcn_ballot_w64(t.phase == PH_STEP && t.cur == c0);
                    distinct++;
                }
                ustat[distinct <= 1 ? 0 : (distinct == 2 ? 1 : (distinct <= 4 ? 2 : 3))]++;
            }
#endif
            if (t.phase == PH_STEP) box_step<PF>(p, r, t);
            // second step under the same vote: halves the per-step loop overhead (ballots, branch, copies)
            if (t.phase == PH_STEP) box_step<PF>(p, r, t);
        }
        if ((stepping | parked) == 0) break;
        // ---------------------------------------------------- leaf phase (Tracer.cu:333-337, 293-306)
        nleaf++;
        if ((t.phase - 1u) < 2u) {
            t.tri_tests++;
            const uint32_t li = t.leaf & kIndexMask;
            const uint4* tp = reinterpret_cast<const uint4*>(p.leaves + li);
            const uint4 l0 = tp[0], l1 = tp[1], l2 = tp[2], l3 = tp[3];
            bool hit_tri = intersect_tri(__uint_as_float(l0.x), __uint_as_float(l0.y), __uint_as_float(l0.z),
                                         __uint_as_float(l1.x), __uint_as_float(l1.y), __uint_as_float(l1.z),
                                         __uint_as_float(l2.x), __uint_as_float(l2.y), __uint_as_float(l2.z),
                                         r, h, li << 1, l0.w);
            // triangle B = (v2, v1, v3) is requested whenever count > 0; for a single triangle v3 == v2
            // bit for bit, B's edge2 is exactly 0, a == 0 and the reference rejects it: skipped, same result.
            if ((t.leaf >> 29) > 0 && (l3.x != l2.x || l3.y != l2.y || l3.z != l2.z))
                hit_tri |= intersect_tri(__uint_as_float(l2.x), __uint_as_float(l2.y), __uint_as_float(l2.z),
                                         __uint_as_float(l1.x), __uint_as_float(l1.y), __uint_as_float(l1.z),
                                         __uint_as_float(l3.x), __uint_as_float(l3.y), __uint_as_float(l3.z),
                                         r, h, (li << 1) + 1, l1.w);
            tri_hit |= hit_tri;
            const bool was_first = t.phase == PH_LEAF0;
            t.phase = PH_STEP;
            if (was_first) t.second_slot(r.tmin, r.tmax);
            if (t.phase == PH_STEP) { t.advance(); prefetch_pair<PF>(p, t); }
        }
    }
#ifdef RT_EXP_UNIFORM_STATS
    steps[0] += ustat[RT_EXP_UNIFORM_STATS == 1 ? 0 : 2];   // votes whose stepping lanes sit on 1 (arm 1) / 3-4 (arm 2) distinct pairs
    steps[1] += ustat[RT_EXP_UNIFORM_STATS == 1 ? 1 : 3];   // ... 2 (arm 1) / more than 4 (arm 2)
    (void)nbox; (void)nleaf;
#elif !defined(RT_TRACE_NO_STEPS)
    steps[0] += nbox;
    steps[1] += nleaf;
#endif
    return tri_hit;
}

__device__ __forceinline__ float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); }

// Tracer.cu:15-41 (float rgb 0..255 before the uchar truncation)
__device__ __forceinline__ void hsv_to_rgb255(float h, float s, float v, float& R, float& G, float& B)
{
    h = clampf(h, 0.f, 1.f) * 360.0f;
    s = clampf(s, 0.f, 1.f);
    v = clampf(v, 0.f, 1.f);
    const float c = s * v;
    const float x = c * (1 - fabsf(((int)h % 120) / 60.0f - 1));
    const float m = v - c;
    float r, g, b;
    if (h >= 0 && h < 60) { r = c; g = x; b = 0; }
    else if (h >= 60 && h < 120) { r = x; g = c; b = 0; }
    else if (h >= 120 && h < 180) { r = 0; g = c; b = x; }
    else if (h >= 180 && h < 240) { r = 0; g = x; b = c; }
    else if (h >= 240 && h < 300) { r = x; g = 0; b = c; }
    else { r = c; g = 0; b = x; }
    R = (r + m) * 255; G = (g + m) * 255; B = (b + m) * 255;
}

// ---------------------------------------------------------------------------------------------
// float -> unsigned char as CUDA converts it (cvt.rzi.u8.f32: NaN -> 0, saturating); bilinear weights at a texture
// border leave [0, 255]
__device__ __forceinline__ uint32_t sat_u8(float v) { return !(v > 0.0f) ? 0u : (v >= 255.0f ? 255u : (uint32_t)v); }
struct F2 { float x, y; };
struct U8x4 { uint32_t c[4]; };
__device__ __forceinline__ float fracf1(float v) { return v - floorf(v); }   // helper_math.h:1367

// Sample(Texture&, int2, lod) (Tracer.cu:103-108)
__device__ __forceinline__ void tex_fetch(const rt_texture& t, int x, int y, int lod, float out[4])
{
    const int sx = t.size_x[lod], sy = t.size_y[lod];
    x = max(0, min(x, sx - 1));
    y = max(0, min(y, sy - 1));
    const uint32_t w = t.mips[lod][(size_t)y * sx + x];
    out[0] = (float)(w & 255u); out[1] = (float)((w >> 8) & 255u); out[2] = (float)((w >> 16) & 255u); out[3] = (float)(w >> 24);
}
// BilinearSample (Tracer.cu:122-140)
__device__ __forceinline__ U8x4 bilinear_sample(const rt_texture& t, F2 uv, int lod)
{
    float cx = fracf1(uv.x) * (float)t.size_x[lod] - 0.5f;
    float cy = fracf1(uv.y) * (float)t.size_y[lod] - 0.5f;
    cy = (float)t.size_y[lod] - cy;
    const int ix = (int)cx, iy = (int)cy;
    const float dx = cx - (float)ix, dy = cy - (float)iy;
    const float w0 = (1.0f - dx) * dy, w1 = dx * dy, w2 = (1.0f - dx) * (1.0f - dy), w3 = dx * (1.0f - dy);
    float s0[4], s1[4], s2[4], s3[4];
    tex_fetch(t, ix, iy, lod, s0);
    tex_fetch(t, ix + 1, iy, lod, s1);
    tex_fetch(t, ix, iy - 1, lod, s2);
    tex_fetch(t, ix + 1, iy - 1, lod, s3);
    U8x4 o;
#pragma unroll
    for (int c = 0; c < 4; c++) o.c[c] = sat_u8(((s0[c] * w0 + s1[c] * w1) + s2[c] * w2) + s3[c] * w3);
    return o;
}
// TrilinearSample (Tracer.cu:142-155)
__device__ __forceinline__ U8x4 trilinear_sample(const rt_texture& t, F2 uv, float lod)
{
    uint32_t min_lod = (uint32_t)floorf(lod), max_lod = min_lod + 1;
    min_lod = min(min_lod, t.max_lod);
    max_lod = min(max_lod, t.max_lod);
    const U8x4 a = bilinear_sample(t, uv, (int)min_lod), b = bilinear_sample(t, uv, (int)max_lod);
    const float frac = fracf1(lod);
    U8x4 o;
#pragma unroll
    for (int c = 0; c < 4; c++) o.c[c] = sat_u8((float)a.c[c] * (1.0f - frac) + (float)b.c[c] * frac);
    return o;
}
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 v3(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vscale(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 vnormalize(V3 v) { return vscale(v, 1.0f / sqrtf(vdot(v, v))); }

struct Surface {   // what the textured shaders need about the hit
    V3 tri[3];
    float uv[3][2];
    V3 n[3];
};
__device__ __forceinline__ F2 interp_uv(const Surface& s, float bu, float bv)   // InterpolateUVs (Tracer.cu:43-48)
{
    const float w0 = 1 - bu - bv;
    return {(s.uv[0][0] * w0 + s.uv[1][0] * bu) + s.uv[2][0] * bv, (s.uv[0][1] * w0 + s.uv[1][1] * bu) + s.uv[2][1] * bv};
}
// ComputeLOD (Tracer.cu:237-254) with RayTriangleGradients (:202-235) inlined
__device__ __forceinline__ float compute_lod(const Ray& r, const Hit& h, float spread, const Surface& s, const rt_texture& tex)
{
    const V3 o = v3(r.ox, r.oy, r.oz), d = v3(r.dx, r.dy, r.dz);
    const V3 edge1 = vsub(s.tri[1], s.tri[0]), edge2 = vsub(s.tri[2], s.tri[0]);
    const V3 sv = vsub(o, s.tri[0]);
    const V3 q = vcross(sv, edge1);
    const V3 x = vscale(vscale(vnormalize(vcross(d, v3(0, 1, 0))), r.tmax), spread);
    const V3 y = vscale(vscale(vnormalize(vcross(d, x)), r.tmax), spread);
    const V3 hit_point = vadd(o, vscale(d, r.tmax));
    const V3 dirx = vnormalize(vsub(vadd(hit_point, x), o)), diry = vnormalize(vsub(vadd(hit_point, y), o));
    const V3 h0 = vcross(dirx, edge2);
    const float f0 = 1.0f / vdot(edge1, h0);
    const float bu0 = f0 * vdot(sv, h0), bv0 = f0 * vdot(dirx, q);
    const V3 h1 = vcross(diry, edge2);
    const float f1 = 1.0f / vdot(edge1, h1);
    const float bu1 = f1 * vdot(sv, h1), bv1 = f1 * vdot(diry, q);
    const F2 uvs = interp_uv(s, h.bu, h.bv), ux = interp_uv(s, bu0, bv0), uy = interp_uv(s, bu1, bv1);
    const float sx = (float)tex.size_x[0], sy = (float)tex.size_y[0];
    const float dxx = fabsf(ux.x - uvs.x) * sx, dxy = fabsf(ux.y - uvs.y) * sy;
    const float dyx = fabsf(uy.x - uvs.x) * sx, dyy = fabsf(uy.y - uvs.y) * sy;
    const float max_change = fmaxf(sqrtf(dxx * dxx + dxy * dxy), sqrtf(dyx * dyx + dyy * dyy));
    return fmaxf(0.0f, fminf(rt_log2f(max_change), (float)tex.max_lod));   // log2f: rt_math.h (bit-identical to the oracle)
}
// TangentMatrix (Tracer.cu:84-101)
__device__ __forceinline__ void tangent_matrix(const Surface& s, V3 rows[3])
{
    const V3 e1 = vsub(s.tri[1], s.tri[0]), e2 = vsub(s.tri[2], s.tri[0]);
    const float d1x = s.uv[1][0] - s.uv[0][0], d1y = s.uv[1][1] - s.uv[0][1];
    const float d2x = s.uv[2][0] - s.uv[0][0], d2y = s.uv[2][1] - s.uv[0][1];
    const float f = 1.0f / (d1x * d2y - d1y * d2x);
    const V3 normal = vnormalize(vcross(e1, e2));
    const V3 tangent = vnormalize(vscale(vsub(vscale(e1, d2y), vscale(e2, d1y)), f));
    const V3 bitangent = vnormalize(vscale(vsub(vscale(e2, d1x), vscale(e1, d2x)), f));
    rows[0] = v3(tangent.x, bitangent.x, normal.x);
    rows[1] = v3(tangent.y, bitangent.y, normal.y);
    rows[2] = v3(tangent.z, bitangent.z, normal.z);
}
// Bump2Normal (Tracer.cu:157-185)
__device__ __forceinline__ V3 bump2normal(const rt_texture& tex, const V3 tbn[3], F2 uv, float lod)
{
    const float texel_step = rt_exp2f(lod);   // powf(2.0f, lod): rt_math.h
    const float stx = texel_step / (float)tex.size_x[0], sty = texel_step / (float)tex.size_y[0];
    const U8x4 a = trilinear_sample(tex, F2{uv.x - stx * 0.5f, uv.y - sty * 0.5f}, lod);
    const U8x4 b = trilinear_sample(tex, F2{uv.x + stx * 0.5f, uv.y + 0.0f}, lod);
    const U8x4 c = trilinear_sample(tex, F2{uv.x + 0.0f, uv.y + sty * 0.5f}, lod);
    const float gx = (float)b.c[0] - (float)a.c[0], gy = (float)c.c[0] - (float)a.c[0];
    const float d = 4.0f;
    V3 n = vnormalize(vcross(v3(1, 0, d * gx / (texel_step * 256.0f)), v3(0, 1, d * gy / (texel_step * 256.0f))));
    n = v3(vdot(tbn[0], n), vdot(tbn[1], n), vdot(tbn[2], n));
    return vnormalize(n);
}

constexpr bool render_is_lit(int r) { return r == RT_RENDER_DIFFUSE || r == RT_RENDER_TEXTURE_LIT || r == RT_RENDER_TEXTURE_LIT_SHADOWS; }
constexpr bool render_uses_surface(int r) { return r == RT_RENDER_LODS || r == RT_RENDER_TEXTURE || r == RT_RENDER_TEXTURE_LIT || r == RT_RENDER_TEXTURE_LIT_SHADOWS; }

// one sample of one pixel -> float colour 0..255 per channel + alpha (TraceRays body, Tracer.cu:482-593).
// Every lane of the wave calls this (inactive lanes trace nothing) because trace_ray votes with ballots; the shadow
// ray of kTextureLitShadows is a second wave-level traversal over the lanes that hit something.
template <int RENDER, bool PF>
__device__ __forceinline__ void shade_sample(const TraceParams& p, const rt_camera& cam, uint32_t x, uint32_t y,
                                             float ox, float oy, Trav& t, bool active, uint32_t& box_acc,
                                             uint32_t& tri_acc, uint32_t* steps, float& R, float& G, float& B, float& A)
{
    const float ndcx = 2 * (((float)x + ox) / (float)p.w) - 1;
    const float ndcy = 2 * (((float)y + oy) / (float)p.h) - 1;
    const float px = (ndcx * cam.u.x + ndcy * cam.v.x) + 1.0f * cam.w.x;
    const float py = (ndcx * cam.u.y + ndcy * cam.v.y) + 1.0f * cam.w.y;
    const float pz = (ndcx * cam.u.z + ndcy * cam.v.z) + 1.0f * cam.w.z;
    const float inv_len = 1.0f / sqrtf(px * px + py * py + pz * pz);  // normalize = v * rsqrtf(dot) (helper_math.h:1318)
    const float max_depth = cam.max_depth;
    Ray r;
    r.dx = px * inv_len; r.dy = py * inv_len; r.dz = pz * inv_len;
    r.ox = cam.position.x; r.oy = cam.position.y; r.oz = cam.position.z;
    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    r.tmin = 0.00001f;
    r.tmax = max_depth;
    Hit h = {0u, 0u, 0.f, 0.f};
    const bool hit = trace_ray<PF>(p, r, h, t, active, steps);
    const uint32_t box_tests = t.box_tests, tri_tests = t.tri_tests;
    box_acc += box_tests;
    tri_acc += tri_tests;
    const float depth = hit ? r.tmax : 0.0f;
    R = G = B = 0;
    A = 255.0f;

    if (RENDER == RT_RENDER_DEPTH) {
        R = G = B = fminf(1.0f, depth / max_depth) * 255;
        return;
    }
    if (RENDER == RT_RENDER_BOXTESTS) {
        G = B = fminf(box_tests / 180.0f, 1.0f) * 255;
        return;
    }
    if (RENDER == kRenderDebugBoxCount) {  // tuning aid: R carries the raw count (bit pattern), see trace_kernel
        R = __uint_as_float(box_tests);
        return;
    }
    if (RENDER == RT_RENDER_TRIANGLE_TESTS) {
        const float g = fminf(tri_tests / 32.0f, 1.0f);
        R = g * 100; G = g * 255; B = g * 100;
        return;
    }
    // ---- modes that look at the surface.  The reference fetches attributes / material before testing `hit`
    // (Tracer.cu:506-509: primitive 0 on a miss); only kLODs depends on that (magenta unless textured AND hit).
    const bool lit = active && hit;
    const bool fetch = active && (hit || RENDER == RT_RENDER_LODS);
    rt_material mat = {};
    Surface s = {};
    int material_id = 0;
    if (fetch) {
        // RotateAttributes (Tracer.cu:57-82)
        const rt_triangle_pair* pair = p.leaves + (h.tri_id >> 1);
        const bool second = h.tri_id & 1;
        const uint32_t rot = second ? pair->rotations[1] : pair->rotations[0];
        const rt_attributes* at = p.attributes + h.primitive_id;
        const int i0 = rot == 1 ? 2 : (rot == 2 ? 1 : 0);
        const int i1 = rot == 1 ? 0 : (rot == 2 ? 2 : 1);
        const int i2 = rot == 1 ? 1 : (rot == 2 ? 0 : 2);
        material_id = at->material_id;
        // indices from scene data are range-checked (the reference is not: FileIO.cpp:191 gives faces before the first
        // usemtl material_id -1): an id outside the table shades as material 0, a texture index outside the texture
        // table reads as -1 (untextured)
        mat = p.materials[(uint32_t)material_id < p.num_materials ? (uint32_t)material_id : 0u];
        if ((uint32_t)mat.texture >= p.num_textures) mat.texture = -1;
        if ((uint32_t)mat.bump >= p.num_textures) mat.bump = -1;
        if ((uint32_t)mat.disp >= p.num_textures) mat.disp = -1;
        const rt_float3 n0 = at->normal[i0], n1 = at->normal[i1], n2 = at->normal[i2];
        s.n[0] = v3(n0.x, n0.y, n0.z); s.n[1] = v3(n1.x, n1.y, n1.z); s.n[2] = v3(n2.x, n2.y, n2.z);
        if (render_uses_surface(RENDER)) {
            s.uv[0][0] = at->uv[i0][0]; s.uv[0][1] = at->uv[i0][1];
            s.uv[1][0] = at->uv[i1][0]; s.uv[1][1] = at->uv[i1][1];
            s.uv[2][0] = at->uv[i2][0]; s.uv[2][1] = at->uv[i2][1];
            const rt_float3 a = second ? pair->v2 : pair->v0, b = pair->v1, c = second ? pair->v3 : pair->v2;
            s.tri[0] = v3(a.x, a.y, a.z); s.tri[1] = v3(b.x, b.y, b.z); s.tri[2] = v3(c.x, c.y, c.z);
        }
    }
    const float spread = 2.0f / p.w;
    if (RENDER == RT_RENDER_LODS) {                       // Tracer.cu:543-556
        if (!active) return;
        if (mat.texture != -1 && hit) {
            const float lod = compute_lod(r, h, spread, s, p.textures[mat.texture]);
            R = G = B = A = (float)(((uint32_t)((int)lod * 20)) & 255u);
        } else {
            R = 255; G = 0; B = 255;
        }
        return;
    }
    if (RENDER == RT_RENDER_MATERIAL_ID) {
        if (lit) hsv_to_rgb255((float)material_id / p.num_materials, 1.0f, 1.0f, R, G, B);
        return;
    }
    if (RENDER == RT_RENDER_TEXTURE) {                    // Tracer.cu:557-578
        if (!lit) return;
        if (mat.texture != -1) {
            const rt_texture& tex = p.textures[mat.texture];
            const float lod = compute_lod(r, h, spread, s, tex);
            const U8x4 c = trilinear_sample(tex, interp_uv(s, h.bu, h.bv), lod);
            R = (float)c.c[0]; G = (float)c.c[1]; B = (float)c.c[2]; A = (float)c.c[3];
        } else {
            R = mat.diffuse.x * 255; G = mat.diffuse.y * 255; B = mat.diffuse.z * 255;
        }
        return;
    }
    // ---- AmbientShader (Tracer.cu:376-469): kDiffuse (no textures), kTextureLit (textures + bump), kTextureLitShadows
    constexpr bool use_textures = RENDER == RT_RENDER_TEXTURE_LIT || RENDER == RT_RENDER_TEXTURE_LIT_SHADOWS;
    constexpr bool use_bump = use_textures;
    constexpr bool use_shadows = RENDER == RT_RENDER_TEXTURE_LIT_SHADOWS;
    const float hx = r.ox + r.dx * r.tmax, hy = r.oy + r.dy * r.tmax, hz = r.oz + r.dz * r.tmax;
    float lx = p.light[0] - hx, ly = p.light[1] - hy, lz = p.light[2] - hz;
    const float to_light = sqrtf(lx * lx + ly * ly + lz * lz);      // length(light_pos - hit_pos) (:455)
    const float linv = 1.0f / to_light;
    lx *= linv; ly *= linv; lz *= linv;
    bool shadowed = false;
    if (use_shadows) {                                    // (:447-462) a second traversal, wave-wide
        Ray sr;
        sr.ox = hx; sr.oy = hy; sr.oz = hz;
        sr.dx = lx; sr.dy = ly; sr.dz = lz;
        sr.ix = 1.0f / lx; sr.iy = 1.0f / ly; sr.iz = 1.0f / lz;
        sr.tmin = 0.001f;
        sr.tmax = to_light;
        Hit sh = {0u, 0u, 0.f, 0.f};
        shadowed = trace_ray<PF>(p, sr, sh, t, lit, steps);   // its test counts are not reported (shadow_stats, :451)
    }
    if (!lit) return;
    const float w0 = 1 - h.bu - h.bv;
    V3 n = vadd(vadd(vscale(s.n[0], w0), vscale(s.n[1], h.bu)), vscale(s.n[2], h.bv));   // InterpolateNormals (:50-56)
    if (use_bump && mat.disp != -1) {                     // displacement map read as a normal map (:388-403)
        const rt_texture& disp = p.textures[mat.disp];
        const float lod = compute_lod(r, h, spread, s, disp);
        V3 tbn[3];
        tangent_matrix(s, tbn);
        const U8x4 smp = trilinear_sample(disp, interp_uv(s, h.bu, h.bv), lod);
        n = v3((float)smp.c[0] / 255.0f, (float)smp.c[1] / 255.0f, (float)smp.c[2] / 255.0f);
        n = vnormalize(v3(n.x * 2.0f - 1.0f, n.y * 2.0f - 1.0f, n.z * 2.0f - 1.0f));
        n = vnormalize(v3(vdot(tbn[0], n), vdot(tbn[1], n), vdot(tbn[2], n)));
    } else if (use_bump && mat.bump != -1) {              // (:405-415)
        const rt_texture& bump = p.textures[mat.bump];
        const float lod = compute_lod(r, h, spread, s, bump);
        V3 tbn[3];
        tangent_matrix(s, tbn);
        n = bump2normal(bump, tbn, interp_uv(s, h.bu, h.bv), lod);
    }
    const float nx = n.x, ny = n.y, nz = n.z;
    const float lcx = 1.0f, lcy = 0.9f, lcz = 0.8f;
    float dterm = 1.0f * fmaxf(nx * lx + ny * ly + nz * lz, 0.0f);
    // reflect(-l, n) = -l - 2.0f * n * dot(n, -l)   (helper_math.h:1435-1438)
    const float nlx = -lx, nly = -ly, nlz = -lz;
    const float ndl = nx * nlx + ny * nly + nz * nlz;
    const float rx = nlx - (nx * 2.0f) * ndl, ry = nly - (ny * 2.0f) * ndl, rz = nlz - (nz * 2.0f) * ndl;
    // pow(max(dot(-dir, refl), 0.0), Ns): double max, double pow, narrowed by operator*(float, float3)
    const double sb = fmax((double)((-r.dx) * rx + (-r.dy) * ry + (-r.dz) * rz), 0.0);
    float sp = (float)(1.0f * rt_pow_d(sb, (double)mat.specular_exp));   // pow: rt_math.h
    float odx = mat.diffuse.x, ody = mat.diffuse.y, odz = mat.diffuse.z;
    if (use_textures && mat.texture != -1) {              // (:432-445): BilinearSample(tex, uv, (int)lod)
        const rt_texture& tex = p.textures[mat.texture];
        const float lod = compute_lod(r, h, spread, s, tex);
        const U8x4 smp = bilinear_sample(tex, interp_uv(s, h.bu, h.bv), (int)lod);
        odx = (float)smp.c[0] / 255; ody = (float)smp.c[1] / 255; odz = (float)smp.c[2] / 255;
    }
    float dfx = lcx * dterm, dfy = lcy * dterm, dfz = lcz * dterm;
    float spx = lcx * sp, spy = lcy * sp, spz = lcz * sp;
    if (shadowed) { dfx = dfy = dfz = 0.0f; spx = spy = spz = 0.0f; }
    const float cr = (dfx * odx + (lcx * 0.2f) * mat.ambient.x) + spx * mat.specular.x;
    const float cg = (dfy * ody + (lcy * 0.2f) * mat.ambient.y) + spy * mat.specular.y;
    const float cb = (dfz * odz + (lcz * 0.2f) * mat.ambient.z) + spz * mat.specular.z;
    R = clampf(cr, 0.0f, 1.0f) * 255;
    G = clampf(cg, 0.0f, 1.0f) * 255;
    B = clampf(cb, 0.0f, 1.0f) * 255;
}

// PF (pair prefetch): for trees that do not fit the caches.  On the 10M-triangle scene (1.28 GB of nodes + leaves, 5 x the
// Infinity Cache; L2 hit 94 %) a wave waits for L2 misses, not for the address path: issuing the NEXT pair's loads right
// after advance() -- one more pair in flight per lane across the wave's vote -- at 96 VGPRs / 5 waves per SIMD is +8 % serial,
// +16 % with frames in flight on camera A, +6 / +14 % on camera B; on the cache-resident 1M tree the same kernel is 25 % SLOWER
// (five waves instead of eight feed the address path), at 4.5M -7 % / +6 %; at 64 - 80 VGPRs the sixteen extra registers
// spill and it is 2 - 5 x slower everywhere (profiles/r04_trace_10m_experiments.txt).  launch_trace picks it by scene size.
template <int RENDER, bool PF>
// kDepth / kBoxtests / kTriangleTests end in a one-line colour conversion: they fit 64 VGPRs (8 waves per SIMD); the
// shading of the other render types would spill there, they keep 72 VGPRs (7 waves)
// (RT_TRACE_LEAN_EXTRA = 0 and RT_TRACE_NO_STEPS are the compile-time arms of tools/trace_spill_experiment.sh, which priced
// the spills of the 64-VGPR instantiations: see DESIGN section 5)
#ifndef RT_TRACE_LEAN_EXTRA
#define RT_TRACE_LEAN_EXTRA 1
#endif
#ifndef RT_TRACE_PF_WAVES
#define RT_TRACE_PF_WAVES 5
#endif
__global__ __launch_bounds__(kTraceWaves * 64, PF ? RT_TRACE_PF_WAVES : ((RENDER <= 2 || RENDER == kRenderDebugBoxCount) ? RT_TRACE_MIN_WAVES + RT_TRACE_LEAN_EXTRA : RT_TRACE_MIN_WAVES))
void trace_kernel(TraceParams p)
{
    __shared__ uint32_t stack_lds[kTraceWaves][kStackLds][64];
    __shared__ unsigned long long csum[4];   // the workgroup's test counters (see the end of the kernel)
    __shared__ uint32_t carrive;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.counters) {                         // (kernel argument: the same for every thread)
        if (threadIdx.x < 4) csum[threadIdx.x] = 0ull;
        if (threadIdx.x == 4) carrive = 0u;
        __syncthreads();
    }

    // XCD-aware remap: hardware deals workgroup b to XCD b % 8; give XCD x a contiguous run of tiles.
    // Tile order vs XCDs: hardware deals workgroup b to XCD b % 8.  XCD x takes chunks of kXcdChunk consecutive workgroups
    // (8 x 4 tiles = a 256 x 8 pixel run: neighbouring rays meet in one L2) dealt round-robin over the whole frame, so
    // every XCD sees every region of the image.  (Round 1 gave each XCD one contiguous eighth of the frame: fine for a
    // uniform view, but a view whose cost is concentrated in part of the frame -- camera B: the foreground rows -- then
    // runs on two or three XCDs while the others idle: 1297 vs 3490 Mrays/s serial, 2548 vs 5397 with frames in flight.)
    // RT_TRACE_XCD_CHUNK overrides the chunk at compile time (tools/xcd_chunk_experiment.sh).
#ifndef RT_TRACE_XCD_CHUNK
#define RT_TRACE_XCD_CHUNK 8
#endif
    const uint32_t nb = gridDim.x, bid = blockIdx.x;
    constexpr uint32_t kXcdChunk = RT_TRACE_XCD_CHUNK;
    const uint32_t full = nb / (8u * kXcdChunk) * (8u * kXcdChunk);   // (the last < 8 * chunk workgroups keep their index)
    const uint32_t xcd = bid & 7u, loc = bid >> 3;
    const uint32_t vb = bid < full ? (loc / kXcdChunk) * (8u * kXcdChunk) + xcd * kXcdChunk + (loc % kXcdChunk) : bid;
    const uint32_t tile = vb * kTraceWaves + wave;

    // lane -> pixel inside the 8x8 tile, Morton order
    const uint32_t lx = (lane & 1) | ((lane >> 1) & 2) | ((lane >> 2) & 4);
    const uint32_t ly = ((lane >> 1) & 1) | ((lane >> 2) & 2) | ((lane >> 3) & 4);
    const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const uint32_t x = tx * 8 + lx;
    uint32_t y = p.y0 + ty * 8 + ly, out_y = y;      // row band: rendered in place
    if (p.strip_tiles) {                            // interleaved strips: rendered compactly
        const uint32_t strip = p.strip_first + (ty / p.strip_tiles) * p.strip_stride;
        y = (strip * p.strip_tiles + ty % p.strip_tiles) * 8 + ly;
        out_y = ty * 8 + ly;
    }
    const bool active = tile < p.num_tiles && x < p.w && y < p.y1;

    const rt_camera cam = *p.camera;
    SpillArray spill;
    Trav t;
    t.lds = (lds_u32*)&stack_lds[wave][0][lane];
    t.spill = spill;
    uint32_t box_acc = 0, tri_acc = 0;
    uint32_t steps[2] = {0u, 0u};
    float R, G, B, A;
    if (p.spp <= 1) {
        shade_sample<RENDER, PF>(p, cam, x, y, 0.5f, 0.5f, t, active, box_acc, tri_acc, steps, R, G, B, A);
    } else {
        float ar = 0, ag = 0, ab = 0, aa = 0;
        const uint32_t side = p.spp == 4 ? 2u : 4u;   // stratified side x side sub-pixel grid (2 x 2 or 4 x 4)
        for (uint32_t s = 0; s < p.spp; s++) {
            const float ox = ((float)(s % side) + 0.5f) / (float)side, oy = ((float)((s / side) % side) + 0.5f) / (float)side;
            shade_sample<RENDER, PF>(p, cam, x, y, ox, oy, t, active, box_acc, tri_acc, steps, R, G, B, A);
            ar += R; ag += G; ab += B; aa += A;
        }
        R = ar / (float)p.spp; G = ag / (float)p.spp; B = ab / (float)p.spp; A = aa / (float)p.spp;
    }
    if (active) {
        uint32_t px = sat_u8(R) | (sat_u8(G) << 8) | (sat_u8(B) << 16) | (sat_u8(A) << 24);
        if (RENDER == kRenderDebugBoxCount) px = __float_as_uint(R);
        reinterpret_cast<uint32_t*>(p.rgba8)[(size_t)out_y * p.w + x] = px;
    }
    if (p.counters) {
        // The workgroup's waves add into LDS, and the LAST wave to finish (an LDS arrival count: no barrier, nobody waits)
        // publishes the four sums (round 2 issued four device atomics per wave on the caller's counters: 130 k queued
        // same-address atomics = 1.4 ms per 1080p frame).
        const uint32_t bsum = wave_sum_u32(box_acc), tsum = wave_sum_u32(tri_acc);  // <= 64 * 2^26: no overflow per wave
        uint32_t last = 0;
        if (lane == 0) {
            atomicAdd(&csum[0], (unsigned long long)bsum);
            atomicAdd(&csum[1], (unsigned long long)tsum);
            atomicAdd(&csum[2], (unsigned long long)steps[0]);  // wave-level box-phase steps (profiling)
            atomicAdd(&csum[3], (unsigned long long)steps[1]);  // wave-level leaf-phase steps
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            last = atomicAdd(&carrive, 1u) == (uint32_t)kTraceWaves - 1u ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (__builtin_amdgcn_readfirstlane((int)last)) {   // the workgroup's last wave: publish (see g_ctr above)
            CtrSlot& S = g_ctr[p.ctr_slot];
            const uint32_t sub = blockIdx.x % kCtrSub;
            if (lane < 4) {
                const unsigned long long v = *(volatile unsigned long long*)&csum[lane];
                // a RETURNING atomic: its value coming back means the add has been performed at the memory side, so the
                // ticket below cannot overtake it
                const unsigned long long old = v ? __hip_atomic_fetch_add(&S.part[sub][lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                asm volatile("" ::"v"((uint32_t)old), "v"((uint32_t)(old >> 32)) : "memory");
            }
            uint32_t fin = 0;
            if (lane == 0) {
                const uint32_t members = (gridDim.x - sub + kCtrSub - 1) / kCtrSub;
                if (__hip_atomic_fetch_add(&S.part[sub][4], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull == members) {
                    const uint32_t rows = min(gridDim.x, kCtrSub);
                    fin = __hip_atomic_fetch_add(&S.top, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull == rows ? 1u : 0u;
                }
            }
            if (__builtin_amdgcn_readfirstlane((int)fin)) {
                // every workgroup of the launch has added its sums: fold the rows (lane = row * 4 + counter), hand them to the
                // caller, leave the slot zeroed for its next user
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                unsigned long long* cell = &S.part[lane >> 2][lane & 3];
                unsigned long long v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane < (int)kCtrSub) __hip_atomic_store(&S.part[lane][4], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lane == 0) __hip_atomic_store(&S.top, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
                if (lane < 4 && v) atomicAdd(&p.counters[lane], v);
            }
        }
    }
}

hipError_t launch_trace(const TraceLaunch& t, hipStream_t st)
{
    if (t.y1 <= t.y0 || t.w == 0) return hipSuccess;
    TraceParams p;
    p.nodes = t.as.nodes;
    p.leaves = t.as.triangles;
    p.attributes = t.scene.attributes;
    p.materials = t.scene.materials;
    p.textures = t.scene.textures;
    p.camera = t.scene.camera;
    p.light[0] = t.scene.light[0]; p.light[1] = t.scene.light[1]; p.light[2] = t.scene.light[2];
    p.root = t.as.root;
    p.count = t.as.count;
    p.num_materials = t.scene.num_materials;
    p.num_textures = t.scene.textures ? t.scene.num_textures : 0;
    p.rgba8 = t.rgba8;
    p.w = t.w; p.h = t.h; p.y0 = t.y0; p.y1 = t.y1; p.spp = t.spp;
    p.counters = reinterpret_cast<unsigned long long*>(t.counters);
    static std::atomic<uint32_t> ctr_seq{0};
    p.ctr_slot = t.counters ? ctr_seq.fetch_add(1u, std::memory_order_relaxed) % kCtrSlots : 0u;
    p.tiles_x = (t.w + 7) / 8;
    uint32_t tiles_y = (t.y1 - t.y0 + 7) / 8;
    p.strip_tiles = t.strip_rows / 8;
    p.strip_first = t.strip_first;
    p.strip_stride = t.strip_stride;
    if (p.strip_tiles) {   // this launch's strips: first, first + stride, ... < ceil(h / strip_rows)
        const uint32_t nstrips = (t.h + t.strip_rows - 1) / t.strip_rows;
        if (t.strip_first >= nstrips) return hipSuccess;
        tiles_y = ((nstrips - t.strip_first + t.strip_stride - 1) / t.strip_stride) * p.strip_tiles;
    }
    p.num_tiles = p.tiles_x * tiles_y;
    p.park_num = kParkNum;
    p.park_den = kParkDen;
#ifdef RT_TRACE_TUNING
    static const int* park = [] {
        static int v[2] = {kParkNum, kParkDen};
        if (const char* e = getenv("RT_TRACE_PARK")) {
            int a = 0, b = 0;
            if (sscanf(e, "%d,%d", &a, &b) == 2 && a > 0 && b > 0) { v[0] = a; v[1] = b; }
        }
        return v;
    }();
    p.park_num = park[0];
    p.park_den = park[1];
#endif
    const uint32_t blocks = (p.num_tiles + kTraceWaves - 1) / kTraceWaves;
    const dim3 grid(blocks), block(kTraceWaves * 64);
    // scene size (DeviceScene::num_attributes, filled by the caller as main.cu:166 does; 0 = unknown): a tree of kPrefetchMinPrims
    // primitives is 1 GB of nodes + leaves, four times the Infinity Cache
    const bool pf = t.scene.num_attributes >= kPrefetchMinPrims;
#define RT_TRACE_CASE(R) case R: if (pf) trace_kernel<R, true><<<grid, block, 0, st>>>(p); else trace_kernel<R, false><<<grid, block, 0, st>>>(p); break;
    switch (t.render_type) {
    RT_TRACE_CASE(RT_RENDER_DEPTH)
    RT_TRACE_CASE(RT_RENDER_BOXTESTS)
    RT_TRACE_CASE(RT_RENDER_TRIANGLE_TESTS)
    RT_TRACE_CASE(RT_RENDER_MATERIAL_ID)
    RT_TRACE_CASE(RT_RENDER_DIFFUSE)
    RT_TRACE_CASE(RT_RENDER_LODS)
    RT_TRACE_CASE(RT_RENDER_TEXTURE)
    RT_TRACE_CASE(RT_RENDER_TEXTURE_LIT)
    RT_TRACE_CASE(RT_RENDER_TEXTURE_LIT_SHADOWS)
#ifdef RT_TRACE_TUNING
    case kRenderDebugBoxCount: trace_kernel<kRenderDebugBoxCount, false><<<grid, block, 0, st>>>(p); break;
#endif
    default: return hipErrorInvalidValue;
    }
#undef RT_TRACE_CASE
    return hipGetLastError();
}

}  // namespace rt
