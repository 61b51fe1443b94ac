/*
 * rt_oracle.c -- see rt_oracle.h.  TEST INFRASTRUCTURE ONLY (checker + timed CPU baseline).
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -shared (oracle/Makefile).
 * All file:line citations are relative to /root/reference/src.
 */
#include "rt_oracle.h"
#include "rt_math.h"   /* gpu-raytracing_amd/csrc: log2f / exp2f / pow as plain IEEE arithmetic, the same text the kernels compile */

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;
void ora_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int ora_get_threads(void) { return g_threads; }

/* ------------------------------------------------------------------ float3 helpers
 * helper_math.h: componentwise + - * / (:990-1013), dot (:1266) = x*x + y*y + z*z left to right,
 * cross (:1444), normalize = v * rsqrtf(dot(v,v)) (:1318) with rsqrtf := 1/sqrtf (:50). */
static inline ora_f3 f3(float x, float y, float z) { ora_f3 r = {x, y, z}; return r; }
static inline ora_f3 add3(ora_f3 a, ora_f3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline ora_f3 sub3(ora_f3 a, ora_f3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline ora_f3 mul3(ora_f3 a, ora_f3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline ora_f3 div3(ora_f3 a, ora_f3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline ora_f3 scale3(ora_f3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
static inline ora_f3 min3(ora_f3 a, ora_f3 b) { return f3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static inline ora_f3 max3(ora_f3 a, ora_f3 b) { return f3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static inline float dot3(ora_f3 a, ora_f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline ora_f3 cross3(ora_f3 a, ora_f3 b)
{
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline ora_f3 normalize3(ora_f3 v)
{
    float inv_len = 1.0f / sqrtf(dot3(v, v));
    return scale3(v, inv_len);
}
/* Triangle::Centre() (Common.cuh:240-242): (v0 + v1 + v2) / 3.0f, componentwise, left to right */
static inline ora_f3 tri_centre(const ora_triangle* t)
{
    ora_f3 c = add3(add3(t->v0, t->v1), t->v2);
    return f3(c.x / 3.0f, c.y / 3.0f, c.z / 3.0f);
}
static inline float clampf(float f, float a, float b) { return fmaxf(a, fminf(f, b)); } /* helper_math.h:1161 */

/* ------------------------------------------------------------------ DeviceUtils.cuh:3-13 */
int32_t ora_float_to_ordered_int(float f)
{
    int32_t i;
    memcpy(&i, &f, 4);
    return (i >= 0) ? i : i ^ 0x7FFFFFFF;
}
float ora_ordered_int_to_float(int32_t i)
{
    int32_t j = (i >= 0) ? i : i ^ 0x7FFFFFFF;
    float f;
    memcpy(&f, &j, 4);
    return f;
}

/* ------------------------------------------------------------------ Multiblock.cu:104-114
 * CalculateSceneAabb: per triangle box, folded with integer atomicMin/Max on the ordered-int
 * encoding (AtomicConvertCombine :34-42) into a box initialised to the ordered-int "empty"
 * {0x7f7fffff x3, 0x80800000 x3} (BuildWrapper.cu:288-289).  Order independent.  The fold is done
 * entirely in the ordered-int domain here (differs from per-triangle fminf only in which of -0/+0
 * survives inside one triangle; the integer order -0 < +0 is what the reference's atomics use). */
void ora_scene_aabb(const ora_triangle* tris, uint32_t n, int32_t out[6])
{
    int32_t lo[3] = {0x7f7fffff, 0x7f7fffff, 0x7f7fffff};
    int32_t hi[3] = {(int32_t)0x80800000, (int32_t)0x80800000, (int32_t)0x80800000};
    const float* f = (const float*)tris;
    for (size_t i = 0; i < (size_t)n * 9; i++) {
        int32_t v = ora_float_to_ordered_int(f[i]);
        int a = (int)(i % 3);
        if (v < lo[a]) lo[a] = v;
        if (v > hi[a]) hi[a] = v;
    }
    out[0] = lo[0]; out[1] = lo[1]; out[2] = lo[2];
    out[3] = hi[0]; out[4] = hi[1]; out[5] = hi[2];
}

/* ------------------------------------------------------------------ BottomUpBuilder.cu:12-32 */
static inline uint32_t expand_bits(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t morton3d(float x, float y, float z)
{
    x = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    y = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    z = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    uint32_t xx = expand_bits((uint32_t)x);
    uint32_t yy = expand_bits((uint32_t)y);
    uint32_t zz = expand_bits((uint32_t)z);
    return xx * 4 + yy * 2 + zz;
}

/* BottomUpBuilder.cu:98-115 GenerateMortonCodes */
void ora_morton_codes(const ora_triangle* tris, uint32_t n, const int32_t aabb[6], uint32_t* codes,
                      uint32_t* values)
{
    ora_f3 smin = f3(ora_ordered_int_to_float(aabb[0]), ora_ordered_int_to_float(aabb[1]),
                     ora_ordered_int_to_float(aabb[2]));
    ora_f3 smax = f3(ora_ordered_int_to_float(aabb[3]), ora_ordered_int_to_float(aabb[4]),
                     ora_ordered_int_to_float(aabb[5]));
    ora_f3 ext = sub3(smax, smin);
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        ora_f3 c = tri_centre(&tris[i]);
        c = div3(sub3(c, smin), ext);
        c = f3(clampf(c.x, 0.0f, 1.0f), clampf(c.y, 0.0f, 1.0f), clampf(c.z, 0.0f, 1.0f));
        codes[i] = morton3d(c.x, c.y, c.z);
        values[i] = (uint32_t)i;
    }
}

/* ------------------------------------------------------------------ RadixSort.cu:171-225
 * Contract: 4 LSD passes x 8 bits, each a stable counting scatter (stable because Distribute ranks
 * equal digits in input order, :153-158); ping-pong keys<->tmp so the result lands back in keys/vals.
 * Parallel form: per-thread chunk histograms, (digit, chunk)-ordered exclusive scan -- the same
 * [digit][segment] table the reference scans (:196-201) with chunks in place of its 128 segments. */
void ora_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tk, uint32_t* tv, uint32_t n)
{
    int T = g_threads;
    uint32_t* hist = (uint32_t*)malloc((size_t)T * 256 * sizeof(uint32_t));
    uint32_t *sk = keys, *sv = vals, *dk = tk, *dv = tv;
    for (int pass = 0; pass < 4; pass++) {
        int shift = pass * 8;
        memset(hist, 0, (size_t)T * 256 * sizeof(uint32_t));
#pragma omp parallel num_threads(T)
        {
#ifdef _OPENMP
            int t = omp_get_thread_num();
#else
            int t = 0;
#endif
            size_t b = (size_t)n * t / T, e = (size_t)n * (t + 1) / T;
            uint32_t* h = hist + (size_t)t * 256;
            for (size_t i = b; i < e; i++) h[(sk[i] >> shift) & 0xFF]++;
#pragma omp barrier
#pragma omp single
            {
                uint32_t sum = 0;
                for (int d = 0; d < 256; d++)
                    for (int c = 0; c < T; c++) {
                        uint32_t v = hist[(size_t)c * 256 + d];
                        hist[(size_t)c * 256 + d] = sum;
                        sum += v;
                    }
            }
            for (size_t i = b; i < e; i++) {
                uint32_t k = sk[i];
                uint32_t pos = h[(k >> shift) & 0xFF]++;
                dk[pos] = k;
                dv[pos] = sv[i];
            }
        }
        uint32_t* x;
        x = sk; sk = dk; dk = x;
        x = sv; sv = dv; dv = x;
    }
    free(hist);
}

/* ------------------------------------------------------------------ BottomUpBuilder.cu:34-38 */
static inline int clz32(uint32_t x) { return x ? __builtin_clz(x) : 32; }
static inline int cpl(const uint32_t* codes, uint32_t i, uint32_t j)
{
    return codes[i] == codes[j] ? 32 + clz32(i ^ j) : clz32(codes[i] ^ codes[j]);
}
static inline int sign_i(int a) { return a >= 0 ? 1 : -1; }

/* BottomUpBuilder.cu:42-68.  `i + lmax*d < count` is an int-vs-unsigned compare in the reference;
 * it is only evaluated after `>= 0` passed, so it equals the plain bounds test used here. */
static void determine_range(const uint32_t* codes, uint32_t count, int i, int* first, int* last)
{
    if (i == 0) { *first = 0; *last = (int)count - 1; return; }
    int d = sign_i(cpl(codes, i, i + 1) - cpl(codes, i, i - 1));
    int cpl_min = cpl(codes, i, i - d);
    int64_t lmax = 2;
    while ((i + lmax * d) >= 0 && (i + lmax * d) < (int64_t)count && cpl(codes, i, (uint32_t)(i + lmax * d)) > cpl_min)
        lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax >> 1; t; t >>= 1) {
        if ((i + (l + t) * d) >= 0 && (i + (l + t) * d) < (int64_t)count &&
            cpl(codes, i, (uint32_t)(i + (l + t) * d)) > cpl_min)
            l += t;
    }
    int j = (int)(i + l * d);
    if (d > 0) { *first = i; *last = j; } else { *first = j; *last = i; }
}

/* BottomUpBuilder.cu:70-96 */
static int find_split(const uint32_t* codes, int first, int last)
{
    int common_prefix = cpl(codes, first, last);
    int split = first;
    int step = last - first;
    do {
        step = (step + 1) >> 1;
        int new_split = split + step;
        if (new_split < last) {
            int split_prefix = cpl(codes, first, new_split);
            if (split_prefix > common_prefix) split = new_split;
        }
    } while (step > 1);
    return split;
}

#define NODE_PARENT_MASK 0x1FFFFFFFu
static inline void set_child_type(ora_node* nd, uint32_t child, uint32_t type) { nd->w28 = (child & NODE_PARENT_MASK) | (type << 29); }
static inline void set_parent(ora_node* nd, uint32_t parent) { nd->w12 = (nd->w12 & ~NODE_PARENT_MASK) | (parent & NODE_PARENT_MASK); }
static inline void set_count(ora_node* nd, uint32_t count) { nd->w12 = (nd->w12 & NODE_PARENT_MASK) | (count << 29); }

/* BottomUpBuilder.cu:167-215 GenerateHierarchy.  Parallel over internal nodes; every 32-bit word is
 * written by exactly one node (child/type word by the owner, parent word of a pair by its parent;
 * count bits are written later by ora_generate_aabbs) so threads never share a word at the same time:
 * the parent word write below is a read-modify-write of w12, whose count bits nobody touches in this
 * phase. */
void ora_generate_hierarchy(ora_node* nodes, uint32_t* leaf_indices, const uint32_t* codes, uint32_t L)
{
    if (L < 2) return;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t ii = 0; ii < (int64_t)L - 1; ii++) {
        uint32_t idx = (uint32_t)ii;
        int first, last;
        determine_range(codes, L, (int)idx, &first, &last);
        int split = find_split(codes, first, last);
        uint32_t child_a = (split == first) ? (uint32_t)split : (uint32_t)split * 2;
        uint32_t type_a = (split == first) ? ORA_TYPE_TRI : ORA_TYPE_BOX;
        uint32_t child_b = (split + 1 == last) ? (uint32_t)split + 1 : ((uint32_t)split + 1) * 2;
        uint32_t type_b = (split + 1 == last) ? ORA_TYPE_TRI : ORA_TYPE_BOX;
        set_child_type(&nodes[idx * 2 + 0], child_a, type_a);
        set_child_type(&nodes[idx * 2 + 1], child_b, type_b);
        if (type_a == ORA_TYPE_BOX) {
            set_parent(&nodes[child_a + 0], idx << 1);
            set_parent(&nodes[child_a + 1], idx << 1);
        } else
            leaf_indices[split] = idx << 1;
        if (type_b == ORA_TYPE_BOX) {
            set_parent(&nodes[child_b + 0], (idx << 1) + 1);
            set_parent(&nodes[child_b + 1], (idx << 1) + 1);
        } else
            leaf_indices[split + 1] = (idx << 1) + 1;
    }
}

/* BottomUpBuilder.cu:287-312 GenerateTriangles (non-pair branch).  Q1: the reference leaves
 * primitive_id_0/1, rotations, pad3 uninitialised; defined here as id = original triangle index,
 * everything else 0.  Q2: the reference's read of triangle index+1 is not reproduced. */
void ora_generate_triangles(const uint32_t* sorted_indices, const ora_triangle* tris, ora_triangle_pair* out, uint32_t L)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t g = 0; g < (int64_t)L; g++) {
        uint32_t index = sorted_indices[g] & 0x7FFFFFFFu;
        ora_triangle_pair r;
        memset(&r, 0, sizeof r);
        r.v0 = tris[index].v0;
        r.v1 = tris[index].v1;
        r.v2 = tris[index].v2;
        r.v3 = r.v2;
        r.primitive_id_0 = index;
        out[g] = r;
    }
}

/* BottomUpBuilder.cu:247-285 GenerateAABBs + UpdateAABB (:217-233).  locks must be zero on entry
 * (BuildWrapper.cu:338).  f3min/f3max are min()/max() on floats = fminf/fmaxf on the device. */
void ora_generate_aabbs(ora_node* nodes, const uint32_t* leaf_indices, const uint32_t* sorted_indices,
                        uint32_t* locks, const ora_triangle_pair* leaves, uint32_t L)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (int64_t g = 0; g < (int64_t)L; g++) {
        uint32_t leaf_index = leaf_indices[g];
        int is_pair = (int)(sorted_indices[g] >> 31);
        ora_f3 bmin = min3(min3(leaves[g].v0, leaves[g].v1), leaves[g].v2);
        ora_f3 bmax = max3(max3(leaves[g].v0, leaves[g].v1), leaves[g].v2);
        if (is_pair) {
            bmin = min3(bmin, leaves[g].v3);
            bmax = max3(bmax, leaves[g].v3);
        }
        nodes[leaf_index].min = bmin;
        nodes[leaf_index].max = bmax;
        set_count(&nodes[leaf_index], 1);
        uint32_t index = leaf_index;
        while (index > 1) {
            /* first arrival at the pair stops, second continues (atomicAdd(&locks[index>>1],1)) */
            uint32_t old = __atomic_fetch_add(&locks[index >> 1], 1u, __ATOMIC_SEQ_CST);
            if (!old) break;
            uint32_t p = nodes[index].w12 & NODE_PARENT_MASK;
            if ((nodes[p].w28 >> 29) == ORA_TYPE_BOX) {
                uint32_t child = nodes[p].w28 & NODE_PARENT_MASK;
                uint32_t right = index & 1;
                bmin = min3(bmin, nodes[child + 1 - right].min);
                bmax = max3(bmax, nodes[child + 1 - right].max);
                nodes[p].min = bmin;
                nodes[p].max = bmax;
            }
            set_count(&nodes[p], 2);
            index = p;
        }
    }
}

/* BuildWrapper.cu:253-348 RunBottomUpBuild (pairs off, hybrid off).  Differences, all documented in
 * nodes are zeroed first (Q3: the reference never initialises nodes_out, so
 * parent of slots 0/1 and every padding bit is defined as 0 here); n < 2 is special-cased (Q8):
 * n == 1 -> slot 0 is the leaf descriptor, slot 1 stays type None; n == 0 -> both None. */
void ora_build(const ora_triangle* tris, uint32_t n, ora_node* nodes, ora_triangle_pair* leaves,
               uint32_t* codes_sorted, uint32_t* indices_sorted, int32_t* aabb_ordered)
{
    uint32_t slots = 2 * (n > 1 ? n - 1 : 1);
    memset(nodes, 0, (size_t)slots * sizeof(ora_node));
    int32_t aabb[6];
    ora_scene_aabb(tris, n, aabb);
    if (aabb_ordered) memcpy(aabb_ordered, aabb, sizeof aabb);
    if (n == 0) return;

    uint32_t* codes = (uint32_t*)malloc((size_t)n * 4);
    uint32_t* vals = (uint32_t*)malloc((size_t)n * 4);
    uint32_t* t1 = (uint32_t*)calloc((size_t)n, 4); /* "locks"        used as sort temp (BuildWrapper.cu:330-332) */
    uint32_t* t2 = (uint32_t*)calloc((size_t)n, 4); /* "leaf_indices" used as sort temp */
    ora_morton_codes(tris, n, aabb, codes, vals);
    ora_radix_sort(codes, vals, t1, t2, n);
    if (codes_sorted) memcpy(codes_sorted, codes, (size_t)n * 4);
    if (indices_sorted) memcpy(indices_sorted, vals, (size_t)n * 4);

    ora_generate_triangles(vals, tris, leaves, n);
    if (n == 1) {
        ora_f3 bmin = min3(min3(leaves[0].v0, leaves[0].v1), leaves[0].v2);
        ora_f3 bmax = max3(max3(leaves[0].v0, leaves[0].v1), leaves[0].v2);
        nodes[0].min = bmin;
        nodes[0].max = bmax;
        nodes[0].w12 = 1u << 29;
        nodes[0].w28 = 0u | ((uint32_t)ORA_TYPE_TRI << 29);
    } else {
        ora_generate_hierarchy(nodes, t2, codes, n);
        memset(t1, 0, (size_t)n * 4);
        ora_generate_aabbs(nodes, t2, vals, t1, leaves, n);
    }
    free(codes); free(vals); free(t1); free(t2);
}

/* ================================================================== --pairs (Pairing.cuh)
 * FindSharedEdge (:26-33): does triangle t contain the directed edge a->b?  returns the rotation of t, or -1 */
static inline int vequal(ora_f3 a, ora_f3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static int find_shared_edge(ora_f3 a, ora_f3 b, const ora_triangle* t)
{
    if (vequal(a, t->v0) && vequal(b, t->v1)) return 0;
    if (vequal(a, t->v1) && vequal(b, t->v2)) return 2;
    if (vequal(a, t->v2) && vequal(b, t->v0)) return 1;
    return -1;
}
static inline ora_f3 tri_vertex(const ora_triangle* t, int i) { return i == 0 ? t->v0 : i == 1 ? t->v1 : t->v2; }
/* CanFormTrianglePair (:42-58) */
static int can_form_pair(const ora_triangle* a, const ora_triangle* b, int* rot_a, int* rot_b)
{
    int t0 = 3, t1 = -1;
    for (int u = 2, v = 0; v < 3; u = v, v++) {
        t1 = find_shared_edge(tri_vertex(a, v), tri_vertex(a, u), b);
        t0--;
        if (t1 != -1) break;
    }
    if (t1 == -1) return 0;
    *rot_a = t0;
    *rot_b = t1;
    return 1;
}
static inline void tri_box(const ora_triangle* t, float* b)   /* AABB(Triangle), Common.cuh:265-269 */
{
    ora_f3 lo = min3(min3(t->v0, t->v1), t->v2), hi = max3(max3(t->v0, t->v1), t->v2);
    b[0] = lo.x; b[1] = lo.y; b[2] = lo.z; b[3] = hi.x; b[4] = hi.y; b[5] = hi.z;
}
static inline float sa6(const float* b)
{
    float lx = b[3] - b[0], ly = b[4] - b[1], lz = b[5] - b[2];
    return 2.0f * (lx * ly + lx * lz + ly * lz);
}
/* merge decision of candidate (2k, 2k+1) (BottomUpBuilder.cu:127-138, ShouldFormTrianglePair Pairing.cuh:35-39) */
static int pair_merges(const ora_triangle* tris, uint32_t n, uint32_t tid)
{
    if (tid + 1 >= n) return 0;
    const ora_triangle *a = &tris[tid], *b = &tris[tid + 1];
    int ra, rb;
    if (!can_form_pair(a, b, &ra, &rb)) return 0;
    float ab[6], bb[6], cb[6];
    tri_box(a, ab);
    tri_box(b, bb);
    for (int k = 0; k < 3; k++) { cb[k] = fminf(ab[k], bb[k]); cb[3 + k] = fmaxf(ab[3 + k], bb[3 + k]); }
    return sa6(cb) * 0.5f < sa6(ab) + sa6(bb);
}

uint32_t ora_build_pairs(const ora_triangle* tris, uint32_t n, ora_node* nodes, ora_triangle_pair* leaves,
                         uint32_t* codes_sorted, uint32_t* indices_sorted, int32_t* aabb_ordered)
{
    uint32_t slots = 2 * (n > 1 ? n - 1 : 1);
    memset(nodes, 0, (size_t)slots * sizeof(ora_node));
    int32_t aabb[6];
    ora_scene_aabb(tris, n, aabb);
    if (aabb_ordered) memcpy(aabb_ordered, aabb, sizeof aabb);
    if (n == 0) return 0;
    ora_f3 smin = f3(ora_ordered_int_to_float(aabb[0]), ora_ordered_int_to_float(aabb[1]), ora_ordered_int_to_float(aabb[2]));
    ora_f3 smax = f3(ora_ordered_int_to_float(aabb[3]), ora_ordered_int_to_float(aabb[4]), ora_ordered_int_to_float(aabb[5]));
    ora_f3 ext = sub3(smax, smin);

    uint32_t* codes = (uint32_t*)malloc((size_t)n * 4);
    uint32_t* vals = (uint32_t*)malloc((size_t)n * 4);
    uint32_t* t1 = (uint32_t*)calloc((size_t)n, 4);
    uint32_t* t2 = (uint32_t*)calloc((size_t)n, 4);
    /* GenerateMortonCodesPairs with slots assigned in input order */
    uint32_t L = 0;
    for (uint32_t tid = 0; tid < n; tid += 2) {
        const int second_valid = tid + 1 < n;
        const int merge = pair_merges(tris, n, tid);
        const ora_triangle *a = &tris[tid], *b = second_valid ? &tris[tid + 1] : &tris[tid];
        ora_f3 centre = tri_centre(a);
        ora_f3 centre2 = tri_centre(b);
        if (merge) centre = scale3(add3(centre, centre2), 0.5f);
        ora_f3 c = div3(sub3(centre, smin), ext);
        c = f3(clampf(c.x, 0.0f, 1.0f), clampf(c.y, 0.0f, 1.0f), clampf(c.z, 0.0f, 1.0f));
        vals[L] = merge ? (tid | 0x80000000u) : tid;
        codes[L] = morton3d(c.x, c.y, c.z);
        L++;
        if (second_valid && !merge) {
            ora_f3 c2 = div3(sub3(centre2, smin), ext);
            c2 = f3(clampf(c2.x, 0.0f, 1.0f), clampf(c2.y, 0.0f, 1.0f), clampf(c2.z, 0.0f, 1.0f));
            vals[L] = tid + 1;
            codes[L] = morton3d(c2.x, c2.y, c2.z);
            L++;
        }
    }
    ora_radix_sort(codes, vals, t1, t2, L);
    if (codes_sorted) memcpy(codes_sorted, codes, (size_t)L * 4);
    if (indices_sorted) memcpy(indices_sorted, vals, (size_t)L * 4);
    /* GenerateTriangles with the pair branch (BottomUpBuilder.cu:294-309, CreateTrianglePair Pairing.cuh:60-77) */
    for (uint32_t g = 0; g < L; g++) {
        const uint32_t index = vals[g] & 0x7FFFFFFFu;
        ora_triangle_pair r;
        memset(&r, 0, sizeof r);
        if (vals[g] >> 31) {
            const ora_triangle *a = &tris[index], *b = &tris[index + 1];
            int ra = 0, rb = 0;
            can_form_pair(a, b, &ra, &rb);
            ora_triangle ar = *a;                       /* RotateTriangle (:9-21) */
            if (ra == 1) { ar.v0 = a->v2; ar.v1 = a->v0; ar.v2 = a->v1; }
            else if (ra == 2) { ar.v0 = a->v1; ar.v1 = a->v2; ar.v2 = a->v0; }
            r.v0 = ar.v0; r.v1 = ar.v1; r.v2 = ar.v2;
            r.v3 = rb == 2 ? b->v0 : rb == 1 ? b->v1 : b->v2;
            r.primitive_id_0 = index;
            r.primitive_id_1 = index + 1;
            r.rot_x = (uint16_t)ra;
            r.rot_y = (uint16_t)rb;
        } else {
            r.v0 = tris[index].v0; r.v1 = tris[index].v1; r.v2 = tris[index].v2;
            r.v3 = r.v2;
            r.primitive_id_0 = index;
        }
        leaves[g] = r;
    }
    if (L == 1) {
        ora_f3 bmin = min3(min3(min3(leaves[0].v0, leaves[0].v1), leaves[0].v2), leaves[0].v3);
        ora_f3 bmax = max3(max3(max3(leaves[0].v0, leaves[0].v1), leaves[0].v2), leaves[0].v3);
        nodes[0].min = bmin; nodes[0].max = bmax;
        nodes[0].w12 = 1u << 29;
        nodes[0].w28 = 0u | ((uint32_t)ORA_TYPE_TRI << 29);
    } else {
        ora_generate_hierarchy(nodes, t2, codes, L);
        memset(t1, 0, (size_t)n * 4);
        ora_generate_aabbs(nodes, t2, vals, t1, leaves, L);   /* includes v3 when the MSB of the value is set */
    }
    free(codes); free(vals); free(t1); free(t2);
    return L;
}

/* ================================================================== hybrid top tree
 * ExtractDepth (BottomUpBuilder.cu:314-371): thread tid walks from pair 0 following bit d of tid at level d
 * (bit set -> the child of slot cur, bit clear -> the child of slot cur+1), 8 levels; a pair with a Tri slot stops
 * the walk and is emitted once (by the thread whose remaining high bits are zero). */
static uint32_t extract_depth(const ora_node* nodes, uint32_t* subroots, float (*boxes)[6])
{
    uint32_t k = 0;
    for (uint32_t tid = 0; tid < 256; tid++) {
        uint32_t cur = 0, d = 0;
        int emit = 1;
        for (; d < 8; d++) {
            if ((nodes[cur].w28 >> 29) == ORA_TYPE_TRI || (nodes[cur + 1].w28 >> 29) == ORA_TYPE_TRI ||
                (nodes[cur].w28 >> 29) == ORA_TYPE_NONE || (nodes[cur + 1].w28 >> 29) == ORA_TYPE_NONE) {
                emit = (tid >> d) == 0;   /* (None only occurs in the n < 2 trees this build defines, Q8) */
                break;
            }
            uint32_t direction = (tid >> d) & 1u;
            cur = (direction ? nodes[cur].w28 : nodes[cur + 1].w28) & NODE_PARENT_MASK;
        }
        if (!emit) continue;
        subroots[k] = cur;
        ora_f3 lo = nodes[cur].min, hi = nodes[cur].max;
        if ((nodes[cur + 1].w28 >> 29) != ORA_TYPE_NONE) {
            lo = min3(lo, nodes[cur + 1].min);
            hi = max3(hi, nodes[cur + 1].max);
        }
        boxes[k][0] = lo.x; boxes[k][1] = lo.y; boxes[k][2] = lo.z;
        boxes[k][3] = hi.x; boxes[k][4] = hi.y; boxes[k][5] = hi.z;
        k++;
    }
    return k;
}

static inline float box_sa(const float* b)  /* Common.cuh:293-297 sa() */
{
    float lx = b[3] - b[0], ly = b[4] - b[1], lz = b[5] - b[2];
    return 2.0f * (lx * ly + lx * lz + ly * lz);
}
static inline void box_reset(float* b) { b[0] = b[1] = b[2] = 3.402823466e+38f; b[3] = b[4] = b[5] = -3.402823466e+38f; }
static inline void box_grow(float* b, const float* o)
{
    for (int k = 0; k < 3; k++) { b[k] = fminf(b[k], o[k]); b[3 + k] = fmaxf(b[3 + k], o[3 + k]); }
}
static inline void box_grow_pt(float* b, const float* c)
{
    for (int k = 0; k < 3; k++) { b[k] = fminf(b[k], c[k]); b[3 + k] = fmaxf(b[3 + k], c[k]); }
}
static inline void put_node(ora_node* n, const float* b, uint32_t child, uint32_t count, uint32_t type)
{
    n->min = f3(b[0], b[1], b[2]);
    n->max = f3(b[3], b[4], b[5]);
    n->w12 = count << 29;                      /* parent is never written by SharedTaskBuild: defined as 0 */
    n->w28 = (child & NODE_PARENT_MASK) | (type << 29);
}

typedef struct { float c[6], p[6]; uint32_t start, end, parent_idx, buf; } top_task;

uint32_t ora_build_hybrid_top(ora_node* nodes, uint32_t L, const int32_t aabb[6], uint32_t* subroots_out)
{
    uint32_t subroots[256];
    float boxes[256][6];
    const uint32_t K = extract_depth(nodes, subroots, boxes);
    if (subroots_out) memcpy(subroots_out, subroots, K * 4);
    const uint32_t slots = 2 * (L > 1 ? L - 1 : 1);   /* LBVH slots in use; the reference writes at 2L (BuildWrapper.cu:360) */
    uint32_t base = 2 * L;
    if (base < slots) base = slots;
    uint32_t write_index = base;
    top_task* queue = (top_task*)malloc(sizeof(top_task) * 1024);
    uint32_t qh = 0, qt = 0;
    uint32_t ids[2][256];
    for (uint32_t i = 0; i < 256; i++) ids[0][i] = i;     /* tmp_ids = 0..511 (BuildWrapper.cu:292-293,303) */

    top_task root;
    box_reset(root.c);
    for (uint32_t i = 0; i < K; i++) box_grow(root.c, boxes[i]);   /* "centroid" bounds = union of the BOXES (:341-346) */
    for (int k = 0; k < 6; k++) root.p[k] = ora_ordered_int_to_float(aabb[k]);
    root.start = 0; root.end = K; root.parent_idx = write_index++; root.buf = 0;
    if (K == 1) {
        /* the reference writes the single leaf INTO slot 2L and then traces from (2L+1, 2): garbage.  Defined
         * here: 2L = Box{2L+1, count 1}, 2L+1 = the leaf descriptor, 2L+2 = None. */
        put_node(&nodes[base], root.p, base + 1, 1, ORA_TYPE_BOX);
        put_node(&nodes[base + 1], boxes[0], subroots[0], 2, ORA_TYPE_BOX);
        memset(&nodes[base + 2], 0, sizeof(ora_node));
        free(queue);
        return K;
    }
    queue[qt++] = root;
    while (qh < qt) {
        top_task t = queue[qh++];
        const uint32_t count = t.end - t.start;
        const uint32_t* in = ids[t.buf];
        uint32_t* out = ids[t.buf ^ 1];
        if (count <= 2) {                                   /* SharedTaskBuilder.cu:396-464, LEAF_THRESHOLD 2 */
            uint32_t child = t.parent_idx;
            if (count != 1) { child = write_index; write_index += count; }
            for (uint32_t i = 0; i < count; i++) {
                uint32_t prim = in[t.start + i];
                put_node(&nodes[child + i], boxes[prim], subroots[prim], 2, ORA_TYPE_BOX);
            }
            if (count > 1) put_node(&nodes[t.parent_idx], t.p, child, count, ORA_TYPE_BOX);
            continue;
        }
        float cc[2][6], cp[2][6];
        uint32_t mid;
        int done = 0;
        if (!(box_sa(t.c) <= 0.0f)) {                       /* binned SAH (:206-350) */
            float lx = t.c[3] - t.c[0], ly = t.c[4] - t.c[1], lz = t.c[5] - t.c[2];
            int axis = 2 * (lz > lx && lz > ly) + 1 * (ly > lx && ly >= lz);
            const float epsilon = 1.1920929e-7f;
            float k1 = 8 * (1 - epsilon) / (t.c[3 + axis] - t.c[axis]);
            float bc[8][6], bp[8][6];
            uint32_t bn[8];
            int binof[256];
            for (int b = 0; b < 8; b++) { box_reset(bc[b]); box_reset(bp[b]); bn[b] = 0; }
            for (uint32_t i = t.start; i < t.end; i++) {
                const float* bx = boxes[in[i]];
                float centre[3] = {(bx[0] + bx[3]) * 0.5f, (bx[1] + bx[4]) * 0.5f, (bx[2] + bx[5]) * 0.5f};
                int bin = (int)(k1 * (centre[axis] - t.c[axis]));
                if (bin < 0) bin = 0;                       /* the reference reports an error and aborts the build */
                if (bin > 7) bin = 7;
                binof[i] = bin;
                box_grow(bp[bin], bx);
                box_grow_pt(bc[bin], centre);
                bn[bin]++;
            }
            /* SelectPlane (:297-350): prefix left->right, sweep right->left, strict <, both sides non-empty */
            float lc[7][6], lp[7][6];
            uint32_t ln[7];
            memcpy(lc[0], bc[0], 24); memcpy(lp[0], bp[0], 24); ln[0] = bn[0];
            for (int i = 1; i < 7; i++) {
                memcpy(lc[i], lc[i - 1], 24); memcpy(lp[i], lp[i - 1], 24);
                box_grow(lc[i], bc[i]); box_grow(lp[i], bp[i]);
                ln[i] = ln[i - 1] + bn[i];
            }
            float rc[6], rp[6];
            uint32_t rn = bn[7];
            memcpy(rc, bc[7], 24); memcpy(rp, bp[7], 24);
            float best = 3.402823466e+38f;
            int plane = -1;
            for (int i = 6; i >= 0; i--) {
                float score = box_sa(lp[i]) * ln[i] + box_sa(rp) * rn;
                if (score < best && ln[i] && rn) {
                    best = score; plane = i;
                    memcpy(cp[0], lp[i], 24); memcpy(cp[1], rp, 24);
                    memcpy(cc[0], lc[i], 24); memcpy(cc[1], rc, 24);
                }
                box_grow(rc, bc[i]); box_grow(rp, bp[i]); rn += bn[i];
            }
            if (plane >= 0) {                               /* PartitionIds (:352-380), stable here */
                uint32_t w = t.start;
                for (uint32_t i = t.start; i < t.end; i++) if (binof[i] <= plane) out[w++] = in[i];
                mid = w;
                for (uint32_t i = t.start; i < t.end; i++) if (binof[i] > plane) out[w++] = in[i];
                done = 1;
            }   /* else: "failed to find valid partition" in the reference; falls through to the median split */
        }
        if (!done) {                                        /* object split at the midpoint (:465-510) */
            mid = t.start + (count >> 1);
            box_reset(cc[0]); box_reset(cc[1]); box_reset(cp[0]); box_reset(cp[1]);
            for (uint32_t i = t.start; i < t.end; i++) {
                const float* bx = boxes[in[i]];
                float centre[3] = {(bx[3] + bx[0]) * 0.5f, (bx[4] + bx[1]) * 0.5f, (bx[5] + bx[2]) * 0.5f};
                int side = i >= mid;
                box_grow(cp[side], bx);
                box_grow_pt(cc[side], centre);
                out[i] = in[i];
            }
        }
        uint32_t child_index = write_index;                 /* (:544-606) */
        write_index += 2;
        put_node(&nodes[t.parent_idx], t.p, child_index, 2, ORA_TYPE_BOX);
        top_task l, r;
        memcpy(l.c, cc[0], 24); memcpy(l.p, cp[0], 24);
        l.start = t.start; l.end = mid; l.parent_idx = child_index; l.buf = t.buf ^ 1;
        memcpy(r.c, cc[1], 24); memcpy(r.p, cp[1], 24);
        r.start = mid; r.end = t.end; r.parent_idx = child_index + 1; r.buf = t.buf ^ 1;
        queue[qt++] = l;
        queue[qt++] = r;
    }
    free(queue);
    return K;
}

/* ================================================================== SAH path
 * RunSahBuild (BuildWrapper.cu:140-251), splits off: Setup (Multiblock.cu:139-207) -> GridBlockCounts / GridBlockScan /
 * GridBlockDistribute (:427-546): leaves bucketed by centroid into a 4 x 4 x 4 grid over the centroid bounds ->
 * SharedTaskBuild per grid cell (SharedTaskBuilder.cu:93-607: top-down binned SAH, 8 bins on the longest centroid axis,
 * leaf threshold 2, object-median split when the centroid box has no area) -> SharedTaskBuild(top_of_tree) over the
 * non-empty cells.  Trace root = (slot 0, count 1) (main.cu:222-223).
 *
 * The reference numbers leaves, cell members, id order inside a partition and node slots by atomic arrival order
 * (SURVEY 0.5); the tree it builds is the same up to that numbering.  DETERMINISTIC restatement, shared with the HIP
 * kernels (gpu-raytracing_amd/csrc/sah_build.hip):
 *   - leaf slot = input order (prefix sum of the per-candidate leaf counts with pairs on);
 *   - cell members in ascending leaf index; non-empty cells in ascending cell index; partitions are stable;
 *   - item positions: the L items (leaves, or leaf references with splits) occupy positions [0, L) of the id array
 *     (cell b = [start_b, end_b)), the top tree's items (cells) positions [B, B+K), B = n (+ n/5 with splits): the
 *     host-known upper bound of L;
 *   - node slots: the two child slots of a split between positions mid-1 | mid (or of a 2-item leaf group at
 *     start, start+1 -> mid = start+1) sit at BIAS + 2*mid, BIAS = 128 for the cell trees and -2B for the top tree
 *     (so the top tree lives in slots [0, 128), like the reference's 2*NUM_BLOCKS offset); the root descriptor of
 *     cell b is slot 128 + 2*start_b (its sibling slot is unused, type None), the top root is slot 0.  Every slot
 *     is written exactly once, independent of processing order.
 * Box unions are taken on the ordered-int encoding everywhere (what the atomics do; differs from fminf only for
 * -0 vs +0).  Two reference defects are not reproduced: a cell with ONE leaf has a Tri sub-root whose type the top
 * tree overwrites with Box (SharedTaskBuilder.cu:432-446: garbage traversal) -- the type is copied here; and the
 * grid cell index is clamped to [0, 3] (rounding of a merged pair's centre can leave the centroid bounds). */
#define SAH_CELLS 64
#define SAH_GRID 4
typedef struct { int32_t v[6]; } ibox;   /* ordered-int box */
static inline void ibox_reset(ibox* b) { b->v[0] = b->v[1] = b->v[2] = 0x7f7fffff; b->v[3] = b->v[4] = b->v[5] = (int32_t)0x80800000; }
static inline void ibox_grow_box(ibox* b, const float* o)
{
    for (int k = 0; k < 3; k++) {
        int32_t lo = ora_float_to_ordered_int(o[k]), hi = ora_float_to_ordered_int(o[3 + k]);
        if (lo < b->v[k]) b->v[k] = lo;
        if (hi > b->v[3 + k]) b->v[3 + k] = hi;
    }
}
static inline void ibox_grow_pt(ibox* b, const float* c)
{
    for (int k = 0; k < 3; k++) {
        int32_t q = ora_float_to_ordered_int(c[k]);
        if (q < b->v[k]) b->v[k] = q;
        if (q > b->v[3 + k]) b->v[3 + k] = q;
    }
}
static inline void ibox_merge(ibox* b, const ibox* o)
{
    for (int k = 0; k < 3; k++) {
        if (o->v[k] < b->v[k]) b->v[k] = o->v[k];
        if (o->v[3 + k] > b->v[3 + k]) b->v[3 + k] = o->v[3 + k];
    }
}
static inline void ibox_to_float(const ibox* b, float* f) { for (int k = 0; k < 6; k++) f[k] = ora_ordered_int_to_float(b->v[k]); }
/* float -> int as the device converts it (cvt.rzi.s32.f32 / v_cvt_i32_f32): NaN -> 0, saturating */
static inline int32_t cvt_rzi(float f)
{
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)f;
}
static inline void box_centre(const float* b, float* c) { for (int k = 0; k < 3; k++) c[k] = (b[k] + b[3 + k]) * 0.5f; }

typedef struct {
    ora_node* nodes;
    const float (*aabbs)[6];      /* item boxes: [0, L) leaves, [n, n+64) cells */
    uint32_t* ids[2];             /* item ids per position */
    uint32_t L;                   /* top_base: the first cell item (>= number of leaf items) */
    const uint32_t* cell_start;   /* top tree leaves point at the cell sub-roots */
    const uint32_t* item_leaf;    /* leaf item -> TrianglePair index | (two triangles ? 1u << 31 : 0) */
} sah_ctx;
typedef struct { uint32_t start, end, parent_idx, buf; float c[6]; int has_c; } sah_task;

static void sah_leaf_desc(const sah_ctx* x, ora_node* out, uint32_t idv)
{
    const uint32_t id = idv;
    if (id < x->L) {                                          /* (:405-421) leaf_type Tri */
        put_node(out, x->aabbs[id], x->item_leaf[id] & 0x7FFFFFFFu, (x->item_leaf[id] >> 31) ? 2 : 1, ORA_TYPE_TRI);
    } else {                                                  /* top_of_tree (:422-446): copy the cell's sub-root */
        const ora_node* sub = &x->nodes[2 * SAH_CELLS + 2 * x->cell_start[id - x->L]];
        put_node(out, x->aabbs[id], sub->w28 & NODE_PARENT_MASK, sub->w12 >> 29, sub->w28 >> 29);
    }
}

/* one sub-tree: the task loop of SharedTaskBuild (:924-967) as a depth-first walk (slot numbering does not depend on
 * the order).  c (centroid bounds) is given for root tasks, derived from the bins / halves for child tasks. */
static void sah_build_range(const sah_ctx* x, sah_task root, int64_t bias)
{
    sah_task* stack = (sah_task*)malloc(sizeof(sah_task) * 64);
    uint32_t cap = 64, sp = 0;
    stack[sp++] = root;
    while (sp) {
        sah_task t = stack[--sp];
        const uint32_t count = t.end - t.start;
        const uint32_t* in = x->ids[t.buf];
        uint32_t* out = x->ids[t.buf ^ 1];
        /* the task's primitive box: union of its items (== Task::p_aabb: cell box / bin prefix boxes) */
        ibox pi; ibox_reset(&pi);
        ibox ci; ibox_reset(&ci);
        for (uint32_t i = t.start; i < t.end; i++) {
            const float* bx = x->aabbs[in[i]];
            float ctr[3]; box_centre(bx, ctr);
            ibox_grow_box(&pi, bx);
            ibox_grow_pt(&ci, ctr);
        }
        float pbox[6], cbox[6];
        ibox_to_float(&pi, pbox);
        if (t.has_c) memcpy(cbox, t.c, 24); else ibox_to_float(&ci, cbox);
        if (count <= 2) {                                     /* (:396-464) LEAF_THRESHOLD 2 */
            if (count == 1) { sah_leaf_desc(x, &x->nodes[t.parent_idx], in[t.start]); continue; }
            const uint32_t child = (uint32_t)(bias + 2 * (int64_t)(t.start + 1));
            for (uint32_t i = 0; i < count; i++) sah_leaf_desc(x, &x->nodes[child + i], in[t.start + i]);
            put_node(&x->nodes[t.parent_idx], pbox, child, count, ORA_TYPE_BOX);
            continue;
        }
        uint32_t mid = 0;
        int done = 0;
        if (!(box_sa(cbox) <= 0.0f)) {                        /* binned SAH (:206-350) */
            const float lx = cbox[3] - cbox[0], ly = cbox[4] - cbox[1], lz = cbox[5] - cbox[2];
            const int axis = 2 * (lz > lx && lz > ly) + 1 * (ly > lx && ly >= lz);
            const float epsilon = 1.1920929e-7f;
            const float k1 = 8 * (1 - epsilon) / (cbox[3 + axis] - cbox[axis]);
            ibox bp[8];
            uint32_t bn[8];
            for (int b = 0; b < 8; b++) { ibox_reset(&bp[b]); bn[b] = 0; }
            uint8_t* binof = (uint8_t*)malloc(count);
            for (uint32_t i = t.start; i < t.end; i++) {
                const float* bx = x->aabbs[in[i]];
                float ctr[3]; box_centre(bx, ctr);
                int32_t bin = cvt_rzi(k1 * (ctr[axis] - cbox[axis]));
                if (bin < 0) bin = 0;                         /* the reference reports an error and aborts the build */
                if (bin > 7) bin = 7;
                binof[i - t.start] = (uint8_t)bin;
                ibox_grow_box(&bp[bin], bx);
                bn[bin]++;
            }
            /* SelectPlane (:297-350): prefix left->right, sweep right->left, score = sa(L)*nL + sa(R)*nR, strict <,
             * both sides non-empty, planes 6 -> 0 */
            ibox lp[7];
            uint32_t ln[7];
            lp[0] = bp[0]; ln[0] = bn[0];
            for (int i = 1; i < 7; i++) { lp[i] = lp[i - 1]; ibox_merge(&lp[i], &bp[i]); ln[i] = ln[i - 1] + bn[i]; }
            ibox rp = bp[7];
            uint32_t rn = bn[7];
            float best = 3.402823466e+38f;
            int plane = -1;
            for (int i = 6; i >= 0; i--) {
                float lf[6], rf[6];
                ibox_to_float(&lp[i], lf); ibox_to_float(&rp, rf);
                const float score = box_sa(lf) * ln[i] + box_sa(rf) * rn;
                if (score < best && ln[i] && rn) { best = score; plane = i; }
                ibox_merge(&rp, &bp[i]); rn += bn[i];
            }
            if (plane >= 0) {                                 /* PartitionIds (:352-380), stable here */
                uint32_t w = t.start;
                for (uint32_t i = t.start; i < t.end; i++) if (binof[i - t.start] <= plane) out[w++] = in[i];
                mid = w;
                for (uint32_t i = t.start; i < t.end; i++) if (binof[i - t.start] > plane) out[w++] = in[i];
                done = 1;
            }   /* else "failed to find valid partition" in the reference: falls through to the median split */
            free(binof);
        }
        if (!done) {                                          /* object split at the midpoint (:465-510) */
            mid = t.start + (count >> 1);
            for (uint32_t i = t.start; i < t.end; i++) out[i] = in[i];
        }
        const uint32_t child_index = (uint32_t)(bias + 2 * (int64_t)mid);   /* (:544-606) */
        put_node(&x->nodes[t.parent_idx], pbox, child_index, 2, ORA_TYPE_BOX);
        sah_task l = {t.start, mid, child_index, t.buf ^ 1, {0}, 0};
        sah_task r = {mid, t.end, child_index + 1, t.buf ^ 1, {0}, 0};
        if (sp + 2 > cap) { cap *= 2; stack = (sah_task*)realloc(stack, sizeof(sah_task) * cap); }
        stack[sp++] = r;
        stack[sp++] = l;
    }
    free(stack);
}

/* fmaxf / fminf of AABB::Intersection (Common.cuh:269-272) on the ordered-int encoding */
static inline float fmin_ord(float a, float b) { return ora_float_to_ordered_int(a) < ora_float_to_ordered_int(b) ? a : b; }
static inline float fmax_ord(float a, float b) { return ora_float_to_ordered_int(a) > ora_float_to_ordered_int(b) ? a : b; }
static inline void box_intersection(const float* a, const float* b, float* out)
{
    for (int k = 0; k < 3; k++) { out[k] = fmax_ord(a[k], b[k]); out[3 + k] = fmin_ord(a[3 + k], b[3 + k]); }
}
static inline int box_valid(const float* b) { return b[3] >= b[0] && b[4] >= b[1] && b[5] >= b[2]; }   /* Common.cuh:274-277 */
/* CalculateGridcell (Multiblock.cu:86-91) */
static inline void grid_cell(const float* p, const float* g, int32_t* c)
{
    for (int k = 0; k < 3; k++) {
        int32_t q = cvt_rzi(floorf((p[k] - g[k]) * 4.0f / (g[3 + k] - g[k])));
        c[k] = q < 0 ? 0 : (q > 3 ? 3 : q);
    }
}
/* CellToBounds (Multiblock.cu:93-102) */
static inline void cell_bounds(const int32_t* c, const float* g, float* out)
{
    for (int k = 0; k < 3; k++) {
        const float step = (g[3 + k] - g[k]) / 4.0f;
        out[k] = g[k] + (float)c[k] * step;
        out[3 + k] = g[k] + (float)(c[k] + 1) * step;
    }
}
static void make_leaf(ora_triangle_pair* r, const ora_triangle* a, const ora_triangle* b, uint32_t ida)
{
    memset(r, 0, sizeof *r);
    if (b) {                                                 /* CreateTrianglePair (Pairing.cuh:60-77) */
        int ra = 0, rb = 0;
        can_form_pair(a, b, &ra, &rb);
        ora_triangle ar = *a;                                /* RotateTriangle (Pairing.cuh:9-21) */
        if (ra == 1) { ar.v0 = a->v2; ar.v1 = a->v0; ar.v2 = a->v1; }
        else if (ra == 2) { ar.v0 = a->v1; ar.v1 = a->v2; ar.v2 = a->v0; }
        r->v0 = ar.v0; r->v1 = ar.v1; r->v2 = ar.v2;
        r->v3 = rb == 2 ? b->v0 : rb == 1 ? b->v1 : b->v2;
        r->primitive_id_0 = ida; r->primitive_id_1 = ida + 1;
        r->rot_x = (uint16_t)ra; r->rot_y = (uint16_t)rb;
    } else {
        r->v0 = a->v0; r->v1 = a->v1; r->v2 = a->v2; r->v3 = a->v2;
        r->primitive_id_0 = ida;
    }
}

uint32_t ora_build_sah(const ora_triangle* tris, uint32_t n, int enable_pairs, int enable_splits, ora_node* nodes,
                       ora_triangle_pair* leaves, uint32_t* cell_counts_out, uint32_t* num_leaf_records_out)
{
    /* ---- Setup / SetupSplits / SetupPairSplits (Multiblock.cu:139-425): leaf records, the items of the build (one per
     * leaf, or one per (leaf, grid cell) reference with splits), their boxes, global primitive / centroid bounds.
     * The split budget (extra_leaves, an atomic counter in the reference) is consumed in input order. */
    const uint32_t thresh = n / 5;                               /* extra_leaves_threshold (BuildWrapper.cu:143) */
    const uint32_t top_base = n + (enable_splits ? thresh : 0);  /* items < top_base; the cells' items start here */
    float (*aabbs)[6] = (float (*)[6])malloc(((size_t)top_base + SAH_CELLS) * 24);
    uint32_t* item_leaf = (uint32_t*)malloc(((size_t)top_base + 1) * 4);
    ibox gp, gc;
    ibox_reset(&gp); ibox_reset(&gc);
    float grid[6] = {0, 0, 0, 0, 0, 0};
    if (enable_splits) {                                         /* CalculateSceneAabb first (BuildWrapper.cu:189-192) */
        int32_t sb[6];
        ora_scene_aabb(tris, n, sb);
        for (int k = 0; k < 6; k++) { gp.v[k] = sb[k]; grid[k] = ora_ordered_int_to_float(sb[k]); }
    }
    uint32_t L = 0, R = 0, budget = 0;                           /* items, leaf records, extra_leaves counter */
    for (uint32_t tid = 0; tid < n; tid += (enable_pairs || !enable_splits) ? 2 : 1) {
        /* SetupSplits walks single triangles; Setup and SetupPairSplits walk candidates (2k, 2k+1) */
        const int by_pairs = enable_pairs || !enable_splits;
        const int second_valid = by_pairs && tid + 1 < n;
        const ora_triangle *a = &tris[tid], *b = second_valid ? &tris[tid + 1] : &tris[tid];
        float ab[6], bb[6], ac[3], bc[3];
        tri_box(a, ab); tri_box(b, bb);
        box_centre(ab, ac); box_centre(bb, bc);
        const int merge = enable_pairs && second_valid && pair_merges(tris, n, tid);
        float ub[6];
        { ibox u; ibox_reset(&u); ibox_grow_box(&u, ab); ibox_grow_box(&u, bb); ibox_to_float(&u, ub); }
        if (!enable_splits) {
            ibox_grow_box(&gp, ab); ibox_grow_box(&gp, bb);      /* AtomicConvertCombine(*s_p_aabb, Combine(a, b)) */
            ibox_grow_pt(&gc, ac); ibox_grow_pt(&gc, bc);        /* c_aabb = {min(a_c, b_c), max(a_c, b_c)} */
        }
        const int nleaf = merge ? 1 : 1 + second_valid;
        for (int s = 0; s < nleaf; s++) {
            const float* box = merge ? ub : (s ? bb : ab);
            make_leaf(&leaves[R], s ? b : a, merge ? b : NULL, tid + (uint32_t)s);
            const uint32_t leafv = R | (merge ? 0x80000000u : 0u);
            R++;
            int split = 0;
            int32_t lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
            if (enable_splits) {
                grid_cell(box, grid, lo);
                grid_cell(box + 3, grid, hi);
                split = lo[0] != hi[0] || lo[1] != hi[1] || lo[2] != hi[2];
                const uint32_t extra = (uint32_t)((hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1) - 1);
                if (split) { split = budget + extra < thresh; budget += extra; }   /* atomicAdd happens either way */
            }
            if (!split) {
                memcpy(aabbs[L], box, 24);
                item_leaf[L] = leafv;
                if (enable_splits) { float c[3]; box_centre(box, c); ibox_grow_pt(&gc, c); }
                L++;
                continue;
            }
            int32_t c[3];
            for (c[2] = lo[2]; c[2] <= hi[2]; c[2]++)            /* GridNextCell order: x fastest (Multiblock.cu:118-134) */
                for (c[1] = lo[1]; c[1] <= hi[1]; c[1]++)
                    for (c[0] = lo[0]; c[0] <= hi[0]; c[0]++) {
                        float cb[6], out[6];
                        cell_bounds(c, grid, cb);
                        if (merge) {                             /* (:362-378) */
                            float ia[6], ibx[6];
                            box_intersection(ab, cb, ia);
                            box_intersection(bb, cb, ibx);
                            if (!box_valid(ia) && !box_valid(ibx)) continue;
                            ibox u; ibox_reset(&u); ibox_grow_box(&u, ia); ibox_grow_box(&u, ibx); ibox_to_float(&u, out);
                        } else {
                            box_intersection(box, cb, out);
                        }
                        memcpy(aabbs[L], out, 24);
                        item_leaf[L] = leafv;
                        float ctr[3]; box_centre(out, ctr);
                        ibox_grow_box(&gp, out);
                        ibox_grow_pt(&gc, ctr);
                        L++;
                    }
        }
    }
    if (num_leaf_records_out) *num_leaf_records_out = R;
    /* every slot this build can leave unwritten is defined as type None */
    memset(nodes, 0, sizeof(ora_node) * ((size_t)2 * SAH_CELLS + 2 * (size_t)L + 2));
    uint32_t cell_count[SAH_CELLS] = {0}, cell_start[SAH_CELLS];
    if (L == 0) { free(aabbs); free(item_leaf); if (cell_counts_out) memcpy(cell_counts_out, cell_count, sizeof cell_count); return 0; }

    /* ---- GridBlockCounts / Scan / Distribute (:427-546) */
    float gcf[6], gpf[6];
    ibox_to_float(&gc, gcf); ibox_to_float(&gp, gpf);
    const float epsilon = 1.1920929e-7f;
    const float gscale = SAH_GRID * (1 - epsilon);
    uint8_t* cell_of = (uint8_t*)malloc(L);
    ibox cell_p[SAH_CELLS], cell_c[SAH_CELLS];
    for (int b = 0; b < SAH_CELLS; b++) { ibox_reset(&cell_p[b]); ibox_reset(&cell_c[b]); }
    for (uint32_t i = 0; i < L; i++) {
        float ctr[3]; box_centre(aabbs[i], ctr);
        int32_t q[3];
        for (int k = 0; k < 3; k++) {
            q[k] = cvt_rzi((ctr[k] - gcf[k]) * gscale / (gcf[3 + k] - gcf[k]));
            if (q[k] < 0) q[k] = 0;
            if (q[k] > SAH_GRID - 1) q[k] = SAH_GRID - 1;
        }
        const int cell = q[0] + q[1] * SAH_GRID + q[2] * SAH_GRID * SAH_GRID;
        cell_of[i] = (uint8_t)cell;
        cell_count[cell]++;
        ibox_grow_box(&cell_p[cell], aabbs[i]);
        ibox_grow_pt(&cell_c[cell], ctr);
    }
    uint32_t run = 0;
    for (int b = 0; b < SAH_CELLS; b++) { cell_start[b] = run; run += cell_count[b]; }
    if (cell_counts_out) memcpy(cell_counts_out, cell_count, sizeof cell_count);
    uint32_t* ids0 = (uint32_t*)malloc(((size_t)top_base + SAH_CELLS) * 4);
    uint32_t* ids1 = (uint32_t*)malloc(((size_t)top_base + SAH_CELLS) * 4);
    {
        uint32_t cur[SAH_CELLS];
        memcpy(cur, cell_start, sizeof cur);
        for (uint32_t i = 0; i < L; i++) ids0[cur[cell_of[i]]++] = i;
    }
    uint32_t K = 0;
    for (int b = 0; b < SAH_CELLS; b++) {
        ibox_to_float(&cell_p[b], aabbs[top_base + b]);
        if (cell_count[b]) ids0[top_base + K++] = top_base + (uint32_t)b;
    }
    sah_ctx x = {nodes, (const float (*)[6])aabbs, {ids0, ids1}, top_base, cell_start, item_leaf};
    /* ---- SharedTaskBuild per cell: root descriptor at 2*64 + 2*start_b (SharedTaskBuilder.cu:116-127) */
    for (int b = 0; b < SAH_CELLS; b++) {
        if (!cell_count[b]) continue;
        sah_task t = {cell_start[b], cell_start[b] + cell_count[b], 2 * SAH_CELLS + 2 * cell_start[b], 0, {0}, 1};
        ibox_to_float(&cell_c[b], t.c);
        sah_build_range(&x, t, 2 * SAH_CELLS);
    }
    /* ---- SharedTaskBuild(top_of_tree): items = non-empty cells, c / p = the global bounds, root descriptor slot 0 */
    {
        sah_task t = {top_base, top_base + K, 0, 0, {0}, 1};
        memcpy(t.c, gcf, 24);
        sah_build_range(&x, t, -2 * (int64_t)top_base);
    }
    free(aabbs); free(item_leaf); free(cell_of); free(ids0); free(ids1);
    return L;
}

/* ------------------------------------------------------------------ Utilities.cpp:8-44
 * (recursion restated with an explicit stack so deep trees cannot overflow the C stack) */
void ora_count_nodes(const ora_node* nodes, uint32_t root, uint32_t count, int32_t out[3])
{
    int32_t num_nodes = 0, num_leaf = 0, num_tree = 0;
    size_t cap = 1024, sp = 0;
    uint32_t* st = (uint32_t*)malloc(cap * 4);
    for (uint32_t i = count; i-- > 0;)
        if ((nodes[root + i].w28 >> 29) == ORA_TYPE_BOX) st[sp++] = root + i;
    while (sp) {
        uint32_t idx = st[--sp];
        num_nodes++;
        uint32_t type = nodes[idx].w28 >> 29;
        if (type == ORA_TYPE_TRI) num_leaf++;
        else if (type == ORA_TYPE_BOX) {
            num_tree++;
            uint32_t c = nodes[idx].w28 & NODE_PARENT_MASK, k = nodes[idx].w12 >> 29;
            if (sp + k + 1 > cap) { cap *= 2; st = (uint32_t*)realloc(st, cap * 4); }
            for (uint32_t i = k; i-- > 0;) st[sp++] = c + i;
        }
    }
    free(st);
    out[0] = num_nodes; out[1] = num_leaf; out[2] = num_tree;
}

/* Utilities.cpp:46-83: a Box slot's min/max must EXACTLY equal the union of its `count` children;
 * on failure the reference prints and does not descend.  Returns the number of failures. */
int ora_verify_hierarchy(const ora_node* nodes, uint32_t root, uint32_t count)
{
    int errors = 0;
    size_t cap = 1024, sp = 0;
    uint32_t* st = (uint32_t*)malloc(cap * 4);
    for (uint32_t i = 0; i < count; i++)
        if ((nodes[root + i].w28 >> 29) == ORA_TYPE_BOX) st[sp++] = root + i;
    while (sp) {
        uint32_t idx = st[--sp];
        if ((nodes[idx].w28 >> 29) != ORA_TYPE_BOX) continue;
        uint32_t c = nodes[idx].w28 & NODE_PARENT_MASK, k = nodes[idx].w12 >> 29;
        ora_f3 cmin = f3(3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f);
        ora_f3 cmax = f3(-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f);
        for (uint32_t i = 0; i < k; i++) {
            cmin = min3(cmin, nodes[c + i].min);
            cmax = max3(cmax, nodes[c + i].max);
        }
        if (nodes[idx].min.x != cmin.x || nodes[idx].min.y != cmin.y || nodes[idx].min.z != cmin.z ||
            nodes[idx].max.x != cmax.x || nodes[idx].max.y != cmax.y || nodes[idx].max.z != cmax.z) {
            errors++;
            continue;
        }
        if (sp + k + 1 > cap) { cap *= 2; st = (uint32_t*)realloc(st, cap * 4); }
        for (uint32_t i = 0; i < k; i++) st[sp++] = c + i;
    }
    free(st);
    return errors;
}

/* ================================================================== Tracer.cu */
typedef struct { ora_f3 origin; float tmin; ora_f3 direction; float tmax; } ray_t;       /* Tracer.cuh:9-14 */
typedef struct { uint32_t primitive_id, tri_id; float bu, bv; } ray_result_t;            /* Tracer.cu:4-8   */
typedef struct { uint32_t box_tests, tri_tests, max_stack, dropped; } stats_t;           /* Tracer.cuh:4-7 (+ 2 test aids) */

/* Tracer.cu:187-200 IntersectRayAabb */
static inline int intersect_ray_aabb(const ora_node* node, const ray_t* ray, float* distance)
{
    ora_f3 inv_dir = f3(1.0f / ray->direction.x, 1.0f / ray->direction.y, 1.0f / ray->direction.z);
    ora_f3 t1 = mul3(sub3(node->min, ray->origin), inv_dir);
    ora_f3 t2 = mul3(sub3(node->max, ray->origin), inv_dir);
    ora_f3 tmin = min3(t1, t2);
    ora_f3 tmax = max3(t1, t2);
    float front = fmaxf(fmaxf(tmin.x, tmin.y), tmin.z);
    float back = fminf(fminf(tmax.x, tmax.y), tmax.z);
    *distance = front;
    return back >= front && front <= ray->tmax && back >= ray->tmin;
}

/* Tracer.cu:256-291 IntersectRayTriangle */
static inline int intersect_ray_triangle(ora_f3 v0, ora_f3 v1, ora_f3 v2, ray_t* ray, ray_result_t* rr,
                                         uint32_t tri_id, uint32_t prim_id)
{
    const float epsilon = 0.000000001f;
    ora_f3 edge1 = sub3(v1, v0);
    ora_f3 edge2 = sub3(v2, v0);
    ora_f3 h = cross3(ray->direction, edge2);
    float a = dot3(edge1, h);
    if (a > -epsilon && a < epsilon) return 0;
    float f = 1.0f / a;
    ora_f3 s = sub3(ray->origin, v0);
    float u = f * dot3(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    ora_f3 q = cross3(s, edge1);
    float v = f * dot3(ray->direction, q);
    if (v < 0.0f || (u + v) > 1.0f) return 0;
    float t = f * dot3(edge2, q);
    if (t < ray->tmin || t > ray->tmax) return 0;
    ray->tmax = t;
    rr->primitive_id = prim_id;
    rr->tri_id = tri_id;
    rr->bu = u;
    rr->bv = v;
    return 1;
}

/* Tracer.cu:293-306 IntersectRayTrianglePair */
static inline int intersect_ray_triangle_pair(const ora_triangle_pair* tp, ray_t* ray, ray_result_t* rr,
                                              uint32_t pair_id, int pair)
{
    int hit_a = intersect_ray_triangle(tp->v0, tp->v1, tp->v2, ray, rr, pair_id << 1, tp->primitive_id_0);
    int hit_b = pair ? intersect_ray_triangle(tp->v2, tp->v1, tp->v3, ray, rr, (pair_id << 1) + 1, tp->primitive_id_1) : 0;
    return hit_a || hit_b;
}

/* Tracer.cu:308-374 TraceRay.  The reference prints "stack overflow" when stack_size reaches 64 and then writes
 * out of bounds (undefined); here EVERY push onto a full stack -- the nearest child's final push included -- is
 * dropped and counted in stats->dropped (cannot happen on a binary LBVH, SURVEY A; a SAH tree over a scene spanning
 * 100+ octaves gets there, tests/test_gpu_traversal_edges.py).  trace_kernel.hip follows the same rule. */
typedef struct { uint32_t index, count; } stack_entry_t;
static uint32_t* g_visit_counts = 0;   /* analysis aid (tools/visit_histogram.py): per-slot visit counter, or NULL */
void ora_set_visit_counts(uint32_t* p) { g_visit_counts = p; }
#define PUSH(e) do { if (sp < 64) stack[sp++] = (e); else stats->dropped++; if (sp > stats->max_stack) stats->max_stack = sp; } while (0)
static int trace_ray(const ora_triangle_pair* leaves, const ora_node* nodes, uint32_t root, uint32_t count,
                     ray_t* ray, ray_result_t* rr, stats_t* stats)
{
    int tri_hit = 0;
    unsigned sp = 1;
    stack_entry_t stack[64];
    stack[0].index = root;
    stack[0].count = count;
    while (sp) {
        stack_entry_t entry = stack[--sp];
        if (g_visit_counts) __atomic_fetch_add(&g_visit_counts[entry.index], 1u, __ATOMIC_RELAXED);
        unsigned num_hits = 0;
        stack_entry_t child_buffer = {0, 0};
        float child_dist = 0.0f;
        for (unsigned i = 0; i < entry.count; i++) {
            const ora_node* node = &nodes[entry.index + i];
            uint32_t type = node->w28 >> 29, child = node->w28 & NODE_PARENT_MASK, ncount = node->w12 >> 29;
            if (type == ORA_TYPE_NONE) continue;
            float dist;
            int hit = intersect_ray_aabb(node, ray, &dist);
            int is_leaf = type == ORA_TYPE_TRI;
            stats->box_tests++;
            if (hit && is_leaf) {
                stats->tri_tests++;
                int hit_tri = intersect_ray_triangle_pair(&leaves[child], ray, rr, child, ncount > 0);
                tri_hit |= hit_tri;
            } else if (hit && num_hits == 0) {
                child_buffer.index = child;
                child_buffer.count = ncount;
                child_dist = dist;
                num_hits++;
            } else if (hit) {
                if (dist < child_dist || (dist == child_dist && child > child_buffer.index)) {
                    stack_entry_t tmp = child_buffer;
                    child_buffer.index = child;
                    child_buffer.count = ncount;
                    child_dist = dist;
                    PUSH(tmp);
                } else {
                    stack_entry_t e = {child, ncount};
                    PUSH(e);
                }
            }
        }
        if (num_hits > 0) PUSH(child_buffer);
    }
    return tri_hit;
}

/* Tracer.cu:57-82 RotateAttributes */
static ora_attributes rotate_attributes(const ora_attributes* in, int second_tri, uint16_t rx, uint16_t ry)
{
    uint16_t r = second_tri ? ry : rx;
    ora_attributes o = *in;
    if (r == 1) {
        o.normal[0] = in->normal[2]; o.normal[1] = in->normal[0]; o.normal[2] = in->normal[1];
        o.uv[0][0] = in->uv[2][0]; o.uv[0][1] = in->uv[2][1];
        o.uv[1][0] = in->uv[0][0]; o.uv[1][1] = in->uv[0][1];
        o.uv[2][0] = in->uv[1][0]; o.uv[2][1] = in->uv[1][1];
    } else if (r == 2) {
        o.normal[0] = in->normal[1]; o.normal[1] = in->normal[2]; o.normal[2] = in->normal[0];
        o.uv[0][0] = in->uv[1][0]; o.uv[0][1] = in->uv[1][1];
        o.uv[1][0] = in->uv[2][0]; o.uv[1][1] = in->uv[2][1];
        o.uv[2][0] = in->uv[0][0]; o.uv[2][1] = in->uv[0][1];
    }
    return o;
}

/* Tracer.cu:15-41 HsvToRgb -> float rgb in [0,255] before the uchar truncation */
static ora_f3 hsv_to_rgb255(float h, float s, float v)
{
    h = clampf(h, 0.f, 1.f) * 360.0f;
    s = clampf(s, 0.f, 1.f);
    v = clampf(v, 0.f, 1.f);
    float c = s * v;
    float x = c * (1 - fabsf(((int)h % 120) / 60.0f - 1));
    float m = v - c;
    ora_f3 rgb;
    if (h >= 0 && h < 60) rgb = f3(c, x, 0);
    else if (h >= 60 && h < 120) rgb = f3(x, c, 0);
    else if (h >= 120 && h < 180) rgb = f3(0, c, x);
    else if (h >= 180 && h < 240) rgb = f3(0, x, c);
    else if (h >= 240 && h < 300) rgb = f3(x, 0, c);
    else rgb = f3(c, 0, x);
    return f3((rgb.x + m) * 255, (rgb.y + m) * 255, (rgb.z + m) * 255);
}

/* float -> unsigned char as the device does it (cvt.rzi.u8.f32: NaN -> 0, out of range saturates); a plain C cast is
 * undefined outside [0, 256) and bilinear weights at a texture border do leave that range */
static inline uint8_t sat_u8(float v) { return !(v > 0.0f) ? 0 : (v >= 255.0f ? 255 : (uint8_t)v); }

/* ================================================================== textures (Tracer.cu:84-254) */
static const ora_texture* g_textures = 0;
static uint32_t g_num_textures = 0;
void ora_set_textures(const ora_texture* t, uint32_t n) { g_textures = t; g_num_textures = n; }

/* Texture::GenerateLODs (FileIO.cpp:121-150): sizes halve rounding up until 1x1; each texel = trunc(0.25 * sum of the
 * 2x2 source texels, coordinates clamped (ReadTexel :109-114)) */
uint32_t ora_lod_sizes(int32_t sx0, int32_t sy0, int32_t* sx, int32_t* sy)
{
    uint32_t lod = 0;
    sx[0] = sx0; sy[0] = sy0;
    while ((sx[lod] > 1 || sy[lod] > 1) && lod + 1 < ORA_NUM_LODS) {
        sx[lod + 1] = (sx[lod] + 1) / 2;
        sy[lod + 1] = (sy[lod] + 1) / 2;
        lod++;
    }
    return lod;
}
void ora_generate_lod(const uint32_t* src, int32_t sx, int32_t sy, uint32_t* dst)
{
    const int32_t dx = (sx + 1) / 2, dy = (sy + 1) / 2;
    for (int32_t j = 0; j < dy; j++)
        for (int32_t i = 0; i < dx; i++) {
            float acc[4] = {0, 0, 0, 0};
            for (int k = 0; k < 4; k++) {
                int32_t x = i * 2 + (k & 1), y = j * 2 + (k >> 1);
                if (x > sx - 1) x = sx - 1;
                if (y > sy - 1) y = sy - 1;
                const uint8_t* t = (const uint8_t*)&src[(size_t)y * sx + x];
                for (int c = 0; c < 4; c++) acc[c] = k == 0 ? (float)t[c] : acc[c] + (float)t[c];
            }
            uint8_t* o = (uint8_t*)&dst[(size_t)j * dx + i];
            for (int c = 0; c < 4; c++) o[c] = sat_u8(acc[c] * 0.25f);
        }
}

typedef struct { float x, y; } f2_t;
static inline float fracf1(float v) { return v - floorf(v); }                                  /* helper_math.h:1367 */
/* Sample(Texture&, int2, lod) (:103-108) -> float4 of the texel bytes */
static inline void tex_fetch(const ora_texture* t, int x, int y, int lod, float out[4])
{
    const int sx = t->size_x[lod], sy = t->size_y[lod];
    x = x > sx - 1 ? sx - 1 : x; x = x < 0 ? 0 : x;       /* clamp(xy, 0, size-1) = max(0, min(xy, size-1)) */
    y = y > sy - 1 ? sy - 1 : y; y = y < 0 ? 0 : y;
    const uint8_t* p = (const uint8_t*)&t->mips[lod][(size_t)y * sx + x];
    out[0] = p[0]; out[1] = p[1]; out[2] = p[2]; out[3] = p[3];
}
/* BilinearSample (:122-140) */
static void bilinear_sample(const ora_texture* t, f2_t uv, int lod, uint8_t out[4])
{
    float cx = fracf1(uv.x) * (float)t->size_x[lod] - 0.5f;
    float cy = fracf1(uv.y) * (float)t->size_y[lod] - 0.5f;
    cy = (float)t->size_y[lod] - cy;
    const int ix = (int)cx, iy = (int)cy;
    const float dx = cx - (float)ix, dy = cy - (float)iy;
    const float w0 = (1.0f - dx) * dy, w1 = dx * dy, w2 = (1.0f - dx) * (1.0f - dy), w3 = dx * (1.0f - dy);
    float s0[4], s1[4], s2[4], s3[4];
    tex_fetch(t, ix, iy, lod, s0);
    tex_fetch(t, ix + 1, iy, lod, s1);
    tex_fetch(t, ix, iy - 1, lod, s2);
    tex_fetch(t, ix + 1, iy - 1, lod, s3);
    for (int c = 0; c < 4; c++) out[c] = sat_u8(((s0[c] * w0 + s1[c] * w1) + s2[c] * w2) + s3[c] * w3);
}
/* TrilinearSample (:142-155) */
static void trilinear_sample(const ora_texture* t, f2_t uv, float lod, uint8_t out[4])
{
    uint32_t min_lod = (uint32_t)floorf(lod), max_lod = min_lod + 1;
    min_lod = min_lod > t->max_lod ? t->max_lod : min_lod;
    max_lod = max_lod > t->max_lod ? t->max_lod : max_lod;
    uint8_t a[4], b[4];
    bilinear_sample(t, uv, (int)min_lod, a);
    bilinear_sample(t, uv, (int)max_lod, b);
    const float frac = fracf1(lod);
    for (int c = 0; c < 4; c++) out[c] = sat_u8((float)a[c] * (1.0f - frac) + (float)b[c] * frac);
}
static inline f2_t interp_uv(const ora_attributes* at, float bu, float bv)               /* InterpolateUVs (:43-48) */
{
    const float w0 = 1 - bu - bv;
    f2_t r;
    r.x = (at->uv[0][0] * w0 + at->uv[1][0] * bu) + at->uv[2][0] * bv;
    r.y = (at->uv[0][1] * w0 + at->uv[1][1] * bu) + at->uv[2][1] * bv;
    return r;
}
/* RayTriangleGradients (:202-235): barycentrics of the hit with the ray moved by one pixel in x and in y */
static void ray_triangle_gradients(const ora_f3 v[3], const ray_t* ray, float spread, float out[4])
{
    ora_f3 edge1 = sub3(v[1], v[0]), edge2 = sub3(v[2], v[0]);
    ora_f3 s = sub3(ray->origin, v[0]);
    ora_f3 q = cross3(s, edge1);
    ora_f3 x = scale3(scale3(normalize3(cross3(ray->direction, f3(0, 1, 0))), ray->tmax), spread);
    ora_f3 y = scale3(scale3(normalize3(cross3(ray->direction, x)), ray->tmax), spread);
    ora_f3 hit_point = add3(ray->origin, scale3(ray->direction, ray->tmax));
    ora_f3 dirx = normalize3(sub3(add3(hit_point, x), ray->origin));
    ora_f3 diry = normalize3(sub3(add3(hit_point, y), ray->origin));
    ora_f3 h0 = cross3(dirx, edge2);
    float a0 = dot3(edge1, h0), f0 = 1.0f / a0;
    out[0] = f0 * dot3(s, h0);
    out[1] = f0 * dot3(dirx, q);
    ora_f3 h1 = cross3(diry, edge2);
    float a1 = dot3(edge1, h1), f1 = 1.0f / a1;
    out[2] = f1 * dot3(s, h1);
    out[3] = f1 * dot3(diry, q);
}
/* ComputeLOD (:237-254) */
static float compute_lod(const ray_t* ray, const ray_result_t* rr, float spread, const ora_f3 tri[3],
                         const ora_attributes* at, const ora_texture* tex)
{
    float g[4];
    ray_triangle_gradients(tri, ray, spread, g);
    f2_t uvs = interp_uv(at, rr->bu, rr->bv), ux = interp_uv(at, g[0], g[1]), uy = interp_uv(at, g[2], g[3]);
    const float sx = (float)tex->size_x[0], sy = (float)tex->size_y[0];
    const float dxx = fabsf(ux.x - uvs.x) * sx, dxy = fabsf(ux.y - uvs.y) * sy;
    const float dyx = fabsf(uy.x - uvs.x) * sx, dyy = fabsf(uy.y - uvs.y) * sy;
    const float max_change = fmaxf(sqrtf(dxx * dxx + dxy * dxy), sqrtf(dyx * dyx + dyy * dyy));
    return clampf(rt_log2f(max_change), 0.0f, (float)tex->max_lod);   /* log2f: rt_math.h (bit-identical on host and device) */
}
/* TangentMatrix (:84-101): rows of the tangent/bitangent/normal frame */
static void tangent_matrix(const ora_f3 tri[3], const ora_attributes* at, ora_f3 rows[3])
{
    ora_f3 e1 = sub3(tri[1], tri[0]), e2 = sub3(tri[2], tri[0]);
    const float d1x = at->uv[1][0] - at->uv[0][0], d1y = at->uv[1][1] - at->uv[0][1];
    const float d2x = at->uv[2][0] - at->uv[0][0], d2y = at->uv[2][1] - at->uv[0][1];
    const float f = 1.0f / (d1x * d2y - d1y * d2x);
    ora_f3 normal = normalize3(cross3(e1, e2));
    ora_f3 tangent = normalize3(scale3(sub3(scale3(e1, d2y), scale3(e2, d1y)), f));
    ora_f3 bitangent = normalize3(scale3(sub3(scale3(e2, d1x), scale3(e1, d2x)), f));
    rows[0] = f3(tangent.x, bitangent.x, normal.x);
    rows[1] = f3(tangent.y, bitangent.y, normal.y);
    rows[2] = f3(tangent.z, bitangent.z, normal.z);
}
/* Bump2Normal (:157-185) */
static ora_f3 bump2normal(const ora_texture* tex, const ora_f3 tbn[3], f2_t uv, float lod)
{
    const float texel_step = rt_exp2f(lod);   /* powf(2.0f, lod): rt_math.h */
    const float stx = texel_step / (float)tex->size_x[0], sty = texel_step / (float)tex->size_y[0];
    uint8_t a[4], b[4], c[4];
    f2_t ua = {uv.x - stx * 0.5f, uv.y - sty * 0.5f}, ub = {uv.x + stx * 0.5f, uv.y + 0.0f}, uc = {uv.x + 0.0f, uv.y + sty * 0.5f};
    trilinear_sample(tex, ua, lod, a);
    trilinear_sample(tex, ub, lod, b);
    trilinear_sample(tex, uc, lod, c);
    const float gx = (float)b[0] - a[0], gy = (float)c[0] - a[0];
    const float d = 4.0f;
    ora_f3 n = normalize3(cross3(f3(1, 0, d * gx / (texel_step * 256.0f)), f3(0, 1, d * gy / (texel_step * 256.0f))));
    n = f3(dot3(tbn[0], n), dot3(tbn[1], n), dot3(tbn[2], n));
    return normalize3(n);
}

/* Tracer.cu:376-469 AmbientShader.  `max(dot, 0.0)` and pow() are evaluated in double on the device (float/double
 * overloads), the product `1.0f * pow(...)` is narrowed to float by operator*(float, float3) -- restated literally. */
typedef struct {
    const ora_triangle_pair* leaves; const ora_node* nodes; uint32_t root, count;
} accel_t;
static ora_f3 ambient_shader255(const accel_t* as, const ray_t* ray, const ray_result_t* rr, const ora_material* mat,
                                const ora_attributes* at, const float light[3], float spread, int use_textures,
                                int use_shadows, int use_bump)
{
    ora_f3 light_colour = f3(1.0f, 0.9f, 0.8f);
    ora_f3 light_pos = f3(light[0], light[1], light[2]);
    ora_f3 hit_pos = add3(ray->origin, scale3(ray->direction, ray->tmax));
    /* InterpolateNormals (:50-56) */
    float w0 = 1 - rr->bu - rr->bv;
    ora_f3 normal = add3(add3(scale3(at->normal[0], w0), scale3(at->normal[1], rr->bu)), scale3(at->normal[2], rr->bv));
    const ora_triangle_pair* pair = &as->leaves[rr->tri_id >> 1];
    ora_f3 tri[3];
    if (rr->tri_id & 1) { tri[0] = pair->v2; tri[1] = pair->v1; tri[2] = pair->v3; }
    else { tri[0] = pair->v0; tri[1] = pair->v1; tri[2] = pair->v2; }
    if (use_bump && mat->disp != -1) {                    /* displacement map used as a normal map (:388-403) */
        const ora_texture* disp = &g_textures[mat->disp];
        f2_t uvs = interp_uv(at, rr->bu, rr->bv);
        float lod = compute_lod(ray, rr, spread, tri, at, disp);
        ora_f3 tbn[3];
        tangent_matrix(tri, at, tbn);
        uint8_t smp[4];
        trilinear_sample(disp, uvs, lod, smp);
        normal = f3(smp[0] / 255.0f, smp[1] / 255.0f, smp[2] / 255.0f);
        normal = normalize3(f3(normal.x * 2.0f - 1.0f, normal.y * 2.0f - 1.0f, normal.z * 2.0f - 1.0f));
        normal = normalize3(f3(dot3(tbn[0], normal), dot3(tbn[1], normal), dot3(tbn[2], normal)));
    } else if (use_bump && mat->bump != -1) {             /* (:405-415) */
        const ora_texture* bump = &g_textures[mat->bump];
        f2_t uvs = interp_uv(at, rr->bu, rr->bv);
        float lod = compute_lod(ray, rr, spread, tri, at, bump);
        ora_f3 tbn[3];
        tangent_matrix(tri, at, tbn);
        normal = bump2normal(bump, tbn, uvs, lod);
    }
    ora_f3 light_dir = normalize3(sub3(light_pos, hit_pos));
    ora_f3 ambient = scale3(light_colour, 0.2f);
    ora_f3 diffuse = scale3(light_colour, 1.0f * fmaxf(dot3(normal, light_dir), 0.0f));
    ora_f3 neg_l = f3(-light_dir.x, -light_dir.y, -light_dir.z);
    /* reflect(i, n) = i - 2.0f * n * dot(n, i)   (helper_math.h:1435-1438) */
    ora_f3 refl = sub3(neg_l, scale3(scale3(normal, 2.0f), dot3(normal, neg_l)));
    ora_f3 neg_d = f3(-ray->direction.x, -ray->direction.y, -ray->direction.z);
    double sp_base = fmax((double)dot3(neg_d, refl), 0.0);
    float sp = (float)(1.0f * rt_pow_d(sp_base, (double)mat->specular_exp));   /* pow: rt_math.h */
    ora_f3 specular = scale3(light_colour, sp);
    ora_f3 object_diffuse = mat->diffuse;
    if (use_textures && mat->texture != -1) {             /* (:432-445): BilinearSample(tex, uv, (int)lod) */
        const ora_texture* tex = &g_textures[mat->texture];
        float lod = compute_lod(ray, rr, spread, tri, at, tex);
        uint8_t smp[4];
        bilinear_sample(tex, interp_uv(at, rr->bu, rr->bv), (int)lod, smp);
        object_diffuse = f3((float)smp[0] / 255, (float)smp[1] / 255, (float)smp[2] / 255);
    }
    if (use_shadows) {                                    /* (:447-462) */
        ray_t shadow;
        ray_result_t sres = {0, 0, 0.f, 0.f};
        stats_t sstats = {0, 0, 0, 0};
        shadow.origin = hit_pos;
        shadow.direction = light_dir;
        shadow.tmin = 0.001f;
        ora_f3 to_light = sub3(light_pos, hit_pos);
        shadow.tmax = sqrtf(dot3(to_light, to_light));
        if (trace_ray(as->leaves, as->nodes, as->root, as->count, &shadow, &sres, &sstats)) {
            diffuse = f3(0, 0, 0);
            specular = f3(0, 0, 0);
        }
    }
    ora_f3 colour = add3(add3(mul3(diffuse, object_diffuse), mul3(ambient, mat->ambient)), mul3(specular, mat->specular));
    colour = f3(clampf(colour.x, 0.0f, 1.0f), clampf(colour.y, 0.0f, 1.0f), clampf(colour.z, 0.0f, 1.0f));
    return f3(colour.x * 255, colour.y * 255, colour.z * 255);
}

/* Tracer.cu:471-595 TraceRays.  One pixel; returns the float colour (0..255 per channel, before the
 * uchar truncation of :512-514 / :468) so that spp>1 can average. */
static ora_f3 shade_pixel(const ora_triangle_pair* leaves, const ora_node* nodes, uint32_t root, uint32_t count,
                          const ora_attributes* attributes, const ora_material* materials, uint32_t num_materials,
                          const ora_camera* cam, const float light[3], int render_type, uint32_t x, uint32_t y,
                          uint32_t w, uint32_t h, float ox, float oy, stats_t* stats, float* alpha)
{
    *alpha = 255.0f;
    float cx = (float)x, cy = (float)y;
    float ndcx = 2 * ((cx + ox) / (float)w) - 1;
    float ndcy = 2 * ((cy + oy) / (float)h) - 1;
    ora_f3 p = add3(add3(scale3(cam->u, ndcx), scale3(cam->v, ndcy)), scale3(cam->w, 1.0f));
    float max_depth = cam->max_depth;
    ray_t ray;
    ray.direction = normalize3(p);
    ray.origin = cam->position;
    ray.tmin = 0.00001f;
    ray.tmax = max_depth;
    ray_result_t rr = {0, 0, 0.f, 0.f};
    stats_t st = {0, 0, 0, 0};
    int hit = trace_ray(leaves, nodes, root, count, &ray, &rr, &st);
    float depth = hit ? ray.tmax : 0.0f;
    stats->box_tests += st.box_tests;
    stats->tri_tests += st.tri_tests;
    if (st.max_stack > stats->max_stack) stats->max_stack = st.max_stack;
    stats->dropped += st.dropped;

    switch (render_type) {
    case ORA_DEPTH: {
        float g = fminf(1.0f, depth / max_depth) * 255;
        return f3(g, g, g);
    }
    case ORA_BOXTESTS: {
        float g = fminf(st.box_tests / 180.0f, 1.0f) * 255;
        return f3(0, g, g);
    }
    case ORA_TRITESTS: {
        float g = fminf(st.tri_tests / 32.0f, 1.0f);
        return f3(g * 100, g * 255, g * 100);
    }
    default: break;
    }
    /* attributes / material are fetched before the hit test in the reference (:506-509): primitive 0 on a miss */
    int second_tri = rr.tri_id & 1;
    const ora_triangle_pair* pair = &leaves[rr.tri_id >> 1];
    ora_attributes at = rotate_attributes(&attributes[rr.primitive_id], second_tri, pair->rot_x, pair->rot_y);
    /* indices from scene data are range-checked (the reference reads out of bounds, e.g. material_id -1 for faces before
     * the first usemtl, FileIO.cpp:191): a material id outside [0, num_materials) shades as material 0, a texture index
     * outside the texture table reads as -1 (untextured).  Same rule in trace_kernel.hip. */
    ora_material mat_checked = materials[(uint32_t)at.material_id < num_materials ? (uint32_t)at.material_id : 0u];
    if ((uint32_t)mat_checked.texture >= g_num_textures) mat_checked.texture = -1;
    if ((uint32_t)mat_checked.bump >= g_num_textures) mat_checked.bump = -1;
    if ((uint32_t)mat_checked.disp >= g_num_textures) mat_checked.disp = -1;
    const ora_material* mat = &mat_checked;
    const accel_t as = {leaves, nodes, root, count};
    const float spread = 2.0f / w;
    ora_f3 tri[3];
    if (second_tri) { tri[0] = pair->v2; tri[1] = pair->v1; tri[2] = pair->v3; }
    else { tri[0] = pair->v0; tri[1] = pair->v1; tri[2] = pair->v2; }
    if (render_type == ORA_LODS) {                        /* (:543-556) */
        if (mat->texture != -1 && hit) {
            float lod = compute_lod(&ray, &rr, 2.0f / w, tri, &at, &g_textures[mat->texture]);
            float v = (float)(uint8_t)((int)lod * 20);   /* (unsigned char)(int(lod) * 20): integer wrap, <= 240 for 13 LODs */
            *alpha = v;
            return f3(v, v, v);
        }
        return f3(255, 0, 255);
    }
    if (!hit) return f3(0, 0, 0);
    if (render_type == ORA_MATERIALID)
        return hsv_to_rgb255((float)at.material_id / num_materials, 1.0f, 1.0f);
    if (render_type == ORA_TEXTURE) {                     /* (:557-578) */
        if (mat->texture != -1) {
            const ora_texture* tex = &g_textures[mat->texture];
            float lod = compute_lod(&ray, &rr, 2.0f / w, tri, &at, tex);
            uint8_t c[4];
            trilinear_sample(tex, interp_uv(&at, rr.bu, rr.bv), lod, c);
            *alpha = c[3];
            return f3(c[0], c[1], c[2]);
        }
        return f3(mat->diffuse.x * 255, mat->diffuse.y * 255, mat->diffuse.z * 255);
    }
    if (render_type == ORA_TEXTURE_LIT) return ambient_shader255(&as, &ray, &rr, mat, &at, light, spread, 1, 0, 1);
    if (render_type == ORA_TEXTURE_LIT_SHADOWS) return ambient_shader255(&as, &ray, &rr, mat, &at, light, spread, 1, 1, 1);
    /* ORA_DIFFUSE */
    return ambient_shader255(&as, &ray, &rr, mat, &at, light, spread, 0, 0, 0);
}

int ora_trace(const ora_triangle_pair* leaves, const ora_node* nodes, uint32_t root, uint32_t count,
              const ora_attributes* attributes, const ora_material* materials, uint32_t num_materials,
              const ora_camera* camera, const float light[3], int render_type, uint8_t* rgba8, uint32_t w,
              uint32_t h, uint32_t y0, uint32_t y1, uint32_t spp, uint64_t* counters)
{
    if (render_type < 0 || render_type > ORA_TEXTURE_LIT_SHADOWS) return -1;
    if (spp < 1) spp = 1;
    uint64_t box = 0, tri = 0, dropped = 0;
    uint32_t maxst = 0;
#pragma omp parallel for num_threads(g_threads) schedule(dynamic, 4) reduction(+ : box, tri, dropped) reduction(max : maxst)
    for (int64_t yy = y0; yy < (int64_t)y1; yy++) {
        uint32_t y = (uint32_t)yy;
        for (uint32_t x = 0; x < w; x++) {
            stats_t st = {0, 0, 0, 0};
            ora_f3 c;
            float alpha = 255.0f;
            if (spp == 1) {
                c = shade_pixel(leaves, nodes, root, count, attributes, materials, num_materials, camera, light,
                                render_type, x, y, w, h, 0.5f, 0.5f, &st, &alpha);
            } else {
                ora_f3 acc = f3(0, 0, 0);
                float aacc = 0.0f;
                /* stratified side x side sub-pixel grid: 2 x 2 for 4 spp, 4 x 4 for 16 spp (the spp extension is this
                 * build's, for BASELINE config 5; the reference traces one centred sample) */
                const uint32_t side = spp == 4 ? 2u : 4u;
                for (uint32_t s = 0; s < spp; s++) {
                    float ox = ((float)(s % side) + 0.5f) / (float)side, oy = ((float)((s / side) % side) + 0.5f) / (float)side, a1;
                    acc = add3(acc, shade_pixel(leaves, nodes, root, count, attributes, materials, num_materials,
                                                camera, light, render_type, x, y, w, h, ox, oy, &st, &a1));
                    aacc += a1;
                }
                c = f3(acc.x / (float)spp, acc.y / (float)spp, acc.z / (float)spp);
                alpha = aacc / (float)spp;
            }
            uint8_t* px = rgba8 + ((size_t)y * w + x) * 4;
            px[0] = sat_u8(c.x);
            px[1] = sat_u8(c.y);
            px[2] = sat_u8(c.z);
            px[3] = sat_u8(alpha);
            box += st.box_tests;
            tri += st.tri_tests;
            dropped += st.dropped;
            if (st.max_stack > maxst) maxst = st.max_stack;
        }
    }
    if (counters) {
        counters[0] += box;
        counters[1] += tri;
        if (maxst > counters[2]) counters[2] = maxst;
        counters[3] += dropped;   /* pushes dropped on a full stack (primary rays) */
    }
    return 0;
}

/* test hooks for rt_math.h (tests/test_oracle_cpu.py compares them with libm) */
float ora_rt_log2f(float x) { return rt_log2f(x); }
float ora_rt_exp2f(float x) { return rt_exp2f(x); }
double ora_rt_pow(double x, double y) { return rt_pow_d(x, y); }
double ora_rt_log2(double x) { return rt_log2_pos(x); }
double ora_rt_exp2(double x) { return rt_exp2_d(x); }

/* ------------------------------------------------------------------ the small arithmetic the reference keeps in
 * __host__ __device__ functions of Common.cuh, exported so that tests can hold the oracle's own helpers (the ones the
 * build paths above call) against the reference's code compiled from its tree (oracle/ref_pairing_driver.cpp). */
void ora_triangle_centre(const ora_triangle* t, float out[3])      /* Triangle::Centre, Common.cuh:240-242 */
{
    const ora_f3 c = tri_centre(t);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void ora_triangle_box(const ora_triangle* t, float out[6]) { tri_box(t, out); }   /* AABB(Triangle), Common.cuh:263-267 */
void ora_box_centre(const float b[6], float out[3]) { box_centre(b, out); }       /* AABB::Centre, Common.cuh:279 */
void ora_box_combine(const float a[6], const float b[6], float out[6])            /* Combine, Common.cuh:299-305 */
{
    for (int k = 0; k < 6; k++) out[k] = a[k];
    box_grow(out, b);
}
int ora_box_intersection(const float a[6], const float b[6], float out[6])        /* AABB::Intersection + Valid, :269-277 */
{
    box_intersection(a, b, out);
    return box_valid(out);
}
