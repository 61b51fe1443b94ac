// hybrid_top.hip -- the hybrid build's SAH top tree above the LBVH.
//
// Replaces ExtractDepth (BottomUpBuilder.cu:314-371) + SharedTaskBuild as launched at
// BuildWrapper.cu:350-361 (SharedTaskBuilder.cu:93-607, 909-967): <= 256 LBVH sub-roots 8 levels below
// the root are re-organised by a top-down binned-SAH build (8 bins, longest centroid axis, leaf
// threshold 2) written at slots >= 2L; the tracer then starts at (2L+1, 2) (main.cu:222-223).
//
// The reference emits sub-roots and allocates node slots in atomic-arrival order, so its numbering is
// not reproducible.  This kernel is DETERMINISTIC: sub-roots in thread-id order, tasks
// level by level in left-to-right order, slots allocated by prefix sums in that order, ids partitioned
// stably -- the same rules as oracle/rt_oracle.c: ora_build_hybrid_top, so the two agree bit for bit.
// One workgroup, one thread per sub-root position, everything in LDS; per level: classify tasks ->
// bin (LDS integer atomics on the ordered-int encoding, order independent) -> select planes -> one
// block scan for the stable partition, one (packed) for bin slots, node slots and queue positions -> emit nodes and
// child tasks.  Inside the level loop the barriers order LDS only: the kernel reads back nothing it stores to memory.
#include <mutex>

#include "rt_device.hpp"
#include "rt_launch.hpp"

namespace rt {

constexpr int kTopMax = 256;            // sub-roots (ExtractDepth runs 256 threads, depth 8)
constexpr int kTopSplitMax = 88;        // tasks with > 2 prims alive in one level (<= 256 / 3)
constexpr float kFltMax = 3.402823466e+38f;

struct TopSmem {
    uint32_t sub[kTopMax];
    float box[kTopMax][6];
    uint32_t ids[2][kTopMax];
    // task queues (structure of arrays), two levels
    float tc[2][kTopMax][6], tp[2][kTopMax][6];
    uint32_t tstart[2][kTopMax], tend[2][kTopMax], tparent[2][kTopMax];
    // per-level scratch
    int kind[kTopMax];                  // 0 leaf, 1 binned split, 2 median split
    int axis[kTopMax], plane[kTopMax], binslot[kTopMax];
    float k1[kTopMax];
    uint32_t mid[kTopMax];
    float cc[kTopMax][2][6], cp[kTopMax][2][6];   // child centroid / primitive boxes
    int bins[kTopSplitMax][8][13];      // ordered-int p box [6], c box [6], count
    int binof[kTopMax];                 // bin of the primitive at a position
    int child_q[kTopMax], child_m[kTopMax];   // per task that splits: first child's place in the next queue, first position of the right child
    uint32_t lex[kTopMax + 1];          // exclusive scan of the "goes left" flags
    uint32_t ws[16];
    uint32_t num_tasks, write_index;
};

__device__ __forceinline__ float box_sa(const float* b)   // Common.cuh:293-297
{
    const float lx = b[3] - b[0], ly = b[4] - b[1], lz = b[5] - b[2];
    return 2.0f * (lx * ly + lx * lz + ly * lz);
}

__device__ __forceinline__ void put_node(rt_node* n, const float* b, uint32_t child, uint32_t count, uint32_t type)
{
    uint4* o = reinterpret_cast<uint4*>(n);
    o[0] = make_uint4(__float_as_uint(b[0]), __float_as_uint(b[1]), __float_as_uint(b[2]), count << 29);
    o[1] = make_uint4(__float_as_uint(b[3]), __float_as_uint(b[4]), __float_as_uint(b[5]), (child & kIndexMask) | (type << 29));
}

__global__ __launch_bounds__(256) void hybrid_top_kernel(rt_node* nodes, const int* aabb_ordered, uint32_t L,
                                                         const uint32_t* n_dev)
{
    if (n_dev) L = *n_dev;   // --pairs: the leaf count is a device value
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    TopSmem& S = *reinterpret_cast<TopSmem*>(smem_raw);
    const uint32_t tid = threadIdx.x;
    const uint32_t base = max(2u * L, 2u);

    // ---------------- ExtractDepth: thread tid follows bit d of tid at level d (set: child of slot cur,
    // clear: child of slot cur+1); a pair with a leaf (or empty) slot stops the walk, emitted once.
    uint32_t cur = 0;
    bool emit = true;
    for (uint32_t d = 0; d < 8; d++) {
        const uint32_t t0 = nodes[cur].w28 >> 29, t1 = nodes[cur + 1].w28 >> 29;
        if (t0 != RT_CHILD_BOX || t1 != RT_CHILD_BOX) { emit = (tid >> d) == 0; break; }
        cur = (((tid >> d) & 1u) ? nodes[cur].w28 : nodes[cur + 1].w28) & kIndexMask;
    }
    uint32_t K;
    const uint32_t pos = block_excl_scan_u32<256>(emit ? 1u : 0u, S.ws, &K);
    if (emit) {
        S.sub[pos] = cur;
        const rt_node a = nodes[cur], b = nodes[cur + 1];
        const bool has_b = (b.w28 >> 29) != RT_CHILD_NONE;
        S.box[pos][0] = has_b ? fminf(a.min.x, b.min.x) : a.min.x;
        S.box[pos][1] = has_b ? fminf(a.min.y, b.min.y) : a.min.y;
        S.box[pos][2] = has_b ? fminf(a.min.z, b.min.z) : a.min.z;
        S.box[pos][3] = has_b ? fmaxf(a.max.x, b.max.x) : a.max.x;
        S.box[pos][4] = has_b ? fmaxf(a.max.y, b.max.y) : a.max.y;
        S.box[pos][5] = has_b ? fmaxf(a.max.z, b.max.z) : a.max.z;
    }
    S.ids[0][tid] = tid;     // tmp_ids = 0, 1, 2, ... (BuildWrapper.cu:292-293,303)
    __syncthreads();

    // root task: c_aabb = union of the sub-root BOXES (:341-346), p_aabb = the scene box (:324-326).  Min / max in the
    // ordered-int encoding: wave reductions, then one LDS atomic per wave and plane (a serial loop of one thread per plane over
    // 256 boxes was a sixth of this kernel's time)
    int* rb = reinterpret_cast<int*>(&S.cc[0][0][0]);   // six words of level scratch nobody uses yet
    if (tid < 6) rb[tid] = tid < 3 ? 0x7f7fffff : (int)0x80800000;   // ordered-int +FLT_MAX / -FLT_MAX
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int v = tid < K ? float_to_ordered_int(S.box[tid][k]) : (k < 3 ? 0x7f7fffff : (int)0x80800000);
        const int r = k < 3 ? wave_min_i32(v) : wave_max_i32(v);
        if ((tid & 63u) == 0) { if (k < 3) atomicMin(&rb[k], r); else atomicMax(&rb[k], r); }
    }
    __syncthreads();
    if (tid < 6) {
        S.tc[0][0][tid] = ordered_int_to_float(rb[tid]);
        S.tp[0][0][tid] = ordered_int_to_float(aabb_ordered[tid]);
    }
    if (tid == 0) {
        S.tstart[0][0] = 0;
        S.tend[0][0] = K;
        S.tparent[0][0] = base;
        S.write_index = base + 1;
        S.num_tasks = 1;
    }
    __syncthreads();
    if (K == 1) {
        // the reference writes the lone leaf INTO slot 2L and then traces from (2L+1, 2): undefined.
        // Defined: 2L = Box{2L+1, count 1}, 2L+1 = the leaf descriptor, 2L+2 = None.
        if (tid == 0) {
            put_node(&nodes[base], S.tp[0][0], base + 1, 1, RT_CHILD_BOX);
            put_node(&nodes[base + 1], S.box[0], S.sub[0], 2, RT_CHILD_BOX);
            uint4* z = reinterpret_cast<uint4*>(&nodes[base + 2]);
            z[0] = make_uint4(0, 0, 0, 0);
            z[1] = make_uint4(0, 0, 0, 0);
        }
        return;
    }

    int q = 0;  // current queue / ids buffer (alternates per level, like Task.buffer_idx)
    int mytask = tid < K ? 0 : -1;   // the root task holds positions [0, K)
    while (true) {
        const uint32_t T = S.num_tasks;
        if (T == 0) break;
        // ---- A: classify tasks; splitting tasks get a bin slot
        int my_kind = 0;
        uint32_t my_count = 0;
        if (tid < T) {
            my_count = S.tend[q][tid] - S.tstart[q][tid];
            if (my_count > 2) {                                    // LEAF_THRESHOLD 2 (SharedTaskBuilder.cu:13,390)
                const float* c = S.tc[q][tid];
                if (box_sa(c) <= 0.0f) my_kind = 2;                // all centroids coincide: object median split (:465)
                else {
                    my_kind = 1;
                    const float lx = c[3] - c[0], ly = c[4] - c[1], lz = c[5] - c[2];
                    const int ax = 2 * (lz > lx && lz > ly) + 1 * (ly > lx && ly >= lz);   // SelectAxis (:197-204)
                    S.axis[tid] = ax;
                    S.k1[tid] = 8 * (1 - 1.1920929e-7f) / (c[3 + ax] - c[ax]);             // (:209-212)
                }
            }
            S.kind[tid] = my_kind;
        }
        // one scan for three allocations, packed (a task that splits -- binned or by the median -- takes 2 node slots and 2
        // queue entries whatever SelectPlane decides later; a leaf task takes its primitive count): bin slot : 8 | node slots
        // : 12 | queue entries : 12
        // (the fields cannot carry into each other: tasks that split hold >= 3 of the <= kTopMax primitives each, a level
        // allocates at most 2 node slots / queue entries per task)
        static_assert(kTopMax / 3 < (1 << 8) && 2 * kTopMax < (1 << 12), "packed scan: bin slot : 8 | node slots : 12 | queue entries : 12");
        const bool splits = tid < T && my_count > 2;
        const uint32_t need = tid < T ? (splits ? 2u : (my_count == 1 ? 0u : my_count)) : 0u;
        uint32_t packed_total;
        const uint32_t packed = block_excl_scan_lds<256>(((tid < T && my_kind == 1) ? 1u : 0u) | (need << 8) | ((splits ? 2u : 0u) << 20),
                                                         S.ws, &packed_total);
        const uint32_t nsplit = packed_total & 0xFFu, total_need = (packed_total >> 8) & 0xFFFu, total_children = packed_total >> 20;
        const uint32_t slot = packed & 0xFFu, alloc = (packed >> 8) & 0xFFFu, qpos = packed >> 20;
        if (tid < T) S.binslot[tid] = (int)slot;
        for (uint32_t e = tid; e < nsplit * 8 * 13; e += 256) {
            const uint32_t f = e % 13;
            (&S.bins[0][0][0])[e] = f == 12 ? 0 : 0x7f7fffff;   // ordered-int empty ("max" words are kept complemented: ~0x80800000)
        }
        lds_barrier();
        // ---- B: every position bins its primitive (`mytask`: the task whose range holds position tid, carried from level to
        // level -- a position of a task that split goes to one of its two children, see E)
        if (mytask >= 0 && S.kind[mytask] == 1) {
            const float* bx = S.box[S.ids[q][tid]];
            const int ax = S.axis[mytask];
            const float centre[3] = {(bx[0] + bx[3]) * 0.5f, (bx[1] + bx[4]) * 0.5f, (bx[2] + bx[5]) * 0.5f};
            int bin = (int)(S.k1[mytask] * (centre[ax] - S.tc[q][mytask][ax]));   // BinCentroids (:206-264)
            bin = min(max(bin, 0), 7);   // the reference prints "bin out of bounds" and abandons the build
            S.binof[tid] = bin;
            int* B = S.bins[S.binslot[mytask]][bin];
            // the positions of one (task, bin) update the same twelve words: "max" words complemented so that every update is
            // the same ds_min_i32, and thread i starts at word i mod 12 (values rotated to match, as in sah_build.hip)
            int val[12];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                val[k] = float_to_ordered_int(bx[k]);
                val[3 + k] = ~float_to_ordered_int(bx[3 + k]);
                val[6 + k] = float_to_ordered_int(centre[k]);
                val[9 + k] = ~float_to_ordered_int(centre[k]);
            }
            const uint32_t r = tid % 12u;
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
                atomicMin(&B[w], val[k]);
                w = w == 11u ? 0u : w + 1u;
            }
            atomicAdd(&B[12], 1);
        }
        lds_barrier();
        // ---- C: one thread per splitting task: SelectPlane (:297-350) or the median split
        if (tid < T && my_kind != 0) {
            const uint32_t s = S.tstart[q][tid], e = S.tend[q][tid];
            bool done = false;
            if (my_kind == 1) {
                float bp[8][6], bc[8][6];
                uint32_t bn[8];
                const int (*B)[13] = S.bins[S.binslot[tid]];
                for (int b = 0; b < 8; b++) {
                    for (int k = 0; k < 6; k++) {
                        bp[b][k] = ordered_int_to_float(k < 3 ? B[b][k] : ~B[b][k]);
                        bc[b][k] = ordered_int_to_float(k < 3 ? B[b][6 + k] : ~B[b][6 + k]);
                    }
                    bn[b] = (uint32_t)B[b][12];
                }
                float lp[7][6], lc[7][6];
                uint32_t ln[7];
                for (int k = 0; k < 6; k++) { lp[0][k] = bp[0][k]; lc[0][k] = bc[0][k]; }
                ln[0] = bn[0];
                for (int i = 1; i < 7; i++) {
                    for (int k = 0; k < 3; k++) {
                        lp[i][k] = fminf(lp[i - 1][k], bp[i][k]); lp[i][3 + k] = fmaxf(lp[i - 1][3 + k], bp[i][3 + k]);
                        lc[i][k] = fminf(lc[i - 1][k], bc[i][k]); lc[i][3 + k] = fmaxf(lc[i - 1][3 + k], bc[i][3 + k]);
                    }
                    ln[i] = ln[i - 1] + bn[i];
                }
                float rp[6], rc[6];
                for (int k = 0; k < 6; k++) { rp[k] = bp[7][k]; rc[k] = bc[7][k]; }
                uint32_t rn = bn[7];
                float best = kFltMax;
                int plane = -1;
                for (int i = 6; i >= 0; i--) {
                    const float score = box_sa(lp[i]) * ln[i] + box_sa(rp) * rn;
                    if (score < best && ln[i] && rn) {
                        best = score;
                        plane = i;
                        for (int k = 0; k < 6; k++) {
                            S.cp[tid][0][k] = lp[i][k]; S.cp[tid][1][k] = rp[k];
                            S.cc[tid][0][k] = lc[i][k]; S.cc[tid][1][k] = rc[k];
                        }
                    }
                    for (int k = 0; k < 3; k++) {
                        rp[k] = fminf(rp[k], bp[i][k]); rp[3 + k] = fmaxf(rp[3 + k], bp[i][3 + k]);
                        rc[k] = fminf(rc[k], bc[i][k]); rc[3 + k] = fmaxf(rc[3 + k], bc[i][3 + k]);
                    }
                    rn += bn[i];
                }
                S.plane[tid] = plane;
                done = plane >= 0;   // else the reference's "failed to find valid partition": median split instead
            }
            if (!done) {
                S.kind[tid] = 2;
                const uint32_t m = s + ((e - s) >> 1);
                S.mid[tid] = m;
                float cp[2][6], cc[2][6];
                for (int h = 0; h < 2; h++)
                    for (int k = 0; k < 6; k++) { cp[h][k] = k < 3 ? kFltMax : -kFltMax; cc[h][k] = cp[h][k]; }
                for (uint32_t i = s; i < e; i++) {
                    const float* bx = S.box[S.ids[q][i]];
                    const int h = i >= m;
                    for (int k = 0; k < 3; k++) {
                        const float ctr = (bx[3 + k] + bx[k]) * 0.5f;
                        cp[h][k] = fminf(cp[h][k], bx[k]); cp[h][3 + k] = fmaxf(cp[h][3 + k], bx[3 + k]);
                        cc[h][k] = fminf(cc[h][k], ctr);   cc[h][3 + k] = fmaxf(cc[h][3 + k], ctr);
                    }
                }
                for (int h = 0; h < 2; h++)
                    for (int k = 0; k < 6; k++) { S.cp[tid][h][k] = cp[h][k]; S.cc[tid][h][k] = cc[h][k]; }
            }
        }
        lds_barrier();
        // ---- D: stable partition of the ids (PartitionIds :352-380, made stable) via one block scan
        bool left = false;
        if (mytask >= 0) {
            const int kd = S.kind[mytask];
            left = kd == 1 ? (S.binof[tid] <= S.plane[mytask]) : (kd == 2 ? tid < S.mid[mytask] : false);
        }
        uint32_t total_left;
        const uint32_t lx = block_excl_scan_lds<256>(left ? 1u : 0u, S.ws, &total_left);
        S.lex[tid] = lx;
        if (tid == 0) S.lex[kTopMax] = total_left;
        lds_barrier();
        const int nq = q ^ 1;
        if (mytask >= 0 && S.kind[mytask] != 0) {
            const uint32_t s = S.tstart[q][mytask], e = S.tend[q][mytask];
            const uint32_t ls = S.lex[s], nleft = S.lex[e] - ls, mine = S.lex[tid] - ls;
            const uint32_t np = left ? s + mine : s + nleft + ((tid - s) - mine);
            S.ids[nq][np] = S.ids[q][tid];
        }
        // ---- E: slot and queue allocation in task order, then emit nodes and child tasks (:396-464, :544-606)
        const int kd = tid < T ? S.kind[tid] : 0;
        const uint32_t wi = S.write_index;
        if (tid < T) {
            const uint32_t s = S.tstart[q][tid], e = S.tend[q][tid], parent = S.tparent[q][tid];
            if (kd == 0) {
                const uint32_t child = my_count == 1 ? parent : wi + alloc;
                for (uint32_t i = 0; i < my_count; i++) {
                    const uint32_t prim = S.ids[q][s + i];
                    put_node(&nodes[child + i], S.box[prim], S.sub[prim], 2, RT_CHILD_BOX);   // leaf_type = Box, count = 2
                }
                if (my_count > 1) put_node(&nodes[parent], S.tp[q][tid], child, my_count, RT_CHILD_BOX);
            } else {
                const uint32_t child_index = wi + alloc;
                put_node(&nodes[parent], S.tp[q][tid], child_index, 2, RT_CHILD_BOX);
                const uint32_t m = kd == 1 ? s + (S.lex[e] - S.lex[s]) : S.mid[tid];
                S.child_q[tid] = (int)qpos;
                S.child_m[tid] = (int)m;
                for (int h = 0; h < 2; h++) {
                    const uint32_t o = qpos + h;
                    for (int k = 0; k < 6; k++) { S.tc[nq][o][k] = S.cc[tid][h][k]; S.tp[nq][o][k] = S.cp[tid][h][k]; }
                    S.tstart[nq][o] = h ? m : s;
                    S.tend[nq][o] = h ? e : m;
                    S.tparent[nq][o] = child_index + h;
                }
            }
        }
        lds_barrier();
        if (tid == 0) {
            S.write_index = wi + total_need;
            S.num_tasks = total_children;
        }
        if (mytask >= 0) mytask = S.kind[mytask] == 0 ? -1 : S.child_q[mytask] + (tid >= (uint32_t)S.child_m[mytask] ? 1 : 0);
        lds_barrier();
        q = nq;
    }
}

// n == 0: no LBVH, no top tree; the caller still traces from (2n+1, 2) = (1, 2): make those slots None
__global__ void hybrid_empty_kernel(rt_node* nodes)
{
    reinterpret_cast<uint32_t*>(nodes)[threadIdx.x] = 0;   // slots 0..7
}

hipError_t launch_hybrid_top(rt_node* nodes, const int* aabb_ordered, uint32_t n, hipStream_t st, const uint32_t* n_dev)
{
    static PerDeviceOnce once;
    const hipError_t attr_err = once([] {
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&hybrid_top_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TopSmem));
    });
    if (attr_err != hipSuccess) return attr_err;
    if (n == 0) hybrid_empty_kernel<<<1, 64, 0, st>>>(nodes);
    else hybrid_top_kernel<<<1, 256, sizeof(TopSmem), st>>>(nodes, aabb_ordered, n, n_dev);
    return hipGetLastError();
}

}  // namespace rt
