// rt_pairing.hpp -- shared-edge triangle pairing (reference Pairing.cuh:9-77) on 9-float triangles.
#pragma once
#include "rt_device.hpp"

namespace rt {

// 16-byte loads from 4-byte-aligned addresses (global loads need only dword alignment on gfx950)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ void load_tri9(const float* t, float* v)
{
    const f4u a = *reinterpret_cast<const f4u*>(t), b = *reinterpret_cast<const f4u*>(t + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w; v[8] = t[8];
}

__device__ __forceinline__ bool vequal(const float* a, const float* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

// FindSharedEdge (Pairing.cuh:26-33): rotation of t that puts the directed edge a->b first, or -1
__device__ __forceinline__ int find_shared_edge(const float* a, const float* b, const float* t)
{
    if (vequal(a, t + 0) && vequal(b, t + 3)) return 0;
    if (vequal(a, t + 3) && vequal(b, t + 6)) return 2;
    if (vequal(a, t + 6) && vequal(b, t + 0)) return 1;
    return -1;
}

// CanFormTrianglePair (Pairing.cuh:42-58)
__device__ __forceinline__ bool can_form_pair(const float* A, const float* B, int& rot_a, int& rot_b)
{
    int t0 = 3, t1 = -1;
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const int u = v == 0 ? 2 : v - 1;
        if (t1 == -1) {
            t1 = find_shared_edge(A + 3 * v, A + 3 * u, B);
            t0--;
        }
    }
    if (t1 == -1) return false;
    rot_a = t0;
    rot_b = t1;
    return true;
}

__device__ __forceinline__ float sa6(const float* b)   // Common.cuh:293-297
{
    const float lx = b[3] - b[0], ly = b[4] - b[1], lz = b[5] - b[2];
    return 2.0f * (lx * ly + lx * lz + ly * lz);
}

// merge decision of the candidate (A, B) (BottomUpBuilder.cu:127-138; ShouldFormTrianglePair Pairing.cuh:35-39)
__device__ __forceinline__ bool pair_merges(const float* A, const float* B)
{
    int ra, rb;
    if (!can_form_pair(A, B, ra, rb)) return false;
    float ab[6], bb[6], cb[6];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        ab[k] = fminf(fminf(A[k], A[3 + k]), A[6 + k]); ab[3 + k] = fmaxf(fmaxf(A[k], A[3 + k]), A[6 + k]);
        bb[k] = fminf(fminf(B[k], B[3 + k]), B[6 + k]); bb[3 + k] = fmaxf(fmaxf(B[k], B[3 + k]), B[6 + k]);
        cb[k] = fminf(ab[k], bb[k]);                    cb[3 + k] = fmaxf(ab[3 + k], bb[3 + k]);
    }
    return sa6(cb) * 0.5f < sa6(ab) + sa6(bb);
}

}  // namespace rt
