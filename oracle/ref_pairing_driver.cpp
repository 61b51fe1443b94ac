// ref_pairing_driver.cpp -- TEST INFRASTRUCTURE: exposes the reference's own triangle-pairing code to the tests.
// It contains no restatement: it #includes Pairing.cuh from the reference tree where it lies (/root/reference/src,
// pure host-compilable arithmetic; the CUDA runtime headers of the image make __device__ an ignored attribute under
// g++) and forwards to CanFormTrianglePair / ShouldFormTrianglePair / CreateTrianglePair (Pairing.cuh:26-77) with the
// boxes the callers build (Multiblock.cu:157-179, BottomUpBuilder.cu:127-138).  Built into oracle/_ref by
// oracle/Makefile; only tests/ load it.
#include <cstddef>

#include "Pairing.cuh"

extern "C" {

// A, B: 9 floats each.  Returns 1 when the reference merges the two triangles; rot[2] = (rot_a, rot_b).
int ref_pair_decision(const float* A, const float* B, int rot[2])
{
    Triangle a(make_float3(A[0], A[1], A[2]), make_float3(A[3], A[4], A[5]), make_float3(A[6], A[7], A[8]));
    Triangle b(make_float3(B[0], B[1], B[2]), make_float3(B[3], B[4], B[5]), make_float3(B[6], B[7], B[8]));
    AABB a_aabb = {fminf(fminf(a.v0, a.v1), a.v2), fmaxf(fmaxf(a.v0, a.v1), a.v2)};
    AABB b_aabb = {fminf(fminf(b.v0, b.v1), b.v2), fmaxf(fmaxf(b.v0, b.v1), b.v2)};
    AABB p_aabb = Combine(a_aabb, b_aabb);
    Rotations r = {0, 0};
    const bool can = CanFormTrianglePair(a, b, r);
    rot[0] = r.rot_a; rot[1] = r.rot_b;
    return can && ShouldFormTrianglePair(a_aabb, b_aabb, p_aabb);
}

// out: the 64 bytes of CreateTrianglePair(a, b or NULL, a_id, b_id, r)
void ref_create_pair(const float* A, const float* B, unsigned a_id, unsigned b_id, int rot_a, int rot_b, void* out)
{
    Triangle a(make_float3(A[0], A[1], A[2]), make_float3(A[3], A[4], A[5]), make_float3(A[6], A[7], A[8]));
    Rotations r = {(unsigned short)rot_a, (unsigned short)rot_b};
    TrianglePair p;
    if (B) {
        Triangle b(make_float3(B[0], B[1], B[2]), make_float3(B[3], B[4], B[5]), make_float3(B[6], B[7], B[8]));
        p = CreateTrianglePair(&a, &b, a_id, b_id, r);
    } else {
        p = CreateTrianglePair(&a, NULL, a_id, 0, r);
    }
    memcpy(out, &p, sizeof p);
}

// sizeof / offsetof of the reference's PODs as this compiler lays them out (Common.cuh:44-59,152-243)
void ref_struct_layout(int* o)
{
    int k = 0;
    o[k++] = sizeof(Triangle); o[k++] = offsetof(Triangle, v0); o[k++] = offsetof(Triangle, v1); o[k++] = offsetof(Triangle, v2);
    o[k++] = sizeof(Node); o[k++] = offsetof(Node, min); o[k++] = offsetof(Node, max);
    o[k++] = sizeof(TrianglePair); o[k++] = offsetof(TrianglePair, v0); o[k++] = offsetof(TrianglePair, primitive_id_0);
    o[k++] = offsetof(TrianglePair, v1); o[k++] = offsetof(TrianglePair, primitive_id_1); o[k++] = offsetof(TrianglePair, v2);
    o[k++] = offsetof(TrianglePair, rotations); o[k++] = offsetof(TrianglePair, v3); o[k++] = offsetof(TrianglePair, pad3);
    o[k++] = sizeof(Camera); o[k++] = offsetof(Camera, position); o[k++] = offsetof(Camera, pitch); o[k++] = offsetof(Camera, w);
    o[k++] = offsetof(Camera, yaw); o[k++] = offsetof(Camera, u); o[k++] = offsetof(Camera, scale); o[k++] = offsetof(Camera, v);
    o[k++] = offsetof(Camera, max_depth);
    o[k++] = sizeof(Attributes); o[k++] = offsetof(Attributes, normal); o[k++] = offsetof(Attributes, uv); o[k++] = offsetof(Attributes, material_id);
    o[k++] = sizeof(AABB);
}

// a Node written through the reference's own bit-fields (Common.cuh:152-159)
void ref_pack_node(const float* mn, const float* mx, unsigned parent, unsigned count, unsigned child, unsigned type, void* out)
{
    Node n;
    memset(&n, 0, sizeof n);
    n.min = make_float3(mn[0], mn[1], mn[2]);
    n.max = make_float3(mx[0], mx[1], mx[2]);
    n.parent = parent; n.count = count; n.child = child; n.type = type;
    memcpy(out, &n, sizeof n);
}

// ---- Common.cuh's own __host__ __device__ arithmetic, called as the reference wrote it (no restatement): the tests hold
// the oracle's helpers (ora_triangle_centre, ora_triangle_box, ora_box_centre, ora_box_combine, ora_box_intersection)
// against these, bit for bit
static Triangle tri_of(const float* v)
{
    return Triangle(make_float3(v[0], v[1], v[2]), make_float3(v[3], v[4], v[5]), make_float3(v[6], v[7], v[8]));
}
static AABB box_of(const float* b) { return AABB(make_float3(b[0], b[1], b[2]), make_float3(b[3], b[4], b[5])); }
static void put_box(const AABB& a, float* o)
{
    o[0] = a.min.x; o[1] = a.min.y; o[2] = a.min.z; o[3] = a.max.x; o[4] = a.max.y; o[5] = a.max.z;
}
void ref_triangle_centre(const float* v, float* out)        // Triangle::Centre (Common.cuh:240-242)
{
    Triangle t = tri_of(v);
    const float3 c = t.Centre();
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void ref_triangle_box(const float* v, float* out) { put_box(AABB(tri_of(v)), out); }   // AABB(Triangle) (:263-267)
void ref_box_centre(const float* b, float* out)             // AABB::Centre (:279)
{
    AABB a = box_of(b);
    const float3 c = a.Centre();
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void ref_box_combine(const float* a, const float* b, float* out) { put_box(Combine(box_of(a), box_of(b)), out); }   // (:299-305)
int ref_box_intersection(const float* a, const float* b, float* out)   // AABB::Intersection + Valid (:269-277)
{
    AABB x = box_of(a);
    AABB r = x.Intersection(box_of(b));
    put_box(r, out);
    return r.Valid() ? 1 : 0;
}

}  // extern "C"
