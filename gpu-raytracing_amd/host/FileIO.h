// FileIO.h -- mirrors the reference's FileIO.h:11-24 (FileIO.cpp:222-457): Wavefront .obj / .mtl loader.
#pragma once
#include <string>
#include <vector>

#include "Common.h"

struct Scene {
    std::vector<Triangle> triangles;
    std::vector<Attributes> attributes;
    Library library;
    AABB aabb;
    vec3 light;
};

// v / vt / vn / f (fan triangulation (0, i-1, i), negative indices), mtllib, usemtl.  Missing normals -> the flat
// normal normalize(cross(v1-v0, v2-v1)) on all corners; missing uvs -> (0,0); material_id = -1 without usemtl.
// scene.aabb = box of all vertices; scene.light = aabb centre unless a `light.txt` ("x y z") sits next to the .obj.
// Unlike the reference (256-byte lines, exit(1) on a missing file) lines may be any length and a missing file
// throws std::runtime_error.
Scene LoadOBJFromFile(const std::string& filename);
Library LoadMTLFromFile(const std::string& filename);
