// Common.h -- host-side types of the MI355X build, mirroring the reference's Common.cuh (file:line cited per item).
// The device-visible PODs are the C-ABI structs of include/rt_abi.h (byte-identical to the reference layouts).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "rt_abi.h"

using vec3 = rt_float3;   // the role of CUDA float3 in the reference (HIP headers own the name float3)
using Triangle = rt_triangle;            // Common.cuh:199-243
using TrianglePair = rt_triangle_pair;   // Common.cuh:161-197
using Node = rt_node;                    // Common.cuh:152-159 (w12 = parent:29|count:3, w28 = child:29|type:3)
using Camera = rt_camera;                // Common.cuh:44-53
using Attributes = rt_attributes;        // Common.cuh:55-59

enum ChildType { ChildType_None = 0, ChildType_Box = 1, ChildType_Tri = 2, ChildType_Inst = 3, ChildType_Proc = 4 };  // Common.cuh:35-41

inline uint32_t NodeParent(const Node& n) { return n.w12 & 0x1FFFFFFFu; }
inline uint32_t NodeCount(const Node& n) { return n.w12 >> 29; }
inline uint32_t NodeChild(const Node& n) { return n.w28 & 0x1FFFFFFFu; }
inline uint32_t NodeType(const Node& n) { return n.w28 >> 29; }

// minimal vec3 algebra with the operation order of the reference's helper_math.h
inline vec3 make_vec3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 make_vec3(float s) { return vec3{s, s, s}; }
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline vec3 operator*(float b, vec3 a) { return {b * a.x, b * a.y, b * a.z}; }
inline vec3 operator/(vec3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
inline vec3 fminf(vec3 a, vec3 b) { return {::fminf(a.x, b.x), ::fminf(a.y, b.y), ::fminf(a.z, b.z)}; }
inline vec3 fmaxf(vec3 a, vec3 b) { return {::fmaxf(a.x, b.x), ::fmaxf(a.y, b.y), ::fmaxf(a.z, b.z)}; }

struct AABB {  // Common.cuh:245-285
    vec3 min, max;
    vec3 Centre() const { return (min + max) * 0.5f; }
};
inline AABB Combine(const AABB& a, const vec3& b) { return {fminf(a.min, b), fmaxf(a.max, b)}; }  // Common.cuh:307-313

// Material as the host sees it (Common.cuh:93-129); `pod()` is what goes to the device (rt_material)
struct Material {
    std::string name;
    vec3 ambient{0, 0, 0}, diffuse{0, 0, 0}, specular{0, 0, 0};
    float specular_exp = 0.0f;
    int32_t texture = -1, bump = -1, disp = -1;
    std::string texture_file, bump_file, disp_file;  // recorded, not decoded (textured modes: SURVEY 8(f) rank 2)
    explicit Material(std::string s = "") : name(std::move(s)) {}
    rt_material pod() const { return rt_material{ambient, diffuse, specular, specular_exp, texture, bump, disp}; }
};

// Texture (Common.cuh:61-91): an RGBA8 mip chain on the host; GenerateLODs = the box filter of FileIO.cpp:121-150.
// The reference decodes images with the vendored stb_image; this build reads binary PPM (P6) only.
constexpr int NUM_LODS = RT_NUM_LODS;   // Common.cuh:17
struct Texture {
    std::string name;
    std::vector<uint32_t> mips[NUM_LODS];   // texel = r | g << 8 | b << 16 | a << 24, row 0 first
    int size_x[NUM_LODS] = {0}, size_y[NUM_LODS] = {0};
    uint32_t max_lod = 0;
    uint32_t ReadTexel(int x, int y, int lod) const;          // coordinates clamped (FileIO.cpp:109-114)
    void GenerateLODs();
};

struct Library {  // Common.cuh:131-150
    std::vector<Material> materials;
    std::vector<Texture> textures;
    std::map<std::string, uint32_t> name_to_mat;
    std::map<std::string, uint32_t> name_to_tex;
    int32_t AddTexture(const std::string& filename);           // FileIO.cpp:166-184; -1 when the file cannot be decoded
    void AddMaterial(const std::string& name) { name_to_mat[name] = (uint32_t)materials.size(); materials.emplace_back(name); }
    int32_t GetMaterialId(const std::string& name) const
    {
        auto it = name_to_mat.find(name);
        return it == name_to_mat.end() ? -1 : (int32_t)it->second;
    }
};
