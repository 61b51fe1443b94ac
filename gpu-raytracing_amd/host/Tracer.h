// Tracer.h -- the reference's Trace() (main.cu:125-192) / TraceRays (Tracer.cuh:21-23) over the C ABI.
#pragma once
#include <cstdint>
#include <vector>

#include "Arguments.h"
#include "Common.h"

struct DeviceSceneView {             // the fields of DeviceScene (Common.cuh:342-351) Trace() fills at main.cu:159-167
    const Attributes* attributes = nullptr;   // device
    const rt_material* materials = nullptr;   // device
    const rt_texture* textures = nullptr;     // device table, see DeviceTextureTable
    uint32_t num_attributes = 0, num_materials = 0, num_textures = 0;
    vec3 light{0, 0, 0};
};

// The texture part of Scene::CopyToDevice (main.cu:100-113): every mip level of every Library texture in device
// memory plus the rt_texture table that points at them.
struct DeviceTextureTable {
    rt_texture* table = nullptr;              // device
    uint32_t count = 0;
    void Upload(const Library& library);
    void Free();
private:
    std::vector<void*> allocations_;
};

// One frame: rows [y0, y1) of a dims_x x dims_y RGBA8 frame (linear device buffer, row 0 first -- the contents of
// the reference's GL surface).  num_tests: optional device uint64[4] ([0] = sum of box tests as printed by the
// reference at main.cu:180-183).  Asynchronous on `stream`.
void Trace(const TrianglePair* triangles, const Node* nodes, uint8_t* rgba8, int dims_x, int dims_y, const Camera* camera_dev,
           unsigned root, unsigned count, RenderType render_type, const DeviceSceneView& scene, uint64_t* num_tests,
           unsigned y0, unsigned y1, unsigned spp = 1, void* stream = nullptr);

// The interleaved-strip form of the same frame (rt_trace_strips; multi-GPU partition, Partition.h): strips first_strip,
// first_strip + strip_stride, ... rendered in ONE launch and stored compactly in rgba8_compact.
void TraceStrips(const TrianglePair* triangles, const Node* nodes, uint8_t* rgba8_compact, int dims_x, int dims_y,
                 const Camera* camera_dev, unsigned root, unsigned count, RenderType render_type, const DeviceSceneView& scene,
                 uint64_t* num_tests, unsigned strip_rows, unsigned first_strip, unsigned strip_stride, unsigned spp = 1,
                 void* stream = nullptr);
