#include "Tracer.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "MemoryBuffer.h"
#include "RadixSort.h"

static rt_scene SceneOf(const DeviceSceneView& scene, const Camera* camera_dev)
{
    rt_scene sc{};
    sc.attributes = scene.attributes;
    sc.materials = scene.materials;
    sc.camera = camera_dev;
    sc.light[0] = scene.light.x; sc.light[1] = scene.light.y; sc.light[2] = scene.light.z;
    sc.num_attributes = scene.num_attributes;
    sc.num_materials = scene.num_materials;
    sc.textures = scene.textures;
    sc.num_textures = scene.num_textures;
    return sc;
}

void Trace(const TrianglePair* triangles, const Node* nodes, uint8_t* rgba8, int dims_x, int dims_y, const Camera* camera_dev,
           unsigned root, unsigned count, RenderType render_type, const DeviceSceneView& scene, uint64_t* num_tests,
           unsigned y0, unsigned y1, unsigned spp, void* stream)
{
    rt_accel as{triangles, nodes, root, count};
    const rt_scene sc = SceneOf(scene, camera_dev);
    const int rc = rt_trace(&as, &sc, num_tests, (int)render_type, rgba8, (uint32_t)dims_x, (uint32_t)dims_y, y0, y1, spp, stream);
    if (rc != RT_OK) {
        fprintf(stderr, "gpu_assert: Trace: %s (%d)\n", rt_error_string(rc), rc);
        exit(rc < 0 ? -rc : rc);
    }
}

void TraceStrips(const TrianglePair* triangles, const Node* nodes, uint8_t* rgba8_compact, int dims_x, int dims_y,
                 const Camera* camera_dev, unsigned root, unsigned count, RenderType render_type, const DeviceSceneView& scene,
                 uint64_t* num_tests, unsigned strip_rows, unsigned first_strip, unsigned strip_stride, unsigned spp, void* stream)
{
    rt_accel as{triangles, nodes, root, count};
    const rt_scene sc = SceneOf(scene, camera_dev);
    const int rc = rt_trace_strips(&as, &sc, num_tests, (int)render_type, rgba8_compact, (uint32_t)dims_x, (uint32_t)dims_y,
                                   strip_rows, first_strip, strip_stride, spp, stream);
    if (rc != RT_OK) {
        fprintf(stderr, "gpu_assert: TraceStrips: %s (%d)\n", rt_error_string(rc), rc);
        exit(rc < 0 ? -rc : rc);
    }
}

void DeviceTextureTable::Upload(const Library& library)
{
    Free();
    count = (uint32_t)library.textures.size();
    if (!count) return;
    std::vector<rt_texture> host(count);
    for (uint32_t t = 0; t < count; t++) {
        const Texture& tex = library.textures[t];
        rt_texture& d = host[t];
        memset(&d, 0, sizeof(d));
        d.max_lod = tex.max_lod;
        for (uint32_t l = 0; l <= tex.max_lod; l++) {
            void* mip = nullptr;
            const size_t bytes = tex.mips[l].size() * sizeof(uint32_t);
            check(hipMalloc(&mip, bytes));
            check(hipMemcpy(mip, tex.mips[l].data(), bytes, hipMemcpyHostToDevice));
            allocations_.push_back(mip);
            d.mips[l] = static_cast<const uint32_t*>(mip);
            d.size_x[l] = tex.size_x[l];
            d.size_y[l] = tex.size_y[l];
        }
    }
    check(hipMalloc((void**)&table, sizeof(rt_texture) * count));
    allocations_.push_back(table);
    check(hipMemcpy(table, host.data(), sizeof(rt_texture) * count, hipMemcpyHostToDevice));
}

void DeviceTextureTable::Free()
{
    for (void* p : allocations_) (void)hipFree(p);
    allocations_.clear();
    table = nullptr;
    count = 0;
}

size_t RadixSortScratchBytes(uint32_t count) { return rt_radix_sort_scratch_bytes(count); }

void RadixSort(uint32_t* gpu_keys, uint32_t* gpu_values, uint32_t* gpu_temp1, uint32_t* gpu_temp2, uint32_t count,
               void* sort_scratch, void* stream)
{
    const int rc = rt_radix_sort_u32_pairs(gpu_keys, gpu_values, gpu_temp1, gpu_temp2, count, sort_scratch, stream);
    if (rc != RT_OK) {
        fprintf(stderr, "gpu_assert: RadixSort: %s (%d)\n", rt_error_string(rc), rc);
        exit(rc < 0 ? -rc : rc);
    }
}
