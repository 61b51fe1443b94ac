#include "Tracer.h"

#include <cstdio>
#include <cstdlib>

#include "RadixSort.h"

void Trace(const TrianglePair* triangles, const Node* nodes, uint8_t* rgba8, int dims_x, int dims_y, const Camera* camera_dev,
           unsigned root, unsigned count, RenderType render_type, const DeviceSceneView& scene, uint64_t* num_tests,
           unsigned y0, unsigned y1, unsigned spp, void* stream)
{
    rt_accel as{triangles, nodes, root, count};
    rt_scene sc{};
    sc.attributes = scene.attributes;
    sc.materials = scene.materials;
    sc.camera = camera_dev;
    sc.light[0] = scene.light.x; sc.light[1] = scene.light.y; sc.light[2] = scene.light.z;
    sc.num_attributes = scene.num_attributes;
    sc.num_materials = scene.num_materials;
    const int rc = rt_trace(&as, &sc, num_tests, (int)render_type, rgba8, (uint32_t)dims_x, (uint32_t)dims_y, y0, y1, spp, stream);
    if (rc != RT_OK) {
        fprintf(stderr, "gpu_assert: Trace: %s (%d)\n", rt_error_string(rc), rc);
        exit(rc < 0 ? -rc : rc);
    }
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
