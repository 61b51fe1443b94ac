// rt_host_api.cpp -- extern "C" hooks over the C++ host mirror so the Python tests can drive LoadOBJFromFile,
// InitialiseCamera / UpdateCamera, CountNodes and VerifyHierarchy (no GPU involved in any of them).
#include <cstring>
#include <exception>

#include "Camera.h"
#include "FileIO.h"
#include "Utilities.h"

struct HostScene {
    Scene scene;
    std::vector<rt_material> mats;
};

extern "C" {

void* rth_load_obj(const char* path)
{
    try {
        auto* h = new HostScene;
        h->scene = LoadOBJFromFile(path);
        for (const Material& m : h->scene.library.materials) h->mats.push_back(m.pod());
        return h;
    } catch (const std::exception&) {
        return nullptr;
    }
}
void rth_free(void* h) { delete static_cast<HostScene*>(h); }
uint32_t rth_num_triangles(void* h) { return (uint32_t) static_cast<HostScene*>(h)->scene.triangles.size(); }
const void* rth_triangles(void* h) { return static_cast<HostScene*>(h)->scene.triangles.data(); }
const void* rth_attributes(void* h) { return static_cast<HostScene*>(h)->scene.attributes.data(); }
uint32_t rth_num_materials(void* h) { return (uint32_t) static_cast<HostScene*>(h)->mats.size(); }
const void* rth_materials(void* h) { return static_cast<HostScene*>(h)->mats.data(); }
void rth_scene_aabb(void* h, float out[6]) { memcpy(out, &static_cast<HostScene*>(h)->scene.aabb, 24); }
void rth_light(void* h, float out[3]) { memcpy(out, &static_cast<HostScene*>(h)->scene.light, 12); }

void rth_initialise_camera(rt_camera* cam, const float aabb[6])
{
    AABB b;
    memcpy(&b, aabb, 24);
    InitialiseCamera(*cam, b);
}
void rth_update_camera(rt_camera* cam) { UpdateCamera(*cam); }

void rth_count_nodes(rt_node* nodes, unsigned root, unsigned count, int out[3])
{
    const HierarchyStats s = CountNodes(nodes, root, count);
    out[0] = s.numNodes; out[1] = s.numLeafNodes; out[2] = s.numTreeNodes;
}
int rth_verify_hierarchy(rt_node* nodes, unsigned root, unsigned count) { return VerifyHierarchy(nodes, root, count); }

}  // extern "C"
