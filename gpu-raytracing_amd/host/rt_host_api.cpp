// rt_host_api.cpp -- extern "C" hooks over the C++ host mirror so the Python tests can drive LoadOBJFromFile,
// InitialiseCamera / UpdateCamera, CountNodes and VerifyHierarchy (no GPU involved in any of them).
#include <cstring>
#include <exception>

#include "Arguments.h"
#include "Camera.h"
#include "FileIO.h"
#include "Partition.h"
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

uint32_t rth_num_textures(void* h) { return (uint32_t) static_cast<HostScene*>(h)->scene.library.textures.size(); }
uint32_t rth_texture_max_lod(void* h, uint32_t t) { return static_cast<HostScene*>(h)->scene.library.textures[t].max_lod; }
void rth_texture_size(void* h, uint32_t t, uint32_t lod, int out[2])
{
    const Texture& x = static_cast<HostScene*>(h)->scene.library.textures[t];
    out[0] = x.size_x[lod]; out[1] = x.size_y[lod];
}
const void* rth_texture_mip(void* h, uint32_t t, uint32_t lod) { return static_cast<HostScene*>(h)->scene.library.textures[t].mips[lod].data(); }

// Texture::GenerateLODs on a caller-supplied level 0; out_mips[l] (l >= 1) must hold size_x[l] * size_y[l] texels.
// Returns max_lod and fills the size tables (call with out_mips == nullptr first to learn the sizes).
uint32_t rth_generate_lods(const uint32_t* mip0, int sx, int sy, int* size_x, int* size_y, uint32_t** out_mips)
{
    Texture t;
    t.size_x[0] = sx; t.size_y[0] = sy;
    t.mips[0].assign(mip0, mip0 + (size_t)sx * sy);
    t.GenerateLODs();
    for (uint32_t l = 0; l <= t.max_lod; l++) {
        size_x[l] = t.size_x[l]; size_y[l] = t.size_y[l];
        if (out_mips && l >= 1) memcpy(out_mips[l], t.mips[l].data(), t.mips[l].size() * 4);
    }
    return t.max_lod;
}

void rth_initialise_camera(rt_camera* cam, const float aabb[6])
{
    AABB b;
    memcpy(&b, aabb, 24);
    InitialiseCamera(*cam, b);
}
void rth_update_camera(rt_camera* cam) { UpdateCamera(*cam); }
// keys: bit 0..6 = w, a, s, d, q, e, space (Input.cuh:4-15)
void rth_camera_move(rt_camera* cam, unsigned keys)
{
    InputState in;
    in.key_pressed_w = keys & 1; in.key_pressed_a = keys & 2; in.key_pressed_s = keys & 4; in.key_pressed_d = keys & 8;
    in.key_pressed_q = keys & 16; in.key_pressed_e = keys & 32; in.key_pressed_space = keys & 64;
    UpdateCameraPosition(*cam, in);
}
void rth_camera_look(rt_camera* cam, float dx, float dy) { UpdateCameraLookDelta(*cam, dx, dy); }
void rth_camera_zoom(rt_camera* cam, int dir) { UpdateCameraZoom(*cam, dir); }
// ParseCmd (Arguments.cpp:47-63): out = {build_type, enable_splits, enable_pairs, render_type}
void rth_parse_cmd(int argc, char** argv, int out[4])
{
    const Arguments a = ParseCmd(argc, argv);
    out[0] = (int)a.build_type; out[1] = a.enable_splits; out[2] = a.enable_pairs; out[3] = (int)a.render_type;
}

void rth_count_nodes(rt_node* nodes, unsigned root, unsigned count, int out[3])
{
    const HierarchyStats s = CountNodes(nodes, root, count);
    out[0] = s.numNodes; out[1] = s.numLeafNodes; out[2] = s.numTreeNodes;
}
int rth_verify_hierarchy(rt_node* nodes, unsigned root, unsigned count) { return VerifyHierarchy(nodes, root, count); }

// Partition.h (the multi-GPU host path's frame partition; compared with gpu-raytracing_amd/sharding.py on the CPU)
void rth_band_of(unsigned height, unsigned devices, unsigned d, unsigned out[2])
{
    const RowBand b = BandOf(height, devices, d);
    out[0] = b.y0; out[1] = b.y1;
}
unsigned rth_num_strips(unsigned height) { return NumStrips(height); }
unsigned rth_strips_per_device(unsigned height, unsigned devices) { return StripsPerDevice(height, devices); }
unsigned rth_strips_owned(unsigned height, unsigned devices, unsigned d) { return StripsOwned(height, devices, d); }
unsigned rth_compact_rows(unsigned height, unsigned devices) { return CompactRows(height, devices); }
unsigned rth_strip_rows_in_frame(unsigned height, unsigned s) { return StripRowsInFrame(height, s); }
int rth_choose_partition(const double* costs, unsigned devices) { return (int)ChoosePartition(costs, devices); }
// GatherPlan: out[k] = {kind, device, src_off, dst_off, bytes, src_pitch, dst_pitch, pieces} as uint64; returns the op count
unsigned rth_gather_plan(unsigned width, unsigned height, unsigned devices, int strips, uint64_t* out)
{
    GatherOp ops[kMaxGatherOps];
    if (devices == 0 || devices > 64) return 0;
    const unsigned k = GatherPlan(width, height, devices, strips ? Partition::kStrips : Partition::kBands, ops);
    for (unsigned i = 0; i < k; i++) {
        const GatherOp& o = ops[i];
        const uint64_t v[8] = {o.kind, o.device, o.src_off, o.dst_off, o.bytes, o.src_pitch, o.dst_pitch, o.pieces};
        for (int j = 0; j < 8; j++) out[i * 8 + j] = v[j];
    }
    return k;
}

}  // extern "C"
