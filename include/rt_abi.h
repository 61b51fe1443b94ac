/*
 * rt_abi.h -- C ABI of the MI355X-native LBVH builder + primary-ray tracer (librt_amd.so).
 *
 * This is the drop-in boundary for the hot path of gregc-91/GPU-Raytracing.  Every entry point
 * names the reference interface it replaces (file:line relative to /root/reference/src).  All
 * pointers are DEVICE pointers unless said otherwise; the caller allocates and owns every buffer
 * (as the reference's Display() does, main.cu:226-240); `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Nothing here allocates, frees or synchronises, so a call sequence can be
 * captured in a hipGraph.  Return value: 0 on success, RT_ERR_* (<0) on bad arguments, or
 * -(hipError_t) - 1000 when a HIP call failed (the reference calls exit() instead, Common.cuh:358-366).
 *
 * POD layouts are byte-identical to the reference's (sizes checked by static_assert in the library):
 *   rt_triangle 36, rt_node 32, rt_triangle_pair 64, rt_camera 64, rt_attributes 72.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_float3 { float x, y, z; } rt_float3;

/* Common.cuh:199-243 Triangle */
typedef struct rt_triangle { rt_float3 v0, v1, v2; } rt_triangle;

/* Common.cuh:152-159 Node: a child-descriptor slot; slots come in sibling pairs (2i, 2i+1).
 * w12 = parent:29 (LSBs) | count:3     w28 = child:29 (LSBs) | type:3 (rt_child_type) */
typedef struct rt_node { rt_float3 min; uint32_t w12; rt_float3 max; uint32_t w28; } rt_node;

/* Common.cuh:161-197 TrianglePair */
typedef struct rt_triangle_pair {
    rt_float3 v0; uint32_t primitive_id_0;
    rt_float3 v1; uint32_t primitive_id_1;
    rt_float3 v2; uint16_t rotations[2];
    rt_float3 v3; float pad3;
} rt_triangle_pair;

/* Common.cuh:44-53 Camera */
typedef struct rt_camera {
    rt_float3 position; float pitch;
    rt_float3 w;        float yaw;
    rt_float3 u;        float scale;
    rt_float3 v;        float max_depth;
} rt_camera;

/* Common.cuh:55-59 Attributes (float2 is 8-byte aligned in CUDA, hence the pads) */
typedef struct rt_attributes {
    rt_float3 normal[3]; uint32_t pad0;
    float uv[3][2];
    int32_t material_id; uint32_t pad1;
} rt_attributes;

/* POD mirror of the device-read fields of Material (Common.cuh:93-129; the reference memcpy's a
 * struct holding a std::string to the device) */
typedef struct rt_material {
    rt_float3 ambient, diffuse, specular;
    float specular_exp;
    int32_t texture, bump, disp;
} rt_material;

/* POD mirror of the device-read fields of Texture (Common.cuh:61-91): an RGBA8 mip chain.  mips[l] is a DEVICE pointer to
 * size_x[l] * size_y[l] texels (row 0 first), levels 0..max_lod as Texture::GenerateLODs makes them (FileIO.cpp:121-150). */
#define RT_NUM_LODS 13
typedef struct rt_texture {
    const uint32_t* mips[RT_NUM_LODS];
    int32_t size_x[RT_NUM_LODS], size_y[RT_NUM_LODS];
    uint32_t max_lod, pad;
} rt_texture;

typedef enum rt_child_type { RT_CHILD_NONE = 0, RT_CHILD_BOX = 1, RT_CHILD_TRI = 2 } rt_child_type; /* Common.cuh:35-41 */

/* Arguments.h:8-26 */
typedef enum rt_build_type { RT_BUILD_SAH = 0, RT_BUILD_BOTTOM_UP = 1, RT_BUILD_HYBRID = 2, RT_BUILD_NONE = 3 } rt_build_type;
typedef enum rt_render_type {
    RT_RENDER_DEPTH = 0, RT_RENDER_BOXTESTS = 1, RT_RENDER_TRIANGLE_TESTS = 2, RT_RENDER_MATERIAL_ID = 3,
    RT_RENDER_LODS = 4, RT_RENDER_DIFFUSE = 5, RT_RENDER_TEXTURE = 6, RT_RENDER_TEXTURE_LIT = 7,
    RT_RENDER_TEXTURE_LIT_SHADOWS = 8, RT_RENDER_COUNT = 9
} rt_render_type;

/* Arguments.h:28-33 Arguments */
typedef struct rt_arguments { int32_t build_type; int32_t enable_splits; int32_t enable_pairs; int32_t render_type; } rt_arguments;

/* BuildWrapper.cuh:6-12 BuildInput */
typedef struct rt_build_input {
    const rt_triangle* triangles_in;   /* 36 n bytes */
    rt_triangle_pair*  triangles_out;  /* >= 64 n bytes (reference allocates 64*2n, main.cu:232-233) */
    uint32_t           num_triangles;
    rt_node*           nodes_out;      /* >= rt_nodes_bytes(n) */
    void*              scratch;        /* >= rt_bu_memory_requirements(n), 256-byte aligned */
} rt_build_input;

/* Common.cuh:335-340 DeviceAccelerationStructure */
typedef struct rt_accel { const rt_triangle_pair* triangles; const rt_node* nodes; uint32_t root; uint32_t count; } rt_accel;

/* Common.cuh:342-351 DeviceScene.  textures: device array indexed by rt_material.texture / .bump / .disp; only the
 * textured render types (kLODs, kTexture, kTextureLit, kTextureLitShadows) read it. */
typedef struct rt_scene {
    const rt_attributes* attributes;
    const rt_material*   materials;
    const rt_texture*    textures;
    const rt_camera*     camera;        /* device pointer, as in the reference (main.cu:151,161) */
    float                light[3];
    uint32_t             num_attributes, num_materials, num_textures;   /* num_attributes = number of primitives (main.cu:166); also
                                                                          the scene-size hint of rt_trace (0 = unknown): from 8M
                                                                          primitives on the tracer takes its pair-prefetch form */
} rt_scene;

enum {
    RT_OK = 0,
    RT_ERR_INVALID_ARGUMENT = -1,
    RT_ERR_UNSUPPORTED = -2,       /* (no option of the built paths returns it any more; kept for ABI stability) */
    RT_ERR_TOO_LARGE = -3,         /* n exceeds the 29-bit child index of Node (Common.cuh:152-159) */
    RT_ERR_BUILD_INCOMPLETE = -4,  /* (kept for ABI stability: rt_run_sah_build is asynchronous since round 4 and reports through the status word) */
    RT_ERR_HIP_BASE = -1000        /* -(hipError_t) + RT_ERR_HIP_BASE */
};

/* replaces BuMemoryRequirements (BuildWrapper.cu:132-136).  Scratch also holds what the reference
 * cudaMalloc's inside RadixSort (RadixSort.cu:187-190). */
size_t rt_bu_memory_requirements(uint32_t num_triangles);

/* bytes the caller must provide for nodes_out; same rule as main.cu:235-237: 32 * 4 * (n + 512) */
size_t rt_nodes_bytes(uint32_t num_triangles);

/* replaces RunBottomUpBuild (BuildWrapper.cu:253-362).  hybrid != 0 additionally builds the SAH top
 * tree above the 8-level-deep LBVH sub-roots (ExtractDepth + SharedTaskBuild, BuildWrapper.cu:350-361);
 * trace root is then (2n+1, 2) instead of (0, 2) (main.cu:222-223).  The top tree is built deterministically
 * (the reference's numbering depends on atomic arrival order); it occupies slots [2L, 2L + 2*256 + 2).
 * args->enable_pairs (Pairing.cuh, GenerateMortonCodesPairs): triangles 2k, 2k+1 sharing an edge become one quad
 * leaf; leaf slots are assigned by a prefix sum in input order (the reference uses atomicAdd arrival order), the
 * leaf count L lands in scratch (rt_bu_scratch_layout.num_leaves); hybrid + pairs roots at (2L+1, 2). */
int rt_run_bottom_up_build(const rt_build_input* input, const rt_arguments* args, int hybrid, void* stream);

/* replaces SahMemoryRequirements (BuildWrapper.cu:126-130); about 90 bytes per triangle */
size_t rt_sah_memory_requirements(uint32_t num_triangles);

/* replaces RunSahBuild (BuildWrapper.cu:140-251), the reference's default --type: leaves bucketed into a 4x4x4 grid
 * by centroid (Setup, GridBlockCounts/Scan/Distribute, Multiblock.cu:139-207,427-546), one binned-SAH sub-tree per
 * cell and a SAH top tree over the cells (SharedTaskBuild, SharedTaskBuilder.cu:93-607,909-967).  Trace root =
 * (slot 0, count 1) (main.cu:222-223).  input->scratch holds rt_sah_memory_requirements(n) bytes, nodes_out
 * rt_nodes_bytes(n); the tree uses slots [0, 128 + 2L), L = number of items.  args->enable_pairs as in
 * rt_run_bottom_up_build.  args->enable_splits (SetupSplits / SetupPairSplits, Multiblock.cu:209-425): a leaf whose
 * box spans several cells of the 4x4x4 grid over the scene box is referenced once per cell, box clipped to the cell,
 * while the running total of extra references -- taken in input order; an atomic counter in the reference -- stays
 * below n/5 (so L < n + n/5; n <= 2^25 with splits).
 * Same tree as the reference up to numbering, which is deterministic here: leaf slots in input order, node slots
 * = f(split position) (see gpu-raytracing_amd/csrc/sah_build.hip).  The depth of the trees is data dependent and the
 * reference loops on the host (cudaMemcpy of num_leaves, BuildWrapper.cu:229); here the number of launches is fixed by n
 * and the data-dependent tail runs as a device-side loop: the call neither copies nor synchronises (hipGraph-capturable,
 * like rt_run_bottom_up_build).  Error flags: the status word of the scratch (rt_sah_scratch_layout.status), 0 = complete
 * tree, to be read by the caller after the stream has run. */
int rt_run_sah_build(const rt_build_input* input, const rt_arguments* args, void* stream);

typedef struct rt_sah_scratch_layout {
    size_t p_aabb, c_aabb;  /* int32[6] each: ordered-int primitive / centroid bounds of the scene (BuildWrapper.cu:170-176) */
    size_t status;          /* uint32[8]: [0] error flags of the last build (0 = ok), [1] number of items L (leaves, or
                             * leaf references with splits), [2] number of TrianglePair records written */
    size_t num_leaves;      /* = status + 4 (items) */
    size_t cell_counts;     /* uint32[64]: leaves per grid cell (block_counts, Multiblock.cu:427) */
    size_t total;
} rt_sah_scratch_layout;
int rt_sah_scratch_layout_get(uint32_t num_triangles, rt_sah_scratch_layout* out);

/* Where the build's intermediates live inside `scratch` (for parity tests and callers that want the
 * sorted Morton codes).  Offsets in bytes. */
typedef struct rt_bu_scratch_layout {
    size_t p_aabb;          /* int32[6] ordered-int scene box (BuildWrapper.cu:288-289, Multiblock.cu:104) */
    size_t status;          /* uint32[8]: [0] = error flags of the last build, 0 = ok (bit 0: a workgroup found more
                               than 128 unfinished sub-trees -- impossible for a tree of depth <= 62) */
    size_t num_leaves;      /* uint32: number of leaves L of the last build (= n unless args.enable_pairs merged triangles) */
    size_t morton;          /* uint32[n] sorted Morton codes after the build */
    size_t sorted_indices;  /* uint32[n] original triangle index per sorted position */
    size_t total;
} rt_bu_scratch_layout;
int rt_bu_scratch_layout_get(uint32_t num_triangles, rt_bu_scratch_layout* out);

/* replaces the CalculateSceneAabb launch (Multiblock.cu:104-114, BuildWrapper.cu:305-308).
 * aabb_ordered: int32[6]; this call first resets it to the ordered-int empty box. */
int rt_calculate_scene_aabb(const rt_triangle* triangles, uint32_t n, int32_t* aabb_ordered, void* stream);

/* replaces the GenerateMortonCodes launch (BottomUpBuilder.cu:98-115, BuildWrapper.cu:324-328) */
int rt_generate_morton_codes(uint32_t* codes, uint32_t* values, const rt_triangle* triangles,
                             const int32_t* aabb_ordered, uint32_t n, void* stream);

/* replaces RadixSort (RadixSort.cuh:6-7, RadixSort.cu:171-225): stable ascending sort of (key,value)
 * pairs, result in keys/values, tmp_* are n-entry temporaries.  sort_scratch: >= rt_radix_sort_scratch_bytes(n).
 * count <= 0x3FFFFFFF (the kernels address the arrays with 32-bit byte offsets); larger -> RT_ERR_TOO_LARGE, nothing runs. */
size_t rt_radix_sort_scratch_bytes(uint32_t count);
int rt_radix_sort_u32_pairs(uint32_t* keys, uint32_t* values, uint32_t* tmp_keys, uint32_t* tmp_values,
                            uint32_t count, void* sort_scratch, void* stream);
/* The same sort for keys whose bits [key_bits, 32) are all zero (RunBottomUpBuild's Morton codes have 30,
 * BottomUpBuilder.cu:23-32; the reference's RadixSort always runs its four 8-bit passes, RadixSort.cu:192-219).
 * Up to 30 bits and a moderate count it runs three 10-bit passes, which read their input from the temporaries:
 * input_in_tmp = 1 says the unsorted pairs are in tmp_keys / tmp_values (0: in keys / values); when that is not where
 * the chosen pass count reads from, the pairs are copied across first (rt_radix_sort_input_in_tmp tells a caller that
 * wants to avoid the copy where to put them).  The sorted result is in keys / values either way. */
int rt_radix_sort_u32_pairs_bits(uint32_t* keys, uint32_t* values, uint32_t* tmp_keys, uint32_t* tmp_values,
                                 uint32_t count, uint32_t key_bits, int input_in_tmp, void* sort_scratch, void* stream);
int rt_radix_sort_input_in_tmp(uint32_t count, uint32_t key_bits);

/* replaces Trace()/TraceRays (main.cu:125-192, Tracer.cu:471-595) for rows [y0, y1) of a w x h frame.
 * rgba8: full-frame linear RGBA8 buffer, pitch 4*w, row 0 first (= the surface contents, SURVEY A).
 * counters: optional device uint64[4], [0] += sum of box tests, [1] += sum of triangle tests (the reference's
 * num_tests is [0], Tracer.cu:503); [2] / [3] += wave-level box-phase / leaf-phase steps (profiling aid).  spp = 1 is the reference; spp in {4,16} is the
 * SURVEY 8(d) config-5 extension (2x2 / 4x4 stratified sub-pixel offsets, averaged before the u8 truncation). */
int rt_trace(const rt_accel* as, const rt_scene* scene, uint64_t* counters, int render_type, uint8_t* rgba8,
             uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t spp, void* stream);

/* Multi-GPU partition into INTERLEAVED STRIPS (no reference counterpart: the reference traces on one GPU,
 * main.cu:169).  A strip = strip_rows rows (a multiple of 8); this call renders strips first_strip, first_strip +
 * strip_stride, ... of the w x h frame in ONE launch and stores them COMPACTLY: its j-th strip occupies rows
 * [j*strip_rows, (j+1)*strip_rows) of rgba8_compact (pitch 4*w), which must hold
 * ceil(ceil(h / strip_rows) / strip_stride) strips.  Rank r of P calls it with (first_strip, strip_stride) = (r, P);
 * rank 0 gathers the compact buffers and de-interleaves them (gpu-raytracing_amd/sharding.py). */
int rt_trace_strips(const rt_accel* as, const rt_scene* scene, uint64_t* counters, int render_type,
                    uint8_t* rgba8_compact, uint32_t w, uint32_t h, uint32_t strip_rows, uint32_t first_strip,
                    uint32_t strip_stride, uint32_t spp, void* stream);

/* static string for a return code */
const char* rt_error_string(int code);

/* library / kernel configuration, for logs: e.g. "rt_amd gfx950 sort=8bit x4 tile=4096 ..." */
const char* rt_version_string(void);

#ifdef __cplusplus
}
#endif
#endif
