/*
 * rt_oracle.h -- CPU restatement of the reference's LBVH build + primary-ray tracer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The product (gpu-raytracing_amd/) never
 * links, loads or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference/src) it restates.
 * Arithmetic is plain C, compiled -O2 -ffp-contract=off, IEEE fminf/fmaxf (minNum/maxNum),
 * correctly rounded '/' and sqrtf -- the same operation order as the reference's device code
 * without FMA contraction and with rsqrtf(x) := 1.0f/sqrtf(x).
 *
 * PINNING STATUS: the reference ships no tests, fixtures or golden vectors
 * and its CUDA kernels cannot be built in this image (no nvcc / CUDA device runtime), so the
 * arithmetic of Morton codes / sort / traversal is "parity unpinned" against reference-held
 * vectors.  What IS pinned: (1) the reference's own structural check -- VerifyHierarchy and
 * CountNodes from Utilities.cpp compiled unmodified into oracle/_ref -- is run on the oracle's
 * (and the GPU's) Node arrays, for the bottom-up, hybrid, pairs and SAH trees alike; Camera.cu
 * (host functions only) and Arguments.cpp are compiled the same way and pin the camera basis every
 * parity test uses and the host mirror's UpdateCamera / InitialiseCamera / controls / ParseCmd byte
 * for byte (tests/test_host_mirror.py); Pairing.cuh / Common.cuh, #included by the forwarding
 * driver oracle/ref_pairing_driver.cpp, pin the pair decisions, the quad-leaf construction and
 * the layouts of all PODs; (2) the
 * node-count identities of SURVEY.md Appendix A; (3) trees of the same triangles are checked
 * against each other (the SAH and hybrid trees render the same kDepth frame as the LBVH).
 * The hybrid, pairs and SAH builders of the reference number nodes / leaves by atomic arrival
 * order, so for those the restatement fixes one deterministic numbering (stated at each function)
 * and parity with the reference is structural.  The textured render types have no reference
 * frame either: pinned by properties (tests/test_oracle_textures.py).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } ora_f3;

/* Common.cuh:199-243  (36 B) */
typedef struct { ora_f3 v0, v1, v2; } ora_triangle;

/* Common.cuh:152-159  (32 B).  w12 = parent:29 (LSBs) | count:3 ; w28 = child:29 (LSBs) | type:3 */
typedef struct { ora_f3 min; uint32_t w12; ora_f3 max; uint32_t w28; } ora_node;

/* Common.cuh:161-197  (64 B) */
typedef struct {
    ora_f3 v0; uint32_t primitive_id_0;
    ora_f3 v1; uint32_t primitive_id_1;
    ora_f3 v2; uint16_t rot_x, rot_y;
    ora_f3 v3; float pad3;
} ora_triangle_pair;

/* Common.cuh:44-53  (64 B) */
typedef struct {
    ora_f3 position; float pitch;
    ora_f3 w;        float yaw;
    ora_f3 u;        float scale;
    ora_f3 v;        float max_depth;
} ora_camera;

/* Common.cuh:55-59  (72 B: float2 is 8-byte aligned in CUDA) */
typedef struct {
    ora_f3 normal[3]; uint32_t pad0;
    float uv[3][2];
    int32_t material_id; uint32_t pad1;
} ora_attributes;

/* POD mirror of the fields of Material the device reads (Common.cuh:93-129) */
typedef struct {
    ora_f3 ambient, diffuse, specular;
    float specular_exp;
    int32_t texture, bump, disp;
} ora_material;

/* POD mirror of the device-read fields of Texture (Common.cuh:61-91): RGBA8 mip chain, mips[l] has size_x[l] x size_y[l]
 * texels (row 0 first), levels 0..max_lod (Texture::GenerateLODs, FileIO.cpp:121-150) */
#define ORA_NUM_LODS 13
typedef struct {
    const uint32_t* mips[ORA_NUM_LODS];
    int32_t size_x[ORA_NUM_LODS], size_y[ORA_NUM_LODS];
    uint32_t max_lod, pad;
} ora_texture;

enum { ORA_TYPE_NONE = 0, ORA_TYPE_BOX = 1, ORA_TYPE_TRI = 2 };          /* Common.cuh:35-41 */
enum { ORA_DEPTH = 0, ORA_BOXTESTS = 1, ORA_TRITESTS = 2, ORA_MATERIALID = 3, ORA_LODS = 4, ORA_DIFFUSE = 5,
       ORA_TEXTURE = 6, ORA_TEXTURE_LIT = 7, ORA_TEXTURE_LIT_SHADOWS = 8 }; /* Arguments.h:15-26 */

void ora_set_threads(int n);
int  ora_get_threads(void);

/* DeviceUtils.cuh:3-13 */
int32_t ora_float_to_ordered_int(float f);
float   ora_ordered_int_to_float(int32_t i);

/* Multiblock.cu:104-114 + BuildWrapper.cu:288-289 (empty box).  out[6] is ordered-int encoded. */
void ora_scene_aabb(const ora_triangle* tris, uint32_t n, int32_t out[6]);

/* BottomUpBuilder.cu:12-32,98-115 */
void ora_morton_codes(const ora_triangle* tris, uint32_t n, const int32_t aabb_ordered[6],
                      uint32_t* codes, uint32_t* values);

/* RadixSort.cu:171-225 contract: stable ascending LSD sort of (key,value), 4 x 8-bit passes.
 * tmp_keys/tmp_vals: n entries each. */
void ora_radix_sort(uint32_t* keys, uint32_t* vals, uint32_t* tmp_keys, uint32_t* tmp_vals, uint32_t n);

/* BottomUpBuilder.cu:34-96,167-215 */
void ora_generate_hierarchy(ora_node* nodes, uint32_t* leaf_indices, const uint32_t* sorted_codes, uint32_t L);

/* BottomUpBuilder.cu:287-312 with Q1/Q2 resolved (ids defined, no OOB read) */
void ora_generate_triangles(const uint32_t* sorted_indices, const ora_triangle* tris,
                            ora_triangle_pair* out, uint32_t L);

/* BottomUpBuilder.cu:217-285 */
void ora_generate_aabbs(ora_node* nodes, const uint32_t* leaf_indices, const uint32_t* sorted_indices,
                        uint32_t* locks, const ora_triangle_pair* leaves, uint32_t L);

/* BuildWrapper.cu:253-348 (RunBottomUpBuild, pairs off, not hybrid).  nodes: 2*max(n,1) slots (zeroed
 * here first), leaves: n.  Optional outputs (may be NULL): codes_sorted[n], indices_sorted[n], aabb[6]. */
void ora_build(const ora_triangle* tris, uint32_t n, ora_node* nodes, ora_triangle_pair* leaves,
               uint32_t* codes_sorted, uint32_t* indices_sorted, int32_t* aabb_ordered);

/* --pairs: GenerateMortonCodesPairs (BottomUpBuilder.cu:117-164) + Pairing.cuh:9-77 + the pair
 * branches of GenerateTriangles / GenerateAABBs (:259-267, :301-303).  Triangles 2k, 2k+1 that share an edge (exact
 * vertex equality) and pass ShouldFormTrianglePair become one quad leaf.  The reference claims leaf slots with
 * atomicAdd (arrival order); here slot = exclusive prefix sum of the per-candidate leaf counts in input
 * order, which makes the pre-sort order -- and so the whole tree -- deterministic.
 * Same outputs as ora_build; returns the number of leaves L (<= n).  nodes: 2*max(n,1) slots, leaves: n entries. */
uint32_t ora_build_pairs(const ora_triangle* tris, uint32_t n, ora_node* nodes, ora_triangle_pair* leaves,
                         uint32_t* codes_sorted, uint32_t* indices_sorted, int32_t* aabb_ordered);

/* Hybrid top tree: ExtractDepth (BottomUpBuilder.cu:314-371) + SharedTaskBuild as launched at
 * BuildWrapper.cu:350-361, restated DETERMINISTICALLY (the reference emits sub-roots and allocates nodes in
 * atomic-arrival order): sub-roots in ascending thread id; tasks processed first-in
 * first-out; node slots allocated in that order; ids partitioned stably.  nodes must hold 2L + 2*256 + 8 slots and
 * already contain the LBVH of L leaves; scene box as ordered ints.  Writes the top tree at slots >= 2L (top root
 * descriptor at 2L, trace root = (2L+1, 2), main.cu:222-223).  Returns the number of sub-roots (<= 256).
 * subroots_out (may be NULL): the sub-root pair indices in emission order. */
uint32_t ora_build_hybrid_top(ora_node* nodes, uint32_t L, const int32_t aabb_ordered[6], uint32_t* subroots_out);

/* SAH path: RunSahBuild (BuildWrapper.cu:140-251) without spatial splits, restated
 * deterministically -- see the comment above ora_build_sah in rt_oracle.c for the numbering rules.  nodes must hold
 * 2*64 + 2*n + 2 slots (the reference allocates 4*(n+512), main.cu:235-237), leaves n entries.  Trace root =
 * (slot 0, count 1).  enable_splits: SetupSplits / SetupPairSplits (Multiblock.cu:209-425) -- a leaf whose box spans
 * several cells of the 4x4x4 grid over the SCENE box is referenced once per cell with its box clipped to the cell,
 * while the running sum of extra references (taken in input order here; an atomic counter in the reference) stays
 * below n/5.  nodes then needs 2*64 + 2*(n + n/5) + 2 slots.  cell_counts_out (may be NULL): items per grid cell
 * [64]; num_leaf_records_out (may be NULL): TrianglePair records written.  Returns the number of items L (leaves,
 * or leaf references with splits). */
uint32_t ora_build_sah(const ora_triangle* tris, uint32_t n, int enable_pairs, int enable_splits, ora_node* nodes,
                       ora_triangle_pair* leaves, uint32_t* cell_counts_out, uint32_t* num_leaf_records_out);

/* Utilities.cpp:8-44 : out[3] = {numNodes, numLeafNodes, numTreeNodes} */
void ora_count_nodes(const ora_node* nodes, uint32_t root, uint32_t count, int32_t out[3]);
/* Utilities.cpp:46-83 : returns the number of failing Box slots (reference prints one line each) */
int  ora_verify_hierarchy(const ora_node* nodes, uint32_t root, uint32_t count);

/* Tracer.cu:471-595 (TraceRays) over rows [y0,y1) of a w x h frame; rgba8 is the full frame (pitch 4w).
 * spp==1 is the reference; spp>1 is the SURVEY 8(d) config-5 extension.
 * counters (may be NULL, else FOUR words): [0] += sum box_tests, [1] += sum tri_tests, [2] = max stack depth seen
 * (entries, the nearest child's push included), [3] += pushes dropped on a full 64-entry stack.
 * returns 0, or -1 for an unsupported render type. */
int ora_trace(const ora_triangle_pair* leaves, const ora_node* nodes, uint32_t root, uint32_t count,
              const ora_attributes* attributes, const ora_material* materials, uint32_t num_materials,
              const ora_camera* camera, const float light[3], int render_type,
              uint8_t* rgba8, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t spp,
              uint64_t* counters);
/* the textured render types (kLODs, kTexture, kTextureLit, kTextureLitShadows; Tracer.cu:103-254,376-469,543-593) read
 * this table (indexed by Material.texture / .bump / .disp); set it before ora_trace, NULL to clear */
void ora_set_textures(const ora_texture* textures, uint32_t num_textures);
/* Texture::GenerateLODs (FileIO.cpp:121-150): fills mips[1..] (caller-allocated, sizes via ora_lod_sizes) */
uint32_t ora_lod_sizes(int32_t sx0, int32_t sy0, int32_t* size_x, int32_t* size_y);
void ora_generate_lod(const uint32_t* src, int32_t sx, int32_t sy, uint32_t* dst);

/* analysis aid: when set, ora_trace adds 1 to p[first slot] for every sibling pair a ray visits */
void ora_set_visit_counts(uint32_t* p);

/* the oracle's small box / centroid helpers (Common.cuh's __host__ __device__ functions), exported for the tests that pin
 * them to the reference's compiled code */
void ora_triangle_centre(const ora_triangle* t, float out[3]);
void ora_triangle_box(const ora_triangle* t, float out[6]);
void ora_box_centre(const float b[6], float out[3]);
void ora_box_combine(const float a[6], const float b[6], float out[6]);
int  ora_box_intersection(const float a[6], const float b[6], float out[6]);

#ifdef __cplusplus
}
#endif
#endif
