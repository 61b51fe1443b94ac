// rt_abi.hip -- the extern "C" entry points declared in include/rt_abi.h.
// Host orchestration only: scratch carving and kernel order (the role of BuildWrapper.cu:68-136,
// 253-362 and main.cu:125-192 in the reference).  No allocation, no host<->device copies of data: every call is a
// sequence of asynchronous launches on the caller's stream (rt_run_sah_build too: its number of launches is fixed by
// n, the data-dependent tail is a device-side loop, sah_build.hip).
#include "rt_device.hpp"
#include "rt_launch.hpp"

namespace rt {

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

BuLayout bu_layout(uint32_t n)
{
    BuLayout L;
    const size_t nn = n ? n : 1;
    size_t off = 0;
    L.p_aabb = off;         off += 32;
    L.status = off;         off += 32;
    off = align_up(off, 256);
    L.morton = off;         off = align_up(off + nn * 4, 256);
    L.sorted_indices = off; off = align_up(off + nn * 4, 256);
    L.tmp_keys = off;       off = align_up(off + nn * 4, 256);
    L.tmp_vals = off;       off = align_up(off + nn * 4, 256);
    L.sort = off;           off = align_up(off + sort_scratch_layout(n).total, 256);
    L.levels = off;         off = align_up(off + lbvh_level_plan(n).total, 256);
    L.hybrid = off;         off = align_up(off + 64 * 1024, 256);
    L.pair_flags = off;     off = align_up(off + (nn + 1) / 2, 256);
    L.pair_sums = off;      off = align_up(off + ((nn + 1) / 2 / 256 + 2) * 4, 256);
    L.aabb_parts = off;     off = align_up(off + kAabbParts * 6 * 4, 256);
    L.total = off;
    return L;
}

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? RT_OK : RT_ERR_HIP_BASE - (int)e; }

// one launch for the build's tiny initialisations: status words = 0, the ordered-int "empty" scene box
// (BuildWrapper.cu:288-303 does these with 6 memset / memcpy calls) and the LBVH hierarchy's arrival counters = 0
__global__ __launch_bounds__(256) void build_init_kernel(uint32_t* status, int* aabb, int* aabb_parts, uint32_t n,
                                                         uint32_t* arrive, uint32_t arrive_words)
{
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < arrive_words; i += gridDim.x * 256) arrive[i] = 0;
    if (blockIdx.x != 0) return;
    if (threadIdx.x < 8) status[threadIdx.x] = threadIdx.x == 1 ? n : 0;   // [1] = number of leaves (pairs: overwritten)
    if (threadIdx.x < 6) aabb[threadIdx.x] = threadIdx.x < 3 ? 0x7f7fffff : (int)0x80800000;
    for (uint32_t i = threadIdx.x; i < 6; i += blockDim.x) aabb_parts[i] = (i % 6) < 3 ? 0x7f7fffff : (int)0x80800000;
}

}  // namespace rt

using namespace rt;

extern "C" {

size_t rt_bu_memory_requirements(uint32_t num_triangles) { return bu_layout(num_triangles).total; }

size_t rt_nodes_bytes(uint32_t num_triangles)
{
    // main.cu:235-237: sizeof(Node) * (n + max(512, NUM_BLOCKS)) * 2 * 2
    return sizeof(rt_node) * ((size_t)num_triangles + 512) * 4;
}

int rt_bu_scratch_layout_get(uint32_t num_triangles, rt_bu_scratch_layout* out)
{
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    const BuLayout L = bu_layout(num_triangles);
    out->p_aabb = L.p_aabb;
    out->status = L.status;
    out->num_leaves = L.status + 4;
    out->morton = L.morton;
    out->sorted_indices = L.sorted_indices;
    out->total = L.total;
    return RT_OK;
}

int rt_calculate_scene_aabb(const rt_triangle* triangles, uint32_t n, int32_t* aabb_ordered, void* stream)
{
    if (!aabb_ordered || (n && !triangles)) return RT_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = launch_reset_aabb(aabb_ordered, st);
    if (e == hipSuccess) e = launch_scene_aabb(triangles, n, aabb_ordered, st);
    return hip_rc(e);
}

int rt_generate_morton_codes(uint32_t* codes, uint32_t* values, const rt_triangle* triangles,
                             const int32_t* aabb_ordered, uint32_t n, void* stream)
{
    if (n && (!codes || !values || !triangles || !aabb_ordered)) return RT_ERR_INVALID_ARGUMENT;
    return hip_rc(launch_morton(codes, values, triangles, aabb_ordered, n, static_cast<hipStream_t>(stream)));
}

size_t rt_radix_sort_scratch_bytes(uint32_t count) { return sort_scratch_layout(count).total; }

int rt_radix_sort_u32_pairs(uint32_t* keys, uint32_t* values, uint32_t* tmp_keys, uint32_t* tmp_values,
                            uint32_t count, void* sort_scratch, void* stream)
{
    if (count && (!keys || !values || !tmp_keys || !tmp_values || !sort_scratch)) return RT_ERR_INVALID_ARGUMENT;
    if (count > kSortMaxCount) return RT_ERR_TOO_LARGE;   // 32-bit byte offsets of the buffer descriptors (rt_abi.h)
    return hip_rc(launch_radix_sort(keys, values, tmp_keys, tmp_values, count, sort_scratch,
                                    static_cast<hipStream_t>(stream)));
}

int rt_radix_sort_u32_pairs_bits(uint32_t* keys, uint32_t* values, uint32_t* tmp_keys, uint32_t* tmp_values,
                                 uint32_t count, uint32_t key_bits, int input_in_tmp, void* sort_scratch, void* stream)
{
    if (count && (!keys || !values || !tmp_keys || !tmp_values || !sort_scratch)) return RT_ERR_INVALID_ARGUMENT;
    if (key_bits == 0 || key_bits > 32) return RT_ERR_INVALID_ARGUMENT;
    if (count > kSortMaxCount) return RT_ERR_TOO_LARGE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // an odd number of passes (3 x 10 bits) reads its input from the temporaries, an even one (4 x 8) from keys / values:
    // the result is in keys / values either way.  The input is moved only when it sits on the other side.
    const bool wants_tmp = key_bits <= 30 && sort_three_passes(sort_num_tiles(count));
    if (count && wants_tmp != (input_in_tmp != 0)) {
        uint32_t *sk = input_in_tmp ? tmp_keys : keys, *sv = input_in_tmp ? tmp_values : values;
        uint32_t *dk = input_in_tmp ? keys : tmp_keys, *dv = input_in_tmp ? values : tmp_values;
        hipError_t e = hipMemcpyAsync(dk, sk, (size_t)count * 4, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dv, sv, (size_t)count * 4, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) return hip_rc(e);
    }
    return hip_rc(launch_radix_sort(keys, values, tmp_keys, tmp_values, count, sort_scratch, st, nullptr, key_bits));
}

int rt_radix_sort_input_in_tmp(uint32_t count, uint32_t key_bits)
{
    return key_bits <= 30 && sort_three_passes(sort_num_tiles(count)) ? 1 : 0;
}

int rt_run_bottom_up_build(const rt_build_input* input, const rt_arguments* args, int hybrid, void* stream)
{
    if (!input || !input->nodes_out || !input->scratch) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = input->num_triangles;
    if (n && (!input->triangles_in || !input->triangles_out)) return RT_ERR_INVALID_ARGUMENT;
    if (n > (1u << 28)) return RT_ERR_TOO_LARGE;  // 2(n-1) slots must fit the 29-bit child field
    // args->enable_splits is ignored here, as in the reference: only RunSahBuild has a split pre-pass (BuildWrapper.cu:188-210)
    const bool pairs = args && args->enable_pairs;
    if ((reinterpret_cast<uintptr_t>(input->scratch) & 255u) || (reinterpret_cast<uintptr_t>(input->triangles_in) & 15u) ||
        (reinterpret_cast<uintptr_t>(input->triangles_out) & 63u) || (reinterpret_cast<uintptr_t>(input->nodes_out) & 63u))
        return RT_ERR_INVALID_ARGUMENT;

    hipStream_t st = static_cast<hipStream_t>(stream);
    const BuLayout L = bu_layout(n);
    char* s = static_cast<char*>(input->scratch);
    int* p_aabb = reinterpret_cast<int*>(s + L.p_aabb);
    uint32_t* status = reinterpret_cast<uint32_t*>(s + L.status);
    uint32_t* morton = reinterpret_cast<uint32_t*>(s + L.morton);
    uint32_t* sorted = reinterpret_cast<uint32_t*>(s + L.sorted_indices);
    uint32_t* tmpk = reinterpret_cast<uint32_t*>(s + L.tmp_keys);
    uint32_t* tmpv = reinterpret_cast<uint32_t*>(s + L.tmp_vals);

    int* aabb_parts = reinterpret_cast<int*>(s + L.aabb_parts);
    const LevelPlan lp = lbvh_level_plan(n);
    const uint32_t arrive_words = (uint32_t)(lp.arrive_bytes / 4);
    uint32_t* arrive = reinterpret_cast<uint32_t*>(s + L.levels + lp.arrive_off);
    uint32_t nparts = 1;
    hipError_t e;
    if (n) {
        // the first launch: scene bounds (partial boxes, folded by the Morton kernels) + the build's initialisations
        e = launch_scene_aabb_build(input->triangles_in, n, aabb_parts, &nparts, status, arrive, arrive_words, st);
    } else {
        build_init_kernel<<<1, 256, 0, st>>>(status, p_aabb, aabb_parts, n, arrive, arrive_words);
        e = hipGetLastError();
    }
    // with --pairs the number of leaves L <= n is only known on the device (status[1]); the reference copies it
    // back to the host (BuildWrapper.cu:317-321, a sync) -- here the downstream kernels read it from memory and the
    // grids are sized for n
    uint32_t* num_leaves = status + 1;
    const uint32_t* n_dev = pairs ? num_leaves : nullptr;
    // The sort of the 30-bit Morton keys: 3 passes x 10 bits while the tables stay small (kSort3PassMaxTiles), else 4 x 8.
    // An odd number of passes ends in the other buffer pair, so the Morton kernels then write into the temporaries.
    // Without --pairs the Morton kernel also produces the first pass's tile histograms (one launch fewer).
    const bool three = sort_three_passes(sort_num_tiles(n));
    uint32_t* code_dst = three ? tmpk : morton;
    uint32_t* value_dst = three ? tmpv : sorted;
    if (e == hipSuccess) {
        if (pairs)
            e = launch_morton_pairs(code_dst, value_dst, input->triangles_in, aabb_parts, n, reinterpret_cast<uint8_t*>(s + L.pair_flags),
                                    reinterpret_cast<uint32_t*>(s + L.pair_sums), num_leaves, st, nparts, p_aabb);
        else
            // (the values of this path are the identity, BottomUpBuilder.cu:113: not written, the sort's first pass regenerates them)
            e = launch_morton_hist(code_dst, nullptr, input->triangles_in, aabb_parts, n, st, nparts, p_aabb,
                                   sort_hist_table(s + L.sort, n), three ? 10 : 8);
    }
    if (e == hipSuccess)
        e = launch_radix_sort(morton, sorted, tmpk, tmpv, n, s + L.sort, st, n_dev, three ? 30 : 32, !pairs, !pairs);
    if (e == hipSuccess)
        e = launch_lbvh_levels(input->triangles_in, morton, sorted, n, input->triangles_out, input->nodes_out,
                               s + L.levels, status, st, n_dev);
    if (e == hipSuccess && hybrid) e = launch_hybrid_top(input->nodes_out, p_aabb, n, st, n_dev);   // BuildWrapper.cu:350-361
    return hip_rc(e);
}

size_t rt_sah_memory_requirements(uint32_t num_triangles) { return sah_layout(num_triangles).total; }

int rt_sah_scratch_layout_get(uint32_t num_triangles, rt_sah_scratch_layout* out)
{
    if (!out) return RT_ERR_INVALID_ARGUMENT;
    const SahLayout L = sah_layout(num_triangles);
    out->p_aabb = L.header;          // SahHeader starts with gp[6], gc[6]
    out->c_aabb = L.header + 24;
    out->status = L.status;
    out->num_leaves = L.status + 4;
    out->cell_counts = L.cell_counts;
    out->total = L.total;
    return RT_OK;
}

int rt_run_sah_build(const rt_build_input* input, const rt_arguments* args, void* stream)
{
    if (!input || !input->nodes_out || !input->scratch) return RT_ERR_INVALID_ARGUMENT;
    const uint32_t n = input->num_triangles;
    if (n && (!input->triangles_in || !input->triangles_out)) return RT_ERR_INVALID_ARGUMENT;
    const bool splits = args && args->enable_splits;
    if (n > (splits ? (1u << 25) : (1u << 28) - 64)) return RT_ERR_TOO_LARGE;   // splits: the 32-bit budget prefix sums 63 per leaf
    if ((reinterpret_cast<uintptr_t>(input->scratch) & 255u) || (reinterpret_cast<uintptr_t>(input->triangles_in) & 15u) ||
        (reinterpret_cast<uintptr_t>(input->triangles_out) & 63u) || (reinterpret_cast<uintptr_t>(input->nodes_out) & 63u))
        return RT_ERR_INVALID_ARGUMENT;
    // asynchronous, like the bottom-up build: the error flags stay in the scratch status word (rt_sah_scratch_layout.status)
    return hip_rc(launch_sah_build(input->triangles_in, n, args && args->enable_pairs, splits, input->triangles_out,
                                   input->nodes_out, input->scratch, static_cast<hipStream_t>(stream), nullptr, nullptr));
}

static int trace_common(const rt_accel* as, const rt_scene* scene, uint64_t* counters, int render_type, uint8_t* rgba8,
                        uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t spp, uint32_t strip_rows,
                        uint32_t strip_first, uint32_t strip_stride, void* stream)
{
    if (!as || !scene || !rgba8 || !as->nodes || !scene->camera) return RT_ERR_INVALID_ARGUMENT;
    if (w == 0 || h == 0 || y0 > y1 || y1 > h || as->count > 7) return RT_ERR_INVALID_ARGUMENT;
    if (spp == 0) spp = 1;
    if (spp != 1 && spp != 4 && spp != 16) return RT_ERR_INVALID_ARGUMENT;
    switch (render_type) {
    case RT_RENDER_DEPTH: case RT_RENDER_BOXTESTS: case RT_RENDER_TRIANGLE_TESTS: break;
#ifdef RT_TRACE_TUNING
    case 100: break;   // tuning build only: raw box-test count per pixel
#endif
    case RT_RENDER_MATERIAL_ID: case RT_RENDER_DIFFUSE:
        if (!scene->attributes || !scene->materials || scene->num_materials == 0) return RT_ERR_INVALID_ARGUMENT;
        break;
    case RT_RENDER_LODS: case RT_RENDER_TEXTURE: case RT_RENDER_TEXTURE_LIT: case RT_RENDER_TEXTURE_LIT_SHADOWS:
        if (!scene->attributes || !scene->materials || scene->num_materials == 0) return RT_ERR_INVALID_ARGUMENT;
        if (scene->num_textures && !scene->textures) return RT_ERR_INVALID_ARGUMENT;
        break;
    default: return RT_ERR_INVALID_ARGUMENT;
    }
    TraceLaunch t;
    t.as = *as;
    t.scene = *scene;
    t.counters = counters;
    t.render_type = render_type;
    t.rgba8 = rgba8;
    t.w = w; t.h = h; t.y0 = y0; t.y1 = y1; t.spp = spp;
    t.strip_rows = strip_rows; t.strip_first = strip_first; t.strip_stride = strip_stride;
    return hip_rc(launch_trace(t, static_cast<hipStream_t>(stream)));
}

int rt_trace(const rt_accel* as, const rt_scene* scene, uint64_t* counters, int render_type, uint8_t* rgba8,
             uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t spp, void* stream)
{
    return trace_common(as, scene, counters, render_type, rgba8, w, h, y0, y1, spp, 0, 0, 1, stream);
}

int rt_trace_strips(const rt_accel* as, const rt_scene* scene, uint64_t* counters, int render_type, uint8_t* rgba8_compact,
                    uint32_t w, uint32_t h, uint32_t strip_rows, uint32_t first_strip, uint32_t strip_stride,
                    uint32_t spp, void* stream)
{
    if (strip_rows == 0 || (strip_rows & 7u) || strip_stride == 0) return RT_ERR_INVALID_ARGUMENT;
    return trace_common(as, scene, counters, render_type, rgba8_compact, w, h, 0, h, spp, strip_rows, first_strip,
                        strip_stride, stream);
}

const char* rt_error_string(int code)
{
    switch (code) {
    case RT_OK: return "ok";
    case RT_ERR_INVALID_ARGUMENT: return "invalid argument";
    case RT_ERR_UNSUPPORTED: return "unsupported option";
    case RT_ERR_TOO_LARGE: return "too many triangles for the 29-bit node index";
    case RT_ERR_BUILD_INCOMPLETE: return "SAH build incomplete (error flags in the scratch status word)";
    default: break;
    }
    if (code <= RT_ERR_HIP_BASE) return hipGetErrorString(static_cast<hipError_t>(RT_ERR_HIP_BASE - code));
    return "unknown error";
}

const char* rt_version_string(void)
{
    return "rt_amd gfx950 | sort: LSD 3x10bit Morton keys (4x8bit generic), tile 4096 | lbvh: LDS agglomerative, 512 leaves/wg (4 wg per CU), leaves and node pairs staged in LDS and streamed out + one chained launch for all upper levels (last-arriver tickets, fan 48 or 64; passes of <= 1023 open roots by range searches over sparse tables), hybrid SAH top | "
           "sah: 4x4x4 grid + level-synchronous binned SAH (fixed launch count, no host round trip), workgroup-per-task stragglers, wave-per-task below 64 items, pairs, splits | "
           "trace: wave64 8x8 tiles, two-phase schedule, LDS stack 16, XCD chunks of 8 workgroups, pair prefetch from 8M primitives, counters through 16-row slots";
}

}  // extern "C"
