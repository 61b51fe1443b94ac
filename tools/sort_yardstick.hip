// sort_yardstick.hip -- times the library's radix sort beside rocprim::radix_sort_pairs on the SAME keys.
//
// A measuring stick, not a product path: nothing in gpu-raytracing_amd/ includes or links rocPRIM.  The keys are the
// builder's own Morton codes of the bench meshes (grid_mesh(G, 1): G = 708 -> 1,002,528 keys, G = 2237 -> 10,008,338),
// produced through the library's stage entry points, plus uniformly random 32-bit keys.  Both sorts are stable, so their
// outputs must be identical word for word; that is checked before anything is timed.
//
//   tools/bin/sort_yardstick <path to librt_amd*.so> [G ...]        (default G: 708 2237)
//
// Output: one line per (key set, sorter): median / min microseconds over 20 runs (events around the sort alone; the input
// is restored by an untimed device copy before each run) and the rate against the 80 B/key formula of SURVEY 8(d).
#include <dlfcn.h>
#include <string.h>
#include <hip/hip_runtime.h>
#include <cstring>   // rocPRIM's texture_cache_iterator.hpp calls memset unqualified
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef int (*fn_aabb)(const void*, uint32_t, int32_t*, void*);
typedef int (*fn_morton)(uint32_t*, uint32_t*, const void*, const int32_t*, uint32_t, void*);
typedef size_t (*fn_scratch)(uint32_t);
typedef int (*fn_sort)(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, void*, void*);
typedef int (*fn_sort_bits)(uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t, int, void*, void*);
typedef int (*fn_in_tmp)(uint32_t, uint32_t);

static uint32_t pcg_hash(uint32_t x)
{
    const uint32_t state = x * 747796405u + 2891336453u;
    const uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
    return (word >> 22) ^ word;
}

// gpu-raytracing_amd/scenes.py: grid_mesh(G, seed)
static std::vector<float> grid_mesh(uint32_t G, uint32_t seed)
{
    std::vector<float> t((size_t)G * G * 2 * 9);
    auto P = [&](uint32_t i, uint32_t j, float* o) {
        const uint32_t key = i + 0x9E3779B9u * j + seed;
        o[0] = (float)i;
        o[1] = 2.0f * ((float)(pcg_hash(key) >> 8) * (1.0f / 16777216.0f));
        o[2] = (float)j;
    };
    size_t w = 0;
    for (uint32_t j = 0; j < G; j++)
        for (uint32_t i = 0; i < G; i++) {
            float p00[3], p10[3], p01[3], p11[3];
            P(i, j, p00); P(i + 1, j, p10); P(i, j + 1, p01); P(i + 1, j + 1, p11);
            const float* a[6] = {p00, p10, p01, p10, p11, p01};
            for (int k = 0; k < 6; k++) { t[w++] = a[k][0]; t[w++] = a[k][1]; t[w++] = a[k][2]; }
        }
    return t;
}

struct Timing { double med, mn; };

template <class F, class R>
static Timing time_it(F&& run, R&& restore, hipStream_t st, int iters = 20)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<double> us;
    for (int i = 0; i < iters + 3; i++) {
        restore();
        CK(hipEventRecord(e0, st));
        run();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 3) us.push_back(ms * 1000.0);
    }
    std::sort(us.begin(), us.end());
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return Timing{us[us.size() / 2], us[0]};
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <librt_amd.so> [G ...]\n", argv[0]); return 1; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto aabb = (fn_aabb)dlsym(h, "rt_calculate_scene_aabb");
    auto morton = (fn_morton)dlsym(h, "rt_generate_morton_codes");
    auto scratch_bytes = (fn_scratch)dlsym(h, "rt_radix_sort_scratch_bytes");
    auto sort32 = (fn_sort)dlsym(h, "rt_radix_sort_u32_pairs");
    auto sortbits = (fn_sort_bits)dlsym(h, "rt_radix_sort_u32_pairs_bits");
    auto in_tmp = (fn_in_tmp)dlsym(h, "rt_radix_sort_input_in_tmp");
    if (!aabb || !morton || !scratch_bytes || !sort32) { fprintf(stderr, "missing symbols\n"); return 1; }
    std::vector<uint32_t> Gs;
    for (int i = 2; i < argc; i++) Gs.push_back((uint32_t)atoi(argv[i]));
    if (Gs.empty()) { Gs.push_back(708); Gs.push_back(2237); }
    hipStream_t st;
    CK(hipStreamCreate(&st));

    // YARD_QUICK=1 (parameter sweeps): Morton keys only, the library's sort only (after the equality check)
    const bool quick = getenv("YARD_QUICK") && atoi(getenv("YARD_QUICK"));
    for (int set = 0; set < (int)Gs.size() * 2; set++) {
        const uint32_t G = Gs[set / 2];
        const bool random_keys = set & 1;
        if (quick && random_keys) continue;
        const uint32_t n = G * G * 2;
        uint32_t *k0, *v0, *k, *v, *tk, *tv, *rk, *rv;   // pristine input; ours (+ temporaries); rocPRIM's output
        for (uint32_t** p : {&k0, &v0, &k, &v, &tk, &tv, &rk, &rv}) CK(hipMalloc(p, (size_t)n * 4));
        const uint32_t bits = random_keys ? 32 : 30;
        if (!random_keys) {
            std::vector<float> tris = grid_mesh(G, 1);
            float* dt; int32_t* dbox;
            CK(hipMalloc(&dt, tris.size() * 4)); CK(hipMalloc(&dbox, 32));
            CK(hipMemcpy(dt, tris.data(), tris.size() * 4, hipMemcpyHostToDevice));
            if (aabb(dt, n, dbox, st) || morton(k0, v0, dt, dbox, n, st)) { fprintf(stderr, "stage entry point failed\n"); return 2; }
            CK(hipStreamSynchronize(st));
            CK(hipFree(dt)); CK(hipFree(dbox));
        } else {
            std::vector<uint32_t> hk(n), hv(n);
            for (uint32_t i = 0; i < n; i++) { hk[i] = pcg_hash(i * 2654435761u + 12345u); hv[i] = i; }
            CK(hipMemcpy(k0, hk.data(), (size_t)n * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(v0, hv.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        }
        void* scr;
        CK(hipMalloc(&scr, scratch_bytes(n)));
        size_t rp_bytes = 0;
        CK(rocprim::radix_sort_pairs(nullptr, rp_bytes, k0, rk, v0, rv, (size_t)n, 0u, bits, st));
        void* rp_tmp;
        CK(hipMalloc(&rp_tmp, rp_bytes));

        // where the library wants its input for this (n, bits)
        const int want_tmp = (sortbits && in_tmp) ? in_tmp(n, bits) : 0;
        auto restore_ours = [&] {
            CK(hipMemcpyAsync(want_tmp ? tk : k, k0, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(want_tmp ? tv : v, v0, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        };
        auto run_ours = [&] {
            const int rc = sortbits ? sortbits(k, v, tk, tv, n, bits, want_tmp, scr, st) : sort32(k, v, tk, tv, n, scr, st);
            if (rc) { fprintf(stderr, "sort failed: %d\n", rc); exit(2); }
        };
        auto restore_32 = [&] {
            CK(hipMemcpyAsync(k, k0, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
            CK(hipMemcpyAsync(v, v0, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
        };
        auto run_32 = [&] { if (sort32(k, v, tk, tv, n, scr, st)) { fprintf(stderr, "sort failed\n"); exit(2); } };
        auto run_rp = [&] { CK(rocprim::radix_sort_pairs(rp_tmp, rp_bytes, k0, rk, v0, rv, (size_t)n, 0u, bits, st)); };

        // identical outputs first
        restore_ours(); run_ours(); run_rp();
        CK(hipStreamSynchronize(st));
        std::vector<uint32_t> a(n), b(n), c(n), d(n);
        CK(hipMemcpy(a.data(), k, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), rk, (size_t)n * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(c.data(), v, (size_t)n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(d.data(), rv, (size_t)n * 4, hipMemcpyDeviceToHost));
        const bool same = a == b && c == d;
        bool sorted = true;
        for (uint32_t i = 1; i < n && sorted; i++) sorted = a[i - 1] <= a[i];
        printf("# %s n=%u key_bits=%u : outputs %s, %s\n", random_keys ? "random32" : "morton(grid_mesh)", n, bits,
               same ? "IDENTICAL to rocPRIM's" : "DIFFER from rocPRIM's", sorted ? "ascending" : "NOT SORTED");
        if ((!same || !sorted) && !(getenv("YARD_NOCHECK") && atoi(getenv("YARD_NOCHECK")))) return 3;

        const Timing to = time_it(run_ours, restore_ours, st);
        const Timing t32 = quick ? Timing{0, 0} : time_it(run_32, restore_32, st);
        const Timing tr = quick ? Timing{1, 1} : time_it(run_rp, [] {}, st);
        const double tbs = 80.0 * n / (to.med * 1e-6) / 1e12;   // 80 B/key
        printf("%-18s n=%-9u rt_radix_sort_u32_pairs_bits(%u)  median %8.1f us  min %8.1f us  %5.2f TB/s of the 80 B/key formula = %4.1f %% of 8 TB/s\n",
               random_keys ? "random32" : "morton", n, bits, to.med, to.mn, tbs, tbs / 8.0 * 100.0);
        printf("%-18s n=%-9u rt_radix_sort_u32_pairs (32 bits)   median %8.1f us  min %8.1f us\n", random_keys ? "random32" : "morton", n, t32.med, t32.mn);
        printf("%-18s n=%-9u rocprim::radix_sort_pairs [0,%u)     median %8.1f us  min %8.1f us   ours/rocPRIM = %.2f\n",
               random_keys ? "random32" : "morton", n, bits, tr.med, tr.mn, to.med / tr.med);
        fflush(stdout);
        for (uint32_t* p : {k0, v0, k, v, tk, tv, rk, rv}) CK(hipFree(p));
        CK(hipFree(scr)); CK(hipFree(rp_tmp));
    }
    return 0;
}
