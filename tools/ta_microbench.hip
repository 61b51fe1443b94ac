// ta_microbench.hip -- how many texture-path cycles does a wave-wide 16-byte-per-lane load cost as a function of the
// number of distinct 64-byte segments it touches, and does letting the four lanes of a quad fetch ONE 64-byte pair
// per instruction (then exchanging inside the quad) beat four loads of every lane's own pair?
//   hipcc --offload-arch=gfx950 -O3 tools/ta_microbench.hip -o gpurun_out/ta_microbench && gpurun_out/ta_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int CTRL>
__device__ __forceinline__ uint32_t qperm(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }

// MODE 0: each lane reads the four 16-byte pieces of its own pair (what trace_kernel does)
// MODE 1: lane j of a quad reads piece j of member k's pair, k = 0..3 (one segment per quad and instruction)
template <int MODE>
__global__ __launch_bounds__(256) void walk(const uint4* __restrict__ pairs, uint32_t npairs, uint32_t share, uint32_t iters,
                                            uint32_t* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    uint32_t state = wave * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t it = 0; it < iters; it++) {
        state = state * 1664525u + 1013904223u;
        // `share` consecutive lanes visit the same pair; different groups visit unrelated pairs inside a 1 MB window
        // (L2 / L1 resident like the hot part of a BVH)
        const uint32_t grp = lane / share;
        uint32_t h = (state ^ (grp * 0x9E3779B9u));
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        const uint32_t idx = h % npairs;
        uint4 a, b, c, d;
        if (MODE == 0) {
            const uint4* q = pairs + (size_t)idx * 4;
            a = q[0]; b = q[1]; c = q[2]; d = q[3];
        } else {
            const uint32_t j = lane & 3;
            a = pairs[(size_t)qperm<0x00>(idx) * 4 + j];
            b = pairs[(size_t)qperm<0x55>(idx) * 4 + j];
            c = pairs[(size_t)qperm<0xAA>(idx) * 4 + j];
            d = pairs[(size_t)qperm<0xFF>(idx) * 4 + j];
        }
        acc += a.x ^ b.y ^ c.z ^ d.w;
        state ^= acc & 0xFF;   // make the next address depend on the loaded data (a traversal's dependent chain)
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

int main()
{
    // windows: 16 KiB (fits every CU's 32 KiB L1: the tracer's 97 % L1-hit regime), 1 MiB (L2), 64 MiB (Infinity Cache)
    const uint32_t max_pairs = 1u << 20;   // 64 MiB
    std::vector<uint32_t> host((size_t)max_pairs * 16);
    for (size_t i = 0; i < host.size(); i++) host[i] = (uint32_t)(i * 2654435761u);
    uint4* dev; uint32_t* sink;
    hipMalloc(&dev, host.size() * 4); hipMalloc(&sink, 4);
    hipMemcpy(dev, host.data(), host.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t blocks = 256 * 8, iters = 2000;     // 8 workgroups x 4 waves = 32 waves per CU
    printf("# ta_microbench: wave-wide 16-byte-per-lane loads, 4 per iteration (one 64-byte pair per lane), 32 waves per CU,\n"
           "# dependent chain (next address from loaded data).  cycles/iter/CU = kernel time x 2.4 GHz / (wave-iterations per CU);\n"
           "# one iteration = 4 wave-loads, so cycles per wave-load = that / 4 and cycles per distinct lane address = that / 4 / (64 / share).\n");
    for (uint32_t npairs : {256u, 16384u, max_pairs})
        for (uint32_t share : {1u, 2u, 4u, 8u, 16u, 64u})
            for (int mode = 0; mode < 2; mode++) {
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    if (mode == 0) walk<0><<<blocks, 256>>>(dev, npairs, share, iters, sink);
                    else walk<1><<<blocks, 256>>>(dev, npairs, share, iters, sink);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double wave_iters = (double)blocks * 4 * iters;
                const double cyc = ms * 1e-3 * 2.4e9 / (wave_iters / 256);
                printf("window %6u KiB  share %2u lanes/pair (%2u distinct)  mode %d (%s): %8.3f ms  %6.1f cycles/iter/CU  %5.1f cycles/wave-load  %5.2f cycles/distinct address\n",
                       npairs * 64 / 1024, share, 64 / share, mode, mode ? "quad fetch " : "own pair x4", ms, cyc, cyc / 4,
                       cyc / 4 / (64.0 / share));
            }
    return 0;
}
