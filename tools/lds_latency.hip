// lds_latency.hip -- latencies that bound a dependent LDS chain on gfx950 (one wave, nothing else on the CU):
// ds_read pointer chase, ds_wrxchg_rtn chain, the deposit + exchange + sibling-read step of the LBVH climb, and the
// issue cost of a burst of global stores with few lanes active.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_latency tools/lds_latency.hip && tools/bin/lds_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kN = 256;

__global__ void k_read_chase(uint64_t* out, int lanes)
{
    __shared__ uint32_t a[4096];
    for (uint32_t i = threadIdx.x; i < 4096; i += 64) a[i] = (i * 97u + 64u) & 4095u;
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    uint32_t p = threadIdx.x;
    const uint64_t t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kN; i++) p = a[p];
    const uint64_t t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}

__global__ void k_xchg_chase(uint64_t* out, int lanes)
{
    __shared__ uint32_t a[4096];
    for (uint32_t i = threadIdx.x; i < 4096; i += 64) a[i] = (i * 97u + 64u) & 4095u;
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    uint32_t p = threadIdx.x;
    const uint64_t t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kN; i++) p = atomicExch(&a[p], p) & 4095u;
    const uint64_t t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p; }
}

// the climb step: 9 dword deposits, exchange, then 9 dependent dword reads at the returned index
__global__ void k_step(uint64_t* out, int lanes)
{
    __shared__ uint32_t lock[2048];
    __shared__ uint32_t st[9][2048];
    for (uint32_t i = threadIdx.x; i < 2048; i += 64) {
        lock[i] = (i * 97u + 64u) & 2047u;
        for (int k = 0; k < 9; k++) st[k][i] = (i * 31u + k) & 2047u;
    }
    __syncthreads();
    if ((int)threadIdx.x >= lanes) return;
    uint32_t p = threadIdx.x, acc = 0;
    const uint64_t t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kN; i++) {
#pragma unroll
        for (int k = 0; k < 9; k++) st[k][p] = acc + k;
        asm volatile("" ::: "memory");
        const uint32_t o = atomicExch(&lock[p], p) & 2047u;
        asm volatile("" ::: "memory");
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += st[k][o];
        acc = s;
        p = (o + s) & 2047u;
    }
    const uint64_t t1 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = p + acc; }
}

// issue cost of 10 global stores (4 x 16 B + 6 x 4 B, like one merge of an upper pass), `lanes` lanes active, scattered
// 64-byte-aligned targets; no wait for completion inside the timed region except at the very end
__global__ void k_stores(uint64_t* out, uint32_t* dst, int lanes, int wait_each)
{
    if ((int)threadIdx.x >= lanes) return;
    uint32_t p = threadIdx.x * 7919u;
    const uint64_t t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < kN; i++) {
        uint32_t* q = dst + (size_t)((p + i * 613u) & 0xFFFFu) * 16;
        uint4 v = make_uint4(p, i, 2, 3);
        reinterpret_cast<uint4*>(q)[0] = v;
        reinterpret_cast<uint4*>(q)[1] = v;
        reinterpret_cast<uint4*>(q)[2] = v;
        reinterpret_cast<uint4*>(q)[3] = v;
        uint32_t* r = dst + (size_t)((p * 3u + i * 977u) & 0xFFFFu) * 16;
        r[3] = i; r[11] = i;
        uint32_t* w = dst + (size_t)((p * 5u + i * 1013u) & 0xFFFFu) * 16;
        w[3] = i; w[11] = i;
        w[19] = i; w[27] = i;
        if (wait_each) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint64_t t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t2 = clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = t2 - t0; }
}

int main()
{
    uint64_t* d;
    uint32_t* dst;
    CHECK(hipMalloc(&d, 64));
    CHECK(hipMalloc(&dst, (size_t)65536 * 64 + 256));
    uint64_t h[2];
    const int lanes_list[] = {1, 4, 64};
    for (int lanes : lanes_list) {
        for (int rep = 0; rep < 2; rep++) k_read_chase<<<1, 64>>>(d, lanes);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("ds_read_b32 dependent chain      lanes %2d: %6.1f clk per op\n", lanes, (double)h[0] / kN);
        for (int rep = 0; rep < 2; rep++) k_xchg_chase<<<1, 64>>>(d, lanes);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("ds_wrxchg_rtn_b32 dependent chain lanes %2d: %6.1f clk per op\n", lanes, (double)h[0] / kN);
        for (int rep = 0; rep < 2; rep++) k_step<<<1, 64>>>(d, lanes);
        CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("deposit(9) + xchg + read(9) step  lanes %2d: %6.1f clk per step\n", lanes, (double)h[0] / kN);
        for (int w = 0; w < 2; w++) {
            for (int rep = 0; rep < 2; rep++) k_stores<<<1, 64>>>(d, dst, lanes, w);
            CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
            printf("10 global stores per step %s lanes %2d: %6.1f clk per step issued, %6.1f incl. final drain\n",
                   w ? "(vmcnt(0) each step)" : "(fire and forget)   ", lanes, (double)h[0] / kN, (double)h[1] / kN);
        }
    }
    return 0;
}
