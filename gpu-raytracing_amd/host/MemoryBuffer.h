// MemoryBuffer.h -- host vector + device allocation pair (role of the reference's MemoryBuffer.h:1-85), on HIP.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

// check(): same convention as the reference's GpuAssert (Common.cuh:354-366): print and exit on a runtime error
#define check(ans) GpuAssert((ans), __FILE__, __LINE__)
inline void GpuAssert(hipError_t code, const char* file, int line)
{
    if (code != hipSuccess) {
        fprintf(stderr, "gpu_assert: %s %s %d\n", hipGetErrorString(code), file, line);
        exit((int)code);
    }
}

template <typename T>
class MemoryBuffer {
public:
    explicit MemoryBuffer(size_t count) : host_(count) { check(hipMalloc(reinterpret_cast<void**>(&device_), bytes())); }
    ~MemoryBuffer() { if (device_) (void)hipFree(device_); }
    MemoryBuffer(const MemoryBuffer&) = delete;
    MemoryBuffer& operator=(const MemoryBuffer&) = delete;
    T* data() { return host_.data(); }
    T* gpu() { return device_; }
    T& operator[](size_t i) { return host_[i]; }
    size_t size() const { return host_.size(); }
    size_t bytes() const { return host_.size() * sizeof(T); }
    void toDevice(hipStream_t st = nullptr) { check(hipMemcpyAsync(device_, host_.data(), bytes(), hipMemcpyHostToDevice, st)); }
    void toHost(hipStream_t st = nullptr)
    {
        check(hipMemcpyAsync(host_.data(), device_, bytes(), hipMemcpyDeviceToHost, st));
        check(hipStreamSynchronize(st));
    }
private:
    std::vector<T> host_;
    T* device_ = nullptr;
};
