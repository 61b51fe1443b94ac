// RadixSort.h -- the reference's RadixSort.cuh:6-7 over the C ABI.
#pragma once
#include <cstdint>

// Stable ascending sort of (key, value) pairs on the device; result in gpu_keys / gpu_values.  The reference
// allocates its tables inside the call (RadixSort.cu:187-190); here `sort_scratch` (>= RadixSortScratchBytes(count)
// device bytes) comes from the caller so the call stays asynchronous.
size_t RadixSortScratchBytes(uint32_t count);
void RadixSort(uint32_t* gpu_keys, uint32_t* gpu_values, uint32_t* gpu_temp1, uint32_t* gpu_temp2, uint32_t count,
               void* sort_scratch, void* stream = nullptr);
