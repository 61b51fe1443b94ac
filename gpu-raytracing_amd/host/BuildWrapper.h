// BuildWrapper.h -- the reference's BuildWrapper.cuh:6-20 entry points over the C ABI (include/rt_abi.h).
#pragma once
#include <cstddef>
#include <cstdint>

#include "Arguments.h"
#include "Common.h"

struct BuildInput {              // BuildWrapper.cuh:6-12 (all device pointers, owned by the caller)
    Triangle* triangles_in;
    TrianglePair* triangles_out;
    unsigned num_triangles;
    Node* nodes_out;
    void* scratch;
};

size_t BuMemoryRequirements(uint32_t num_triangles);                       // BuildWrapper.cu:132-136
// Launches the LBVH build on `stream` (default stream when null) and returns without synchronising (the reference
// synchronises after every kernel through its run() macro).  Error convention of the reference: message + exit.
void RunBottomUpBuild(BuildInput input, Arguments args, bool hybrid, void* stream = nullptr);  // BuildWrapper.cu:253-362
size_t SahMemoryRequirements(uint32_t num_triangles);                      // BuildWrapper.cu:126-130
// SAH build (args.enable_pairs, args.enable_splits as in the reference).  Trace root = (0, 1) (main.cu:222-223).
// Asynchronous on `stream` like RunBottomUpBuild (the reference reads num_leaves back and loops on the host,
// BuildWrapper.cu:229; here the data-dependent tail is a device-side loop).  Error flags: the scratch status word.
void RunSahBuild(BuildInput input, Arguments args, void* stream = nullptr);  // BuildWrapper.cu:140-251
