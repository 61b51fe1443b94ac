// Arguments.h -- the command-line options of the reference (its Arguments.h:8-37, parsed by Arguments.cpp:42-63), kept
// source compatible for callers (`args.build_type == kHybrid`, `args.enable_pairs` ...) and tied to the C ABI: every
// enumerator takes its value from include/rt_abi.h, and abi() yields the rt_arguments the library entry points take.
#pragma once
#include <string>

#include "rt_abi.h"

extern std::string g_filename;   // argv[1], as in the reference

enum BuildType {
    kSAH = RT_BUILD_SAH,             // the default, like the reference
    kBottomUp = RT_BUILD_BOTTOM_UP,
    kHybrid = RT_BUILD_HYBRID,
    kNone = RT_BUILD_NONE,
};

enum RenderType {
    kDepth = RT_RENDER_DEPTH,
    kBoxtests = RT_RENDER_BOXTESTS,
    kTriangleTests = RT_RENDER_TRIANGLE_TESTS,
    kMaterialId = RT_RENDER_MATERIAL_ID,
    kLODs = RT_RENDER_LODS,
    kDiffuse = RT_RENDER_DIFFUSE,
    kTexture = RT_RENDER_TEXTURE,
    kTextureLit = RT_RENDER_TEXTURE_LIT,
    kTextureLitShadows = RT_RENDER_TEXTURE_LIT_SHADOWS,
    kCount = RT_RENDER_COUNT,
};

struct Arguments {
    BuildType build_type = kSAH;
    bool enable_splits = false;      // --splits: SAH builds only (RunBottomUpBuild ignores it, as in the reference)
    bool enable_pairs = false;       // --pairs
    RenderType render_type = kDepth;

    rt_arguments abi() const
    {
        return rt_arguments{(int32_t)build_type, enable_splits ? 1 : 0, enable_pairs ? 1 : 0, (int32_t)render_type};
    }
};

// `<file.obj> [--pairs] [--splits] [--type sah|bottom-up|hybrid]`; an unknown --type gives kNone (the reference asserts)
Arguments ParseCmd(int argc, char** argv);
std::string BuildTypeToString(BuildType b);
