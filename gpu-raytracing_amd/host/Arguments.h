// Arguments.h -- mirrors the reference's Arguments.h:8-37 / Arguments.cpp:42-63.
#pragma once
#include <string>

extern std::string g_filename;

enum BuildType { kSAH, kBottomUp, kHybrid, kNone };
enum RenderType { kDepth = 0, kBoxtests = 1, kTriangleTests = 2, kMaterialId = 3, kLODs = 4, kDiffuse = 5,
                  kTexture = 6, kTextureLit = 7, kTextureLitShadows = 8, kCount = 9 };

struct Arguments {
    BuildType build_type = kSAH;   // the reference's default (Arguments.h:29); only kBottomUp / kHybrid are built here
    bool enable_splits = false;
    bool enable_pairs = false;
    RenderType render_type = RenderType::kDepth;
};

// `<file.obj> [--pairs] [--splits] [--type sah|bottom-up|hybrid]`; unknown --type -> kNone (the reference asserts)
Arguments ParseCmd(int argc, char** argv);
std::string BuildTypeToString(BuildType b);
