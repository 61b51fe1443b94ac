#include "Arguments.h"

#include <cstdio>

std::string g_filename = "";

std::string BuildTypeToString(BuildType b)
{
    switch (b) {
    case kHybrid: return "hybrid";
    case kSAH: return "sah";
    case kBottomUp: return "bottom-up";
    default: return "none";
    }
}

static BuildType ParseType(const std::string& s)
{
    if (s == "hybrid") return kHybrid;
    if (s == "sah") return kSAH;
    if (s == "bottom-up") return kBottomUp;
    return kNone;
}

Arguments ParseCmd(int argc, char** argv)
{
    Arguments args;
    if (argc >= 2) g_filename = argv[1];
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--pairs") args.enable_pairs = true;
        else if (a == "--splits") args.enable_splits = true;
        else if (a == "--type" && i + 1 < argc) args.build_type = ParseType(argv[++i]);
    }
    printf("Arguments:\n  BuildType: %s\n  Pairs: %s\n  Splits: %s\n\n", BuildTypeToString(args.build_type).c_str(),
           args.enable_pairs ? "true" : "false", args.enable_splits ? "true" : "false");
    return args;
}
