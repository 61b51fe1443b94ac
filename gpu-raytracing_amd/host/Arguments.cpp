#include "Arguments.h"

#include <cstdio>

std::string g_filename = "";

namespace {
// the --type spellings of the reference (Arguments.cpp:9-33), one table for both directions
struct TypeName { const char* name; BuildType type; };
constexpr TypeName kTypeNames[] = {{"sah", kSAH}, {"bottom-up", kBottomUp}, {"hybrid", kHybrid}};

BuildType TypeFromName(const std::string& s)
{
    for (const TypeName& t : kTypeNames)
        if (s == t.name) return t.type;
    return kNone;
}
}  // namespace

std::string BuildTypeToString(BuildType b)
{
    for (const TypeName& t : kTypeNames)
        if (t.type == b) return t.name;
    return "none";
}

Arguments ParseCmd(int argc, char** argv)
{
    Arguments args;
    if (argc >= 2) g_filename = argv[1];
    for (int i = 2; i < argc; i++) {
        const std::string opt = argv[i];
        if (opt == "--pairs") args.enable_pairs = true;
        else if (opt == "--splits") args.enable_splits = true;
        else if (opt == "--type" && i + 1 < argc) args.build_type = TypeFromName(argv[++i]);
    }
    // the reference's PrintArgs block (Arguments.cpp:35-40)
    printf("Arguments:\n  BuildType: %s\n  Pairs: %s\n  Splits: %s\n\n", BuildTypeToString(args.build_type).c_str(),
           args.enable_pairs ? "true" : "false", args.enable_splits ? "true" : "false");
    return args;
}
