#include "BuildWrapper.h"

#include <cstdio>
#include <cstdlib>

static void Die(const char* what, int rc)
{
    fprintf(stderr, "gpu_assert: %s: %s (%d)\n", what, rt_error_string(rc), rc);
    exit(rc < 0 ? -rc : rc);
}

size_t BuMemoryRequirements(uint32_t num_triangles) { return rt_bu_memory_requirements(num_triangles); }

void RunBottomUpBuild(BuildInput input, Arguments args, bool hybrid, void* stream)
{
    rt_build_input in{input.triangles_in, input.triangles_out, input.num_triangles, input.nodes_out, input.scratch};
    const rt_arguments a = args.abi();
    const int rc = rt_run_bottom_up_build(&in, &a, hybrid ? 1 : 0, stream);
    if (rc != RT_OK) Die("RunBottomUpBuild", rc);
}

size_t SahMemoryRequirements(uint32_t num_triangles) { return rt_sah_memory_requirements(num_triangles); }

void RunSahBuild(BuildInput input, Arguments args, void* stream)
{
    rt_build_input in{input.triangles_in, input.triangles_out, input.num_triangles, input.nodes_out, input.scratch};
    const rt_arguments a = args.abi();
    const int rc = rt_run_sah_build(&in, &a, stream);
    if (rc != RT_OK) Die("RunSahBuild", rc);
}
