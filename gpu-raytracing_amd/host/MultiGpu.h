// MultiGpu.h -- Trace() across the GPUs of one node from ONE host process (north_star: "Host code stays C++ ... the path
// shards by image tile across the 8 GPUs of one node with the BVH replicated and a final RCCL gather of tile buffers over
// xGMI"; SURVEY 5 last row / 8(e)).  The reference has one GPU, one stream and no collective (main.cu:125-192); this is
// the multi-device form of the same Display() steps:
//   * every device holds a replica of the scene and builds the identical (deterministic) tree itself -- the build does not
//     shard ("replicas only"), and rebuilding costs less than broadcasting nodes + leaves;
//   * per frame every device traces its part on its own HIP stream (rt_trace row band, or rt_trace_strips), then the parts
//     travel to device 0 with ONE grouped RCCL send/recv per frame (direct xGMI links into GPU 0, not a ring) and the
//     per-device test counters are summed by one ncclReduce;
//   * one host thread drives all devices (ncclCommInitAll, as SURVEY 5 sketches); nothing synchronises until the caller
//     asks for the frame.
// The kernels' per-device function attributes are handled inside librt_amd.so (PerDeviceOnce, csrc/rt_launch.hpp).
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <vector>

#include "Arguments.h"
#include "BuildWrapper.h"
#include "Common.h"
#include "FileIO.h"
#include "Partition.h"
#include "Tracer.h"

struct ncclComm;   // <rccl/rccl.h>

class MultiGpuTracer {
public:
    // devices <= hipGetDeviceCount(); creates one stream per device and the RCCL communicators (ncclCommInitAll)
    explicit MultiGpuTracer(int devices);
    ~MultiGpuTracer();
    MultiGpuTracer(const MultiGpuTracer&) = delete;
    MultiGpuTracer& operator=(const MultiGpuTracer&) = delete;

    int devices() const { return (int)dev_.size(); }
    // replicate triangles, attributes, materials and textures on every device (Scene::CopyToDevice, main.cu:421-456)
    void UploadScene(const Scene& scene);
    // the same build on every device, concurrently; returns the slowest device's build time (ms).  Synchronises.
    float Build(const Arguments& args);
    // frame buffers for a width x height frame (full frame + compact strip buffer per device, staging on device 0)
    void Resize(int width, int height);
    // One frame into device 0's frame buffer.  Asynchronous: returns after enqueueing; Frame() / Counters() synchronise.
    // `partition`: bands or strips (kAuto: bands for the first frame, then whatever the measured band costs say).
    void TraceFrame(const Camera& camera, RenderType render_type, unsigned root, unsigned count, unsigned spp, Partition partition);
    // waits for the frame; returns device 0's RGBA8 frame (device pointer) / copies it to the host
    const uint8_t* Frame();
    void FrameToHost(std::vector<uint8_t>& out);
    // sum over devices of the test counters of the last frame ([0] = box tests: "TraceRays number of tests", main.cu:180-183)
    void Counters(uint64_t out[4]);
    // per-device trace time of the last frame (ms, events on each device's stream) and the partition it used
    const std::vector<float>& DeviceMs();
    Partition LastPartition() const { return last_partition_; }
    // device 0's replica of the build (for the read-back / verify steps of frame 0, main.cu:248-259)
    const BuildInput& Replica0() const { return dev_[0].in; }
    hipStream_t Stream0() const { return dev_[0].stream; }

private:
    struct Replica {
        int device = 0;
        hipStream_t stream = nullptr;
        BuildInput in{};
        Camera* camera = nullptr;
        uint8_t* frame = nullptr;       // full frame (bands are traced in place)
        uint8_t* compact = nullptr;     // this device's strips (device 0: slot 0 of `staging`)
        uint64_t* num_tests = nullptr;
        Attributes* attributes = nullptr;
        rt_material* materials = nullptr;
        uint32_t num_materials = 0;
        DeviceTextureTable textures;
        vec3 light{0, 0, 0};
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };
    void SetDevice(const Replica& r) const;
    std::vector<Replica> dev_;
    std::vector<ncclComm*> comms_;
    uint8_t* staging_ = nullptr;        // device 0: the P compact buffers back to back (strips)
    uint64_t* totals_ = nullptr;        // device 0: summed counters
    int width_ = 0, height_ = 0;
    unsigned num_triangles_ = 0;
    bool sah_ = false;
    Partition last_partition_ = Partition::kBands;
    Partition decided_ = Partition::kAuto;   // kAuto until the first frame's band costs are known
    std::vector<float> device_ms_;
    bool timed_ = false;
};
