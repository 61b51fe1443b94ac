// MultiGpu.h -- Trace() across the GPUs of one node from ONE host process (north_star: "Host code stays C++ ... the path
// shards by image tile across the 8 GPUs of one node with the BVH replicated and a final RCCL gather of tile buffers over
// xGMI"; SURVEY 5 last row / 8(e)).  The reference has one GPU, one stream and no collective (main.cu:125-192); this is
// the multi-device form of the same Display() steps:
//   * every device holds a replica of the scene and builds the identical (deterministic) tree itself -- the build does not
//     shard ("replicas only"), and rebuilding costs less than broadcasting nodes + leaves;
//   * per frame every device traces its part on a HIP stream (rt_trace row band, or rt_trace_strips), then the parts travel
//     to device 0 with ONE grouped RCCL send/recv per frame (direct xGMI links into GPU 0, not a ring) and the per-device
//     test counters are summed by one ncclReduce;
//   * FRAMES IN FLIGHT: the tracer owns `inflight` slots; a slot is, per device, a stream + camera + frame / compact buffers
//     + counters, on device 0 also the staging area, and its own set of RCCL communicators (operations on one communicator
//     must stay ordered; slots must not order each other).  Frame i goes to slot i mod inflight: a 1/8 band is 4,050 waves
//     on a machine with 8,192 wave slots -- one half-empty, latency-bound round per launch -- so the next frame's waves fill
//     what this frame leaves idle (one-GPU band timings, DESIGN section 5: 0.40 ms serial vs 0.26 ms in flight per 1/8 band);
//   * one host thread drives all devices (ncclCommInitAll, as SURVEY 5 sketches); nothing synchronises until the caller
//     asks for a frame.
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
    // devices <= hipGetDeviceCount(); creates inflight streams per device and inflight sets of RCCL communicators
    // (ncclCommInitAll each)
    // virtual_devices: all `devices` replicas live on device 0 (they share ONE scene and tree; each has its own streams and
    // frame / compact buffers) and the gather runs as device-to-device copies ordered by events instead of RCCL send / recv.
    // Not a scaling mode: it exists so that the P > 1 frame pipeline -- band offsets, strip ownership, staging slots, the
    // de-interleave, frames in flight -- runs on hardware that has a single GPU (tests/test_gpu_golden.py).
    explicit MultiGpuTracer(int devices, int inflight = 1, bool virtual_devices = false);
    ~MultiGpuTracer();
    MultiGpuTracer(const MultiGpuTracer&) = delete;
    MultiGpuTracer& operator=(const MultiGpuTracer&) = delete;

    int devices() const { return (int)dev_.size(); }
    int inflight() const { return (int)slot_.size(); }
    // replicate triangles, attributes, materials and textures on every device (Scene::CopyToDevice, main.cu:421-456)
    void UploadScene(const Scene& scene);
    // the same build on every device, concurrently (both builders are asynchronous launch sequences); returns the slowest
    // device's build time (ms).  Synchronises.
    float Build(const Arguments& args);
    // frame buffers for a width x height frame (per slot and device: full frame + compact strip buffer; staging on device 0)
    void Resize(int width, int height);
    // One frame into the next slot's frame buffer on device 0; returns the slot.  Asynchronous: returns after enqueueing and
    // never waits (a slot's previous frame is ordered before the new one by the slot's streams; the CALLER must have taken
    // what it wants from that frame).  `partition`: bands or strips; kAuto: bands until the per-device times of an earlier
    // kAuto frame are available (polled, not waited for), then whatever they say.
    int TraceFrame(const Camera& camera, RenderType render_type, unsigned root, unsigned count, unsigned spp, Partition partition);
    // wait for the frame of `slot` (-1: the last one issued); device 0's RGBA8 frame (device pointer) / a host copy
    const uint8_t* Frame(int slot = -1);
    void FrameToHost(std::vector<uint8_t>& out, int slot = -1);
    // sum over devices of the test counters of that frame ([0] = box tests: "TraceRays number of tests", main.cu:180-183)
    void Counters(uint64_t out[4], int slot = -1);
    // per-device trace time of that frame (ms, events on each device's stream; frames in flight overlap) / its partition
    const std::vector<float>& DeviceMs(int slot = -1);
    Partition LastPartition(int slot = -1) const { return slot_[(size_t)Resolve(slot)].partition; }
    void WaitAll();
    // device 0's replica of the build (for the read-back / verify steps of frame 0, main.cu:248-259)
    const BuildInput& Replica0() const { return dev_[0].in; }

private:
    struct DevSlot {                    // one device's share of a slot
        hipStream_t stream = nullptr;
        Camera* camera = nullptr;
        uint8_t* frame = nullptr;       // full frame (bands are traced in place)
        uint8_t* compact = nullptr;     // this device's strips (device 0: slot 0 of the staging area)
        uint64_t* num_tests = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
    };
    struct Replica {
        int device = 0;
        BuildInput in{};
        Attributes* attributes = nullptr;
        rt_material* materials = nullptr;
        uint32_t num_materials = 0;
        DeviceTextureTable textures;
        vec3 light{0, 0, 0};
    };
    struct Slot {
        std::vector<DevSlot> dev;       // [device]
        std::vector<ncclComm*> comms;   // [device]
        uint8_t* staging = nullptr;     // device 0: the P compact buffers back to back (strips)
        uint64_t* totals = nullptr;     // device 0: summed counters
        hipEvent_t gathered = nullptr;  // virtual devices: device 0 has copied this slot's parts (the senders may overwrite them)
        Partition partition = Partition::kBands;
        std::vector<float> device_ms;
        bool timed = false, used = false;
    };
    void SetDevice(const Replica& r) const;
    int Resolve(int slot) const { return slot < 0 ? last_slot_ : slot; }
    void PollAutoDecision();
    std::vector<Replica> dev_;
    std::vector<Slot> slot_;
    int width_ = 0, height_ = 0;
    unsigned num_triangles_ = 0;
    bool sah_ = false;
    int next_slot_ = 0, last_slot_ = 0;
    bool virtual_ = false;
    Partition decided_ = Partition::kAuto;   // kAuto until the band costs of a probe frame are known
    int probe_slot_ = -1;                    // a frame issued as bands under kAuto whose times will decide
};
