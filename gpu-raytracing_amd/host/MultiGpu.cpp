#include "MultiGpu.h"

#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "MemoryBuffer.h"

#define nccl_check(ans) NcclAssert((ans), __FILE__, __LINE__)
static void NcclAssert(ncclResult_t r, const char* file, int line)
{
    if (r != ncclSuccess) {
        fprintf(stderr, "gpu_assert: RCCL: %s %s %d\n", ncclGetErrorString(r), file, line);   // the reference's convention: print and exit
        exit((int)r);
    }
}

void MultiGpuTracer::SetDevice(const Replica& r) const { check(hipSetDevice(r.device)); }

MultiGpuTracer::MultiGpuTracer(int devices, int inflight, bool virtual_devices) : virtual_(virtual_devices)
{
    int have = 0;
    check(hipGetDeviceCount(&have));
    if (devices < 1 || (!virtual_ && devices > have) || devices > 64) {
        fprintf(stderr, "gpu_assert: --gpus %d but %d device(s) visible\n", devices, have);
        exit(2);
    }
    if (inflight < 1 || inflight > 16) {
        fprintf(stderr, "gpu_assert: --inflight %d (1 .. 16)\n", inflight);
        exit(2);
    }
    dev_.resize((size_t)devices);
    std::vector<int> list((size_t)devices);
    for (int d = 0; d < devices; d++) dev_[(size_t)d].device = list[(size_t)d] = virtual_ ? 0 : d;
    slot_.resize((size_t)inflight);
    for (Slot& s : slot_) {
        s.dev.resize((size_t)devices);
        for (int d = 0; d < devices; d++) {
            DevSlot& x = s.dev[(size_t)d];
            SetDevice(dev_[(size_t)d]);
            check(hipStreamCreateWithFlags(&x.stream, hipStreamNonBlocking));
            check(hipEventCreate(&x.e0));
            check(hipEventCreate(&x.e1));
            check(hipMalloc((void**)&x.camera, sizeof(Camera)));
            check(hipMalloc((void**)&x.num_tests, sizeof(uint64_t) * 4));
        }
        // one set of communicators per slot: RCCL orders the operations of a communicator, slots must stay independent
        s.comms.assign((size_t)devices, nullptr);
        if (!virtual_) nccl_check(ncclCommInitAll(reinterpret_cast<ncclComm_t*>(s.comms.data()), devices, list.data()));
        SetDevice(dev_[0]);
        check(hipMalloc((void**)&s.totals, sizeof(uint64_t) * 4));
        check(hipEventCreateWithFlags(&s.gathered, hipEventDisableTiming));
        s.device_ms.assign((size_t)devices, 0.0f);
    }
}

void MultiGpuTracer::WaitAll()
{
    for (Slot& s : slot_)
        for (size_t d = 0; d < dev_.size(); d++) {
            SetDevice(dev_[d]);
            check(hipStreamSynchronize(s.dev[d].stream));
        }
}

MultiGpuTracer::~MultiGpuTracer()
{
    for (Slot& s : slot_)
        for (size_t d = 0; d < dev_.size(); d++) {
            (void)hipSetDevice(dev_[d].device);
            (void)hipStreamSynchronize(s.dev[d].stream);
        }
    for (Slot& s : slot_) {
        for (ncclComm* c : s.comms) if (c) (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(c));
        for (size_t d = 0; d < dev_.size(); d++) {
            DevSlot& x = s.dev[d];
            (void)hipSetDevice(dev_[d].device);
            for (void* p : {(void*)x.camera, (void*)x.frame, (void*)(d == 0 ? nullptr : x.compact), (void*)x.num_tests})
                if (p) (void)hipFree(p);
            if (x.e0) (void)hipEventDestroy(x.e0);
            if (x.e1) (void)hipEventDestroy(x.e1);
            if (x.stream) (void)hipStreamDestroy(x.stream);
        }
        (void)hipSetDevice(dev_[0].device);
        if (s.staging) (void)hipFree(s.staging);
        if (s.totals) (void)hipFree(s.totals);
        if (s.gathered) (void)hipEventDestroy(s.gathered);
    }
    for (size_t d = 0; d < dev_.size(); d++) {
        Replica& r = dev_[d];
        if (virtual_ && d > 0) break;          // virtual devices alias replica 0's scene
        (void)hipSetDevice(r.device);
        r.textures.Free();
        for (void* p : {(void*)r.in.triangles_in, (void*)r.in.triangles_out, (void*)r.in.nodes_out, r.in.scratch, (void*)r.attributes,
                        (void*)r.materials})
            if (p) (void)hipFree(p);
    }
    (void)hipSetDevice(dev_.empty() ? 0 : dev_[0].device);
}

void MultiGpuTracer::UploadScene(const Scene& scene)
{
    const unsigned n = num_triangles_ = (unsigned)scene.triangles.size();
    std::vector<rt_material> mats;
    for (const Material& m : scene.library.materials) mats.push_back(m.pod());
    for (size_t d = 0; d < dev_.size(); d++) {
        Replica& r = dev_[d];
        hipStream_t st = slot_[0].dev[d].stream;
        SetDevice(r);
        if (virtual_ && d > 0) {               // one scene, one tree: the virtual devices read replica 0's
            const int keep = r.device;
            r = dev_[0];
            r.device = keep;
            continue;
        }
        // the four build buffers of Display() frame 0 (main.cu:226-240); the scratch covers either builder
        const size_t bu = BuMemoryRequirements(n), sah = SahMemoryRequirements(n);
        r.in.num_triangles = n;
        check(hipMalloc((void**)&r.in.triangles_in, sizeof(Triangle) * (n ? n : 1)));
        check(hipMalloc((void**)&r.in.triangles_out, sizeof(TrianglePair) * (size_t)(n ? n : 1) * 2));
        check(hipMalloc(&r.in.scratch, bu > sah ? bu : sah));
        check(hipMalloc((void**)&r.in.nodes_out, rt_nodes_bytes(n)));
        if (n) {
            check(hipMemcpyAsync(r.in.triangles_in, scene.triangles.data(), sizeof(Triangle) * n, hipMemcpyHostToDevice, st));
            check(hipMalloc((void**)&r.attributes, sizeof(Attributes) * n));
            check(hipMemcpyAsync(r.attributes, scene.attributes.data(), sizeof(Attributes) * n, hipMemcpyHostToDevice, st));
        }
        r.num_materials = (uint32_t)mats.size();
        if (!mats.empty()) {
            check(hipMalloc((void**)&r.materials, sizeof(rt_material) * mats.size()));
            check(hipMemcpyAsync(r.materials, mats.data(), sizeof(rt_material) * mats.size(), hipMemcpyHostToDevice, st));
        }
        r.textures.Upload(scene.library);
        r.light = scene.light;
    }
    WaitAll();   // the host vectors may go away
}

float MultiGpuTracer::Build(const Arguments& args)
{
    sah_ = args.build_type == kSAH;
    const bool hybrid = args.build_type == kHybrid;
    // Both builds are sequences of asynchronous launches (RunSahBuild since round 4: its data-dependent tail is a device-side
    // loop): one host thread issues them on every device and the devices build concurrently.
    auto build_one = [&](size_t d) {
        Replica& r = dev_[d];
        DevSlot& x = slot_[0].dev[d];
        check(hipSetDevice(r.device));
        check(hipEventRecord(x.e0, x.stream));
        if (sah_) RunSahBuild(r.in, args, x.stream); else RunBottomUpBuild(r.in, args, hybrid, x.stream);
        check(hipEventRecord(x.e1, x.stream));
    };
    for (size_t d = 0; d < (virtual_ ? 1 : dev_.size()); d++) build_one(d);
    if (virtual_) {                            // (every virtual device's streams start behind the one build)
        check(hipStreamSynchronize(slot_[0].dev[0].stream));
        for (size_t d = 1; d < dev_.size(); d++) { check(hipEventRecord(slot_[0].dev[d].e0, slot_[0].dev[d].stream)); check(hipEventRecord(slot_[0].dev[d].e1, slot_[0].dev[d].stream)); }
    }
    float worst = 0;
    for (size_t d = 0; d < dev_.size(); d++) {
        SetDevice(dev_[d]);
        check(hipEventSynchronize(slot_[0].dev[d].e1));
        float ms = 0;
        check(hipEventElapsedTime(&ms, slot_[0].dev[d].e0, slot_[0].dev[d].e1));
        worst = ms > worst ? ms : worst;
    }
    return worst;
}

void MultiGpuTracer::Resize(int width, int height)
{
    WaitAll();
    width_ = width; height_ = height;
    const unsigned P = (unsigned)dev_.size();
    const size_t frame_bytes = (size_t)width * height * 4;
    const size_t compact_bytes = (size_t)CompactRows((unsigned)height, P) * width * 4;
    for (Slot& s : slot_) {
        SetDevice(dev_[0]);
        if (s.staging) check(hipFree(s.staging));
        check(hipMalloc((void**)&s.staging, compact_bytes * P));
        for (unsigned d = 0; d < P; d++) {
            DevSlot& x = s.dev[d];
            SetDevice(dev_[d]);
            if (x.frame) check(hipFree(x.frame));
            check(hipMalloc((void**)&x.frame, frame_bytes));
            check(hipMemsetAsync(x.frame, 0, frame_bytes, x.stream));
            if (d == 0) {
                x.compact = s.staging;     // device 0 renders its strips straight into slot 0 of the staging area
            } else {
                if (x.compact) check(hipFree(x.compact));
                check(hipMalloc((void**)&x.compact, compact_bytes));
            }
        }
        s.used = false;
    }
    decided_ = Partition::kAuto;
    probe_slot_ = -1;
    next_slot_ = last_slot_ = 0;
}

// kAuto: the per-device times of the probe frame decide -- once they are there.  Never waits.
void MultiGpuTracer::PollAutoDecision()
{
    if (decided_ != Partition::kAuto || probe_slot_ < 0) return;
    Slot& s = slot_[(size_t)probe_slot_];
    const unsigned P = (unsigned)dev_.size();
    for (unsigned d = 0; d < P; d++) {
        SetDevice(dev_[d]);
        const hipError_t q = hipEventQuery(s.dev[d].e1);
        if (q == hipErrorNotReady) return;
        check(q);
    }
    std::vector<double> cost(P);
    for (unsigned d = 0; d < P; d++) {
        float ms = 0;
        SetDevice(dev_[d]);
        check(hipEventElapsedTime(&ms, s.dev[d].e0, s.dev[d].e1));
        cost[d] = ms;
    }
    decided_ = ChoosePartition(cost.data(), P);
    probe_slot_ = -1;
}

int MultiGpuTracer::TraceFrame(const Camera& camera, RenderType render_type, unsigned root, unsigned count, unsigned spp,
                               Partition partition)
{
    const unsigned P = (unsigned)dev_.size(), W = (unsigned)width_, H = (unsigned)height_;
    Partition use = partition;
    if (partition == Partition::kAuto) {
        if (P == 1) decided_ = Partition::kBands;   // (explicit strips on one device: the same code path, de-interleave = identity)
        PollAutoDecision();
        use = decided_ == Partition::kAuto ? Partition::kBands : decided_;
    }
    const int si = next_slot_;
    next_slot_ = (next_slot_ + 1) % (int)slot_.size();
    last_slot_ = si;
    Slot& s = slot_[(size_t)si];
    if (probe_slot_ == si) probe_slot_ = -1;          // the probe's events are about to be re-recorded
    if (virtual_ && s.used)   // the slot's previous gather (copies on device 0's stream) must have read the parts before they are overwritten
        for (unsigned d = 1; d < P; d++) check(hipStreamWaitEvent(s.dev[d].stream, s.gathered, 0));
    s.partition = use;
    s.timed = false;
    s.used = true;

    // ---- every device: camera (64 B, main.cu:151), counters, its part of the frame
    for (unsigned d = 0; d < P; d++) {
        Replica& r = dev_[d];
        DevSlot& x = s.dev[d];
        SetDevice(r);
        check(hipMemcpyAsync(x.camera, &camera, sizeof(Camera), hipMemcpyHostToDevice, x.stream));   // (pageable source: staged before the call returns)
        check(hipMemsetAsync(x.num_tests, 0, sizeof(uint64_t) * 4, x.stream));
        DeviceSceneView view;
        view.attributes = r.attributes;
        view.materials = r.materials;
        view.num_attributes = num_triangles_;
        view.num_materials = r.num_materials;
        view.textures = r.textures.table;
        view.num_textures = r.textures.count;
        view.light = r.light;
        check(hipEventRecord(x.e0, x.stream));
        if (use == Partition::kBands) {
            const RowBand b = BandOf(H, P, d);
            if (b.y1 > b.y0)
                Trace(r.in.triangles_out, r.in.nodes_out, x.frame, width_, height_, x.camera, root, count, render_type, view,
                      x.num_tests, b.y0, b.y1, spp, x.stream);
        } else if (StripsOwned(H, P, d)) {
            TraceStrips(r.in.triangles_out, r.in.nodes_out, x.compact, width_, height_, x.camera, root, count, render_type, view,
                        x.num_tests, kStripRows, d, P, spp, x.stream);
        }
        check(hipEventRecord(x.e1, x.stream));
    }
    if (partition == Partition::kAuto && decided_ == Partition::kAuto && probe_slot_ < 0) probe_slot_ = si;

    // ---- the parts travel to device 0 (Partition.h GatherPlan): ONE grouped send / recv per frame (point-to-point over xGMI
    // into GPU 0), and the counters are summed on device 0 (ncclReduce runs on a 1-device communicator too)
    GatherOp ops[kMaxGatherOps];
    const unsigned nops = GatherPlan(W, H, P, use, ops);
    if (virtual_) {
        // the same plan with copies: a sender's part is complete at its e1 event; device 0's slot stream waits for it and copies
        for (unsigned k = 0; k < nops; k++) {
            const GatherOp& o = ops[k];
            if (o.kind != GatherOp::kRecvBand && o.kind != GatherOp::kRecvCompact) continue;
            DevSlot& x = s.dev[o.device];
            const uint8_t* src = (o.kind == GatherOp::kRecvBand ? x.frame : x.compact) + o.src_off;
            uint8_t* dst = (o.kind == GatherOp::kRecvBand ? s.dev[0].frame : s.staging) + o.dst_off;
            check(hipStreamWaitEvent(s.dev[0].stream, x.e1, 0));
            check(hipMemcpyAsync(dst, src, o.bytes, hipMemcpyDeviceToDevice, s.dev[0].stream));
        }
    } else {
    nccl_check(ncclGroupStart());
    for (unsigned d = 0; d < P; d++) {
        DevSlot& x = s.dev[d];
        nccl_check(ncclReduce(x.num_tests, d == 0 ? s.totals : x.num_tests, 4, ncclUint64, ncclSum, 0,
                              reinterpret_cast<ncclComm_t>(s.comms[d]), x.stream));
    }
    ncclComm_t comm0 = reinterpret_cast<ncclComm_t>(s.comms[0]);
    for (unsigned k = 0; k < nops; k++) {
        const GatherOp& o = ops[k];
        if (o.kind != GatherOp::kRecvBand && o.kind != GatherOp::kRecvCompact) continue;
        DevSlot& x = s.dev[o.device];
        const uint8_t* src = (o.kind == GatherOp::kRecvBand ? x.frame : x.compact) + o.src_off;
        uint8_t* dst = (o.kind == GatherOp::kRecvBand ? s.dev[0].frame : s.staging) + o.dst_off;
        nccl_check(ncclSend(src, o.bytes, ncclUint8, 0, reinterpret_cast<ncclComm_t>(s.comms[o.device]), x.stream));
        nccl_check(ncclRecv(dst, o.bytes, ncclUint8, (int)o.device, comm0, s.dev[0].stream));
    }
    nccl_check(ncclGroupEnd());
    }

    // ---- strips: de-interleave on device 0 (after the receives, on the same stream)
    SetDevice(dev_[0]);
    for (unsigned k = 0; k < nops; k++) {
        const GatherOp& o = ops[k];
        if (o.kind == GatherOp::kCopyStrips)
            check(hipMemcpy2DAsync(s.dev[0].frame + o.dst_off, o.dst_pitch, s.staging + o.src_off, o.src_pitch, o.bytes, o.pieces,
                                   hipMemcpyDeviceToDevice, s.dev[0].stream));
        else if (o.kind == GatherOp::kCopyCut)
            check(hipMemcpyAsync(s.dev[0].frame + o.dst_off, s.staging + o.src_off, o.bytes, hipMemcpyDeviceToDevice, s.dev[0].stream));
    }
    if (virtual_) check(hipEventRecord(s.gathered, s.dev[0].stream));
    return si;
}

const std::vector<float>& MultiGpuTracer::DeviceMs(int slot)
{
    Slot& s = slot_[(size_t)Resolve(slot)];
    if (!s.timed && s.used) {
        for (size_t d = 0; d < dev_.size(); d++) {
            SetDevice(dev_[d]);
            check(hipEventSynchronize(s.dev[d].e1));
            check(hipEventElapsedTime(&s.device_ms[d], s.dev[d].e0, s.dev[d].e1));
        }
        s.timed = true;
    }
    return s.device_ms;
}

const uint8_t* MultiGpuTracer::Frame(int slot)
{
    Slot& s = slot_[(size_t)Resolve(slot)];
    // every device's stream of this slot: the sends are on them (device 0's holds the receives and the de-interleave)
    for (size_t d = 0; d < dev_.size(); d++) { SetDevice(dev_[d]); check(hipStreamSynchronize(s.dev[d].stream)); }
    SetDevice(dev_[0]);
    return s.dev[0].frame;
}

void MultiGpuTracer::FrameToHost(std::vector<uint8_t>& out, int slot)
{
    const uint8_t* f = Frame(slot);
    out.resize((size_t)width_ * height_ * 4);
    check(hipMemcpy(out.data(), f, out.size(), hipMemcpyDeviceToHost));
}

void MultiGpuTracer::Counters(uint64_t out[4], int slot)
{
    (void)Frame(slot);
    Slot& s = slot_[(size_t)Resolve(slot)];
    if (!virtual_) {
        check(hipMemcpy(out, s.totals, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost));
        return;
    }
    for (int k = 0; k < 4; k++) out[k] = 0;
    for (size_t d = 0; d < dev_.size(); d++) {       // (no RCCL in the virtual mode: the host adds the per-device counters)
        uint64_t v[4];
        check(hipMemcpy(v, s.dev[d].num_tests, sizeof v, hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; k++) out[k] += v[k];
    }
}
