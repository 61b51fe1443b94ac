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

MultiGpuTracer::MultiGpuTracer(int devices)
{
    int have = 0;
    check(hipGetDeviceCount(&have));
    if (devices < 1 || devices > have) {
        fprintf(stderr, "gpu_assert: --gpus %d but %d device(s) visible\n", devices, have);
        exit(2);
    }
    dev_.resize((size_t)devices);
    std::vector<int> list((size_t)devices);
    for (int d = 0; d < devices; d++) {
        Replica& r = dev_[(size_t)d];
        r.device = list[(size_t)d] = d;
        SetDevice(r);
        check(hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking));
        check(hipEventCreate(&r.e0));
        check(hipEventCreate(&r.e1));
        check(hipMalloc((void**)&r.camera, sizeof(Camera)));
        check(hipMalloc((void**)&r.num_tests, sizeof(uint64_t) * 4));
    }
    comms_.resize((size_t)devices);
    nccl_check(ncclCommInitAll(reinterpret_cast<ncclComm_t*>(comms_.data()), devices, list.data()));
    SetDevice(dev_[0]);
    check(hipMalloc((void**)&totals_, sizeof(uint64_t) * 4));
    device_ms_.assign((size_t)devices, 0.0f);
}

MultiGpuTracer::~MultiGpuTracer()
{
    for (Replica& r : dev_) {
        (void)hipSetDevice(r.device);
        (void)hipStreamSynchronize(r.stream);
    }
    for (ncclComm* c : comms_) if (c) (void)ncclCommDestroy(reinterpret_cast<ncclComm_t>(c));
    for (size_t d = 0; d < dev_.size(); d++) {
        Replica& r = dev_[d];
        (void)hipSetDevice(r.device);
        r.textures.Free();
        for (void* p : {(void*)r.in.triangles_in, (void*)r.in.triangles_out, (void*)r.in.nodes_out, r.in.scratch, (void*)r.camera,
                        (void*)r.frame, (void*)(d == 0 ? nullptr : r.compact), (void*)r.num_tests, (void*)r.attributes, (void*)r.materials})
            if (p) (void)hipFree(p);
        if (r.e0) (void)hipEventDestroy(r.e0);
        if (r.e1) (void)hipEventDestroy(r.e1);
        if (r.stream) (void)hipStreamDestroy(r.stream);
    }
    (void)hipSetDevice(dev_.empty() ? 0 : dev_[0].device);
    if (staging_) (void)hipFree(staging_);
    if (totals_) (void)hipFree(totals_);
}

void MultiGpuTracer::UploadScene(const Scene& scene)
{
    const unsigned n = num_triangles_ = (unsigned)scene.triangles.size();
    std::vector<rt_material> mats;
    for (const Material& m : scene.library.materials) mats.push_back(m.pod());
    for (Replica& r : dev_) {
        SetDevice(r);
        // the four build buffers of Display() frame 0 (main.cu:226-240); the scratch covers either builder
        const size_t bu = BuMemoryRequirements(n), sah = SahMemoryRequirements(n);
        r.in.num_triangles = n;
        check(hipMalloc((void**)&r.in.triangles_in, sizeof(Triangle) * (n ? n : 1)));
        check(hipMalloc((void**)&r.in.triangles_out, sizeof(TrianglePair) * (size_t)(n ? n : 1) * 2));
        check(hipMalloc(&r.in.scratch, bu > sah ? bu : sah));
        check(hipMalloc((void**)&r.in.nodes_out, rt_nodes_bytes(n)));
        if (n) {
            check(hipMemcpyAsync(r.in.triangles_in, scene.triangles.data(), sizeof(Triangle) * n, hipMemcpyHostToDevice, r.stream));
            check(hipMalloc((void**)&r.attributes, sizeof(Attributes) * n));
            check(hipMemcpyAsync(r.attributes, scene.attributes.data(), sizeof(Attributes) * n, hipMemcpyHostToDevice, r.stream));
        }
        r.num_materials = (uint32_t)mats.size();
        if (!mats.empty()) {
            check(hipMalloc((void**)&r.materials, sizeof(rt_material) * mats.size()));
            check(hipMemcpyAsync(r.materials, mats.data(), sizeof(rt_material) * mats.size(), hipMemcpyHostToDevice, r.stream));
        }
        r.textures.Upload(scene.library);
        r.light = scene.light;
    }
    for (Replica& r : dev_) { SetDevice(r); check(hipStreamSynchronize(r.stream)); }   // the host vectors may go away
}

float MultiGpuTracer::Build(const Arguments& args)
{
    sah_ = args.build_type == kSAH;
    const bool hybrid = args.build_type == kHybrid;
    // the bottom-up build is a sequence of asynchronous launches: all devices build concurrently.  (RunSahBuild synchronises
    // its stream -- data-dependent level count, like the reference -- so SAH replicas build one after the other.)
    for (Replica& r : dev_) {
        SetDevice(r);
        check(hipEventRecord(r.e0, r.stream));
        if (sah_) RunSahBuild(r.in, args, r.stream); else RunBottomUpBuild(r.in, args, hybrid, r.stream);
        check(hipEventRecord(r.e1, r.stream));
    }
    float worst = 0;
    for (Replica& r : dev_) {
        SetDevice(r);
        check(hipEventSynchronize(r.e1));
        float ms = 0;
        check(hipEventElapsedTime(&ms, r.e0, r.e1));
        worst = ms > worst ? ms : worst;
    }
    return worst;
}

void MultiGpuTracer::Resize(int width, int height)
{
    width_ = width; height_ = height;
    const unsigned P = (unsigned)dev_.size();
    const size_t frame_bytes = (size_t)width * height * 4;
    const size_t compact_bytes = (size_t)CompactRows((unsigned)height, P) * width * 4;
    SetDevice(dev_[0]);
    if (staging_) check(hipFree(staging_));
    check(hipMalloc((void**)&staging_, compact_bytes * P));
    for (unsigned d = 0; d < P; d++) {
        Replica& r = dev_[d];
        SetDevice(r);
        if (r.frame) check(hipFree(r.frame));
        check(hipMalloc((void**)&r.frame, frame_bytes));
        check(hipMemsetAsync(r.frame, 0, frame_bytes, r.stream));
        if (d == 0) {
            r.compact = staging_;      // device 0 renders its strips straight into slot 0 of the staging area
        } else {
            if (r.compact) check(hipFree(r.compact));
            check(hipMalloc((void**)&r.compact, compact_bytes));
        }
    }
    decided_ = Partition::kAuto;
}

void MultiGpuTracer::TraceFrame(const Camera& camera, RenderType render_type, unsigned root, unsigned count, unsigned spp,
                                Partition partition)
{
    const unsigned P = (unsigned)dev_.size(), W = (unsigned)width_, H = (unsigned)height_;
    Partition use = partition;
    if (partition == Partition::kAuto) use = decided_ == Partition::kAuto ? Partition::kBands : decided_;
    if (P == 1 && partition == Partition::kAuto) use = Partition::kBands;   // (explicit strips on one device: the same code path, de-interleave = identity)
    last_partition_ = use;
    const size_t row = (size_t)W * 4;
    const unsigned J = StripsPerDevice(H, P);
    const size_t compact_bytes = (size_t)J * kStripRows * row;

    // ---- every device: camera (64 B, main.cu:151), counters, its part of the frame
    for (unsigned d = 0; d < P; d++) {
        Replica& r = dev_[d];
        SetDevice(r);
        check(hipMemcpyAsync(r.camera, &camera, sizeof(Camera), hipMemcpyHostToDevice, r.stream));
        check(hipMemsetAsync(r.num_tests, 0, sizeof(uint64_t) * 4, r.stream));
        DeviceSceneView view;
        view.attributes = r.attributes;
        view.materials = r.materials;
        view.num_attributes = num_triangles_;
        view.num_materials = r.num_materials;
        view.textures = r.textures.table;
        view.num_textures = r.textures.count;
        view.light = r.light;
        check(hipEventRecord(r.e0, r.stream));
        if (use == Partition::kBands) {
            const RowBand b = BandOf(H, P, d);
            if (b.y1 > b.y0)
                Trace(r.in.triangles_out, r.in.nodes_out, r.frame, width_, height_, r.camera, root, count, render_type, view,
                      r.num_tests, b.y0, b.y1, spp, r.stream);
        } else if (StripsOwned(H, P, d)) {
            TraceStrips(r.in.triangles_out, r.in.nodes_out, r.compact, width_, height_, r.camera, root, count, render_type, view,
                        r.num_tests, kStripRows, d, P, spp, r.stream);
        }
        check(hipEventRecord(r.e1, r.stream));
    }
    timed_ = false;

    // ---- the parts travel to device 0: ONE grouped send / recv per frame (point-to-point over xGMI into GPU 0), and the
    // counters are summed on device 0 (ncclReduce runs on a 1-device communicator too)
    nccl_check(ncclGroupStart());
    for (unsigned d = 0; d < P; d++) {
        Replica& r = dev_[d];
        ncclComm_t comm = reinterpret_cast<ncclComm_t>(comms_[d]);
        nccl_check(ncclReduce(r.num_tests, d == 0 ? totals_ : r.num_tests, 4, ncclUint64, ncclSum, 0, comm, r.stream));
        if (d == 0) continue;
        ncclComm_t comm0 = reinterpret_cast<ncclComm_t>(comms_[0]);
        if (use == Partition::kBands) {
            const RowBand b = BandOf(H, P, d);
            const size_t bytes = (size_t)(b.y1 - b.y0) * row;
            if (!bytes) continue;
            nccl_check(ncclSend(r.frame + b.y0 * row, bytes, ncclUint8, 0, comm, r.stream));
            nccl_check(ncclRecv(dev_[0].frame + b.y0 * row, bytes, ncclUint8, (int)d, comm0, dev_[0].stream));
        } else {
            nccl_check(ncclSend(r.compact, compact_bytes, ncclUint8, 0, comm, r.stream));
            nccl_check(ncclRecv(staging_ + d * compact_bytes, compact_bytes, ncclUint8, (int)d, comm0, dev_[0].stream));
        }
    }
    nccl_check(ncclGroupEnd());

    if (use == Partition::kStrips) {
        // de-interleave on device 0: local strip j of device d is global strip d + j*P -- one strided copy per source device
        // for its strips that lie wholly inside the frame, one plain copy for a strip the frame's edge cuts
        SetDevice(dev_[0]);
        const size_t strip_bytes = kStripRows * row;
        for (unsigned d = 0; d < P; d++) {
            const unsigned owned = StripsOwned(H, P, d);
            if (!owned) continue;
            const unsigned last = d + (owned - 1) * P;
            const bool cut = StripRowsInFrame(H, last) < kStripRows;
            const unsigned whole = cut ? owned - 1 : owned;
            const uint8_t* src = staging_ + d * compact_bytes;
            if (whole)
                check(hipMemcpy2DAsync(dev_[0].frame + (size_t)d * strip_bytes, (size_t)P * strip_bytes, src, strip_bytes, strip_bytes, whole,
                                       hipMemcpyDeviceToDevice, dev_[0].stream));
            if (cut)
                check(hipMemcpyAsync(dev_[0].frame + (size_t)last * strip_bytes, src + (size_t)(owned - 1) * strip_bytes,
                                     (size_t)StripRowsInFrame(H, last) * row, hipMemcpyDeviceToDevice, dev_[0].stream));
        }
    }

    if (partition == Partition::kAuto && decided_ == Partition::kAuto && P > 1) {
        // the first frame went out as bands: its per-device times decide the partition of the following frames
        const std::vector<float>& ms = DeviceMs();
        std::vector<double> cost(ms.begin(), ms.end());
        decided_ = ChoosePartition(cost.data(), P);
    }
}

const std::vector<float>& MultiGpuTracer::DeviceMs()
{
    if (!timed_) {
        for (size_t d = 0; d < dev_.size(); d++) {
            SetDevice(dev_[d]);
            check(hipEventSynchronize(dev_[d].e1));
            check(hipEventElapsedTime(&device_ms_[d], dev_[d].e0, dev_[d].e1));
        }
        timed_ = true;
    }
    return device_ms_;
}

const uint8_t* MultiGpuTracer::Frame()
{
    for (Replica& r : dev_) { SetDevice(r); check(hipStreamSynchronize(r.stream)); }
    SetDevice(dev_[0]);
    return dev_[0].frame;
}

void MultiGpuTracer::FrameToHost(std::vector<uint8_t>& out)
{
    const uint8_t* f = Frame();
    out.resize((size_t)width_ * height_ * 4);
    check(hipMemcpy(out.data(), f, out.size(), hipMemcpyDeviceToHost));
}

void MultiGpuTracer::Counters(uint64_t out[4])
{
    (void)Frame();
    check(hipMemcpy(out, totals_, sizeof(uint64_t) * 4, hipMemcpyDeviceToHost));
}
