#!/usr/bin/env python3
"""bench.py -- Mrays/s at 1920x1080 primary rays (+ BVH build ms) on the 1M-triangle procedural mesh.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (one
rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or invoked directly, in which case this
process starts the N rank processes itself (it touches no GPU before doing so) and relays rank 0's JSON line.

A "step" is one frame of the hot path over synthetic input already resident in HBM: every rank traces its part of
the 1920x1080 frame (row bands, or interleaved 8-row strips when the bands are unbalanced: gpu-raytracing_amd/sharding.py)
against its own replica of the LBVH (the build is deterministic, so replicas are identical) and, for N > 1, the parts
are gathered into rank 0's frame buffer with one RCCL gather per frame.  The frame is fixed as N grows (strong
scaling, BASELINE config 3).  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): grid_mesh(G=708, seed=1) = 1,002,528 triangles, camera A
("top-down", 100 % coverage), kDepth, 1 spp.  The build time and the CPU baseline are reported as extra fields of
the same line; --other-camera adds camera B ("oblique").

What `value` is: K frames / wall time of the timed region with `--inflight` (default 8) frames in flight on separate HIP
streams (the stream argument of rt_trace; the frames of a camera path are independent).  The one-frame-at-a-time rate
(`serial_mrays`: median of >= 20 single launches timed with events) is on the same line, and
`--inflight 1` makes it the headline.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)
L1_PEAK_GBS = 256 * 64 * 2.4     # vector L1 (TCP) data path: 256 CUs x 64 B/clk x 2.4 GHz = 39,321.6 GB/s
# measured ceiling of DIVERGENT 16-byte-per-lane gathers through the same path (profiles/r02_ta_microbench.txt): one
# lane request per 0.69 cycles unless the four lanes of a quad agree -> 16 B / 0.69 clk x 256 CUs x 2.4 GHz
GATHER_CEILING_GBS = 256 * 16 / 0.69 * 2.4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=708, help="G of grid_mesh (708 -> 1,002,528 triangles)")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=1)
    ap.add_argument("--camera", choices=["a", "b"], default="a")
    ap.add_argument("--render-type", type=int, default=0, choices=[0, 1, 2],
                    help="kDepth (headline), kBoxtests, kTriangleTests -- the render types that need no materials")
    ap.add_argument("--build-reps", type=int, default=10)
    ap.add_argument("--type", choices=["bottom-up", "bottom-up-pairs", "hybrid", "sah", "sah-pairs"], default="bottom-up",
                    help="tree the rays are traced through: the LBVH of the headline metric (default) or the SAH tree "
                         "(rt_run_sah_build, the reference's default --type; reported as a separate workload)")
    ap.add_argument("--inflight", type=int, default=8,
                    help="frames in flight: step i is launched on HIP stream i mod INFLIGHT into its own frame buffer, so "
                         "the next frame's waves fill the CUs the previous frame's last waves leave idle (1 = serial)")
    ap.add_argument("--partition", choices=["auto", "bands", "strips"], default="auto",
                    help="N > 1: contiguous row bands, interleaved 8-row strips, or (auto) strips when the measured band "
                         "costs have max/mean > 1.15 (SURVEY 8(e))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the build timing")
    ap.add_argument("--other-camera", action="store_true",
                    help="also time the other camera (off by default so that every trace_kernel launch of the default "
                         "command is the headline workload and rocprof's per-kernel average matches the line)")
    ap.add_argument("--preset", choices=["config2", "config4", "config5"], default=None,
                    help="BASELINE.json configs: config2 = 1M tris 1080p (default); config4 = 10M-triangle scene "
                         "(full LBVH rebuild, builder-bound); config5 = 1M tris, 3840x2160, 16 spp (traversal-bound)")
    a = ap.parse_args()
    if a.preset == "config4":
        a.grid = 2237
    elif a.preset == "config5":
        a.width, a.height, a.spp = 3840, 2160, 16
        a.steps = min(a.steps, 10)
    return a


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N rank processes (this process has made no GPU call and
    imports no torch), give rank 0 our stdout, wait for all of them.  Children are killed by exact PID if one fails."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RT_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # only rank 0 prints the JSON line; its stdout is filtered below (communication libraries may chat on stdout)
        out = subprocess.PIPE if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out,
                                      text=(r == 0)))
    import threading

    def relay():
        for line in procs[0].stdout:
            (sys.stdout if line.startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    th.join(timeout=10)
    return rc


# ------------------------------------------------------------------------------------------------ one rank
def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    ndev = torch.cuda.device_count()          # (does not initialise the GPU)
    if ndev == 0:
        print("bench.py: no GPU visible", file=sys.stderr)
        sys.exit(2)
    # one rank per GPU over RCCL.  With fewer GPUs than ranks (the one-GPU rehearsal box) the ranks share cards and the
    # collectives go through gloo with host staging: same control flow, says so in the line, not a scaling measurement.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl" if ndev >= world else "gloo")
    dev = local_rank % ndev
    torch.cuda.set_device(dev)
    use_dist = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST") == "1"   # FORCE: 1-rank RCCL init + collectives (API check)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # (single-rank RCCL check: any free port)
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    ctl_dev = "cuda" if backend == "nccl" else "cpu"    # where the small control tensors of the collectives live

    rt = importlib.import_module("gpu-raytracing_amd")
    scenes = importlib.import_module("gpu-raytracing_amd.scenes")
    sharding = importlib.import_module("gpu-raytracing_amd.sharding")
    rt.lib()  # fail loudly if the HIP library is missing

    G, W, H = args.grid, args.width, args.height
    tris = scenes.grid_mesh(G, 1)
    n = tris.shape[0]
    sah = args.type in ("sah", "sah-pairs")
    hybrid = args.type == "hybrid"
    inp = rt.BuildInput.allocate(tris, sah=sah)
    sah_args = rt.Arguments(build_type=rt.kSAH, enable_pairs=args.type == "sah-pairs")
    ROOT_IDX, ROOT_CNT = (0, 1) if sah else ((2 * n + 1, 2) if hybrid else (0, 2))      # main.cu:222-223

    def build():
        if sah:
            rt.RunSahBuild(inp, sah_args)
        elif args.type == "bottom-up-pairs":
            rt.RunBottomUpBuild(inp, rt.Arguments(build_type=rt.kBottomUp, enable_pairs=True))
        else:
            rt.RunBottomUpBuild(inp, hybrid=hybrid)

    def ev():
        return torch.cuda.Event(enable_timing=True)

    # ---- build (replicated on every rank); timed with events on the launch stream, scratch preallocated
    build()
    torch.cuda.synchronize()
    build_ms = None
    if not args.no_extras:
        times = []
        for _ in range(args.build_reps):
            e0, e1 = ev(), ev()
            e0.record()
            build()
            e1.record()
            e1.synchronize()
            times.append(e0.elapsed_time(e1))
        build_ms = statistics.median(times)
        # the same builds queued back to back (events around each, one synchronisation at the end): what a per-frame rebuild
        # loop sees, and what the per-kernel sum of a rocprofv3 trace adds up to -- an isolated build also pays the GPU's
        # start from idle
        evs = [(ev(), ev()) for _ in range(args.build_reps)]
        for a_, b_ in evs:
            a_.record()
            build()
            b_.record()
        torch.cuda.synchronize()
        build_b2b_ms = statistics.median(a_.elapsed_time(b_) for a_, b_ in evs[1:]) if len(evs) > 1 else None

    # ---- the build's radix sort on its own (the stage entry points of the C ABI): the scene's Morton codes, sorted on 30
    # bits exactly as the builder does; HIP events around the sort alone, input restored (untimed) before every run
    sort_us = None
    if not args.no_extras and args.type == "bottom-up" and world == 1 and n > 0:
        import numpy as np  # noqa: PLC0415
        d_aabb = torch.zeros(8, dtype=torch.int32, device="cuda")
        k0 = torch.empty(n, dtype=torch.int32, device="cuda")
        v0 = torch.empty_like(k0)
        rt.CalculateSceneAabb(inp.triangles_in, n, d_aabb)
        rt.GenerateMortonCodes(k0, v0, inp.triangles_in, d_aabb, n)
        kk, vv, tk, tv = (torch.empty_like(k0) for _ in range(4))
        scr = rt.device_bytes(rt.RadixSortScratchBytes(n))
        in_tmp = bool(rt.lib().rt_radix_sort_input_in_tmp(n, 30))
        ts = []
        for it in range(13):
            (tk if in_tmp else kk).copy_(k0)
            (tv if in_tmp else vv).copy_(v0)
            e0, e1 = ev(), ev()
            e0.record()
            rt.RadixSortBits(kk, vv, tk, tv, n, 30, input_in_tmp=in_tmp, sort_scratch=scr)
            e1.record()
            e1.synchronize()
            if it >= 3:
                ts.append(e0.elapsed_time(e1) * 1e3)
        sort_us = statistics.median(ts)
        assert bool((kk[1:] >= kk[:-1]).all()), "sorted Morton codes are not ascending"
        del k0, v0, kk, vv, tk, tv, scr

    cams = {"a": scenes.camera_a(G), "b": scenes.camera_b(G)}
    cam_dev = {k: rt.to_device(v) for k, v in cams.items()}
    cam = args.camera
    S = max(1, args.inflight)
    y0, y1 = sharding.my_band(H, world, rank)

    def trace_band(frame, cam_key, render_type, counters=None, rows=None):
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (W, H), cam_dev[cam_key], ROOT_IDX, ROOT_CNT,
                 render_type=render_type, counters=counters, rows=rows if rows is not None else (y0, y1), spp=args.spp,
                 num_primitives=n)

    def trace_strips(compact, cam_key, render_type, counters=None):
        rt.Trace(inp.triangles_out, inp.nodes_out, compact, (W, H), cam_dev[cam_key], ROOT_IDX, ROOT_CNT,
                 render_type=render_type, counters=counters, spp=args.spp, strips=(sharding.STRIP_ROWS, rank, world),
                 num_primitives=n)

    # ---- partition (N > 1): measure every rank's band (serial launches, events), share the costs, choose
    scratch_frame = torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda")
    partition, band_costs = "bands", None
    if world > 1:
        ts = []
        for _ in range(5):
            e0, e1 = ev(), ev()
            e0.record()
            trace_band(scratch_frame, cam, args.render_type)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        mine = torch.tensor([statistics.median(ts[1:])], dtype=torch.float64, device=ctl_dev)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        band_costs = [round(float(c.item()), 4) for c in allc]
        partition = sharding.choose_partition(band_costs) if args.partition == "auto" else args.partition
    elif args.partition == "strips":        # one rank, strips forced: exercises rt_trace_strips + the de-interleave
        partition = "strips"
    strips = partition == "strips"

    # INFLIGHT frame buffers, one HIP stream each: a traced frame is ~4.5 rounds of waves (one band of an 8-GPU run
    # is less than one), so a launch ends with CUs idling behind its slowest waves; with several frames in flight the
    # next frame's waves take those CUs (and with N > 1 the gather of frame i overlaps the trace of frame i+1)
    frames = [torch.zeros(H * W * 4, dtype=torch.uint8, device="cuda") for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    compact = staging = None
    if strips:
        cbytes = sharding.compact_rows(H, world) * W * 4
        compact = [torch.zeros(cbytes, dtype=torch.uint8, device="cuda") for _ in range(S)]
        staging = [torch.zeros(cbytes * world, dtype=torch.uint8, device="cuda") for _ in range(S)] if rank == 0 else [None] * S
        if not use_dist:
            staging = compact                  # nothing to gather: de-interleave straight from the compact buffer
    frame = frames[0]
    pending = [None] * S        # the collective that last used buffer k (or True when only the de-interleave is owed)
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    # the reference's Trace() resets and passes its num_tests buffer every frame (main.cu:153-157, Tracer.cu:503), and so
    # does rt_cli: every timed frame does the same (one buffer per frame in flight, cleared on the frame's stream)
    frame_counters = [torch.zeros(4, dtype=torch.int64, device="cuda") for _ in range(S)]
    step_no = [0]

    def finish(k):
        """(on stream k) the gather that last used buffer k has completed and, for strips, rank 0's frame is in image order"""
        if pending[k] is None:
            return
        if pending[k] is not True:
            pending[k].wait()
        if strips and rank == 0:
            sharding.deinterleave(staging[k], frames[k], W, H, world)
        pending[k] = None

    def step(cam_key, with_counters=False, events=None):
        k = step_no[0] % S
        step_no[0] += 1
        with torch.cuda.stream(streams[k]):
            finish(k)
            if events is not None:
                events[0].record()
            ctr = counters if with_counters else frame_counters[k]
            if not with_counters:
                ctr.zero_()
            if strips:
                trace_strips(compact[k], cam_key, args.render_type, ctr)
            else:
                trace_band(frames[k], cam_key, args.render_type, ctr)
            if events is not None:
                events[1].record()
            if use_dist:
                if strips:
                    h = sharding.gather_strips(compact[k], staging[k], world, rank, dist, async_op=True, force=True)
                else:
                    h = sharding.gather_bands(frames[k], W, H, world, rank, dist, async_op=True, force=True)
                pending[k] = h if h is not None else True
            elif strips:
                pending[k] = True

    def drain():
        for k in range(S):
            with torch.cuda.stream(streams[k]):
                finish(k)
            streams[k].synchronize()

    def timed(cam_key, steps, warmup):
        for k in range(S):
            streams[k].wait_stream(torch.cuda.current_stream())
        for _ in range(warmup):
            step(cam_key)
        drain()
        evs = [(ev(), ev()) for _ in range(steps)]
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(cam_key, events=evs[i])
        drain()                              # every gather (and de-interleave) of the K timed frames has completed
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kern_ms = statistics.mean(a.elapsed_time(b) for a, b in evs)
        return dt, kern_ms

    def test_counts(cam_key):
        counters.zero_()
        torch.cuda.synchronize()
        step(cam_key, with_counters=True)
        drain()
        torch.cuda.synchronize()
        c = counters.clone()
        if world > 1:
            c = c.to(ctl_dev)
            dist.all_reduce(c)
        return [int(v) for v in c.tolist()]

    box, tri, wsteps_box, wsteps_leaf = test_counts(cam)   # whole-frame sums (all ranks)
    dt, kern_ms = timed(cam, args.steps, args.warmup)
    rays = W * H * args.spp
    value = rays * args.steps / dt / 1e6
    gpu_frame = frames[(step_no[0] - 1) % S].cpu().numpy().reshape(H, W, 4) if rank == 0 else None   # last timed frame

    # ---- this rank's part alone on the GPU, one launch at a time (events around the trace kernel,
    # median of >= 20): the duration the kernel's own quality is judged by.  Issued as the OTHER colour conversion of the
    # same traversal (kBoxtests <-> kDepth: same rays, same tests, a different one-line colour conversion) so that a
    # profiler's per-kernel average keeps the in-flight launches of the timed region and these serial ones apart.
    counters.zero_()
    if strips:
        trace_strips(compact[0], cam, args.render_type, counters)
    else:
        trace_band(frame, cam, args.render_type, counters)
    torch.cuda.synchronize()
    lbox, ltri = int(counters[0].item()), int(counters[1].item())
    my_rows = (min(H, len(sharding.my_strips(H, world, rank)) * sharding.STRIP_ROWS) if strips else (y1 - y0))
    alg_bytes = 32 * lbox + 64 * ltri + 4 * W * my_rows
    iso_rt = 1 if args.render_type == 0 else 0
    iso = []
    for _ in range(24):
        e0, e1 = ev(), ev()
        counters.zero_()
        e0.record()
        if strips:
            trace_strips(compact[0], cam, iso_rt, counters)
        else:
            trace_band(scratch_frame, cam, iso_rt, counters)
        e1.record()
        e1.synchronize()
        iso.append(e0.elapsed_time(e1))
    serial_ms = statistics.median(iso[2:])
    serial_l1 = alg_bytes / (serial_ms * 1e-3) / 1e9       # GB/s of node + leaf bytes delivered to the lanes
    if world > 1:                                          # the slowest rank's part bounds a serial multi-GPU frame
        t = torch.tensor([serial_ms], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_ms_max = float(t.item())
    else:
        serial_ms_max = serial_ms

    # ---- N > 1 diagnostics (outside the timed region): one frame at a time -- this rank's launch, then the gather alone
    # (events on the launch stream; the gather's time is rank 0's view of the whole collective).  They tell a reader of a
    # scaling record where a step's time goes: the slowest rank's part, or the collection on rank 0.
    diag = None
    if world > 1:
        tr, ga = [], []
        for _ in range(6):
            dist.barrier()
            torch.cuda.synchronize()
            e0, e1, e2 = ev(), ev(), ev()
            e0.record()
            if strips:
                trace_strips(compact[0], cam, args.render_type)
            else:
                trace_band(frames[0], cam, args.render_type)
            e1.record()
            if strips:
                h = sharding.gather_strips(compact[0], staging[0], world, rank, dist, async_op=True)
            else:
                h = sharding.gather_bands(frames[0], W, H, world, rank, dist, async_op=True)
            if h is not None:
                h.wait()
            if strips and rank == 0:
                sharding.deinterleave(staging[0], frames[0], W, H, world)
            e2.record()
            e2.synchronize()
            tr.append(e0.elapsed_time(e1))
            ga.append(e1.elapsed_time(e2))
        mine = torch.tensor([statistics.median(tr[1:]), statistics.median(ga[1:])], dtype=torch.float64, device=ctl_dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        diag = {"serial_trace_ms_per_rank": [round(float(v[0]), 4) for v in allv],
                "serial_gather_ms_per_rank": [round(float(v[1]), 4) for v in allv],
                "is": "one frame at a time after a barrier: each rank's trace launch, then the gather (+ de-interleave on rank 0) "
                      "as seen from that rank; the timed region overlaps these across frames"}

    extras = {}
    if not args.no_extras:
        extras = {
            "build_ms": round(build_ms, 4),
            "build_gbps_algorithmic": round(512.0 * n / (build_ms * 1e-3) / 1e9, 1),  # 512 B/triangle, SURVEY 8(d)
            "build_frac_of_hbm_peak": round(512.0 * n / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "build": build_record(n, build_ms, sort_us, args.type, G, build_b2b_ms),
        }
    if not args.no_extras and args.type == "bottom-up" and world == 1:
        # the SAH builder (the reference's default --type) on the same triangles: build time only here, so that every
        # trace_kernel launch of the default command stays the headline workload; `--type sah` traces through it
        sinp = rt.BuildInput.allocate(tris, sah=True)
        rt.RunSahBuild(sinp)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = ev(), ev()
            e0.record()
            rt.RunSahBuild(sinp)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        extras["sah_build_ms"] = round(statistics.median(ts), 4)
        del sinp
    if not args.no_extras and args.type == "bottom-up" and world == 1:
        # (1) rebuild + trace per frame as one HIP graph (nothing in that path synchronises or allocates);
        # (2) an on-box stream ceiling (device-to-device copy, read + write bytes) beside the 8 TB/s spec figure
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                build()
                trace_band(frame, cam, iso_rt)
                side.synchronize()
                with torch.cuda.graph(gr, stream=side):
                    build()
                    trace_band(frame, cam, iso_rt)
            torch.cuda.current_stream().wait_stream(side)
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = ev(), ev()
            e0.record()
            for _ in range(10):
                gr.replay()
            e1.record()
            e1.synchronize()
            extras["graph_rebuild_plus_trace_ms"] = round(e0.elapsed_time(e1) / 10, 4)
        except Exception as ex:  # noqa: BLE001  (an extra, never the headline)
            extras["graph_rebuild_plus_trace_ms"] = f"unavailable: {type(ex).__name__}"
        src = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(5):
            dst.copy_(src)
        e1.record()
        e1.synchronize()
        extras["stream_copy_gbps"] = round(2 * 5 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
        del src, dst
    if args.other_camera:
        other = "b" if cam == "a" else "a"
        obox, otri, _, _ = test_counts(other)
        odt, okern = timed(other, max(args.steps // 2, 5), 2)
        extras[f"camera_{other}_mrays"] = round(rays * max(args.steps // 2, 5) / odt / 1e6, 2)
        extras[f"camera_{other}_box_tests_per_ray"] = round(obox / rays, 2)

    rc = 0
    if rank == 0:
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md "HBM"): a counter
        # pass cannot run inside this process, so the figure is the committed result of `tools/collect.sh run <round> pmc` on this same
        # command; it is attached only to the exact workload it was collected on and labelled with its source.
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "trace_traffic.json" if G == 708 else "trace_traffic_10m.json")
        if os.path.exists(tpath) and world == 1 and cam == "a" and (W, H, args.spp) == (1920, 1080, 1) and G in (708, 2237) and args.type == "bottom-up":
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{os.path.basename(tpath)} ({tj.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command')})"
            except Exception:
                traffic = None
        # utilisation of the per-CU vector-memory address path (TA) and the L1 hit rate, from the committed PMC pass of this
        # command (tools/collect.sh run <round> pmc -> tools/pmc_l1_json.py): attached to the exact workload only
        l1_pmc = None
        lpath = os.path.join(ROOT, "profiles", "trace_l1_pmc.json" if G == 708 else "trace_l1_pmc_10m.json")
        if os.path.exists(lpath) and world == 1 and cam == "a" and (W, H, args.spp) == (1920, 1080, 1) and G in (708, 2237) and args.type == "bottom-up":
            try:
                l1_pmc = json.load(open(lpath))
            except Exception:
                l1_pmc = None
        roof = {
            # what bounds trace_kernel: the per-CU vector-memory (TA -> L1) path, not HBM (the 128 MB BVH is cache
            # resident).  achieved = algorithmic node + leaf + frame bytes of one launch / the launch's duration ALONE
            # on the GPU; peak = 256 CUs x 64 B/clk x 2.4 GHz of L1 data path.
            "bound": "l1", "kernel": "trace_kernel", "achieved": round(serial_l1, 1), "peak": round(L1_PEAK_GBS, 1),
            "unit": "GB/s", "frac": round(serial_l1 / L1_PEAK_GBS, 4),
            "peak_formula": "256 CUs x 64 B/clk x 2.4 GHz (vector L1 data path) -- a NOMINAL figure: MI355X_MICROARCH.md does not "
                            "state it; the measured ceilings of this access pattern are in measured_ceilings",
            "reference_rates": {"fully_divergent_16B_gather_GBps": round(GATHER_CEILING_GBS, 1),
                                "fully_divergent_16B_gather_is": "tools/ta_microbench.hip, profiles/r02_ta_microbench.txt: 0.69 clk per lane request when "
                                                                 "NO two lanes of a quad agree.  A rate, NOT an upper bound: lanes of a quad that visit the "
                                                                 "same node share one request (config 5 and camera-coherent frames exceed it)",
                                "achieved_over_fully_divergent": round(serial_l1 / GATHER_CEILING_GBS, 4),
                                "l2_served_row_gather_guide_GBps": [16800, 18800],
                                "l2_served_row_gather_source": "MI355X_MICROARCH.md, 'Indexed rows: gather into LDS' (2,048 rows shared by every workgroup)",
                                "achieved_over_guide_l2_gather_low": round(serial_l1 / 16800.0, 4)},
            "address_path_utilisation": l1_pmc,
            "algorithmic_bytes_per_launch": alg_bytes, "formula": "32*sum(box_tests) + 64*sum(tri_tests) + 4*W*rows",
            "kernel_ms": round(serial_ms, 4),
            "kernel_ms_is": f"median of 22 serial launches of this rank's part, HIP events on the launch stream "
                            f"(trace_kernel<{iso_rt}>: same rays and tests as the timed region's trace_kernel<{args.render_type}>)",
            "algorithmic_over_hbm_peak": round(serial_l1 / HBM_PEAK_GBS, 4),
            "algorithmic_over_hbm_peak_is": "> 1 means cache reuse: these bytes cannot all have come from HBM",
            "traffic": traffic, "traffic_source": traffic_src,
            "hbm_counter_frac": (round(traffic / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
            "north_star_hbm_target": ("not meetable as worded: HBM carries a few % of its peak during traversal because the "
                                      "whole BVH stays in L2 / Infinity Cache; the kernel is limited by L1 address rate "
                                      "(address_path_utilisation: profiles/trace_l1_pmc.json; profiles/r02_ta_microbench.txt)" if G <= 1000 else
                                      "the one regime where HBM can matter (BVH 1.28 GB > 256 MiB Infinity Cache): see hbm_counter_frac "
                                      "and profiles/r04_trace_pmc_10m.txt"),
            "inflight_mean_launch_ms": round(kern_ms, 4),
            "inflight_mean_launch_ms_is": f"mean start-to-end time of the timed region's launches, {S} of which overlap",
        }
        out = {
            "metric": "Mrays/s at 1920x1080 primary rays + BVH build ms, 1M-tri scene",
            "value": round(value, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "value_is": f"K frames / wall time with {S} frame(s) in flight on {S} HIP stream(s)"
                        + ("" if S > 1 else " (= one frame at a time)"),
            "inflight": S,
            "serial_mrays": round(rays / (serial_ms_max * 1e-3) / 1e6, 2),
            "serial_ms_per_frame": round(serial_ms_max, 4),
            "counters_in_timed_region": True,
            "counters_is": "every timed and every serial launch resets and passes the num_tests buffer, as the reference's Trace() does "
                           "(main.cu:153-157, Tracer.cu:503)",
            "serial_is": "one launch at a time (the reference's frame loop): median of 22 event-timed launches"
                         + ("; slowest rank's part, gather not included" if world > 1 else ""),
            "config": {"workload": f"grid_mesh(G={G}, seed=1) = {n} triangles, {W}x{H}, {args.spp} spp, camera "
                                   f"{cam.upper()} ({'top-down' if cam == 'a' else 'oblique'}), render_type {args.render_type}; "
                                   + ({"bottom-up": "LBVH", "bottom-up-pairs": "LBVH with triangle pairs", "hybrid": "LBVH + SAH top tree (hybrid)", "sah": "SAH tree",
                                      "sah-pairs": "SAH tree with triangle pairs"}[args.type])
                                   + " replicated per GPU",
                       "parallelism": (f"row-bands x{world}" if not strips else f"interleaved {sharding.STRIP_ROWS}-row strips x{world}")
                                      + f", {S} frames in flight on {S} HIP streams"
                                      + (f" + one {'RCCL' if backend == 'nccl' else backend + ' (host-staged)'} gather to rank 0 per frame" if world > 1 else "")},
            "ranks": {"world_size": (dist.get_world_size() if use_dist else 1), "backend": backend if use_dist else None,
                      "gpus_visible": ndev, "self_launched": os.environ.get("RT_BENCH_SELF_LAUNCHED") == "1",
                      "partition": partition if world > 1 else None, "band_costs_ms": band_costs, "diagnostics": diag,
                      "note": (None if backend == "nccl" or world == 1 else
                               f"REHEARSAL: {world} ranks share {ndev} GPU(s), collectives over gloo with host staging -- "
                               "exercises the N > 1 control flow, not a scaling measurement")},
            "box_tests_per_ray": round(box / rays, 2), "tri_tests_per_ray": round(tri / rays, 3),
            "wave_steps": {"box_phase": wsteps_box, "leaf_phase": wsteps_leaf,
                           "lane_utilisation_box_phase": round(box / 2 / max(wsteps_box, 1) / 64, 3)},
            "roofline": roof,
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ok = cpu_baseline(tris, cams[cam], W, H, args.spp, args.render_type, G, args.type,
                                                   gpu_frame, (box, tri))
            if not ok:
                rc = 3
        print(json.dumps(out), flush=True)
        if rc:
            print("bench.py: the GPU frame / test counts differ from the oracle's (cpu_baseline.parity)", file=sys.stderr)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def build_record(n, build_ms, sort_us, tree, G, b2b_ms=None):
    """The `build` object of the line: the build half of the metric against ITS roofline (HBM, 512 B/triangle),
    the radix sort against its own 80 B/key formula (timed live in this run through the stage entry points), and the
    per-kernel times of the same build from the committed rocprofv3 --kernel-trace --stats summary (profiles/)."""
    rec = {"bound": "hbm", "ms": round(build_ms, 4), "algorithmic_bytes": 512 * n, "bytes_per_triangle": 512,
           "achieved": round(512.0 * n / (build_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(512.0 * n / (build_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "ms_is": "median of --build-reps builds, HIP events around all launches of rt_run_bottom_up_build / rt_run_sah_build, "
                    "scratch preallocated, triangles resident"}
    if b2b_ms:
        rec["ms_back_to_back"] = round(b2b_ms, 4)
        rec["frac_back_to_back"] = round(512.0 * n / (b2b_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        rec["ms_back_to_back_is"] = "median of the same builds queued without a synchronisation in between (events around each build)"
    if sort_us:
        rec["sort"] = {"us": round(sort_us, 1), "keys": n, "key_bits": 30, "bytes_per_key": 80,
                       "achieved": round(80.0 * n / (sort_us * 1e-6) / 1e9, 1), "unit": "GB/s",
                       "frac": round(80.0 * n / (sort_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                       "us_is": "median of 10 runs of rt_radix_sort_u32_pairs_bits(30) on this scene's Morton codes, events around the sort alone",
                       "yardstick": "rocprim::radix_sort_pairs on the same keys: profiles/r04_sort_yardstick.txt (tools/sort_yardstick.hip)"}
    stats = os.path.join(ROOT, "profiles", {708: "r04_build1m_kernel_stats.txt", 2237: "r04_build10m_kernel_stats.txt"}.get(G, ""))
    if tree == "bottom-up" and os.path.isfile(stats):
        kern = {}
        for line in open(stats):
            t = line.split()
            if "calls" in t and "avg" in t:
                kern[" ".join(t[:t.index("calls")])] = float(t[t.index("avg") + 1])
        rec["kernels_avg_us"] = kern
        rec["kernels_source"] = f"profiles/{os.path.basename(stats)} (rocprofv3 --kernel-trace --stats of tools/build_loop.py on this scene)"
    return rec


def cpu_baseline(tris, cam, W, H, spp, render_type, G, tree, gpu_frame, gpu_counts):
    """The oracle (C port of the reference algorithm, OpenMP) timed on this box's host cores on a bounded sample of the
    same workload (one full frame at 1 spp; a band of rows sized for ~20 s otherwise).  Reported baseline only.  The
    oracle's rows are also compared with the GPU frame of the timed region (and the whole-frame test counts when the
    sample is the whole frame): returns (record, parity_ok)."""
    import numpy as np
    from oracle import oracle_py as ora
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # cgroup v2 CPU quota (the GPU box gives each job a 16-CPU share of a 256-thread host)
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, -(-int(quota) // int(period))))
    except Exception:
        pass
    def build_tree():
        if tree == "bottom-up":
            return ora.build_bvh(tris)
        if tree == "bottom-up-pairs":
            return ora.build_pairs(tris)
        if tree == "hybrid":
            return ora.build_hybrid(tris)
        return ora.build_sah(tris, tree == "sah-pairs")
    # the oracle's builders are OpenMP-parallel: timed with all `cores` threads and with one
    ora.set_threads(1)
    t0 = time.perf_counter()
    o = build_tree()
    t_build_one = time.perf_counter() - t0
    ora.set_threads(cores)
    t0 = time.perf_counter()
    o = build_tree()
    t_build = time.perf_counter() - t0
    root, count = o.get("root", 0), o.get("count", 2)
    # bounded sample: the whole frame when it is ~2 M rays; otherwise rows from the middle of the frame worth ~2 M rays
    full = W * H * spp <= 4_000_000
    r0, r1 = (0, H) if full else (H // 2 - max(8, 2_000_000 // (W * spp)) // 2, H // 2 + max(8, 2_000_000 // (W * spp)) // 2)
    t0 = time.perf_counter()
    img, oc = ora.trace(o["leaves"], o["nodes"], root, count, cam, W, H, render_type=render_type, spp=spp, rows=(r0, r1))
    t_trace = time.perf_counter() - t0
    frame_equal = bool((img[r0:r1] == gpu_frame[r0:r1]).all())
    counts_equal = (int(oc[0]) == gpu_counts[0] and int(oc[1]) == gpu_counts[1]) if full else None
    # the same port on ONE thread, on a bounded sample of the frame (every 16th row band of 8 rows)
    ora.set_threads(1)
    rows1 = 0
    t0 = time.perf_counter()
    for yb in range(0, H - 7, 128 * max(1, spp)):
        ora.trace(o["leaves"], o["nodes"], root, count, cam, W, H, render_type=render_type, spp=spp, rows=(yb, yb + 8))
        rows1 += 8
    t_one = time.perf_counter() - t0
    ora.set_threads(cores)
    names = {"bottom-up": "LBVH", "bottom-up-pairs": "LBVH with pairs", "hybrid": "LBVH + SAH top tree"}
    rec = {"value": round(W * (r1 - r0) * spp / t_trace / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
           "one_thread_mrays": round(W * rows1 * spp / t_one / 1e6, 3),
           "sample": (f"1 full {W}x{H} frame" if full else f"rows [{r0}, {r1}) of the {W}x{H} frame") +
                     f" ({spp} spp) of the same scene and camera, oracle/liboracle.so (-O2 -ffp-contract=off, OpenMP over "
                     f"rows); {names.get(tree, 'SAH')} build of the same {tris.shape[0]} triangles with {cores} OpenMP threads "
                     f"(build_ms) and with one (build_ms_one_thread)",
           "build_ms": round(t_build * 1e3, 1), "build_threads": cores, "build_ms_one_thread": round(t_build_one * 1e3, 1),
           "trace_s": round(t_trace, 3),
           "parity": {"gpu_frame_rows_equal_oracle": frame_equal, "rows": [r0, r1],
                      "sum_box_tri_tests_equal_oracle": counts_equal,
                      "tolerance": "byte-exact (every render type: the transcendental calls of the shaded modes are csrc/rt_math.h on both sides)"}}
    ok = frame_equal and counts_equal is not False
    return rec, ok


if __name__ == "__main__":
    main()
