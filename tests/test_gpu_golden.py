"""GPU tests against the committed golden fixtures (tests/golden) and of the C++ host path (rt_cli) end to end."""
import hashlib
import importlib
import json
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(GOLD, "golden.json")))


@pytest.fixture(scope="module")
def fixtures():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fixture_scenes()


@pytest.mark.parametrize("name", ["cornell34", "grid24", "soup2048", "flat12"])
def test_gpu_matches_golden(name, golden, fixtures):
    from helpers import gpu_build, gpu_trace
    tris, cam, w, h, at, mats, light = fixtures[name]
    g = golden[name]
    b = gpu_build(tris)
    assert [int(x) for x in b["aabb"]] == g["aabb_ordered"]
    assert sha(b["codes"]) == g["codes_sha256"] and sha(b["indices"]) == g["indices_sha256"]
    assert sha(b["nodes"]) == g["nodes_sha256"], "Node[] bytes"
    assert sha(b["leaves"]) == g["leaves_sha256"], "TrianglePair[] bytes"
    for rtype, fr in g["frames"].items():
        img, cnt = gpu_trace(b, cam, w, h, int(rtype), attributes=at, materials=mats, light=light)
        assert (int(cnt[0]), int(cnt[1])) == (fr["box_tests"], fr["tri_tests"]), f"render {rtype} counters"
        assert sha(img) == fr["sha256"], f"render type {rtype} frame bytes"
    if name == "cornell34":
        exp = np.load(os.path.join(GOLD, "cornell34_frame_r5.npz"))["rgba"]
        img, _ = gpu_trace(b, cam, w, h, 5, attributes=at, materials=mats, light=light)
        assert (img == exp).all()     # kDiffuse: byte-exact (rt_math.h pow on both sides)
        assert (np.load(os.path.join(GOLD, "cornell34_frame_r0.npz"))["rgba"] == gpu_trace(b, cam, w, h, 0)[0]).all()


@pytest.mark.parametrize("build_type", ["bottom-up", "hybrid", "sah"])
def test_rt_cli_end_to_end(tmp_path, rt, ora, build_type):
    """The C++ host path the reference's main() takes: LoadOBJFromFile -> InitialiseCamera -> RunBottomUpBuild ->
    CountNodes / VerifyHierarchy -> Trace, through gpu-raytracing_amd/host/rt_cli, checked against the oracle."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    out = str(tmp_path / "f.ppm")
    obj = os.path.join(GOLD, "cornell34.obj")
    p = subprocess.run([cli, obj, "--type", build_type, "--render", "diffuse", "--width", "320", "--height", "200",
                        "--pos", "5", "5", "-5.25", "--yaw", "0", "--pitch", "0", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    if build_type == "bottom-up":
        assert "num nodes: 66" in p.stdout and "num tree nodes: 32" in p.stdout
    assert "num leaf nodes: 34" in p.stdout
    assert "Invalid hierarchy" not in p.stderr
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [5, 5, -5.25], 0, 0
    cam = host.UpdateCamera(cam)
    if build_type == "sah":
        assert "num nodes: 67" in p.stdout and "RunSahBuild time elapsed" in p.stdout
    o = {"bottom-up": ora.build_bvh, "hybrid": ora.build_hybrid, "sah": ora.build_sah}[build_type](s["triangles"])
    exp, cnt = ora.trace(o["leaves"], o["nodes"], o.get("root", 0), o.get("count", 2), cam, 320, 200, render_type=5, attributes=s["attributes"],
                         materials=s["materials"], light=tuple(s["light"]))
    assert int(re.search(r"TraceRays number of tests (\d+)", p.stdout).group(1)) == int(cnt[0])
    raw = open(out, "rb").read()
    assert raw.startswith(b"P6\n320 200\n255\n")
    got = np.frombuffer(raw[len(b"P6\n320 200\n255\n"):], np.uint8).reshape(200, 320, 3)
    assert (got == exp[..., :3]).all()
    assert (got.max(axis=2) > 0).mean() > 0.5


@pytest.mark.parametrize("render,rtype", [("texture", 6), ("texturelit", 7), ("shadows", 8), ("lods", 4)])
def test_rt_cli_textured_scene(tmp_path, rt, ora, render, rtype):
    """tiles.obj (map_Kd + bump PPM textures) end to end through the C++ host path: the .mtl loader decodes and mip-maps
    the textures, rt_cli uploads the rt_texture table, the textured render types run; frame against the oracle fed by
    the same loader output, byte for byte."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    out = str(tmp_path / "f.ppm")
    obj = os.path.join(GOLD, "tiles", "tiles.obj")
    p = subprocess.run([cli, obj, "--type", "bottom-up", "--render", render, "--width", "320", "--height", "200",
                        "--pos", "-1", "5", "-2", "--yaw", "-0.75", "--pitch", "0.5", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "num leaf nodes: 76" in p.stdout
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [-1, 5, -2], -0.75, 0.5
    cam = host.UpdateCamera(cam)
    o = ora.build_bvh(s["triangles"])
    exp, cnt = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, 320, 200, render_type=rtype, attributes=s["attributes"],
                         materials=s["materials"], light=tuple(s["light"]), textures=s["textures"])
    assert int(re.search(r"TraceRays number of tests (\d+)", p.stdout).group(1)) == int(cnt[0])
    raw = open(out, "rb").read()
    got = np.frombuffer(raw[len(b"P6\n320 200\n255\n"):], np.uint8).reshape(200, 320, 3)
    assert (got == exp[..., :3]).all(), np.abs(got.astype(int) - exp[..., :3].astype(int)).max()
    assert len(np.unique(got.reshape(-1, 3), axis=0)) >= (3 if rtype == 4 else 100)


def test_rt_cli_sah_pairs_splits(tmp_path, rt, ora):
    """rt_cli --type sah --pairs --splits on the textured fixture: flags reach RunSahBuild, the frame equals the oracle's."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    out = str(tmp_path / "f.ppm")
    obj = os.path.join(GOLD, "tiles", "tiles.obj")
    p = subprocess.run([cli, obj, "--type", "sah", "--pairs", "--splits", "--render", "depth", "--width", "320", "--height", "200",
                        "--pos", "-1", "5", "-2", "--yaw", "-0.75", "--pitch", "0.5", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert "Pairs: true" in p.stdout and "Splits: true" in p.stdout and "Invalid hierarchy" not in p.stderr
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [-1, 5, -2], -0.75, 0.5
    cam = host.UpdateCamera(cam)
    o = ora.build_sah(s["triangles"], True, True)
    assert f"num leaf nodes: {o['L']}" in p.stdout
    exp, cnt = ora.trace(o["leaves"], o["nodes"], 0, 1, cam, 320, 200, render_type=0)
    assert int(re.search(r"TraceRays number of tests (\d+)", p.stdout).group(1)) == int(cnt[0])
    raw = open(out, "rb").read()
    got = np.frombuffer(raw[len(b"P6\n320 200\n255\n"):], np.uint8).reshape(200, 320, 3)
    assert (got == exp[..., :3]).all()


def _replay_path(host, cam, events, frames):
    """The camera of every frame of `rt_cli --path`: the same host library calls (librt_host.so, pinned byte for byte
    against the compiled reference Camera.cu in test_host_mirror.py) in the order main.cpp / the GLUT callbacks apply them."""
    import ctypes
    H = ctypes.CDLL(os.path.join(ROOT, "gpu-raytracing_amd", "host", "librt_host.so"))
    H.rth_camera_look.argtypes = [ctypes.c_void_p, ctypes.c_float, ctypes.c_float]
    H.rth_camera_zoom.argtypes = [ctypes.c_void_p, ctypes.c_int]
    H.rth_camera_move.argtypes = [ctypes.c_void_p, ctypes.c_uint]
    H.rth_update_camera.argtypes = [ctypes.c_void_p]
    cam = np.ascontiguousarray(cam).copy()
    p = cam.ctypes.data_as(ctypes.c_void_p)
    bits = {"w": 1, "a": 2, "s": 4, "d": 8, "q": 16, "e": 32, "_": 64}
    out, render_steps = [], 0
    renders = []
    for f in range(frames):
        ev, keys, i = events[f % len(events)], 0, 0
        while i < len(ev):
            c = ev[i]
            if c in bits:
                keys |= bits[c]
            elif c == "m":
                render_steps += 1
            elif c == "z":
                H.rth_camera_zoom(p, 1 if ev[i + 1] == "i" else -1)
                i += 1
            elif c == "l":
                m = re.match(r"l(-?\d+):(-?\d+)", ev[i:])
                H.rth_camera_look(p, float(m.group(1)), float(m.group(2)))
                H.rth_update_camera(p)
                i += len(m.group(0)) - 1
            i += 1
        H.rth_camera_move(p, keys)
        out.append(cam.copy())
        renders.append(render_steps)
    return out, renders


def test_rt_cli_frame_loop_with_input_path(tmp_path, rt, ora):
    """The reference's Display() loop (main.cu:215-292), headless: build at frame 0, then per frame the input callbacks,
    UpdateCameraPosition, Trace, present.  8 frames over the cornell fixture driven by keys, mouse drags, the wheel and
    the 'm' render-type key; every frame's PPM and printed sum of box tests against the oracle on the replayed camera."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    obj = os.path.join(GOLD, "cornell34.obj")
    out = str(tmp_path / "walk.ppm")
    path = "-,w,wd,l25:-10,zi,m,a_l-40:5,zoe"
    events, K, W, H = path.split(","), 8, 256, 160
    p = subprocess.run([cli, obj, "--type", "bottom-up", "--render", "depth", "--width", str(W), "--height", str(H),
                        "--pos", "5", "5", "-5.25", "--yaw", "0", "--pitch", "0", "--frames", str(K), "--path", path,
                        "--out", out], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [5, 5, -5.25], 0, 0
    cam = host.UpdateCamera(cam)
    cams, renders = _replay_path(host, cam, events, K)
    o = ora.build_bvh(s["triangles"])
    lines = re.findall(r"frame (\d+): TraceRays \S+ms  box tests (\d+)  triangle tests (\d+) .* render (\d+)  pos", p.stdout)
    assert len(lines) == K
    seen = set()
    for f in range(K):
        rtype = renders[f] % 9
        assert int(lines[f][3]) == rtype
        exp, cnt = ora.trace(o["leaves"], o["nodes"], 0, 2, cams[f], W, H, render_type=rtype, attributes=s["attributes"],
                             materials=s["materials"], light=tuple(s["light"]))
        assert (int(lines[f][1]), int(lines[f][2])) == (int(cnt[0]), int(cnt[1])), f"frame {f} test counts"
        raw = open(str(tmp_path / f"walk_{f:04d}.ppm"), "rb").read()
        hdr = f"P6\n{W} {H}\n255\n".encode()
        got = np.frombuffer(raw[len(hdr):], np.uint8).reshape(H, W, 3)
        assert (got == exp[..., :3]).all(), f"frame {f}"
        seen.add(got.tobytes())
    assert len(seen) >= 6, "the camera path changes the picture"
    assert f"{K} frames: mean TraceRays" in p.stdout


def test_rt_cli_rebuild_every_frame(tmp_path, rt, ora):
    """--rebuild: the LBVH is rebuilt before every frame of the loop (asynchronous launches behind the trace of the
    previous frame); pictures and counters must not change."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    obj = os.path.join(GOLD, "tiles", "tiles.obj")
    res = []
    for extra in ([], ["--rebuild"]):
        p = subprocess.run([cli, obj, "--type", "bottom-up", "--render", "boxtests", "--width", "192", "--height", "128",
                            "--frames", "4", "--path", "w,l10:3"] + extra, capture_output=True, text=True, timeout=120)
        assert p.returncode == 0, p.stderr
        res.append(re.findall(r"frame (\d+): TraceRays \S+ms  box tests (\d+)  triangle tests (\d+)", p.stdout))
    assert len(res[0]) == 4 and res[0] == res[1]


@pytest.mark.parametrize("partition", ["bands", "strips", "auto"])
@pytest.mark.parametrize("build_type", ["bottom-up", "sah"])
def test_rt_cli_multi_gpu_host_path(tmp_path, rt, ora, partition, build_type):
    """host/MultiGpu.cpp, the one-process multi-device Trace() (replicated build, one band or strip set per device, grouped RCCL
    send/recv into device 0, ncclReduce of the test counters): `rt_cli --gpus 1` builds a ONE-device RCCL communicator and
    takes exactly that path.  The frame and the counters must equal the single-device path's (and the oracle's).  More than
    one device is unmeasured in this environment (one GPU per box): the partition arithmetic for 1 - 8 devices is unit-tested
    on the CPU (tests/test_host_mirror.py)."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    obj = os.path.join(GOLD, "cornell34.obj")
    outs = {}
    for tag, extra in (("single", []), ("multi", ["--gpus", "1", "--partition", partition])):
        out = str(tmp_path / f"{tag}.ppm")
        p = subprocess.run([cli, obj, "--type", build_type, "--render", "diffuse", "--width", "322", "--height", "203",
                            "--pos", "5", "5", "-5.25", "--yaw", "0", "--pitch", "0", "--out", out, "--frames", "1"] + extra,
                           capture_output=True, text=True, timeout=180)
        assert p.returncode == 0, p.stderr + p.stdout
        assert "Invalid hierarchy" not in p.stderr
        outs[tag] = (open(out, "rb").read(), int(re.search(r"TraceRays number of tests (\d+)", p.stdout).group(1)), p.stdout)
    assert outs["multi"][0] == outs["single"][0], "frame of the multi-device path differs from the single-device path"
    assert outs["multi"][1] == outs["single"][1], "summed box tests differ"
    assert ("strips" if partition == "strips" else "bands") in outs["multi"][2]
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [5, 5, -5.25], 0, 0
    cam = host.UpdateCamera(cam)
    o = {"bottom-up": ora.build_bvh, "sah": ora.build_sah}[build_type](s["triangles"])
    exp, cnt = ora.trace(o["leaves"], o["nodes"], o.get("root", 0), o.get("count", 2), cam, 322, 203, render_type=5,
                         attributes=s["attributes"], materials=s["materials"], light=tuple(s["light"]))
    assert outs["multi"][1] == int(cnt[0])
    hdr = b"P6\n322 203\n255\n"
    got = np.frombuffer(outs["multi"][0][len(hdr):], np.uint8).reshape(203, 322, 3)
    assert (got == exp[..., :3]).all()


@pytest.mark.parametrize("partition,build_type", [("bands", "bottom-up"), ("strips", "sah"), ("auto", "bottom-up")])
def test_rt_cli_multi_gpu_frames_in_flight(tmp_path, rt, ora, partition, build_type):
    """`rt_cli --gpus 1 --inflight 4`: host/MultiGpu.cpp with four slots (streams, frame / compact / staging buffers,
    counters and an RCCL communicator set per slot), frame f enqueued into slot f mod 4 and taken just before the slot is
    reused.  Ten frames of a static camera: every frame's summed test counters and the last frame's pixels equal the
    single-device path's and the oracle's."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    obj = os.path.join(GOLD, "cornell34.obj")
    common = [cli, obj, "--type", build_type, "--render", "diffuse", "--width", "322", "--height", "203",
              "--pos", "5", "5", "-5.25", "--yaw", "0", "--pitch", "0"]
    out = str(tmp_path / "inflight.ppm")
    p = subprocess.run(common + ["--gpus", "1", "--partition", partition, "--inflight", "4", "--frames", "10", "--out", out],
                       capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr + p.stdout
    frames = re.findall(r"frame (\d+): (bands|strips)  \S+ ms \(enqueue to taken\)  box tests (\d+)  triangle tests (\d+)", p.stdout)
    assert [int(f[0]) for f in frames] == list(range(10)), p.stdout
    assert "10 frames, 4 in flight" in p.stdout
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [5, 5, -5.25], 0, 0
    cam = host.UpdateCamera(cam)
    o = {"bottom-up": ora.build_bvh, "sah": ora.build_sah}[build_type](s["triangles"])
    exp, cnt = ora.trace(o["leaves"], o["nodes"], o.get("root", 0), o.get("count", 2), cam, 322, 203, render_type=5,
                         attributes=s["attributes"], materials=s["materials"], light=tuple(s["light"]))
    for f in frames:
        assert (int(f[2]), int(f[3])) == (int(cnt[0]), int(cnt[1])), f"frame {f[0]}: counters"
        assert f[1] == ("strips" if partition == "strips" else "bands")
    hdr = b"P6\n322 203\n255\n"
    got = np.frombuffer(open(out, "rb").read()[len(hdr):], np.uint8).reshape(203, 322, 3)
    assert (got == exp[..., :3]).all()
    single = str(tmp_path / "single.ppm")
    p1 = subprocess.run(common + ["--out", single, "--frames", "1"], capture_output=True, text=True, timeout=180)
    assert p1.returncode == 0, p1.stderr
    assert open(single, "rb").read() == open(out, "rb").read()


@pytest.mark.parametrize("devices,partition,inflight,height", [(2, "bands", 1, 203), (3, "strips", 1, 203), (5, "bands", 3, 201), (8, "strips", 4, 203),
                                                                 (8, "bands", 2, 64), (7, "strips", 2, 57), (4, "auto", 2, 203)])
def test_rt_cli_virtual_devices_run_the_multi_device_pipeline(tmp_path, rt, ora, devices, partition, inflight, height):
    """`rt_cli --gpus N --virtual`: N virtual devices on the one GPU (shared scene and tree; per device and slot its own
    streams and frame / compact buffers; the gather of Partition.h::GatherPlan executed as event-ordered copies instead of
    RCCL send / recv).  The P > 1 frame pipeline of host/MultiGpu.cpp -- band offsets into device 0's frame, strip ownership,
    staging slots, the strided de-interleave and the cut last strip, slots reused with frames in flight -- on hardware:
    every frame's summed counters and the last frame's pixels equal the single-device path's and the oracle's."""
    host = importlib.import_module("gpu-raytracing_amd.host_py")
    cli = os.path.join(ROOT, "gpu-raytracing_amd", "host", "rt_cli")
    obj = os.path.join(GOLD, "cornell34.obj")
    W = 322
    out = str(tmp_path / "virt.ppm")
    p = subprocess.run([cli, obj, "--type", "bottom-up", "--render", "diffuse", "--width", str(W), "--height", str(height),
                        "--pos", "5", "5", "-5.25", "--yaw", "0", "--pitch", "0", "--gpus", str(devices), "--virtual",
                        "--partition", partition, "--inflight", str(inflight), "--frames", "9", "--out", out],
                       capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr + p.stdout
    frames = re.findall(r"frame (\d+): (bands|strips)  \S+ ms(?: \(enqueue to taken\))?  box tests (\d+)  triangle tests (\d+)", p.stdout)
    assert [int(f[0]) for f in frames] == list(range(9)), p.stdout
    s = host.LoadOBJFromFile(obj)
    cam = host.InitialiseCamera(s["aabb"])
    cam["position"], cam["yaw"], cam["pitch"] = [5, 5, -5.25], 0, 0
    cam = host.UpdateCamera(cam)
    o = ora.build_bvh(s["triangles"])
    exp, cnt = ora.trace(o["leaves"], o["nodes"], 0, 2, cam, W, height, render_type=5, attributes=s["attributes"],
                         materials=s["materials"], light=tuple(s["light"]))
    for f in frames:
        assert (int(f[2]), int(f[3])) == (int(cnt[0]), int(cnt[1])), f"frame {f[0]}: summed counters"
        if partition != "auto":
            assert f[1] == partition
    hdr = f"P6\n{W} {height}\n255\n".encode()
    got = np.frombuffer(open(out, "rb").read()[len(hdr):], np.uint8).reshape(height, W, 3)
    assert (got == exp[..., :3]).all(), f"{(got != exp[..., :3]).any(axis=2).sum()} pixels differ"
