"""CPU tests of the C++ host mirror (gpu-raytracing_amd/host): .obj/.mtl loader, camera, hierarchy checks."""
import importlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host():
    return importlib.import_module("gpu-raytracing_amd.host_py")


def test_load_cornell34(host):
    s = host.LoadOBJFromFile(os.path.join(HERE, "golden", "cornell34.obj"))
    t = s["triangles"].reshape(-1, 3, 3)
    assert t.shape[0] == 34                                         # 17 quads, fan-triangulated
    assert (s["aabb"] == [0, 0, 0, 10, 10, 10]).all()
    assert (s["light"] == np.float32([5, 9.5, 4])).all()            # light.txt overrides the box centre
    assert list(s["attributes"]["material_id"]) == [0] * 6 + [1] * 2 + [2] * 2 + [3] * 12 + [4] * 12
    assert s["materials"].shape[0] == 5
    assert np.allclose(s["materials"][1]["diffuse"], [0.65, 0.06, 0.05]) and s["materials"][1]["specular"][2] == np.float32(0.1)
    assert s["materials"][4]["specular_exp"] == 40 and s["materials"][0]["texture"] == -1
    # fan triangulation (0, i-1, i) of the floor quad and its v/vt/vn corners
    assert (t[0] == [[0, 0, 0], [10, 0, 0], [10, 0, 10]]).all() and (t[1] == [[0, 0, 0], [10, 0, 10], [0, 0, 10]]).all()
    assert (s["attributes"]["normal"][0] == [0, 1, 0]).all() and (s["attributes"]["uv"][1] == [[0, 0], [1, 1], [0, 1]]).all()
    # negative indices: the right wall lies in x = 10
    assert (t[8:10, :, 0] == 10).all()
    # faces without vn get the flat normal normalize(cross(v1-v0, v2-v1)) on every corner; without vt uv = 0
    e1, e2 = t[2, 1] - t[2, 0], t[2, 2] - t[2, 1]
    n = np.cross(e1, e2)
    n = n / np.linalg.norm(n)
    assert np.allclose(s["attributes"]["normal"][2], np.broadcast_to(n, (3, 3)), atol=1e-6)
    assert (s["attributes"]["uv"][2] == 0).all()


def test_missing_obj_raises(host):
    with pytest.raises(FileNotFoundError):
        host.LoadOBJFromFile("/nonexistent/file.obj")


def test_obj_edge_cases(host, tmp_path):
    p = tmp_path / "e.obj"
    # no trailing newline, tabs, a polygon with 5 corners, an out-of-range face that must be skipped, no usemtl
    p.write_text("v 0 0 0\nv\t1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 1.5 0\n# c\nf 1 2 3 4 5\nf 1 2 9\nf 3 2 1")
    s = host.LoadOBJFromFile(str(p))
    assert s["triangles"].shape[0] == 4
    assert (s["attributes"]["material_id"] == -1).all()
    assert np.allclose(s["light"], [0.5, 0.75, 0.0])                # no light.txt: box centre


def test_camera(host, scenes):
    cam = host.InitialiseCamera([0, 0, 0, 10, 4, 20])
    c = cam[0]
    assert (c["position"] == [5, 2, 10]).all() and c["max_depth"] == 30 and c["scale"] == 2 and c["pitch"] == 0
    assert abs(float(c["yaw"]) - np.pi / 2) < 1e-6
    assert np.allclose(c["w"], [-1, 0, 0], atol=1e-6) and np.allclose(c["v"], [0, -1, 0], atol=1e-6)
    # the numpy camera of the tests and UpdateCamera agree to rounding, and u, v, w are orthonormal
    ref = scenes.make_camera((1, 2, 3), -0.8, 0.3, 50.0)
    got = host.UpdateCamera(ref)
    for k in "uvw":
        assert np.allclose(got[0][k], ref[0][k], atol=2e-7)
    m = np.stack([got[0]["u"], got[0]["v"], got[0]["w"]])
    assert np.allclose(m @ m.T, np.eye(3), atol=1e-6)
    # pitch is clamped inside (-pi/2, pi/2)
    steep = ref.copy()
    steep["pitch"] = 3.0
    assert float(host.UpdateCamera(steep)[0]["pitch"]) < np.pi / 2


def test_count_and_verify_match_oracle_and_reference(host, ora, scenes):
    b = ora.build_bvh(scenes.grid_mesh(24, 1))
    assert host.CountNodes(b["nodes"], 0, 2) == (2302, 1152, 1150)
    assert host.VerifyHierarchy(b["nodes"], 0, 2) == 0
    bad = b["nodes"].copy()
    bad["min"][77, 2] -= 3.0
    assert host.VerifyHierarchy(bad, 0, 2) == ora.verify_hierarchy(bad, 0, 2) == 1


def test_load_textured_obj(host, ora):
    """map_Kd / bump in the .mtl -> Library::AddTexture -> PPM decode -> GenerateLODs (FileIO.cpp:121-184, 232-258)."""
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiles")
    s = host.LoadOBJFromFile(os.path.join(gold, "tiles.obj"))
    assert s["triangles"].shape[0] == 6 * 6 * 2 + 4
    assert len(s["textures"]) == 2                      # tiles_kd.ppm is shared by both materials
    m = s["materials"]
    assert (m["texture"][-2:] == 0).all() and m["bump"][-2] == -1 and m["bump"][-1] == 1 and (m["disp"] == -1).all()
    kd, bump = s["textures"]
    assert kd[0].shape == (12, 16) and bump[0].shape == (8, 8)
    assert [c.shape for c in kd] == [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    # texel (0, 0) of tiles_kd.ppm: checker cell 0 -> (50, 90, 200), alpha 255
    assert kd[0][0, 0] == (50 | 90 << 8 | 200 << 16 | 255 << 24)
    for chain in (kd, bump):
        exp = ora.generate_lods(chain[0])
        assert len(exp) == len(chain)
        for a, b in zip(chain, exp):
            assert (a == b).all()
    # vt records reach the attributes
    assert s["attributes"]["uv"].max() == pytest.approx(4.5)


def test_unreadable_texture_keeps_material_untextured(host, tmp_path):
    (tmp_path / "a.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    (tmp_path / "a.mtl").write_text("newmtl m\nKd 1 0 0\nmap_Kd missing.png\n")
    s = host.LoadOBJFromFile(str(tmp_path / "a.obj"))
    assert len(s["textures"]) == 0 and s["materials"]["texture"][-1] == -1


# ---------------------------------------------------------------- pinned against the reference's own code (oracle/_ref)
def _hostlib():
    import ctypes
    return ctypes.CDLL(os.path.join(HERE, "..", "gpu-raytracing_amd", "host", "librt_host.so"))


@pytest.mark.skipif(not os.path.exists(os.path.join(HERE, "..", "oracle", "_ref", "libref_camera.so")),
                    reason="oracle/_ref/libref_camera.so (the reference's Camera.cu compiled as C++) is not built")
def test_camera_mirror_matches_the_reference_code(host, ora, scenes):
    """host/Camera.cpp against Camera.cu itself (compiled unmodified into oracle/_ref): UpdateCamera, InitialiseCamera
    and the three interactive controls, byte for byte over a sweep of angles and inputs.  This pins the camera every
    parity test uses (scenes.make_camera is checked against the same code below)."""
    import ctypes
    H = _hostlib()
    rng = np.random.default_rng(3)
    for _ in range(200):
        cam = np.zeros(1, host._pkg.CAMERA)
        cam["position"] = rng.uniform(-50, 50, 3).astype(np.float32)
        cam["yaw"], cam["pitch"] = np.float32(rng.uniform(-4, 4)), np.float32(rng.uniform(-1.55, 1.55))
        cam["scale"], cam["max_depth"] = np.float32(rng.uniform(0.1, 5)), np.float32(100)
        mine = host.UpdateCamera(cam)
        ref = ora.ref_update_camera(cam)
        assert mine.tobytes() == ref.tobytes()
        # scenes.make_camera (numpy float32 sin / cos) produces the same basis
        py = scenes.make_camera(cam["position"][0], float(cam["yaw"][0]), float(cam["pitch"][0]), 100.0, float(cam["scale"][0]))
        for f in ("u", "v", "w"):
            assert np.abs(py[f] - ref.view(host._pkg.CAMERA)[f]).max() <= 2e-7
        keys = int(rng.integers(0, 128))
        a, b = mine.copy(), ref.copy()
        H.rth_camera_move(a.ctypes.data_as(ctypes.c_void_p), keys)
        b = ora.ref_camera_controls(b, keys=[k for i, k in enumerate(("w", "a", "s", "d", "q", "e", "space")) if keys >> i & 1])
        dx, dy = float(rng.uniform(-30, 30)), float(rng.uniform(-30, 30))
        H.rth_camera_look(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_float(dx), ctypes.c_float(dy))
        b = ora.ref_camera_controls(b, look=(dx, dy))
        z = int(rng.integers(0, 2)) * 2 - 1
        H.rth_camera_zoom(a.ctypes.data_as(ctypes.c_void_p), z)
        b = ora.ref_camera_controls(b, zoom=z)
        assert a.tobytes() == b.tobytes()
    for box in ([0, 0, 0, 10, 10, 10], [-3, 2, 5, 40, 2.5, 9], [0, 0, 0, 708, 2, 708]):
        assert host.InitialiseCamera(box).tobytes() == ora.ref_initialise_camera(box).tobytes()


@pytest.mark.skipif(not os.path.exists(os.path.join(HERE, "..", "oracle", "_ref", "libref_arguments.so")),
                    reason="oracle/_ref/libref_arguments.so (the reference's Arguments.cpp) is not built")
def test_parse_cmd_matches_the_reference_code(ora):
    import ctypes
    H = _hostlib()
    for argv in (["rt", "a.obj"], ["rt", "a.obj", "--pairs"], ["rt", "a.obj", "--type", "bottom-up", "--splits"],
                 ["rt", "a.obj", "--type", "hybrid", "--pairs", "--splits"], ["rt", "a.obj", "--type", "sah", "--unknown", "--pairs"]):
        arr = (ctypes.c_char_p * len(argv))(*[a.encode() for a in argv])
        out = (ctypes.c_int * 4)()
        H.rth_parse_cmd(len(argv), arr, out)
        assert tuple(out) == tuple(int(v) for v in ora.ref_parse_cmd(argv)), argv


def test_partition_arithmetic_matches_the_python_harness():
    """host/Partition.h (the C++ multi-device path's frame partition, 1 - 8 devices) against gpu-raytracing_amd/sharding.py
    (what bench.py and the gloo tests use): row bands, interleaved strips, the bands -> strips decision."""
    import ctypes
    import importlib
    sh = importlib.import_module("gpu-raytracing_amd.sharding")
    L = _hostlib()
    out2 = (ctypes.c_uint * 2)()
    for fn in (L.rth_num_strips, L.rth_strips_per_device, L.rth_strips_owned, L.rth_compact_rows, L.rth_strip_rows_in_frame):
        fn.restype = ctypes.c_uint
    L.rth_choose_partition.restype = ctypes.c_int
    L.rth_choose_partition.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_uint]
    for height in (1, 7, 8, 9, 53, 768, 1080, 1081, 2160):
        assert L.rth_num_strips(height) == sh.num_strips(height)
        for world in range(1, 9):
            b = sh.band_bounds(height, world)
            covered = 0
            for d in range(world):
                L.rth_band_of(height, world, d, out2)
                assert (out2[0], out2[1]) == (b[d], b[d + 1])
                covered += out2[1] - out2[0]
                assert L.rth_strips_owned(height, world, d) == len(sh.my_strips(height, world, d))
            assert covered == height
            assert L.rth_strips_per_device(height, world) == sh.strips_per_rank(height, world)
            assert L.rth_compact_rows(height, world) == sh.compact_rows(height, world)
            # every row of the frame lies in exactly one strip of exactly one device
            rows = 0
            for d in range(world):
                for s in sh.my_strips(height, world, d):
                    rows += L.rth_strip_rows_in_frame(height, s)
            assert rows == height
    for costs in ([1.0], [1.0, 1.0], [0.02, 0.02, 0.85, 0.31], [1.0, 1.1, 1.0, 1.05], [0.0, 0.0], [1.0, 1.16, 1.0, 0.84]):
        arr = (ctypes.c_double * len(costs))(*costs)
        assert ("bands", "strips")[L.rth_choose_partition(arr, len(costs))] == sh.choose_partition(costs)


@pytest.mark.parametrize("strips", [False, True])
def test_multi_gpu_gather_plan_writes_every_row_once(strips):
    """host/MultiGpu.cpp::TraceFrame executes Partition.h::GatherPlan -- the receives into device 0 (band rows straight
    into the frame; compact strip buffers into staging slots) and, for strips, one strided copy per source device plus a
    plain copy for a strip the frame's edge cuts.  A multi-GPU node is not available to the tests, so the plan is replayed
    here on numpy buffers for 1 - 8 devices, odd widths and ragged heights: each device 'renders' its rows as (row label,
    device) and after the replay every row of device 0's frame must hold its own label from its owner, written exactly
    once (a wrong offset, pitch, piece count or cut-strip copy shows up as a missing, doubled or misplaced row)."""
    import ctypes
    L = _hostlib()
    L.rth_gather_plan.restype = ctypes.c_uint
    L.rth_gather_plan.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    for fn in (L.rth_strips_owned, L.rth_compact_rows, L.rth_strip_rows_in_frame):
        fn.restype = ctypes.c_uint
    buf = (ctypes.c_uint64 * (8 * 3 * 64))()
    out2 = (ctypes.c_uint * 2)()
    for W in (1, 5, 322):
        row = W * 4
        for H in (1, 7, 8, 9, 53, 203, 1080, 1081):
            for P in range(1, 9):
                nops = L.rth_gather_plan(W, H, P, int(strips), buf)
                ops = np.frombuffer(buf, np.uint64, nops * 8).reshape(nops, 8).astype(np.int64)
                crow = L.rth_compact_rows(H, P)
                # what each device rendered: label = 1 + global row (0 = never written), second byte = device
                frames = [np.zeros((H, row), np.uint16) for _ in range(P)]        # bands: in place in the device's own frame
                compact = [np.zeros((crow, row), np.uint16) for _ in range(P)]   # strips: compact buffers
                owner = np.full(H, -1)
                for d in range(P):
                    if strips:
                        for j in range(L.rth_strips_owned(H, P, d)):
                            s = d + j * P
                            for r in range(L.rth_strip_rows_in_frame(H, s)):
                                compact[d][j * 8 + r] = (s * 8 + r + 1) | (d << 12)
                                owner[s * 8 + r] = d
                    else:
                        L.rth_band_of(H, P, d, out2)
                        for y in range(out2[0], out2[1]):
                            frames[d][y] = (y + 1) | (d << 12)
                            owner[y] = d
                assert (owner >= 0).all()
                dst = frames[0].view(np.uint8).reshape(-1).copy() if not strips else np.zeros(H * row * 2, np.uint8)
                writes = np.zeros(H * row * 2, np.int32)
                if not strips:
                    L.rth_band_of(H, P, 0, out2)
                    writes[out2[0] * row * 2: out2[1] * row * 2] += 1              # device 0's band is in place
                staging = np.zeros(P * crow * row * 2, np.uint8)
                staging[: crow * row * 2] = compact[0].view(np.uint8).reshape(-1)  # device 0 renders into staging slot 0
                for kind, dev, src_off, dst_off, nbytes, sp, dp, pieces in ops:
                    # (offsets are in bytes of RGBA8 rows; the replay's rows are uint16 per byte -> scale by 2)
                    so, do, nb, sp, dp = src_off * 2, dst_off * 2, nbytes * 2, sp * 2, dp * 2
                    if kind == 0:      # kRecvBand: sender's frame -> device 0's frame
                        dst[do:do + nb] = frames[dev].view(np.uint8).reshape(-1)[so:so + nb]
                        writes[do:do + nb] += 1
                    elif kind == 1:    # kRecvCompact: sender's compact buffer -> staging
                        assert dev >= 1 and nb == crow * row * 2
                        staging[do:do + nb] = compact[dev].view(np.uint8).reshape(-1)[so:so + nb]
                    elif kind == 2:    # kCopyStrips: `pieces` pieces of nbytes, strided
                        for q in range(pieces):
                            dst[do + q * dp: do + q * dp + nb] = staging[so + q * sp: so + q * sp + nb]
                            writes[do + q * dp: do + q * dp + nb] += 1
                    else:              # kCopyCut
                        dst[do:do + nb] = staging[so:so + nb]
                        writes[do:do + nb] += 1
                assert (writes == 1).all(), (W, H, P, strips, "a byte of the frame written %d..%d times" % (writes.min(), writes.max()))
                got = dst.view(np.uint16).reshape(H, row)
                exp = ((np.arange(H) + 1) | (owner << 12)).astype(np.uint16)
                assert (got == exp[:, None]).all(), (W, H, P, strips)
