"""ctypes front-end of the CPU oracle (oracle/liboracle.so) and of the reference's own hierarchy checker
(oracle/_ref/libref_utilities.so = /root/reference/src/Utilities.cpp compiled unmodified).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by
the product package.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_PATH = os.path.join(_HERE, "_ref", "libref_utilities.so")

NODE = np.dtype([("min", "<f4", 3), ("w12", "<u4"), ("max", "<f4", 3), ("w28", "<u4")])
TRIANGLE_PAIR = np.dtype([("v0", "<f4", 3), ("primitive_id_0", "<u4"), ("v1", "<f4", 3), ("primitive_id_1", "<u4"),
                          ("v2", "<f4", 3), ("rotations", "<u2", 2), ("v3", "<f4", 3), ("pad3", "<f4")])

_lib = None


def build() -> None:
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.ora_set_threads.argtypes = [ctypes.c_int]
        L.ora_get_threads.restype = ctypes.c_int
        L.ora_scene_aabb.argtypes = [vp, u32, vp]
        L.ora_morton_codes.argtypes = [vp, u32, vp, vp, vp]
        L.ora_radix_sort.argtypes = [vp, vp, vp, vp, u32]
        L.ora_build.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.ora_build_pairs.argtypes = [vp, u32, vp, vp, vp, vp, vp]
        L.ora_build_pairs.restype = u32
        L.ora_build_hybrid_top.argtypes = [vp, u32, vp, vp]
        L.ora_build_hybrid_top.restype = u32
        L.ora_count_nodes.argtypes = [vp, u32, u32, vp]
        L.ora_verify_hierarchy.argtypes = [vp, u32, u32]
        L.ora_verify_hierarchy.restype = ctypes.c_int
        L.ora_trace.argtypes = [vp, vp, u32, u32, vp, vp, u32, vp, vp, ctypes.c_int, vp, u32, u32, u32, u32, u32, vp]
        L.ora_trace.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def set_threads(n: int) -> None:
    lib().ora_set_threads(int(n))


def scene_aabb(tris: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    out = np.zeros(6, dtype=np.int32)
    lib().ora_scene_aabb(_p(t), t.shape[0], _p(out))
    return out


def ordered_to_float(a: np.ndarray) -> np.ndarray:
    a = np.asarray(a, dtype=np.int32)
    return np.where(a >= 0, a, a ^ np.int32(0x7FFFFFFF)).astype(np.int32).view(np.float32)


def morton_codes(tris: np.ndarray, aabb_ordered: np.ndarray):
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    n = t.shape[0]
    codes, vals = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    a = np.ascontiguousarray(aabb_ordered, dtype=np.int32)
    lib().ora_morton_codes(_p(t), n, _p(a), _p(codes), _p(vals))
    return codes, vals


def radix_sort(keys: np.ndarray, vals: np.ndarray):
    k, v = np.ascontiguousarray(keys, np.uint32).copy(), np.ascontiguousarray(vals, np.uint32).copy()
    tk, tv = np.zeros_like(k), np.zeros_like(v)
    lib().ora_radix_sort(_p(k), _p(v), _p(tk), _p(tv), k.shape[0])
    return k, v


def build_bvh(tris: np.ndarray) -> dict:
    """RunBottomUpBuild on the CPU: returns nodes (2*max(n-1,1) slots), leaves, sorted codes/indices, scene box."""
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    n = t.shape[0]
    nodes = np.zeros(2 * max(n - 1, 1), dtype=NODE)
    leaves = np.zeros(max(n, 1), dtype=TRIANGLE_PAIR)
    codes, idx = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    aabb = np.zeros(6, np.int32)
    lib().ora_build(_p(t), n, _p(nodes), _p(leaves), _p(codes), _p(idx), _p(aabb))
    return dict(nodes=nodes, leaves=leaves[:n], codes=codes[:n], indices=idx[:n], aabb=aabb, n=n)


def build_pairs(tris: np.ndarray) -> dict:
    """RunBottomUpBuild with enable_pairs: shared-edge triangle pairs become quad leaves (deterministic slots)."""
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    n = t.shape[0]
    nodes = np.zeros(2 * max(n - 1, 1), dtype=NODE)
    leaves = np.zeros(max(n, 1), dtype=TRIANGLE_PAIR)
    codes, idx = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint32)
    aabb = np.zeros(6, np.int32)
    L = int(lib().ora_build_pairs(_p(t), n, _p(nodes), _p(leaves), _p(codes), _p(idx), _p(aabb)))
    return dict(nodes=nodes[:2 * max(L - 1, 1)], leaves=leaves[:L], codes=codes[:L], indices=idx[:L], aabb=aabb, n=n, L=L)


def build_hybrid(tris: np.ndarray) -> dict:
    """RunBottomUpBuild(hybrid=true): the LBVH plus the deterministic SAH top tree at slots >= 2L.
    Trace root = (root, 2) with root = 2L+1 (main.cu:222-223)."""
    b = build_bvh(tris)
    n = b["n"]
    nodes = np.zeros(2 * max(n, 1) + 2 * 256 + 8, dtype=NODE)
    nodes[:b["nodes"].shape[0]] = b["nodes"]
    sub = np.zeros(256, np.uint32)
    k = lib().ora_build_hybrid_top(_p(nodes), n, _p(b["aabb"]), _p(sub))
    b.update(nodes=nodes, subroots=sub[:k], root=max(2 * n, 2) + 1)
    return b


def build_sah(tris: np.ndarray, pairs: bool = False, splits: bool = False) -> dict:
    """RunSahBuild, deterministic restatement.  Trace root = (0, 1) (main.cu:222-223).
    nodes: the 2*64 + 2L slots in use (top tree in [0, 128), cell trees above); L = items (leaves, or leaf references
    with splits); leaves: the R TrianglePair records."""
    t = np.ascontiguousarray(tris, dtype=np.float32).reshape(-1, 9)
    n = t.shape[0]
    nodes = np.zeros(128 + 2 * (n + n // 5) + 2, dtype=NODE)
    leaves = np.zeros(max(n, 1), dtype=TRIANGLE_PAIR)
    cells = np.zeros(64, np.uint32)
    nrec = np.zeros(1, np.uint32)
    L = lib()
    L.ora_build_sah.restype = ctypes.c_uint32
    L.ora_build_sah.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    nl = int(L.ora_build_sah(_p(t), n, int(bool(pairs)), int(bool(splits)), _p(nodes), _p(leaves), _p(cells), _p(nrec)))
    return dict(nodes=nodes[:128 + 2 * nl], leaves=leaves[:int(nrec[0])], n=n, L=nl, R=int(nrec[0]), root=0, count=1,
                cell_counts=cells)


def count_nodes(nodes: np.ndarray, root: int, count: int) -> tuple:
    out = np.zeros(3, np.int32)
    lib().ora_count_nodes(_p(np.ascontiguousarray(nodes)), root, count, _p(out))
    return int(out[0]), int(out[1]), int(out[2])


def verify_hierarchy(nodes: np.ndarray, root: int, count: int) -> int:
    return int(lib().ora_verify_hierarchy(_p(np.ascontiguousarray(nodes)), root, count))


NUM_LODS = 13


class _Texture(ctypes.Structure):  # ora_texture
    _fields_ = [("mips", ctypes.c_void_p * NUM_LODS), ("size_x", ctypes.c_int32 * NUM_LODS),
                ("size_y", ctypes.c_int32 * NUM_LODS), ("max_lod", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


def generate_lods(mip0: np.ndarray) -> list:
    """Texture::GenerateLODs: mip0 is [sy, sx] uint32 (r | g<<8 | b<<16 | a<<24); returns the list of levels."""
    L = lib()
    L.ora_lod_sizes.restype = ctypes.c_uint32
    L.ora_lod_sizes.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.ora_generate_lod.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    mip0 = np.ascontiguousarray(mip0, np.uint32)
    sx, sy = np.zeros(NUM_LODS, np.int32), np.zeros(NUM_LODS, np.int32)
    max_lod = L.ora_lod_sizes(mip0.shape[1], mip0.shape[0], _p(sx), _p(sy))
    mips = [mip0]
    for l in range(1, max_lod + 1):
        dst = np.zeros((sy[l], sx[l]), np.uint32)
        L.ora_generate_lod(_p(mips[-1]), int(sx[l - 1]), int(sy[l - 1]), _p(dst))
        mips.append(dst)
    return mips


def _texture_table(textures):
    """textures: list of mip chains (each a list of [sy, sx] uint32 arrays).  Returns (ctypes array, keepalive)."""
    arr = (_Texture * len(textures))()
    keep = []
    for t, chain in enumerate(textures):
        assert 1 <= len(chain) <= NUM_LODS
        for l, m in enumerate(chain):
            m = np.ascontiguousarray(m, np.uint32)
            keep.append(m)
            arr[t].mips[l] = m.ctypes.data
            arr[t].size_x[l], arr[t].size_y[l] = m.shape[1], m.shape[0]
        arr[t].max_lod = len(chain) - 1
    return arr, keep


def trace(leaves, nodes, root, count, camera, w, h, *, render_type=0, attributes=None, materials=None, light=(0, 0, 0),
          rows=None, spp=1, textures=None):
    """TraceRays on the CPU.  Returns (rgba8 [h, w, 4] uint8, counters [box, tri, max_stack, dropped_pushes]).
    textures: list of mip chains (see generate_lods) indexed by Material.texture / .bump / .disp."""
    L = lib()
    L.ora_set_textures.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    if textures:
        table, _keep = _texture_table(textures)
        L.ora_set_textures(ctypes.cast(table, ctypes.c_void_p), len(textures))
    else:
        L.ora_set_textures(None, 0)
    try:
        return _trace(leaves, nodes, root, count, camera, w, h, render_type, attributes, materials, light, rows, spp)
    finally:
        L.ora_set_textures(None, 0)


def _trace(leaves, nodes, root, count, camera, w, h, render_type, attributes, materials, light, rows, spp):
    rgba = np.zeros((h, w, 4), np.uint8)
    counters = np.zeros(4, np.uint64)
    y0, y1 = (0, h) if rows is None else rows
    lt = np.asarray(light, np.float32)
    cam = np.ascontiguousarray(camera)
    nm = 0 if materials is None else materials.shape[0]
    rc = lib().ora_trace(_p(np.ascontiguousarray(leaves)), _p(np.ascontiguousarray(nodes)), root, count,
                         _p(None if attributes is None else np.ascontiguousarray(attributes)),
                         _p(None if materials is None else np.ascontiguousarray(materials)), nm, _p(cam), _p(lt),
                         render_type, _p(rgba), w, h, y0, y1, spp, _p(counters))
    if rc != 0:
        raise ValueError(f"oracle: unsupported render type {render_type}")
    return rgba, counters


# ---------------------------------------------------------------- the reference's own checker (oracle/_ref)
class _HierarchyStats(ctypes.Structure):  # Utilities.h:3-7
    _fields_ = [("numNodes", ctypes.c_int), ("numLeafNodes", ctypes.c_int), ("numTreeNodes", ctypes.c_int)]


_ref = None


def ref_available() -> bool:
    return os.path.exists(REF_PATH)


def _reflib():
    global _ref
    if _ref is None:
        R = ctypes.CDLL(REF_PATH)
        R.count = getattr(R, "_Z10CountNodesP4Nodejj")           # HierarchyStats CountNodes(Node*, unsigned, unsigned)
        R.count.restype = _HierarchyStats
        R.count.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        R.verify = getattr(R, "_Z15VerifyHierarchyP4Nodejj")    # void VerifyHierarchy(Node*, unsigned, unsigned)
        R.verify.restype = None
        R.verify.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint]
        _ref = R
    return _ref


def ref_count_nodes(nodes: np.ndarray, root: int, count: int) -> tuple:
    """Reference CountNodes (Utilities.cpp:32-44), the compiled reference code itself."""
    s = _reflib().count(_p(np.ascontiguousarray(nodes)), root, count)
    return s.numNodes, s.numLeafNodes, s.numTreeNodes


def ref_verify_hierarchy(nodes: np.ndarray, root: int, count: int) -> str:
    """Reference VerifyHierarchy (Utilities.cpp:77-83).  It reports by printing to stderr; returns what it printed
    ('' = hierarchy valid)."""
    import sys
    R = _reflib()
    a = np.ascontiguousarray(nodes)
    sys.stderr.flush()
    libc = ctypes.CDLL(None)
    with tempfile.TemporaryFile(mode="w+b") as tmp:
        saved = os.dup(2)
        try:
            os.dup2(tmp.fileno(), 2)
            R.verify(_p(a), root, count)
            libc.fflush(None)
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode(errors="replace")


# ---------------------------------------------------------------- the reference's camera and argument parser (oracle/_ref)
class _InputState(ctypes.Structure):  # Input.cuh:4-15
    _fields_ = [(k, ctypes.c_bool) for k in ("w", "a", "s", "d", "q", "e", "space", "mouse_down")] + \
               [("prev_x", ctypes.c_int), ("prev_y", ctypes.c_int)]


class _AABB(ctypes.Structure):
    _fields_ = [("v", ctypes.c_float * 6)]


class _RefArguments(ctypes.Structure):  # Arguments.h:28-33
    _fields_ = [("build_type", ctypes.c_int), ("enable_splits", ctypes.c_bool), ("enable_pairs", ctypes.c_bool),
                ("render_type", ctypes.c_int)]


def ref_camera_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_camera.so"))


def _refcam():
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_camera.so"))
    vp = ctypes.c_void_p
    R.update = getattr(R, "_Z12UpdateCameraR6Camera"); R.update.argtypes = [vp]
    R.init = getattr(R, "_Z16InitialiseCameraR6Camera4AABB"); R.init.argtypes = [vp, _AABB]
    R.zoom = getattr(R, "_Z16UpdateCameraZoomR6Camerai"); R.zoom.argtypes = [vp, ctypes.c_int]
    R.move = getattr(R, "_Z20UpdateCameraPositionR6Camera10InputState"); R.move.argtypes = [vp, _InputState]
    R.look = getattr(R, "_Z21UpdateCameraLookDeltaR6Cameraff"); R.look.argtypes = [vp, ctypes.c_float, ctypes.c_float]
    return R


def ref_update_camera(cam: np.ndarray) -> np.ndarray:
    """Camera.cu:8-29 (the reference's own code): cam is a 64-byte Camera record; returns the updated copy."""
    c = np.ascontiguousarray(cam).copy()
    _refcam().update(_p(c))
    return c


def ref_initialise_camera(aabb) -> np.ndarray:
    """Camera.cu:62-91"""
    c = np.zeros(64, np.uint8)
    box = _AABB()
    for k in range(6):
        box.v[k] = float(aabb[k])
    _refcam().init(_p(c), box)
    return c


def ref_camera_controls(cam: np.ndarray, keys=(), look=None, zoom=None) -> np.ndarray:
    """UpdateCameraPosition / UpdateCameraLookDelta / UpdateCameraZoom (Camera.cu:31-60) in that order."""
    c = np.ascontiguousarray(cam).copy()
    R = _refcam()
    if keys:
        st = _InputState()
        for k in keys:
            setattr(st, k, True)
        R.move(_p(c), st)
    if look is not None:
        R.look(_p(c), ctypes.c_float(look[0]), ctypes.c_float(look[1]))
    if zoom is not None:
        R.zoom(_p(c), int(zoom))
    return c


def ref_parse_cmd(argv) -> tuple:
    """Arguments.cpp:47-63 ParseCmd (the reference's own code): returns (build_type, enable_splits, enable_pairs, render_type)."""
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_arguments.so"))
    f = getattr(R, "_Z8ParseCmdiPPc")
    f.restype = _RefArguments
    f.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p)]
    arr = (ctypes.c_char_p * len(argv))(*[a.encode() for a in argv])
    a = f(len(argv), arr)
    return int(a.build_type), bool(a.enable_splits), bool(a.enable_pairs), int(a.render_type)


# ---------------------------------------------------------------- the reference's Pairing.cuh (oracle/_ref/libref_pairing.so)
def ref_pairing_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_pairing.so"))


def ref_pair_decision(a9: np.ndarray, b9: np.ndarray) -> tuple:
    """CanFormTrianglePair && ShouldFormTrianglePair (Pairing.cuh:35-58) on two 9-float triangles: (merge, rot_a, rot_b)."""
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_pairing.so"))
    R.ref_pair_decision.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rot = np.zeros(2, np.int32)
    m = R.ref_pair_decision(_p(np.ascontiguousarray(a9, np.float32)), _p(np.ascontiguousarray(b9, np.float32)), _p(rot))
    return bool(m), int(rot[0]), int(rot[1])


def ref_create_pair(a9, b9, a_id: int, b_id: int, rot_a: int, rot_b: int) -> np.ndarray:
    """CreateTrianglePair (Pairing.cuh:60-77): the 64-byte TrianglePair record (pad3, bytes 60..63, is not set by the reference)."""
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_pairing.so"))
    R.ref_create_pair.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p]
    out = np.zeros(64, np.uint8)
    R.ref_create_pair(_p(np.ascontiguousarray(a9, np.float32)), None if b9 is None else _p(np.ascontiguousarray(b9, np.float32)),
                      a_id, b_id, rot_a, rot_b, _p(out))
    return out


def ref_struct_layout() -> np.ndarray:
    """sizeof / offsetof of the reference's Triangle, Node, TrianglePair, Camera, Attributes, AABB (30 ints, see the driver)."""
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_pairing.so"))
    out = np.zeros(30, np.int32)
    R.ref_struct_layout(_p(out))
    return out


def ref_pack_node(mn, mx, parent: int, count: int, child: int, type_: int) -> np.ndarray:
    R = ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_pairing.so"))
    R.ref_pack_node.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_uint] * 4 + [ctypes.c_void_p]
    out = np.zeros(1, NODE)
    R.ref_pack_node(_p(np.asarray(mn, np.float32)), _p(np.asarray(mx, np.float32)), parent, count, child, type_, _p(out))
    return out


# ---------------------------------------------------------------- Common.cuh's small host-callable arithmetic, both sides
def box_helpers(which: str):
    """(triangle_centre, triangle_box, box_centre, box_combine, box_intersection) of the oracle (`which` = "ora": the helpers
    its build paths call) or of the reference's Common.cuh compiled from its tree (`which` = "ref").  Each takes / returns
    float32 arrays; box_intersection returns (box, valid)."""
    L = lib() if which == "ora" else ctypes.CDLL(os.path.join(_HERE, "_ref", "libref_pairing.so"))
    pre = "ora_" if which == "ora" else "ref_"
    vp = ctypes.c_void_p

    def call(name, ins, nout, ret_int=False):
        f = getattr(L, pre + name)
        f.argtypes = [vp] * (len(ins) + 1)
        f.restype = ctypes.c_int if ret_int else None
        arrs = [np.ascontiguousarray(a, np.float32) for a in ins]
        out = np.zeros(nout, np.float32)
        r = f(*[_p(a) for a in arrs], _p(out))
        return (out, bool(r)) if ret_int else out

    return (lambda v9: call("triangle_centre", [v9], 3), lambda v9: call("triangle_box", [v9], 6),
            lambda b6: call("box_centre", [b6], 3), lambda a6, b6: call("box_combine", [a6, b6], 6),
            lambda a6, b6: call("box_intersection", [a6, b6], 6, ret_int=True))
