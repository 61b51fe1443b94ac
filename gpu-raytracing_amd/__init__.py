"""gpu-raytracing_amd -- MI355X-native LBVH builder + primary-ray tracer (hot path of gregc-91/GPU-Raytracing).

Python is only the harness language here (tests, bench): the product is ``csrc/librt_amd.so`` -- hand-written
HIP kernels for gfx950 behind the C ABI of ``include/rt_abi.h`` -- and the C++ host mirror in ``host/``.
This module binds that C ABI with ctypes and mirrors the reference's entry points by name
(``BuildInput``, ``BuMemoryRequirements``, ``RunBottomUpBuild``, ``RadixSort``, ``Trace``; reference
``src/BuildWrapper.cuh:6-20``, ``src/RadixSort.cuh:6-7``, ``src/main.cu:125-127``).  torch is used for device
memory and streams only.  There is NO CPU fallback: if the HIP library is missing, import of the
native symbols fails loudly.

The directory name contains a hyphen, so import it with
``importlib.import_module("gpu-raytracing_amd")``.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_amd.so")

# ---------------------------------------------------------------- POD layouts (include/rt_abi.h)
TRIANGLE = np.dtype([("v0", "<f4", 3), ("v1", "<f4", 3), ("v2", "<f4", 3)])                        # 36 B
NODE = np.dtype([("min", "<f4", 3), ("w12", "<u4"), ("max", "<f4", 3), ("w28", "<u4")])              # 32 B
TRIANGLE_PAIR = np.dtype([("v0", "<f4", 3), ("primitive_id_0", "<u4"), ("v1", "<f4", 3), ("primitive_id_1", "<u4"),
                          ("v2", "<f4", 3), ("rotations", "<u2", 2), ("v3", "<f4", 3), ("pad3", "<f4")])  # 64 B
CAMERA = np.dtype([("position", "<f4", 3), ("pitch", "<f4"), ("w", "<f4", 3), ("yaw", "<f4"),
                   ("u", "<f4", 3), ("scale", "<f4"), ("v", "<f4", 3), ("max_depth", "<f4")])          # 64 B
ATTRIBUTES = np.dtype([("normal", "<f4", (3, 3)), ("pad0", "<u4"), ("uv", "<f4", (3, 2)),
                       ("material_id", "<i4"), ("pad1", "<u4")])                                     # 72 B
MATERIAL = np.dtype([("ambient", "<f4", 3), ("diffuse", "<f4", 3), ("specular", "<f4", 3),
                     ("specular_exp", "<f4"), ("texture", "<i4"), ("bump", "<i4"), ("disp", "<i4")])  # 52 B
assert TRIANGLE.itemsize == 36 and NODE.itemsize == 32 and TRIANGLE_PAIR.itemsize == 64
assert CAMERA.itemsize == 64 and ATTRIBUTES.itemsize == 72 and MATERIAL.itemsize == 52

INDEX_MASK = 0x1FFFFFFF
CHILD_NONE, CHILD_BOX, CHILD_TRI = 0, 1, 2
# Arguments.h:8-26
kSAH, kBottomUp, kHybrid, kNone = 0, 1, 2, 3
kDepth, kBoxtests, kTriangleTests, kMaterialId, kLODs, kDiffuse, kTexture, kTextureLit, kTextureLitShadows = range(9)


class RtError(RuntimeError):
    pass


# ---------------------------------------------------------------- ctypes structs
class _BuildInput(ctypes.Structure):
    _fields_ = [("triangles_in", ctypes.c_void_p), ("triangles_out", ctypes.c_void_p),
                ("num_triangles", ctypes.c_uint32), ("nodes_out", ctypes.c_void_p), ("scratch", ctypes.c_void_p)]


class _Arguments(ctypes.Structure):
    _fields_ = [("build_type", ctypes.c_int32), ("enable_splits", ctypes.c_int32),
                ("enable_pairs", ctypes.c_int32), ("render_type", ctypes.c_int32)]


class _Accel(ctypes.Structure):
    _fields_ = [("triangles", ctypes.c_void_p), ("nodes", ctypes.c_void_p),
                ("root", ctypes.c_uint32), ("count", ctypes.c_uint32)]


class _Scene(ctypes.Structure):
    _fields_ = [("attributes", ctypes.c_void_p), ("materials", ctypes.c_void_p), ("textures", ctypes.c_void_p),
                ("camera", ctypes.c_void_p), ("light", ctypes.c_float * 3),
                ("num_attributes", ctypes.c_uint32), ("num_materials", ctypes.c_uint32),
                ("num_textures", ctypes.c_uint32)]


NUM_LODS = 13  # Common.cuh:17


class _Texture(ctypes.Structure):  # rt_texture
    _fields_ = [("mips", ctypes.c_void_p * NUM_LODS), ("size_x", ctypes.c_int32 * NUM_LODS),
                ("size_y", ctypes.c_int32 * NUM_LODS), ("max_lod", ctypes.c_uint32), ("pad", ctypes.c_uint32)]


assert ctypes.sizeof(_Texture) == 216


class _ScratchLayout(ctypes.Structure):
    _fields_ = [("p_aabb", ctypes.c_size_t), ("status", ctypes.c_size_t), ("num_leaves", ctypes.c_size_t),
                ("morton", ctypes.c_size_t),
                ("sorted_indices", ctypes.c_size_t), ("total", ctypes.c_size_t)]


class _SahScratchLayout(ctypes.Structure):
    _fields_ = [("p_aabb", ctypes.c_size_t), ("c_aabb", ctypes.c_size_t), ("status", ctypes.c_size_t),
                ("num_leaves", ctypes.c_size_t), ("cell_counts", ctypes.c_size_t), ("total", ctypes.c_size_t)]


EXPORTS = ["rt_bu_memory_requirements", "rt_nodes_bytes", "rt_run_bottom_up_build", "rt_bu_scratch_layout_get",
           "rt_sah_memory_requirements", "rt_run_sah_build", "rt_sah_scratch_layout_get",
           "rt_calculate_scene_aabb", "rt_generate_morton_codes", "rt_radix_sort_scratch_bytes",
           "rt_radix_sort_u32_pairs", "rt_radix_sort_u32_pairs_bits", "rt_radix_sort_input_in_tmp", "rt_trace", "rt_trace_strips", "rt_error_string", "rt_version_string"]

_lib = None


def lib() -> ctypes.CDLL:
    """Load csrc/librt_amd.so.  Fails loudly when it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RtError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch ships its own HIP runtime: it must be the one in the process before librt_amd.so resolves its HIP symbols
    # (loaded the other way round, the two runtimes disagree about the device: hipErrorNoDevice at the first launch)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, i32 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int
    L.rt_bu_memory_requirements.restype = ctypes.c_size_t
    L.rt_bu_memory_requirements.argtypes = [u32]
    L.rt_nodes_bytes.restype = ctypes.c_size_t
    L.rt_nodes_bytes.argtypes = [u32]
    L.rt_run_bottom_up_build.restype = i32
    L.rt_run_bottom_up_build.argtypes = [ctypes.POINTER(_BuildInput), ctypes.POINTER(_Arguments), i32, vp]
    L.rt_bu_scratch_layout_get.restype = i32
    L.rt_bu_scratch_layout_get.argtypes = [u32, ctypes.POINTER(_ScratchLayout)]
    L.rt_sah_memory_requirements.restype = ctypes.c_size_t
    L.rt_sah_memory_requirements.argtypes = [u32]
    L.rt_run_sah_build.restype = i32
    L.rt_run_sah_build.argtypes = [ctypes.POINTER(_BuildInput), ctypes.POINTER(_Arguments), vp]
    L.rt_sah_scratch_layout_get.restype = i32
    L.rt_sah_scratch_layout_get.argtypes = [u32, ctypes.POINTER(_SahScratchLayout)]
    L.rt_calculate_scene_aabb.restype = i32
    L.rt_calculate_scene_aabb.argtypes = [vp, u32, vp, vp]
    L.rt_generate_morton_codes.restype = i32
    L.rt_generate_morton_codes.argtypes = [vp, vp, vp, vp, u32, vp]
    L.rt_radix_sort_scratch_bytes.restype = ctypes.c_size_t
    L.rt_radix_sort_scratch_bytes.argtypes = [u32]
    L.rt_radix_sort_u32_pairs.restype = i32
    L.rt_radix_sort_u32_pairs.argtypes = [vp, vp, vp, vp, u32, vp, vp]
    L.rt_radix_sort_u32_pairs_bits.restype = i32
    L.rt_radix_sort_u32_pairs_bits.argtypes = [vp, vp, vp, vp, u32, u32, i32, vp, vp]
    L.rt_radix_sort_input_in_tmp.restype = i32
    L.rt_radix_sort_input_in_tmp.argtypes = [u32, u32]
    L.rt_trace.restype = i32
    L.rt_trace.argtypes = [ctypes.POINTER(_Accel), ctypes.POINTER(_Scene), vp, i32, vp, u32, u32, u32, u32, u32, vp]
    L.rt_trace_strips.restype = i32
    L.rt_trace_strips.argtypes = [ctypes.POINTER(_Accel), ctypes.POINTER(_Scene), vp, i32, vp, u32, u32, u32, u32, u32, u32, vp]
    L.rt_error_string.restype = ctypes.c_char_p
    L.rt_error_string.argtypes = [i32]
    L.rt_version_string.restype = ctypes.c_char_p
    _lib = L
    return L


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RtError(f"{what} failed: {rc} ({lib().rt_error_string(rc).decode()})")


def _torch():
    import torch
    return torch


def _stream_ptr(stream) -> int:
    torch = _torch()
    s = stream if stream is not None else torch.cuda.current_stream()
    return int(s.cuda_stream)


def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


def device_bytes(nbytes: int, device="cuda"):
    """Uninitialised device buffer (256-byte aligned by the caching allocator)."""
    torch = _torch()
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


def to_device(arr: np.ndarray, device="cuda"):
    torch = _torch()
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(device)


def to_host(t, dtype: np.dtype, count: Optional[int] = None, offset: int = 0) -> np.ndarray:
    """Copy (part of) a device byte buffer back as a numpy array of `dtype`."""
    dtype = np.dtype(dtype)
    nbytes = t.numel() - offset if count is None else count * dtype.itemsize
    return t[offset:offset + nbytes].cpu().numpy().view(dtype).copy()


# ---------------------------------------------------------------- reference-named entry points
def BuMemoryRequirements(num_triangles: int) -> int:
    """BuildWrapper.cu:132-136"""
    return int(lib().rt_bu_memory_requirements(num_triangles))


def NodesBytes(num_triangles: int) -> int:
    """main.cu:235-237"""
    return int(lib().rt_nodes_bytes(num_triangles))


@dataclass
class Arguments:
    """Arguments.h:28-33 (defaults as in the reference)"""
    build_type: int = kSAH
    enable_splits: bool = False
    enable_pairs: bool = False
    render_type: int = kDepth


@dataclass
class BuildInput:
    """BuildWrapper.cuh:6-12; fields are torch uint8 device buffers owned by the caller."""
    triangles_in: object
    triangles_out: object
    num_triangles: int
    nodes_out: object
    scratch: object

    @staticmethod
    def allocate(triangles: np.ndarray, device="cuda", sah: bool = False) -> "BuildInput":
        """What Display() does at frame 0 (main.cu:226-240): allocate the four buffers, upload triangles.
        sah: size the scratch with SahMemoryRequirements instead of BuMemoryRequirements (main.cu:227-234)."""
        tri = np.ascontiguousarray(triangles, dtype=np.float32).reshape(-1, 9)
        n = tri.shape[0]
        return BuildInput(triangles_in=to_device(tri, device) if n else device_bytes(64, device),
                          triangles_out=device_bytes(64 * max(n, 1) + 64, device), num_triangles=n,
                          nodes_out=device_bytes(NodesBytes(n), device),
                          scratch=device_bytes(SahMemoryRequirements(n) if sah else BuMemoryRequirements(n), device))


def RunBottomUpBuild(inp: BuildInput, args: Optional[Arguments] = None, hybrid: bool = False, stream=None) -> None:
    """BuildWrapper.cu:253-362.  Asynchronous on `stream` (torch current stream by default)."""
    args = args or Arguments(build_type=kHybrid if hybrid else kBottomUp)
    ci = _BuildInput(_ptr(inp.triangles_in), _ptr(inp.triangles_out), inp.num_triangles, _ptr(inp.nodes_out),
                     _ptr(inp.scratch))
    ca = _Arguments(args.build_type, int(args.enable_splits), int(args.enable_pairs), args.render_type)
    _check(lib().rt_run_bottom_up_build(ctypes.byref(ci), ctypes.byref(ca), int(hybrid), _stream_ptr(stream)),
           "rt_run_bottom_up_build")


def SahMemoryRequirements(num_triangles: int) -> int:
    """BuildWrapper.cu:126-130"""
    return int(lib().rt_sah_memory_requirements(num_triangles))


def RunSahBuild(inp: BuildInput, args: Optional[Arguments] = None, stream=None) -> None:
    """BuildWrapper.cu:140-251.  Trace root = (0, 1).  Asynchronous on `stream` (no copy, no synchronisation: graph-capturable);
    error flags in the scratch status word (sah_scratch_layout(n).status)."""
    args = args or Arguments(build_type=kSAH)
    ci = _BuildInput(_ptr(inp.triangles_in), _ptr(inp.triangles_out), inp.num_triangles, _ptr(inp.nodes_out),
                     _ptr(inp.scratch))
    ca = _Arguments(args.build_type, int(args.enable_splits), int(args.enable_pairs), args.render_type)
    _check(lib().rt_run_sah_build(ctypes.byref(ci), ctypes.byref(ca), _stream_ptr(stream)), "rt_run_sah_build")


def sah_scratch_layout(num_triangles: int) -> _SahScratchLayout:
    out = _SahScratchLayout()
    _check(lib().rt_sah_scratch_layout_get(num_triangles, ctypes.byref(out)), "rt_sah_scratch_layout_get")
    return out


def scratch_layout(num_triangles: int) -> _ScratchLayout:
    out = _ScratchLayout()
    _check(lib().rt_bu_scratch_layout_get(num_triangles, ctypes.byref(out)), "rt_bu_scratch_layout_get")
    return out


def CalculateSceneAabb(triangles_dev, n: int, aabb_dev, stream=None) -> None:
    """Multiblock.cu:104-114 (aabb_dev: 6 x int32, ordered-int encoded)"""
    _check(lib().rt_calculate_scene_aabb(_ptr(triangles_dev), n, _ptr(aabb_dev), _stream_ptr(stream)),
           "rt_calculate_scene_aabb")


def GenerateMortonCodes(codes_dev, values_dev, triangles_dev, aabb_dev, n: int, stream=None) -> None:
    """BottomUpBuilder.cu:98-115"""
    _check(lib().rt_generate_morton_codes(_ptr(codes_dev), _ptr(values_dev), _ptr(triangles_dev), _ptr(aabb_dev), n,
                                          _stream_ptr(stream)), "rt_generate_morton_codes")


def RadixSort(keys, values, temp1, temp2, count: int, sort_scratch=None, stream=None) -> None:
    """RadixSort.cuh:6-7.  The reference mallocs its tables per call (RadixSort.cu:187-190); here the caller may
    pass `sort_scratch` (>= RadixSortScratchBytes(count)), else one is taken from torch's caching allocator."""
    if sort_scratch is None:
        sort_scratch = device_bytes(int(lib().rt_radix_sort_scratch_bytes(count)), keys.device)
    _check(lib().rt_radix_sort_u32_pairs(_ptr(keys), _ptr(values), _ptr(temp1), _ptr(temp2), count,
                                         _ptr(sort_scratch), _stream_ptr(stream)), "rt_radix_sort_u32_pairs")


def RadixSortBits(keys, values, temp1, temp2, count: int, key_bits: int, input_in_tmp: bool = False, sort_scratch=None,
                  stream=None) -> None:
    """rt_radix_sort_u32_pairs_bits: the sort for keys of `key_bits` significant bits (the builder's Morton codes: 30)."""
    if sort_scratch is None:
        sort_scratch = device_bytes(int(lib().rt_radix_sort_scratch_bytes(count)), keys.device)
    _check(lib().rt_radix_sort_u32_pairs_bits(_ptr(keys), _ptr(values), _ptr(temp1), _ptr(temp2), count, key_bits,
                                              int(input_in_tmp), _ptr(sort_scratch), _stream_ptr(stream)),
           "rt_radix_sort_u32_pairs_bits")


def RadixSortScratchBytes(count: int) -> int:
    return int(lib().rt_radix_sort_scratch_bytes(count))


class DeviceTextures:
    """The device-side Texture table of DeviceScene (Common.cuh:342-351, uploaded by main.cu:100-113): one
    rt_texture per texture, mips in device memory.  `chains` is a list of mip chains, each a list of [sy, sx]
    uint32 arrays (r | g<<8 | b<<16 | a<<24), level 0 first."""

    def __init__(self, chains, device="cuda"):
        table = (_Texture * max(1, len(chains)))()
        self._mips = []
        for t, chain in enumerate(chains):
            if not 1 <= len(chain) <= NUM_LODS:
                raise ValueError("a texture has 1..13 mip levels")
            for l, m in enumerate(chain):
                m = np.ascontiguousarray(m, np.uint32)
                d = to_device(m, device)
                self._mips.append(d)
                table[t].mips[l] = _ptr(d)
                table[t].size_x[l], table[t].size_y[l] = m.shape[1], m.shape[0]
            table[t].max_lod = len(chain) - 1
        self.count = len(chains)
        self.table = to_device(np.frombuffer(bytes(table), np.uint8).copy(), device)


def Trace(triangles, nodes, rgba8, dims, camera, root: int, count: int, *, render_type: int = kDepth,
          attributes=None, materials=None, num_materials: int = 0, light=(0.0, 0.0, 0.0), counters=None,
          rows=None, spp: int = 1, stream=None, textures: Optional[DeviceTextures] = None, strips=None,
          num_primitives: int = 0) -> None:
    """main.cu:125-192 Trace(): `camera` is a 64-byte DEVICE buffer, `rgba8` a w*h*4-byte device buffer
    (the reference writes a GL surface; row 0 first).  rows=(y0, y1) restricts to a row band (multi-GPU tiling);
    strips=(strip_rows, first, stride) renders interleaved strips into a COMPACT buffer (rt_trace_strips).
    num_primitives: DeviceScene::num_attributes (main.cu:166 sets it to the triangle count; 0 = unknown) -- the library
    reads it as the scene size when it picks the tracer instantiation (pair prefetch from 8M primitives on)."""
    w, h = int(dims[0]), int(dims[1])
    y0, y1 = (0, h) if rows is None else (int(rows[0]), int(rows[1]))
    if strips is not None:
        a = _Accel(_ptr(triangles), _ptr(nodes), root, count)
        s = _Scene(_ptr(attributes), _ptr(materials), _ptr(textures.table) if textures is not None else 0, _ptr(camera),
                   (ctypes.c_float * 3)(*[float(x) for x in light]),
                   int(num_primitives), num_materials, textures.count if textures is not None else 0)
        _check(lib().rt_trace_strips(ctypes.byref(a), ctypes.byref(s), _ptr(counters), render_type, _ptr(rgba8), w, h,
                                     int(strips[0]), int(strips[1]), int(strips[2]), spp, _stream_ptr(stream)),
               "rt_trace_strips")
        return
    a = _Accel(_ptr(triangles), _ptr(nodes), root, count)
    s = _Scene(_ptr(attributes), _ptr(materials), _ptr(textures.table) if textures is not None else 0, _ptr(camera),
               (ctypes.c_float * 3)(*[float(x) for x in light]),
               int(num_primitives), num_materials, textures.count if textures is not None else 0)
    _check(lib().rt_trace(ctypes.byref(a), ctypes.byref(s), _ptr(counters), render_type, _ptr(rgba8), w, h, y0, y1,
                          spp, _stream_ptr(stream)), "rt_trace")


def version() -> str:
    return lib().rt_version_string().decode()
