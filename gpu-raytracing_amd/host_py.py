"""ctypes view of the C++ host mirror (host/librt_host.so): LoadOBJFromFile, InitialiseCamera / UpdateCamera,
CountNodes, VerifyHierarchy -- the host-side entry points the reference keeps next to the hot path
(reference src/FileIO.cpp:327, src/Camera.cu:8-91, src/Utilities.cpp:8-83).  No GPU is involved."""
from __future__ import annotations

import ctypes
import importlib
import os

import numpy as np

_pkg = importlib.import_module(__package__)
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host", "librt_host.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise _pkg.RtError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
        L = ctypes.CDLL(LIB_PATH)
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        L.rth_load_obj.restype = vp
        L.rth_load_obj.argtypes = [ctypes.c_char_p]
        L.rth_free.argtypes = [vp]
        for f in ("rth_num_triangles", "rth_num_materials", "rth_num_textures"):
            getattr(L, f).restype = u32
            getattr(L, f).argtypes = [vp]
        for f in ("rth_triangles", "rth_attributes", "rth_materials"):
            getattr(L, f).restype = vp
            getattr(L, f).argtypes = [vp]
        L.rth_texture_max_lod.restype = u32
        L.rth_texture_max_lod.argtypes = [vp, u32]
        L.rth_texture_size.argtypes = [vp, u32, u32, vp]
        L.rth_texture_mip.restype = vp
        L.rth_texture_mip.argtypes = [vp, u32, u32]
        L.rth_scene_aabb.argtypes = [vp, vp]
        L.rth_light.argtypes = [vp, vp]
        L.rth_initialise_camera.argtypes = [vp, vp]
        L.rth_update_camera.argtypes = [vp]
        L.rth_count_nodes.argtypes = [vp, u32, u32, vp]
        L.rth_verify_hierarchy.argtypes = [vp, u32, u32]
        L.rth_verify_hierarchy.restype = ctypes.c_int
        _lib = L
    return _lib


def _copy(ptr, dtype, count):
    if not count:
        return np.zeros(0, dtype)
    buf = (ctypes.c_char * (np.dtype(dtype).itemsize * count)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count).copy()


def LoadOBJFromFile(path: str) -> dict:
    L = lib()
    h = L.rth_load_obj(path.encode())
    if not h:
        raise FileNotFoundError(path)
    try:
        n, m = L.rth_num_triangles(h), L.rth_num_materials(h)
        aabb, light = np.zeros(6, np.float32), np.zeros(3, np.float32)
        L.rth_scene_aabb(h, aabb.ctypes.data_as(ctypes.c_void_p))
        L.rth_light(h, light.ctypes.data_as(ctypes.c_void_p))
        textures = []                                        # Library::textures: one mip chain per texture
        for t in range(L.rth_num_textures(h)):
            chain = []
            for l in range(L.rth_texture_max_lod(h, t) + 1):
                sz = np.zeros(2, np.int32)
                L.rth_texture_size(h, t, l, sz.ctypes.data_as(ctypes.c_void_p))
                chain.append(_copy(L.rth_texture_mip(h, t, l), np.uint32, int(sz[0]) * int(sz[1])).reshape(int(sz[1]), int(sz[0])))
            textures.append(chain)
        return dict(triangles=_copy(L.rth_triangles(h), np.float32, n * 9).reshape(n, 9),
                    attributes=_copy(L.rth_attributes(h), _pkg.ATTRIBUTES, n),
                    materials=_copy(L.rth_materials(h), _pkg.MATERIAL, m), aabb=aabb, light=light, textures=textures)
    finally:
        L.rth_free(h)


def InitialiseCamera(aabb) -> np.ndarray:
    cam = np.zeros(1, _pkg.CAMERA)
    a = np.ascontiguousarray(aabb, np.float32)
    lib().rth_initialise_camera(cam.ctypes.data_as(ctypes.c_void_p), a.ctypes.data_as(ctypes.c_void_p))
    return cam


def UpdateCamera(cam: np.ndarray) -> np.ndarray:
    cam = np.ascontiguousarray(cam).copy()
    lib().rth_update_camera(cam.ctypes.data_as(ctypes.c_void_p))
    return cam


def CountNodes(nodes: np.ndarray, root: int, count: int) -> tuple:
    out = np.zeros(3, np.int32)
    a = np.ascontiguousarray(nodes)
    lib().rth_count_nodes(a.ctypes.data_as(ctypes.c_void_p), root, count, out.ctypes.data_as(ctypes.c_void_p))
    return int(out[0]), int(out[1]), int(out[2])


def VerifyHierarchy(nodes: np.ndarray, root: int, count: int) -> int:
    a = np.ascontiguousarray(nodes)
    return int(lib().rth_verify_hierarchy(a.ctypes.data_as(ctypes.c_void_p), root, count))
