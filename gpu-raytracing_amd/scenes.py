"""Synthetic scenes and cameras for the parity tests and the bench.

Every generator is libm-free (integer hash -> float by exact scaling) so the triangles are bit-identical on
every machine.  Cameras are built with float32 numpy arithmetic following UpdateCamera
(reference src/Camera.cu:8-29); they are INPUTS handed identically to the oracle and to the GPU path, so parity
never depends on the host's sin/cos.
"""
from __future__ import annotations

import importlib

import numpy as np

_pkg = importlib.import_module(__package__) if __package__ else None
CAMERA = _pkg.CAMERA if _pkg else None
ATTRIBUTES = _pkg.ATTRIBUTES if _pkg else None
MATERIAL = _pkg.MATERIAL if _pkg else None


def pcg_hash(x: np.ndarray) -> np.ndarray:
    """PCG-RXS-M-XS 32-bit output hash on uint32 arrays."""
    x = x.astype(np.uint32)
    with np.errstate(over="ignore"):
        state = x * np.uint32(747796405) + np.uint32(2891336453)
        word = ((state >> ((state >> np.uint32(28)) + np.uint32(4))) ^ state) * np.uint32(277803737)
        return (word >> np.uint32(22)) ^ word


def grid_mesh(G: int, seed: int = 1) -> np.ndarray:
    """Height-field of G x G cells, 2 triangles per cell: returns float32 [2*G*G, 9].

    P(i,j) = (float(i), 2*h01(i,j), float(j)), h01 = (pcg_hash(i + 0x9E3779B9*j + seed) >> 8) * 2^-24; cells row-major
    (j outer); each cell emits (P00,P10,P01) then (P10,P11,P01).  G=708 -> 1,002,528 tris; G=2237 -> 10,008,338.
    """
    i = np.arange(G + 1, dtype=np.uint32)[None, :]
    j = np.arange(G + 1, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        key = i + np.uint32(0x9E3779B9) * j + np.uint32(seed)
    h01 = (pcg_hash(key) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    P = np.empty((G + 1, G + 1, 3), dtype=np.float32)  # [j, i]
    P[..., 0] = np.arange(G + 1, dtype=np.float32)[None, :]
    P[..., 1] = np.float32(2.0) * h01
    P[..., 2] = np.arange(G + 1, dtype=np.float32)[:, None]
    p00, p10, p01, p11 = P[:-1, :-1], P[:-1, 1:], P[1:, :-1], P[1:, 1:]
    tris = np.empty((G, G, 2, 3, 3), dtype=np.float32)
    tris[:, :, 0, 0], tris[:, :, 0, 1], tris[:, :, 0, 2] = p00, p10, p01
    tris[:, :, 1, 0], tris[:, :, 1, 1], tris[:, :, 1, 2] = p10, p11, p01
    return tris.reshape(-1, 9)


def soup(n: int, seed: int = 7, dup_fraction: float = 0.25, size: float = 0.02) -> np.ndarray:
    """n small random triangles in the unit cube; `dup_fraction` of them are exact copies of other triangles
    (equal centroids -> equal Morton codes -> exercises the index tie-break of cpl, BottomUpBuilder.cu:34-38)."""
    idx = np.arange(n * 9, dtype=np.uint32)
    r = (pcg_hash(idx + np.uint32((seed * 0x01000193) & 0xFFFFFFFF)) >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    r = r.reshape(n, 3, 3)
    centre = r[:, 0, :].copy()
    tris = np.empty((n, 3, 3), dtype=np.float32)
    tris[:, 0] = centre
    tris[:, 1] = centre + (r[:, 1] - np.float32(0.5)) * np.float32(size)
    tris[:, 2] = centre + (r[:, 2] - np.float32(0.5)) * np.float32(size)
    ndup = int(n * dup_fraction)
    if ndup:
        src = (pcg_hash(np.arange(ndup, dtype=np.uint32) + np.uint32(seed + 99)) % np.uint32(max(n - ndup, 1))).astype(np.int64)
        tris[n - ndup:] = tris[src]
    return tris.reshape(n, 9)


def fractal_corner(n: int, seed: int = 3, octaves: int = 59, top_exp: int = 42, size: float = 0.5) -> np.ndarray:
    """A self-similar scene: n triangles whose distance from the origin corner is log-uniform over `octaves` octaves below
    2^top_exp (radius = ldexp(1 + u, e): exact, libm-free), direction in the positive octant, edge length ~ size * radius.
    Deep, skewed trees: the LBVH is a chain down the Morton bits and an index-bit tree below 2^(top_exp-10); the binned SAH
    peels about three octaves per level.  A ray leaving the corner along the diagonal enters the nested ("rest") box
    first at every level and defers the other child: oracle max_stack 26-29 (LBVH), 46-48 (SAH) with 59 octaves, and a
    full 64-entry stack with 140 -- the traversal-stack tests (tests/test_gpu_traversal_edges.py)."""
    idx = np.arange(n * 10, dtype=np.uint32)
    h = pcg_hash(idx + np.uint32((seed * 0x01000193) & 0xFFFFFFFF)).reshape(n, 10)
    r = (h >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    e = np.int32(top_exp) - (h[:, 9] % np.uint32(octaves)).astype(np.int32)
    radius = np.ldexp(np.float32(1.0) + r[:, 0], e).astype(np.float32)
    d = (np.float32(0.05) + r[:, 1:4] * np.float32(0.95)).astype(np.float32)
    c = (d * radius[:, None]).astype(np.float32)
    t = np.empty((n, 3, 3), np.float32)
    t[:, 0] = c
    t[:, 1] = c + (r[:, 4:7] - np.float32(0.5)) * (radius * np.float32(size))[:, None]
    t[:, 2] = c + (r[:, 6:9] - np.float32(0.5)) * (radius * np.float32(size))[:, None]
    return t.reshape(n, 9)


def diagonal_camera(offset: float, max_depth: float) -> np.ndarray:
    """Camera at (-offset, -offset, -offset) looking along (1, 1, 1) with a hand-written basis (w is NOT normalised: the
    ray generator normalises p = ndc.x*u + ndc.y*v + w): with odd frame sizes the centre pixel has ndc = 0 and its ray
    runs exactly down the diagonal through the origin."""
    cam = np.zeros(1, dtype=CAMERA)
    cam["position"] = np.float32(-offset)
    cam["w"] = np.float32(1.0)
    cam["u"] = np.array([1, -1, 0], np.float32) * np.float32(0.70710678)
    cam["v"] = np.array([1, 1, -2], np.float32) * np.float32(0.40824829)
    cam["scale"], cam["max_depth"] = np.float32(1.0), np.float32(max_depth)
    return cam


def flat_mesh(G: int, seed: int = 3) -> np.ndarray:
    """Grid with every y equal: the scene box is flat on y, (c-min)/(max-min) = 0/0 = NaN, clamp -> 1
    (SURVEY 'hard parts': the NaN clamp path of GenerateMortonCodes)."""
    t = grid_mesh(G, seed).reshape(-1, 3, 3)
    t[:, :, 1] = np.float32(0.25)
    return t.reshape(-1, 9)


def _normalize(v: np.ndarray) -> np.ndarray:
    v = v.astype(np.float32)
    d = np.float32(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return (v * (np.float32(1.0) / np.sqrt(d, dtype=np.float32))).astype(np.float32)


def _cross(a, b):
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=np.float32)


def make_camera(position, yaw: float, pitch: float, max_depth: float, scale: float = 1.0) -> np.ndarray:
    """Camera record with the u,v,w basis of UpdateCamera (Camera.cu:8-29)."""
    cam = np.zeros(1, dtype=CAMERA)
    yaw32, pitch32 = np.float32(yaw), np.float32(pitch)
    w = np.array([-np.sin(yaw32) * np.cos(pitch32), -np.sin(pitch32), np.cos(yaw32) * np.cos(pitch32)], dtype=np.float32)
    w = _normalize(w)
    u = _normalize(_cross(w, np.array([0, 1, 0], dtype=np.float32)))
    v = _normalize(_cross(w, u))
    cam["position"] = np.asarray(position, dtype=np.float32)
    cam["pitch"], cam["yaw"], cam["scale"], cam["max_depth"] = pitch32, yaw32, np.float32(scale), np.float32(max_depth)
    cam["w"], cam["u"], cam["v"] = w, u, v
    return cam


def camera_a(G: int) -> np.ndarray:
    """'top-down' headline camera of SURVEY 8(d): 100 % coverage, coherent rays."""
    return make_camera((G / 2, 0.45 * G, G / 2), 0.0, 1.5, 1.5 * G)


def camera_b(G: int) -> np.ndarray:
    """'oblique' divergence-stress camera of SURVEY 8(d): partial coverage."""
    return make_camera((-0.3 * G, 0.5 * G, -0.3 * G), -0.8, 0.3, 1.5 * G)


def camera_for_box(bmin, bmax, yaw=0.6, pitch=0.45, back=1.4) -> np.ndarray:
    """A camera outside an arbitrary scene box looking at its centre region (test scenes)."""
    bmin, bmax = np.asarray(bmin, np.float32), np.asarray(bmax, np.float32)
    centre, ext = (bmin + bmax) * np.float32(0.5), bmax - bmin
    cam = make_camera(centre, yaw, pitch, float(ext.max()) * 4.0)
    cam["position"] = centre - cam["w"][0] * np.float32(back * float(ext.max()))
    return cam


def flat_attributes(triangles: np.ndarray, material_ids=None) -> np.ndarray:
    """Attributes as LoadOBJFromFile makes them without vn/vt (FileIO.cpp:88-93,415-430): the flat normal
    normalize(cross(v1-v0, v2-v1)) on all three corners, uv = 0."""
    t = triangles.reshape(-1, 3, 3).astype(np.float32)
    e1, e2 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1]
    with np.errstate(over="ignore", invalid="ignore"):       # (huge or degenerate triangles: inf / NaN normals, as the loader would give)
        return _flat_attributes(t, e1, e2, material_ids)


def _flat_attributes(t, e1, e2, material_ids):
    c = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                  e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1).astype(np.float32)
    d = (c[:, 0] * c[:, 0] + c[:, 1] * c[:, 1] + c[:, 2] * c[:, 2]).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        nrm = (c * (np.float32(1.0) / np.sqrt(d, dtype=np.float32))[:, None]).astype(np.float32)
    at = np.zeros(t.shape[0], dtype=ATTRIBUTES)
    at["normal"] = nrm[:, None, :]
    at["material_id"] = 0 if material_ids is None else material_ids
    return at


def default_materials(k: int = 3) -> np.ndarray:
    m = np.zeros(k, dtype=MATERIAL)
    pal = np.array([[0.8, 0.3, 0.2], [0.2, 0.7, 0.3], [0.25, 0.35, 0.9], [0.9, 0.8, 0.2]], dtype=np.float32)
    for i in range(k):
        m[i]["ambient"] = pal[i % 4] * np.float32(0.5)
        m[i]["diffuse"] = pal[i % 4]
        m[i]["specular"] = np.float32(0.5)
        m[i]["specular_exp"] = np.float32(10.0 + 7 * i)
        m[i]["texture"] = m[i]["bump"] = m[i]["disp"] = -1
    return m


def planar_uv_attributes(triangles: np.ndarray, material_ids=None, uv_scale: float = 0.25, axes=(0, 2)) -> np.ndarray:
    """flat_attributes plus a planar texture mapping uv = position[axes] * uv_scale per corner (what an OBJ with
    vt records would carry, FileIO.cpp:404-413)."""
    at = flat_attributes(triangles, material_ids)
    t = triangles.reshape(-1, 3, 3).astype(np.float32)
    at["uv"][:, :, 0] = t[:, :, axes[0]] * np.float32(uv_scale)
    at["uv"][:, :, 1] = t[:, :, axes[1]] * np.float32(uv_scale)
    return at


def procedural_texture(sx: int, sy: int, seed: int = 1, kind: str = "checker") -> np.ndarray:
    """[sy, sx] uint32 RGBA8 texels (r | g<<8 | b<<16 | a<<24).  kinds: checker (colour blocks + hash noise),
    noise (hash per texel, used as bump / height map), normal (unit-ish normals around +z, used as a normal map)."""
    x = np.arange(sx, dtype=np.uint32)[None, :]
    y = np.arange(sy, dtype=np.uint32)[:, None]
    with np.errstate(over="ignore"):
        hsh = pcg_hash(x + np.uint32(0x9E3779B9) * y + np.uint32((seed * 0x01000193) & 0xFFFFFFFF))
    if kind == "checker":
        c = (((x // 4) + (y // 4)) & 1).astype(np.uint32)
        r = np.where(c == 1, 220, 40).astype(np.uint32) + (hsh & np.uint32(31))
        g = np.where(c == 1, 60, 200).astype(np.uint32) + ((hsh >> np.uint32(5)) & np.uint32(31))
        b = (x * np.uint32(255) // np.uint32(max(sx - 1, 1)) + np.zeros_like(y)).astype(np.uint32)
        a = np.full((sy, sx), 255, np.uint32) - ((hsh >> np.uint32(10)) & np.uint32(63))
    elif kind == "noise":
        r = g = b = (hsh >> np.uint32(12)) & np.uint32(255)
        a = np.full((sy, sx), 255, np.uint32)
    elif kind == "normal":
        r = np.uint32(96) + ((hsh >> np.uint32(3)) & np.uint32(63))
        g = np.uint32(96) + ((hsh >> np.uint32(11)) & np.uint32(63))
        b = np.uint32(200) + ((hsh >> np.uint32(19)) & np.uint32(31))
        a = np.full((sy, sx), 255, np.uint32)
    else:
        raise ValueError(kind)
    r, g, b, a = (np.broadcast_to(v, (sy, sx)).astype(np.uint32) for v in (r, g, b, a))
    return (r | (g << np.uint32(8)) | (b << np.uint32(16)) | (a << np.uint32(24))).astype(np.uint32)
