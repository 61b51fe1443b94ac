"""CPU checks of the oracle's texture path (Texture::GenerateLODs, FileIO.cpp:121-150; Tracer.cu:58-375 sampling,
LOD selection and the textured render types).  There is no golden frame for these modes in the reference, so they are
pinned by properties: mip sizes and box-filter values by a numpy restatement, constant textures giving constant
colours, kTexture on an untextured material giving the material colour, shadows only ever darkening."""
import numpy as np
import pytest

import texture_scene


def _np_lods(m):
    out = [m]
    while (m.shape[1] > 1 or m.shape[0] > 1) and len(out) < 13:
        sy, sx = (m.shape[0] + 1) // 2, (m.shape[1] + 1) // 2
        jj = np.minimum(np.arange(sy)[:, None] * 2 + np.array([0, 0, 1, 1])[None, :], m.shape[0] - 1)   # [sy, 4]
        ii = np.minimum(np.arange(sx)[:, None] * 2 + np.array([0, 1, 0, 1])[None, :], m.shape[1] - 1)   # [sx, 4]
        t = m[jj[:, None, :], ii[None, :, :]]                                                            # [sy, sx, 4]
        nxt = np.zeros((sy, sx), np.uint32)
        for c in range(4):
            ch = ((t >> np.uint32(8 * c)) & np.uint32(255)).astype(np.float32)
            s = ((ch[..., 0] + ch[..., 1]) + ch[..., 2]) + ch[..., 3]
            nxt |= (s * np.float32(0.25)).astype(np.uint32) << np.uint32(8 * c)
        out.append(nxt)
        m = nxt
    return out


@pytest.mark.parametrize("sx,sy", [(64, 64), (37, 21), (1, 1), (5, 1), (1, 7), (256, 2)])
def test_generate_lods_matches_numpy(scenes, ora, sx, sy):
    m0 = scenes.procedural_texture(sx, sy, 11, "checker")
    got, exp = ora.generate_lods(m0), _np_lods(m0)
    assert len(got) == len(exp)
    for l, (g, e) in enumerate(zip(got, exp)):
        assert g.shape == e.shape, l
        assert (g == e).all(), l
    assert got[-1].shape == (1, 1)


def test_host_mirror_generate_lods_matches_oracle(scenes, ora):
    """host/FileIO.cpp Texture::GenerateLODs (the product's host code) against the oracle, via rth_generate_lods."""
    import ctypes, os
    here = os.path.dirname(os.path.abspath(__file__))
    H = ctypes.CDLL(os.path.join(here, "..", "gpu-raytracing_amd", "host", "librt_host.so"))
    H.rth_generate_lods.restype = ctypes.c_uint32
    H.rth_generate_lods.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]
    for sx, sy in [(64, 64), (37, 21), (1, 1), (5, 1), (300, 7)]:
        m0 = scenes.procedural_texture(sx, sy, 13, "checker")
        exp = ora.generate_lods(m0)
        size_x, size_y = np.zeros(13, np.int32), np.zeros(13, np.int32)
        max_lod = H.rth_generate_lods(m0.ctypes.data, sx, sy, size_x.ctypes.data, size_y.ctypes.data, None)
        assert max_lod == len(exp) - 1
        bufs = [np.zeros((max(size_y[l], 1), max(size_x[l], 1)), np.uint32) for l in range(13)]
        ptrs = (ctypes.c_void_p * 13)(*[b.ctypes.data for b in bufs])
        H.rth_generate_lods(m0.ctypes.data, sx, sy, size_x.ctypes.data, size_y.ctypes.data, ptrs)
        for l in range(1, max_lod + 1):
            assert (size_x[l], size_y[l]) == (exp[l].shape[1], exp[l].shape[0])
            assert (bufs[l] == exp[l]).all(), (sx, sy, l)


def test_textured_modes_properties(scenes, ora):
    sc = texture_scene.make(scenes, ora)
    b = ora.build_bvh(sc["tris"])
    cam, w, h = sc["cameras"]["oblique"], 160, 120
    kw = dict(attributes=sc["attributes"], materials=sc["materials"], light=sc["light"])
    depth, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=0)
    hit = depth[..., 0] > 0
    assert 0.3 < hit.mean() < 1.0
    # constant textures: every textured material samples the same colour at every level -> kTexture is that colour
    const = ora.generate_lods(np.full((8, 8), 0x80FF4020, np.uint32))
    tex, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=6, textures=[const] * 5, **kw)
    textured_px = tex[hit]
    # (the float blends of equal texels may truncate one below: w0 * c + w1 * c < c in float)
    diff = np.array([0x20, 0x40, 0xFF, 0x80], np.int32) - textured_px.astype(np.int32)
    is_const = ((diff >= 0) & (diff <= 1)).all(axis=1)
    d = sc["materials"][3]["diffuse"]
    is_mat3 = (textured_px[:, :3] == (d * np.float32(255)).astype(np.uint8)).all(axis=1) & (textured_px[:, 3] == 255)
    assert (is_const | is_mat3).all()
    assert 0.5 < is_const.mean() < 0.95 and is_mat3.any()
    assert (tex[~hit] == np.array([0, 0, 0, 255], np.uint8)).all()
    # kLODs: miss or untextured -> magenta; else grey = int(lod) * 20
    lods, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=4, textures=sc["textures"], **kw)
    magenta = (lods[..., :3] == np.array([255, 0, 255], np.uint8)).all(axis=-1)
    assert magenta[~hit].all()
    grey = lods[~magenta]
    assert (grey[:, 0] == grey[:, 1]).all() and (grey[:, 0] % 20 == 0).all() and grey[:, 0].max() <= 240
    assert len(np.unique(grey[:, 0])) >= 2, "the oblique view spans several mip levels"
    # shadows only remove light
    lit, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=7, textures=sc["textures"], **kw)
    shd, _ = ora.trace(b["leaves"], b["nodes"], 0, 2, cam, w, h, render_type=8, textures=sc["textures"], **kw)
    assert (shd[..., :3] <= lit[..., :3]).all()
    assert (shd[..., :3] < lit[..., :3]).any(), "some pixel is in shadow"
    assert (shd != lit).any(axis=-1).mean() < 0.8
