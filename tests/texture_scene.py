"""The textured test scene shared by the CPU and GPU texture tests: a 40 x 40 height field with planar uv,
4 materials (diffuse texture only / texture + bump / texture + normal map in `disp` / untextured) and 5 textures
(power-of-two, non-power-of-two, 1 x 1, a noise height map and a normal map)."""
import numpy as np


def make(scenes, ora):
    tris = scenes.grid_mesh(40, 9)
    n = tris.shape[0]
    mat_ids = (np.arange(n, dtype=np.int32) // 2 // 5) % 4           # runs of 5 cells share a material
    at = scenes.planar_uv_attributes(tris, mat_ids, uv_scale=0.21)
    mats = scenes.default_materials(4)
    chains = [ora.generate_lods(scenes.procedural_texture(64, 64, 1, "checker")),
              ora.generate_lods(scenes.procedural_texture(37, 21, 2, "checker")),
              ora.generate_lods(scenes.procedural_texture(1, 1, 3, "checker")),
              ora.generate_lods(scenes.procedural_texture(32, 16, 4, "noise")),
              ora.generate_lods(scenes.procedural_texture(16, 16, 5, "normal"))]
    mats[0]["texture"] = 0
    mats[1]["texture"], mats[1]["bump"] = 1, 3
    mats[2]["texture"], mats[2]["disp"] = 0, 4
    mats[3]["texture"] = -1
    cams = {"oblique": scenes.make_camera((4.0, 9.0, 4.0), -0.785, 0.7, 120.0),
            "top": scenes.make_camera((20.0, 30.0, 20.0), 0.0, 1.5, 120.0),
            "grazing": scenes.make_camera((-3.0, 4.0, 20.0), -1.5708, 0.25, 120.0)}
    light = (20.0, 3.0, -10.0)    # low over the height field: about a quarter of the hits are shadowed
    return dict(tris=tris, attributes=at, materials=mats, textures=chains, cameras=cams, light=light)
