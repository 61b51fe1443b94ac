"""CPU test: the C-ABI library loads and exports every function include/rt_abi.h declares; the pure size queries
(no GPU work) answer sensibly; the Python-side POD dtypes match the header's layouts."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(rt):
    names = declared_functions()
    assert len(names) >= 11 and "rt_trace" in names and "rt_run_bottom_up_build" in names
    L = rt.lib()
    for n in names:
        assert getattr(L, n) is not None, n
    assert sorted(rt.EXPORTS) == names, "the Python binding must cover exactly the header's entry points"


def test_size_queries_without_gpu(rt):
    assert rt.NodesBytes(0) == 32 * 4 * 512                       # main.cu:235-237
    assert rt.NodesBytes(1000) == 32 * 4 * 1512
    sizes = [rt.BuMemoryRequirements(n) for n in (0, 1, 1000, 1 << 20, 10_008_338)]
    assert all(s % 256 == 0 for s in sizes) and sizes == sorted(sizes)
    assert sizes[3] >= 16 * (1 << 20)                              # morton + indices + 2 sort temporaries
    lay = rt.scratch_layout(1000)
    assert lay.p_aabb == 0 and lay.status == 32 and lay.morton % 256 == 0 and lay.sorted_indices >= lay.morton + 4000
    assert lay.total == rt.BuMemoryRequirements(1000)
    assert rt.RadixSortScratchBytes(1 << 20) >= 2 * 256 * 256 * 4
    assert "gfx950" in rt.version()
    assert rt.lib().rt_error_string(-2).decode().startswith("unsupported")


def test_bad_arguments_are_rejected_before_any_gpu_work(rt):
    L = rt.lib()
    assert L.rt_run_bottom_up_build(None, None, 0, None) == -1
    assert L.rt_trace(None, None, None, 0, None, 0, 0, 0, 0, 1, None) == -1
    assert L.rt_radix_sort_u32_pairs(None, None, None, None, 5, None, None) == -1
    assert L.rt_radix_sort_u32_pairs(None, None, None, None, 0, None, None) == 0   # empty input: nothing to do


def test_pod_layouts(rt):
    assert rt.NODE.fields["w12"][1] == 12 and rt.NODE.fields["max"][1] == 16 and rt.NODE.fields["w28"][1] == 28
    tp = rt.TRIANGLE_PAIR.fields
    assert (tp["primitive_id_0"][1], tp["v1"][1], tp["primitive_id_1"][1], tp["v2"][1], tp["rotations"][1], tp["v3"][1],
            tp["pad3"][1]) == (12, 16, 28, 32, 44, 48, 60)
    at = rt.ATTRIBUTES.fields
    assert (at["uv"][1], at["material_id"][1]) == (40, 64)
    cam = rt.CAMERA.fields
    assert (cam["pitch"][1], cam["w"][1], cam["yaw"][1], cam["u"][1], cam["scale"][1], cam["v"][1], cam["max_depth"][1]) == \
        (12, 16, 28, 32, 44, 48, 60)


def test_pod_layouts_match_the_reference_headers():
    """The ABI's PODs against the reference's own struct definitions (Common.cuh compiled by oracle/ref_pairing_driver.cpp):
    sizes, every field offset, and the Node bit-fields (parent:29 | count:3, child:29 | type:3)."""
    import importlib, os
    import numpy as np
    from oracle import oracle_py as ora
    if not ora.ref_pairing_available():
        import pytest
        pytest.skip("oracle/_ref/libref_pairing.so is not built")
    rt = importlib.import_module("gpu-raytracing_amd")
    L = [int(v) for v in ora.ref_struct_layout()]
    tri, node, pair, cam, att = rt.TRIANGLE, rt.NODE, rt.TRIANGLE_PAIR, rt.CAMERA, rt.ATTRIBUTES
    off = lambda d, f: d.fields[f][1]
    assert L[0:4] == [tri.itemsize, off(tri, "v0"), off(tri, "v1"), off(tri, "v2")]
    assert L[4:7] == [node.itemsize, off(node, "min"), off(node, "max")]
    assert L[7:16] == [pair.itemsize, off(pair, "v0"), off(pair, "primitive_id_0"), off(pair, "v1"), off(pair, "primitive_id_1"),
                       off(pair, "v2"), off(pair, "rotations"), off(pair, "v3"), off(pair, "pad3")]
    assert L[16:25] == [cam.itemsize, off(cam, "position"), off(cam, "pitch"), off(cam, "w"), off(cam, "yaw"), off(cam, "u"),
                        off(cam, "scale"), off(cam, "v"), off(cam, "max_depth")]
    assert L[25:29] == [att.itemsize, off(att, "normal"), off(att, "uv"), off(att, "material_id")]
    assert L[29] == 24
    n = ora.ref_pack_node([1, 2, 3], [4, 5, 6], 0x1ABCDEF, 5, 0x1234567, 2)
    assert int(n["w12"][0]) == (0x1ABCDEF | (5 << 29)) and int(n["w28"][0]) == (0x1234567 | (2 << 29))
    assert n["min"][0].tolist() == [1, 2, 3] and n["max"][0].tolist() == [4, 5, 6]
