"""The bottom-up build and the tracer never synchronise, allocate or copy: a whole frame (rt_run_bottom_up_build +
rt_trace) can be captured into a HIP graph and replayed -- the reference synchronises after every kernel
(run() macro, Common.cuh:369-388).  Replays must reproduce the eager frame and follow new input data."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("hybrid", [False, True])
def test_build_and_trace_in_a_hip_graph(hybrid, rt, scenes, ora):
    import torch
    G = 60
    tris = scenes.grid_mesh(G, 3)
    n = tris.shape[0]
    inp = rt.BuildInput.allocate(tris)
    cam = rt.to_device(scenes.camera_b(G))
    w, h = 320, 200
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    root = (max(2 * n, 2) + 1) if hybrid else 0

    def one_frame():
        rt.RunBottomUpBuild(inp, hybrid=hybrid)
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (w, h), cam, root, 2)

    one_frame()
    torch.cuda.synchronize()
    eager = frame.clone()
    assert int((eager.view(-1, 4)[:, 0] > 0).sum()) > 1000

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        one_frame()                       # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            one_frame()
    torch.cuda.current_stream().wait_stream(side)

    frame.zero_()
    inp.nodes_out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(frame, eager)

    # new triangles in the same buffers: the replay rebuilds and re-traces them
    tris2 = scenes.grid_mesh(G, 4)
    inp.triangles_in.copy_(rt.to_device(tris2))
    graph.replay()
    torch.cuda.synchronize()
    o = ora.build_hybrid(tris2) if hybrid else ora.build_bvh(tris2)
    exp, _ = ora.trace(o["leaves"], o["nodes"], o.get("root", 0), 2, scenes.camera_b(G), w, h, render_type=0)
    assert (frame.cpu().numpy().reshape(h, w, 4) == exp).all()


@pytest.mark.parametrize("scene", ["grid", "fractal"])
def test_sah_build_and_trace_in_a_hip_graph(scene, rt, scenes, ora):
    """rt_run_sah_build is a fixed sequence of asynchronous launches (round 4: the level loop's data-dependent tail is a
    device-side loop, sah_finish_kernel -- no copy back, no synchronisation), so SAH build + trace capture into ONE HIP graph
    too.  `grid`: every task finishes inside the level launches; `fractal`: 59 octaves under a binned SAH -- three level
    launches for 62 items per cell, trees 45+ levels deep: almost the whole build runs in the straggler kernel.  Replays
    follow new triangles in the same buffers; Node[] and the frame against the oracle."""
    import torch
    from helpers import assert_nodes_equal
    if scene == "grid":
        sets = [scenes.grid_mesh(60, 3), scenes.grid_mesh(60, 4)]
        cam_h, w, h = scenes.camera_b(60), 320, 200
    else:
        sets = [scenes.fractal_corner(4000, 3), scenes.fractal_corner(4000, 8)]
        cam_h, w, h = scenes.diagonal_camera(2.0 ** -10, 2.0 ** 45), 97, 65
    n = sets[0].shape[0]
    inp = rt.BuildInput.allocate(sets[0], sah=True)
    cam = rt.to_device(cam_h)
    frame = torch.zeros(w * h * 4, dtype=torch.uint8, device="cuda")
    counters = torch.zeros(4, dtype=torch.int64, device="cuda")
    lay = rt.sah_scratch_layout(n)

    def one_frame():
        rt.RunSahBuild(inp)
        rt.Trace(inp.triangles_out, inp.nodes_out, frame, (w, h), cam, 0, 1, render_type=1, counters=counters)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        one_frame()                       # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            one_frame()
    torch.cuda.current_stream().wait_stream(side)
    for tris in (sets[1], sets[0], sets[1]):
        inp.triangles_in.copy_(rt.to_device(tris))
        inp.nodes_out.zero_()
        frame.zero_()
        counters.zero_()
        graph.replay()
        torch.cuda.synchronize()
        status = rt.to_host(inp.scratch, np.uint32, 8, lay.status)
        assert status[0] == 0
        o = ora.build_sah(tris)
        assert int(status[1]) == o["L"]
        assert_nodes_equal(rt.to_host(inp.nodes_out, rt.NODE, 128 + 2 * o["L"]), o["nodes"], f"{scene} replay")
        exp, oc = ora.trace(o["leaves"], o["nodes"], 0, 1, cam_h, w, h, render_type=1)
        assert (frame.cpu().numpy().reshape(h, w, 4) == exp).all()
        assert (counters.cpu().numpy()[:2].astype(np.uint64) == oc[:2]).all()
