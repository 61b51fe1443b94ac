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
