"""Scenes, cameras and hand-made trees for the traversal edge cases (tests/test_gpu_traversal_edges.py on the GPU,
tests/test_oracle_edges.py on the CPU): deep stacks, a full stack, -0.0 / zero-area / repeated-vertex triangles,
axis-parallel rays starting on box planes, nodes wider than a pair."""
import numpy as np


def signed_zero_mesh(scenes, G=20, seed=6):
    """A grid mesh centred on the origin whose zero coordinates are -0.0 or +0.0 by a hash, with heights snapped so that
    some are exactly +-0.0, plus degenerate triangles: all three vertices equal, a repeated vertex, three collinear
    points, and exact copies of each (equal Morton codes)."""
    t = scenes.grid_mesh(G, seed).reshape(-1, 3, 3).copy()
    t[:, :, 0] -= np.float32(G // 2)
    t[:, :, 2] -= np.float32(G // 2)
    t[:, :, 1] = np.where(t[:, :, 1] < np.float32(0.5), np.float32(0.0), t[:, :, 1])
    h = scenes.pcg_hash(np.arange(t.size, dtype=np.uint32) + np.uint32(seed * 7919)).reshape(t.shape)
    neg = (t == 0) & ((h & np.uint32(1)) == 1)
    t[neg] = np.float32(-0.0)
    assert np.signbit(t[t == 0]).any() and (~np.signbit(t[t == 0])).any()
    k = t.shape[0]
    deg = np.empty((12, 3, 3), np.float32)
    for i in range(4):
        a, b = t[(i * 37) % k, 0], t[(i * 53 + 11) % k, 2]
        deg[3 * i] = np.stack([a, a, a])                                  # a point
        deg[3 * i + 1] = np.stack([a, b, b])                              # a repeated vertex (a segment)
        deg[3 * i + 2] = np.stack([a, b, (a + b) * np.float32(0.5)])      # collinear
    out = np.concatenate([t[: k // 2], deg, t[k // 2:], deg[:6]])
    return out.reshape(-1, 9)


def axis_camera(scenes, position, max_depth):
    """yaw = pitch = 0 (UpdateCamera, Camera.cu:8-29): w = (-0, -0, 1), u = (-1, 0, 0), v = (0, -1, 0).  With an odd width
    the centre column has ndc.x = 0, i.e. direction.x = 0 and 1 / direction.x = inf; with an odd height the same for y."""
    return scenes.make_camera(position, 0.0, 0.0, max_depth)


def collapse_wide(nodes, root, count, width, node_dtype):
    """Re-pack a binary tree into nodes of up to `width` (<= 7) slots by pulling grandchildren up wherever they
    fit -- the reference's TraceRay loops over entry.count slots (Tracer.cu:323), and its own SAH builder can emit such
    nodes in principle.  Returns (new nodes, new root, new count).  Slots of a node are contiguous; a node with an odd
    number of slots exercises the tracer's lone-slot step."""
    MASK = 0x1FFFFFFF
    out = []

    def slots_of(index, cnt):
        return [nodes[index + i] for i in range(cnt)]

    def widen(sl):
        while True:                          # pull children up, level by level, while they fit
            res, budget = [], width - len(sl)
            for s in sl:
                typ, child, ccount = int(s["w28"]) >> 29, int(s["w28"]) & MASK, int(s["w12"]) >> 29
                if typ == 1 and ccount - 1 <= budget:
                    res.extend(slots_of(child, ccount))
                    budget -= ccount - 1
                else:
                    res.append(s)
            if len(res) == len(sl):
                return res
            sl = res

    def emit(sl, parent):
        base = len(out)
        out.extend([None] * len(sl))
        for i, s in enumerate(sl):
            rec = np.zeros((), node_dtype)
            rec["min"], rec["max"] = s["min"], s["max"]
            typ, child, ccount = int(s["w28"]) >> 29, int(s["w28"]) & MASK, int(s["w12"]) >> 29
            if typ == 1:
                kids = widen(slots_of(child, ccount))
                cbase = emit(kids, base + i)
                rec["w28"] = (1 << 29) | cbase
                rec["w12"] = (len(kids) << 29) | (parent & MASK)
            else:
                rec["w28"] = s["w28"]
                rec["w12"] = (ccount << 29) | (parent & MASK)
            out[base + i] = rec
        return base

    import sys
    sys.setrecursionlimit(10000)
    top = widen(slots_of(root, count))
    emit(top, 0)
    arr = np.array(out, dtype=node_dtype)
    return arr, 0, len(top)
