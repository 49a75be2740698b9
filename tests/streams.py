"""Hand-made packed streams (SURVEY.md Appendix A layout) for tests that need a tree SHAPE no builder would produce.

A tree is nested 2-tuples; an int is a leaf holding that triangle.  Boxes are the exact float32 min/max union of the
children's (what BVHBuilder.cpp:53-76 computes), nodes are numbered in pre-order, miss links follow BVHBuilder.cpp:222-244."""
import numpy as np

END = 0xFFFFFFFF


def stream_from_tree(tree, tris):
    """tris: (P, 3, 3) float32 vertices.  Returns the packed (5P-2, 4) uint32 stream."""
    tris = np.ascontiguousarray(tris, np.float32)
    P = tris.shape[0]
    N = 2 * P - 1
    nodes = []                                         # (kind, payload, size, lo, hi) in pre-order

    def walk(t):
        at = len(nodes)
        nodes.append(None)
        if isinstance(t, tuple):
            l, r = walk(t[0]), walk(t[1])
            lo = np.minimum(nodes[l][3], nodes[r][3])
            hi = np.maximum(nodes[l][4], nodes[r][4])
            nodes[at] = ("inner", None, nodes[l][2] + nodes[r][2] + 1, lo, hi)
        else:
            v = tris[t]
            nodes[at] = ("leaf", int(t), 1, v.min(0), v.max(0))
        return at

    import sys
    sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
    walk(tree)
    assert len(nodes) == N, (len(nodes), N)
    out = np.zeros((5 * P - 2, 4), np.uint32)
    f = out.view(np.float32)
    seen = set()
    for i, (kind, prim, size, lo, hi) in enumerate(nodes):
        nxt = i + size if i + size < N else END
        if kind == "inner":
            f[2 * i, :3] = lo
            out[2 * i, 3] = END
            f[2 * i + 1, :3] = hi
        else:
            v0, v1, v2 = tris[prim]
            f[2 * i, :3] = v1 - v0
            out[2 * i, 3] = 2 * N + prim
            f[2 * i + 1, :3] = v2 - v0
            f[2 * N + prim, :3] = v0
            seen.add(prim)
        out[2 * i + 1, 3] = nxt
    assert len(seen) == P
    return out
