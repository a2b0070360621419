"""Small triangle meshes for the rigid mesh indenter (`UipcSim.set_indenter_mesh`, indenter kind 4): what the reference's scenes press
into the gelpad are rigid meshes (the rolling ball of envs/ball_rolling_uipc.py is one); these generators stand in for asset files."""
from __future__ import annotations

import numpy as np


def icosphere(radius: float = 1.0, subdivisions: int = 2):
    """(vertices (Nv,3), triangles (Nt,3)) of an icosphere: 20 * 4^subdivisions triangles, vertices ON the sphere."""
    t = (1.0 + 5.0**0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.asarray(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdivisions):
        mid, nf = {}, []

        def midpoint(a, b):
            k = (min(a, b), max(a, b))
            if k not in mid:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                mid[k] = len(v) - 1
            return mid[k]

        for a, b, c in f:
            ab, bc, ca = midpoint(a, b), midpoint(b, c), midpoint(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.asarray(v) * radius, np.asarray(f, np.int32)


def box(half_extents=(1.0, 1.0, 1.0)):
    """(vertices (8,3), triangles (12,3)) of an axis-aligned box centred at the origin: faces, edges and corners for the
    closest-feature cases of the point-triangle distance."""
    h = np.asarray(half_extents, np.float64)
    v = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64) * h
    q = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = []
    for a, b, c, d in q:
        t += [(a, b, c), (a, c, d)]
    return v, np.asarray(t, np.int32)
