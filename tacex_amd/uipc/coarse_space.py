"""Coarse space of the two-level preconditioner of the gelpad Newton system (`tacex_fem_set_coarse_space`).

libuipc solves its Newton systems with a preconditioned CG (`linear_pcg`, uipc_sim.py:86-90); which preconditioner its backend uses
cannot be read here (un-vendored submodule).  This build adds an additive coarse-grid correction to the per-vertex 3x3 block
Jacobi: trilinear hat functions of a small grid over the mesh's bounding box (any tet mesh), Galerkin operator with the
REST-state matrix - constant per mesh and constraint set, so it is assembled and inverted on the host once."""
from __future__ import annotations

import numpy as np


def coarse_grid_dims(points: np.ndarray, target: int = 3) -> tuple[int, int, int]:
    """Cells per axis: `target` along the longest extent, proportionally fewer along the others, at least 1 (a thin pad gets ONE
    cell through its thickness)."""
    ext = np.ptp(np.asarray(points, np.float64), axis=0)
    ext = np.maximum(ext, 1e-300)
    return tuple(int(max(1, round(target * e / ext.max()))) for e in ext)


def build_coarse_space(points: np.ndarray, dims: tuple[int, int, int]):
    """(node (V,8) int32, weight (V,8) float64, num_coarse): the 8 trilinear weights of every vertex on a dims[0] x dims[1] x dims[2]
    cell grid over the bounding box (node id = (i * (ny + 1) + j) * (nz + 1) + k)."""
    P = np.asarray(points, np.float64)
    lo, ext = P.min(0), np.maximum(np.ptp(P, axis=0), 1e-300)
    n = np.asarray(dims, np.int64)
    t = (P - lo) / ext * n                      # grid coordinates in [0, n]
    c = np.minimum(np.floor(t).astype(np.int64), n - 1)
    f = t - c
    V = len(P)
    node = np.zeros((V, 8), np.int32)
    w = np.zeros((V, 8), np.float64)
    k = 0
    for a in (0, 1):
        for b in (0, 1):
            for d in (0, 1):
                i, j, l = c[:, 0] + a, c[:, 1] + b, c[:, 2] + d
                node[:, k] = ((i * (n[1] + 1) + j) * (n[2] + 1) + l).astype(np.int32)
                w[:, k] = (f[:, 0] if a else 1 - f[:, 0]) * (f[:, 1] if b else 1 - f[:, 1]) * (f[:, 2] if d else 1 - f[:, 2])
                k += 1
    return node, w, int((n[0] + 1) * (n[1] + 1) * (n[2] + 1))


def prolongation_matrix(node: np.ndarray, w: np.ndarray, nc: int) -> np.ndarray:
    """Dense (3V, 3 nc) prolongation (the same hat weight for the three components of a vertex)."""
    V = node.shape[0]
    Pn = np.zeros((V, nc))
    np.add.at(Pn, (np.repeat(np.arange(V), 8), node.reshape(-1)), w.reshape(-1))
    return np.kron(Pn, np.eye(3))


def coarse_operator_inverse(element_hessians: np.ndarray, tets: np.ndarray, mass: np.ndarray, constrained: np.ndarray, strength: float,
                            dt: float, node: np.ndarray, w: np.ndarray, nc: int) -> np.ndarray:
    """inverse of P^T A_0 P, A_0 = diag(m (1 + s c)) + dt^2 K_0, from the rest-state element Hessians (T,12,12) of vol * Psi (vertex-major
    12 = 4 vertices x xyz, as tacex_fem_element_terms returns them)."""
    V = len(mass)
    A = np.zeros((3 * V, 3 * V))
    dof = (np.asarray(tets, np.int64)[:, :, None] * 3 + np.arange(3)).reshape(len(tets), 12)
    np.add.at(A, (np.repeat(dof, 12, axis=1).reshape(-1), np.tile(dof, (1, 12)).reshape(-1)), (dt * dt * element_hessians).reshape(-1))
    A[np.arange(3 * V), np.arange(3 * V)] += np.repeat(mass * (1.0 + strength * np.asarray(constrained, np.float64)), 3)
    Pm = prolongation_matrix(node, w, nc)
    Ac = Pm.T @ A @ Pm
    Ac = 0.5 * (Ac + Ac.T)
    return np.linalg.inv(Ac)


def build_vertex_chains(points: np.ndarray, tets: np.ndarray, max_len: int = 16) -> list[list[int]]:
    """Vertex chains for the block-tridiagonal part of the preconditioner (`tacex_fem_set_chains`): the columns of vertices through
    the mesh's THIN direction (the axis of least extent - a gelpad's thickness, where the nearly incompressible material couples the
    layers most strongly).  Vertices that share their two other coordinates form a column, ordered along the thin axis; a column is
    cut wherever two consecutive vertices share no tet (the chain keeps only couplings the matrix has) and at `max_len`.  Works on any
    tet mesh: an unstructured one simply yields chains of one vertex (= plain block Jacobi)."""
    P = np.asarray(points, np.float64)
    T = np.asarray(tets, np.int64)
    ext = np.ptp(P, axis=0)
    ax = int(np.argmin(ext))
    oth = [a for a in range(3) if a != ax]
    scale = max(float(ext.max()), 1e-300)
    key = np.round(P[:, oth] / (1e-7 * scale)).astype(np.int64)
    adj = set()
    for a in range(4):
        for b in range(a + 1, 4):
            lo, hi = np.minimum(T[:, a], T[:, b]), np.maximum(T[:, a], T[:, b])
            adj.update(zip(lo.tolist(), hi.tolist()))
    order = np.lexsort((P[:, ax], key[:, 1], key[:, 0]))
    chains, cur = [], []
    for v in order.tolist():
        if cur and (key[v] == key[cur[-1]]).all() and (min(v, cur[-1]), max(v, cur[-1])) in adj and len(cur) < max_len:
            cur.append(v)
        else:
            if len(cur) > 1:
                chains.append(cur)
            cur = [v]
    if len(cur) > 1:
        chains.append(cur)
    return chains
