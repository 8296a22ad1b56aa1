"""Synthetic structured meshes and state fields for the BASELINE.json configs.

Node numbering is lexicographic (x fastest); element order lexicographic.  Linear
elements are numbered counter-clockwise like the reference's shipped meshes
(tests/sm01/problem.mesh: elements `1 2 3`, `1 3 4`); quadratic elements use
Jive's perimeter order corner-mid-corner-mid.
"""
import numpy as np


def grid_quad4(nx, ny, lx=1.0, ly=1.0):
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    coords = np.empty(((nx + 1) * (ny + 1), 2))
    coords[:, 0] = np.tile(xs, ny + 1)
    coords[:, 1] = np.repeat(ys, nx + 1)
    i = np.arange(nx, dtype=np.int64)[None, :]
    j = np.arange(ny, dtype=np.int64)[:, None]
    n0 = (j * (nx + 1) + i).ravel()
    conn = np.stack([n0, n0 + 1, n0 + nx + 2, n0 + nx + 1], axis=1).astype(np.int32)
    return coords, conn


def grid_tri3(nx, ny, lx=1.0, ly=1.0):
    coords, q = grid_quad4(nx, ny, lx, ly)
    conn = np.empty((2 * q.shape[0], 3), dtype=np.int32)
    conn[0::2] = q[:, [0, 1, 2]]
    conn[1::2] = q[:, [0, 2, 3]]
    return coords, conn


def grid_hex8(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    xs = np.linspace(0.0, lx, nx + 1)
    ys = np.linspace(0.0, ly, ny + 1)
    zs = np.linspace(0.0, lz, nz + 1)
    npx, npy = nx + 1, ny + 1
    coords = np.empty((npx * npy * (nz + 1), 3))
    coords[:, 0] = np.tile(xs, npy * (nz + 1))
    coords[:, 1] = np.tile(np.repeat(ys, npx), nz + 1)
    coords[:, 2] = np.repeat(zs, npx * npy)
    i = np.arange(nx, dtype=np.int64)[None, None, :]
    j = np.arange(ny, dtype=np.int64)[None, :, None]
    k = np.arange(nz, dtype=np.int64)[:, None, None]
    n0 = ((k * npy + j) * npx + i).ravel()
    up = npx * npy
    conn = np.stack([n0, n0 + 1, n0 + npx + 1, n0 + npx,
                     n0 + up, n0 + up + 1, n0 + up + npx + 1, n0 + up + npx], axis=1).astype(np.int32)
    return coords, conn


def _quadratic_from_linear(coords, conn, edges):
    """Insert mid-edge nodes.  edges: list of (a, b) local corner pairs, in output perimeter order."""
    nel = conn.shape[0]
    ea = np.stack([conn[:, a] for a, _ in edges], axis=1).astype(np.int64)
    eb = np.stack([conn[:, b] for _, b in edges], axis=1).astype(np.int64)
    lo, hi = np.minimum(ea, eb), np.maximum(ea, eb)
    key = (lo * coords.shape[0] + hi).ravel()
    uniq, inv = np.unique(key, return_inverse=True)
    mid = (coords[(uniq // coords.shape[0])] + coords[(uniq % coords.shape[0])]) * 0.5
    mid_ids = (coords.shape[0] + inv).reshape(nel, len(edges)).astype(np.int32)
    return np.vstack([coords, mid]), mid_ids


def grid_quad8(nx, ny, lx=1.0, ly=1.0):
    c, q = grid_quad4(nx, ny, lx, ly)
    c2, m = _quadratic_from_linear(c, q, [(0, 1), (1, 2), (2, 3), (3, 0)])
    conn = np.stack([q[:, 0], m[:, 0], q[:, 1], m[:, 1], q[:, 2], m[:, 2], q[:, 3], m[:, 3]], axis=1)
    return c2, conn.astype(np.int32)


def grid_tri6(nx, ny, lx=1.0, ly=1.0):
    c, t = grid_tri3(nx, ny, lx, ly)
    c2, m = _quadratic_from_linear(c, t, [(0, 1), (1, 2), (2, 0)])
    conn = np.stack([t[:, 0], m[:, 0], t[:, 1], m[:, 1], t[:, 2], m[:, 2]], axis=1)
    return c2, conn.astype(np.int32)


def grid_hex20(nx, ny, nz, lx=1.0, ly=1.0, lz=1.0):
    c, h = grid_hex8(nx, ny, nz, lx, ly, lz)
    edges = [(0, 1), (1, 2), (2, 3), (3, 0), (0, 4), (1, 5), (2, 6), (3, 7), (4, 5), (5, 6), (6, 7), (7, 4)]
    c2, m = _quadratic_from_linear(c, h, edges)
    conn = np.stack([h[:, 0], m[:, 0], h[:, 1], m[:, 1], h[:, 2], m[:, 2], h[:, 3], m[:, 3],
                     m[:, 4], m[:, 5], m[:, 6], m[:, 7],
                     h[:, 4], m[:, 8], h[:, 5], m[:, 9], h[:, 6], m[:, 10], h[:, 7], m[:, 11]], axis=1)
    return c2, conn.astype(np.int32)


def distort(coords, conn, amp, seed=12345):
    """Jitter every node by amp * (local element size); keeps elements valid for amp < ~0.25."""
    rng = np.random.default_rng(seed)
    a, b = coords[conn[:, 0]], coords[conn[:, 1]]
    h = np.sqrt(((a - b) ** 2).sum(1)).min()
    return coords + amp * h * rng.uniform(-1.0, 1.0, coords.shape)


def shuffle_nodes(coords, conn, seed=12345):
    rng = np.random.default_rng(seed)
    perm = rng.permutation(coords.shape[0])          # new id -> old id
    inv = np.empty_like(perm)
    inv[perm] = np.arange(perm.size)
    return coords[perm], inv[conn].astype(np.int32), perm


def smooth_state(coords, ndofpn, seed=12345, amp=1e-3, noise=1e-6, notch=True, l0=0.015, phi_noise=0.05):
    """u = amp*(sin pi x cos pi y, cos pi x sin pi y[, sin pi z]) + noise; phi = notch profile.
    Interleaved node-major (dof = node*ndofpn + type), SURVEY.md 8d."""
    rng = np.random.default_rng(seed)
    n, rank = coords.shape
    s = np.zeros((n, ndofpn))
    x, y = coords[:, 0], coords[:, 1]
    s[:, 0] = amp * np.sin(np.pi * x) * np.cos(np.pi * y)
    s[:, 1] = amp * np.cos(np.pi * x) * np.sin(np.pi * y) * 1.3
    if rank == 3:
        s[:, 2] = amp * np.sin(np.pi * coords[:, 2]) * np.cos(np.pi * x)
    s[:, :rank] += noise * rng.uniform(-1.0, 1.0, (n, rank))
    if ndofpn > rank:
        mid = coords[:, rank - 1]
        phi = np.exp(-np.abs(mid - 0.5) / l0) * (x < 0.5) if notch else np.zeros(n)
        phi = phi + phi_noise * rng.uniform(0.0, 1.0, n)      # noise = phi_noise = 0: a function of the coordinates only
        s[:, rank] = np.clip(phi, 0.0, 0.98)
    return s.ravel()
