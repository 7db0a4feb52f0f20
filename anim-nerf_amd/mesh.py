"""Mesh extraction behind the sigma grid (extract_mesh.py:159-173 of the reference): the level set of the thresholded density
volume as an indexed triangle mesh, the rescale of `mcubes_to_world` (extract_mesh.py:37-47) and the .obj export.

The reference calls PyMCubes (`mcubes.marching_cubes(-sigmas, 0.)`), which is not in this image, so this restates the
algorithm itself (Lorensen & Cline's marching cubes) — PARITY-UNPINNED against PyMCubes' output, checked by properties:
  * the vertex SET of marching cubes is determined by the volume alone — one vertex per grid edge whose ends straddle the
    level, at the linear interpolation point — and is what `marching_cubes` returns (order: grid point, then axis);
  * the triangles come from a 256-case table GENERATED here (`case_table`), not typed in: on every cube face the crossing
    points are joined by a rule that looks at the face's four corner signs only (an ambiguous face cuts off its inside
    corners), so two cubes sharing a face always agree and the surface is closed wherever the level set is.
The two passes over the grid are HIP kernels (csrc/mesh.hip); the scans between them are torch.cumsum.
"""
from __future__ import annotations

from functools import lru_cache

import numpy as np
import torch

# corner c of the unit cube sits at (c & 1, (c >> 1) & 1, (c >> 2) & 1) along array axes (0, 1, 2)
CORNERS = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)
# edge e joins corners EDGES[e]; it runs along axis EDGE_AXIS[e] from the grid point at offset EDGE_BASE[e]
EDGES = [(a, b) for a in range(8) for b in range(a + 1, 8) if bin(a ^ b).count("1") == 1]
EDGE_AXIS = [int(np.log2(a ^ b)) for a, b in EDGES]
EDGE_BASE = [tuple(int(v) for v in CORNERS[a]) for a, b in EDGES]
MAX_TRIS = 8


def _faces():
    """the six faces as corner cycles, counter-clockwise seen from OUTSIDE the cube"""
    out = []
    for axis in range(3):
        for side in (0, 1):
            u, v = [a for a in range(3) if a != axis]
            cyc = []
            for du, dv in ((0, 0), (1, 0), (1, 1), (0, 1)):
                p = [0, 0, 0]
                p[axis], p[u], p[v] = side, du, dv
                cyc.append(p[0] | (p[1] << 1) | (p[2] << 2))
            n = np.cross(CORNERS[cyc[1]] - CORNERS[cyc[0]], CORNERS[cyc[3]] - CORNERS[cyc[0]])   # normal of the cycle
            outward = np.zeros(3, dtype=np.int64)
            outward[axis] = 1 if side else -1
            if np.dot(n, outward) < 0:
                cyc = cyc[::-1]
            out.append(cyc)
    return out


def _share_a_face(e1, e2):
    """both cube edges lie in one face of the cube"""
    pts = CORNERS[list(EDGES[e1]) + list(EDGES[e2])]
    return any((pts[:, a] == pts[0, a]).all() for a in range(3))


def _triangulate(loop):
    """Triangles (same orientation as the loop) of a polygon of crossing points such that NO DIAGONAL joins two points of one
    cube face: a loop that crosses an ambiguous face twice has non-consecutive vertices in that face's plane, and a diagonal
    between them would lie in the face — where the neighbouring cube may put one too (two sheets sharing an edge: the
    surface would stop being a 2-manifold there).  Loops have at most 7 vertices: plain search over the triangulations."""
    n = len(loop)
    if n == 3:
        return [tuple(loop)]

    def ok(i, j):                                        # polygon edge, or a diagonal that stays inside the cube
        return (j - i) % n in (1, n - 1) or not _share_a_face(loop[i], loop[j])

    def solve(i, j):                                     # triangulations of the sub-polygon i, i+1, ..., j (indices ascending)
        if j - i < 2:
            return []
        for m in range(i + 1, j):
            if ok(i, m) and ok(m, j):
                left, right = solve(i, m), solve(m, j)
                if left is not None and right is not None:
                    return left + [(loop[i], loop[m], loop[j])] + right
        return None
    for shift in range(n):                               # (the closing edge i-j of the top call is a polygon edge for any rotation)
        rot = loop[shift:] + loop[:shift]
        loop_saved, loop = loop, rot
        res = solve(0, n - 1)
        loop = loop_saved
        if res is not None:
            return res
    raise AssertionError(("no triangulation without an in-face diagonal", loop))


@lru_cache(maxsize=None)
def case_table():
    """(n_tris[256] uint8, tris[256, MAX_TRIS, 3] int8 of edge ids).  Bit c of the case index is set where corner c is INSIDE
    (value < level).  Triangles are oriented with the normal pointing from inside to outside."""
    eid = {frozenset(e): i for i, e in enumerate(EDGES)}
    faces = _faces()
    n_tris = np.zeros(256, dtype=np.uint8)
    tris = np.full((256, MAX_TRIS, 3), -1, dtype=np.int8)
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        nxt = {}                                         # directed segments: edge id -> edge id
        for cyc in faces:
            s = [inside[c] for c in cyc]
            cross = [i for i in range(4) if s[i] != s[(i + 1) % 4]]          # face edge i joins cyc[i], cyc[i+1]
            def fe(i):
                return eid[frozenset((cyc[i % 4], cyc[(i + 1) % 4]))]
            # Orientation: seen from outside the cube, a directed segment runs counter-clockwise around the inside corners
            # it cuts off; the loops these segments close into, fanned into triangles, then have their normals pointing
            # out of the inside region (checked on a single inside corner and on a sphere's volume in tests/test_mesh.py).
            if len(cross) == 2:
                i, j = cross
                # between edge i and edge j (going ccw from i+1 to j) the corners are all of one kind
                if s[(i + 1) % 4]:                       # the corners between edge i and edge j (ccw) are the inside ones
                    nxt[fe(i)] = fe(j)
                else:
                    nxt[fe(j)] = fe(i)
            elif len(cross) == 4:                        # ambiguous face: cut off each inside corner on its own
                for c in range(4):
                    if s[c]:                             # corner cyc[c] lies between face edges c-1 and c
                        nxt[fe(c - 1)] = fe(c)
        seen, k = set(), 0
        for start in sorted(nxt):
            if start in seen:
                continue
            loop, e = [], start
            while e not in seen:
                seen.add(e)
                loop.append(e)
                e = nxt[e]
            assert e == start and len(loop) >= 3, (case, loop)
            for tri in _triangulate(loop):
                tris[case, k] = tri
                k += 1
        assert len(seen) == len(nxt) and k <= MAX_TRIS, (case, k)
        n_tris[case] = k
    return n_tris, tris


def marching_cubes(volume: torch.Tensor, level: float = 0.0):
    """(vertices[V,3] float32 in index coordinates, triangles[T,3] int64) of the surface volume == level; volume[N0,N1,N2]
    float32 on the GPU.  `mcubes.marching_cubes(volume, level)`'s role (extract_mesh.py:165)."""
    from . import ops
    return ops.marching_cubes(volume, level)


def mcubes_to_world(vertices, N, x_range, y_range, z_range):
    """extract_mesh.py:37-47, as written there: index coordinates / N (not N - 1), the first two axes swapped — the grid comes
    from np.meshgrid's default 'xy' indexing (create_grid, :27-35), so array axis 0 runs over y."""
    v = np.asarray(vertices, dtype=np.float64) / N
    out = np.empty_like(v)
    out[:, 0] = (y_range[1] - y_range[0]) * v[:, 1] + y_range[0]
    out[:, 1] = (x_range[1] - x_range[0]) * v[:, 0] + x_range[0]
    out[:, 2] = (z_range[1] - z_range[0]) * v[:, 2] + z_range[0]
    return out


def export_obj(vertices, triangles, path):
    """mcubes.export_obj's layout: `v x y z` lines, then 1-based `f a b c` lines."""
    with open(path, "w") as f:
        for v in np.asarray(vertices):
            f.write(f"v {v[0]} {v[1]} {v[2]}\n")
        for t in np.asarray(triangles):
            f.write(f"f {int(t[0]) + 1} {int(t[1]) + 1} {int(t[2]) + 1}\n")


def gaussian_smooth(volume: torch.Tensor, sigma: float = 1.0) -> torch.Tensor:
    """Separable Gaussian filter of the volume (reflecting borders): what `--smooth` applies before the level set is taken.
    The reference calls `mcubes.smooth` there (extract_mesh.py:162-163; PyMCubes is absent): parity-unpinned."""
    r = max(1, int(round(3 * sigma)))
    x = torch.arange(-r, r + 1, dtype=torch.float32, device=volume.device)
    k = torch.exp(-0.5 * (x / sigma) ** 2)
    k = k / k.sum()
    v = volume[None, None]
    for axis in range(3):
        shape = [1, 1, 1, 1, 1]
        shape[2 + axis] = -1
        pad = [0, 0, 0, 0, 0, 0]
        pad[2 * (2 - axis)] = pad[2 * (2 - axis) + 1] = r
        v = torch.nn.functional.conv3d(torch.nn.functional.pad(v, pad, mode="replicate"), k.view(shape))
    return v[0, 0]
