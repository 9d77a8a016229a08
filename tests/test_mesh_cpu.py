"""tools/extract_mesh.py's pipeline (jnerf_amd/mesh.py) on the CPU: an analytic field stands in for the trained network and a recording stub for the renderer, so the
lattice truncation, the iso-surface, the connected-cluster filter, the normals' orientation, the ray set-up of the colouring pass and the two PLY files are all
checked against closed forms.  (The model and the renderer themselves are the GPU tests' business.)"""
import os
import numpy as np
import pytest
import torch
from jnerf_amd import mesh
from jnerf_amd.utils.isosurface import read_ply

C_BIG, R_BIG = np.array([0.45, 0.5, 0.55]), 0.25
C_SMALL, R_SMALL = np.array([0.9, 0.9, 0.1]), 0.06


class _Field:
    """log-density `inside` in the two balls, -6 elsewhere; columns 0..2 are never read by the mesh code"""
    def __init__(self, inside=6.0):
        self.inside = inside
        self.calls = 0

    def __call__(self, pos, dirs):
        assert pos.shape == dirs.shape and float(dirs.abs().max()) == 0.0          # extract_mesh.py:57: zero directions
        self.calls += 1
        p = pos.double().numpy()
        ins = (np.linalg.norm(p - C_BIG, axis=-1) < R_BIG) | (np.linalg.norm(p - C_SMALL, axis=-1) < R_SMALL)
        out = torch.zeros((pos.shape[0], 4))
        out[:, -1] = torch.from_numpy(np.where(ins, self.inside, -6.0)).float()
        return out


class _Dataset:
    device = "cpu"
    aabb_scale = 4


class _Runner:
    def __init__(self, tmp, inside=6.0):
        self.model = _Field(inside)
        self.dataset = {"train": _Dataset()}
        self.save_path = str(tmp)
        self.background_color = [1.0, 0.5, 0.0]
        self.render_chunk = 1000
        self.rays = None

    def _render_rays(self, ids, o, d, chunk):
        assert ids.dtype == torch.int32 and ids.shape[0] == o.shape[0] and chunk == self.render_chunk
        self.rays = (o.clone(), d.clone())
        return 0.5 + 0.25 * d, torch.full((o.shape[0], 1), 0.75)


def test_extract_mesh_on_an_analytic_field(tmp_path):
    N = 64
    r = _Runner(tmp_path)
    verts, tris, colors = mesh.extract_mesh(r, resolution=N, log=lambda *a: None)
    cell = 1.0 / (N - 1)
    # only the big ball is left, and the surface is where the ball's surface is: integer lattice values 0 | 6 cut at 0.5 put every vertex 1/12 of a lattice edge
    # from its EMPTY end, and the tetrahedra's edges include the cubes' diagonals (up to sqrt(3) cells long)
    dist = np.linalg.norm(verts - C_BIG, axis=-1)
    assert (dist - R_BIG).min() > -0.25 * cell and (dist - R_BIG).max() < 1.75 * cell and len(tris) > 2000
    origin_v, origin_t, origin_c = read_ply(os.path.join(tmp_path, "mesh-origin.ply"))
    assert origin_c is None and len(origin_t) > len(tris)
    d_small = np.linalg.norm(origin_v - C_SMALL, axis=-1)
    assert (d_small < R_SMALL + cell).sum() > 50                               # the small ball IS in mesh-origin.ply
    # closed surface of the largest cluster: every edge is shared by exactly two triangles, Euler characteristic 2
    e = np.sort(np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all() and len(verts) - len(cnt) + len(tris) == 2
    # colouring rays: 0.2 outside along the outward normal, looking in, in world coordinates
    o, d = (t.double().numpy() for t in r.rays)
    radial = (verts - C_BIG) / dist[:, None]
    cosine = ((-d) * radial).sum(-1)                                            # -d is the outward normal; a voxel staircase has side walls, so only the hemisphere
    assert cosine.min() > 0.0 and cosine.mean() > 0.8                           # is certain vertex by vertex (the smoothed surface below is held to 0.95)
    np.testing.assert_allclose(np.linalg.norm(d, axis=-1), 1.0, atol=1e-6)
    start_unit = (o - 0.5) / _Dataset.aabb_scale + 0.5
    np.testing.assert_allclose(start_unit, verts - mesh.RAY_BACKOFF * d, atol=1e-6)
    assert (np.linalg.norm(start_unit - C_BIG, axis=-1) > R_BIG).all()                      # every ray starts OUTSIDE the object
    # colours = rgb + background * (1 - alpha), rounded as the reference rounds
    want = (0.5 + 0.25 * d) + np.array(r.background_color) * 0.25
    np.testing.assert_array_equal(colors, (want * 255 + 0.5).clip(0, 255).astype(np.uint8))
    cv, ct, cc = read_ply(os.path.join(tmp_path, "mesh-color.ply"))
    np.testing.assert_array_equal(cc, colors)
    np.testing.assert_array_equal(ct, tris)
    np.testing.assert_allclose(cv, verts)


def test_smoothed_surface_is_smoother_and_in_place(tmp_path):
    N = 64
    cell = 1.0 / (N - 1)
    rough, _, _ = mesh.extract_mesh(_Runner(tmp_path / "a"), resolution=N, log=lambda *a: None)
    rb = _Runner(tmp_path / "b")
    smooth, tris, _ = mesh.extract_mesh(rb, resolution=N, smooth=True, log=lambda *a: None)
    radial = (smooth - C_BIG) / np.linalg.norm(smooth - C_BIG, axis=-1, keepdims=True)
    assert (-rb.rays[1].double().numpy() * radial).sum(-1).min() > 0.95
    dr = np.linalg.norm(rough - C_BIG, axis=-1) - R_BIG
    ds = np.linalg.norm(smooth - C_BIG, axis=-1) - R_BIG
    assert abs(ds).max() < 1.5 * cell and ds.std() < 0.5 * dr.std() and len(tris) > 2000


def test_lattice_truncates_like_int_cast(tmp_path):
    """jt.maximum(out, 0).int(): log-density 0.99 is empty, 1.0 is occupied, 2.7 counts as 2 (extract_mesh.py:66)"""
    occ = mesh.occupancy_lattice(_Field(2.7), 16, "cpu", batch=1000)
    assert occ.dtype == np.int32 and occ.shape == (16, 16, 16) and set(np.unique(occ)) == {0, 2}
    axis = np.linspace(0, 1, 16)
    p = np.stack(np.meshgrid(axis, axis, axis, indexing="ij"), -1)
    np.testing.assert_array_equal(occ > 0, (np.linalg.norm(p - C_BIG, axis=-1) < R_BIG) | (np.linalg.norm(p - C_SMALL, axis=-1) < R_SMALL))      # [ix, iy, iz] order
    with pytest.raises(RuntimeError, match="no surface"):
        mesh.extract_mesh(_Runner(tmp_path, inside=0.99), resolution=16, log=lambda *a: None)


def test_slabs_cover_the_lattice_once():
    """a lattice larger than one slab (extract_mesh.py:44-46) is assembled from x-slabs in order"""
    old = mesh.LATTICE_CHUNK
    mesh.LATTICE_CHUNK = 16 * 16 * 4
    try:
        f = _Field()
        a = mesh.occupancy_lattice(f, 16, "cpu")
        assert f.calls == 4
    finally:
        mesh.LATTICE_CHUNK = old
    np.testing.assert_array_equal(a, mesh.occupancy_lattice(_Field(), 16, "cpu"))


def test_clusters_are_edge_connected():
    # two triangles sharing an edge + one touching them in a single vertex only + a far pair: Open3D's clusters are edge-connected
    t = np.array([[0, 1, 2], [1, 3, 2], [2, 4, 5], [6, 7, 8], [7, 9, 8], [8, 9, 10]])
    keep = mesh.largest_component(t, 11)
    np.testing.assert_array_equal(keep, t[3:])
    v, f = mesh.drop_unreferenced(np.arange(33, dtype=np.float64).reshape(11, 3), keep)
    assert len(v) == 5 and f.max() == 4 and (v[f] == np.arange(33).reshape(11, 3)[keep]).all()


def test_vertex_normals_of_an_octahedron():
    v = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float64)
    t = np.array([[0, 2, 4], [2, 1, 4], [1, 3, 4], [3, 0, 4], [2, 0, 5], [1, 2, 5], [3, 1, 5], [0, 3, 5]])
    np.testing.assert_allclose(mesh.vertex_normals(v, t), v, atol=1e-12)
