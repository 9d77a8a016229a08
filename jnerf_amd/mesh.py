"""Triangle mesh out of a trained Instant-NGP field - the job of the reference's tools/extract_mesh.py (lines 12-156), without PyMCubes / Open3D / plyfile (none is
installed here):

  1. the field's log-density on an N^3 lattice over the unit cube the encoder lives in, truncated to integers exactly as the reference does
     (`jt.maximum(out, 0).int()`, extract_mesh.py:56-66: a cell counts as occupied from log-density 1 on), evaluated slab by slab through `runner.model(pos, dir)`
     with zero directions;
  2. the 0.5 iso-surface (extract_mesh.py:76) - marching tetrahedra (utils/isosurface.py) in place of `mcubes.marching_cubes`; `smooth=True` stands in for
     `mcubes.smooth` (extract_mesh.py:73-74) with its "gaussian" method: a band-limited signed distance of the occupied set, Gaussian-filtered, surface at 0;
  3. only the largest edge-connected set of triangles is kept (Open3D's `cluster_connected_triangles`, extract_mesh.py:91-95), unreferenced vertices dropped;
  4. vertex normals as Open3D computes them (normalised sum of the unit normals of the adjacent triangles, extract_mesh.py:104-105);
  5. every vertex is coloured by rendering ONE ray through the normal renderer (sampler.sample -> model -> rays2rgb(inference), extract_mesh.py:127-141): the ray
     starts 0.2 in front of the surface and looks at it along the inward normal, in world coordinates `(p - 0.5) * aabb_scale + 0.5` (extract_mesh.py:121-122).
     (The reference reaches the inward direction through a mirror: it writes mesh-origin.ply with x and y exchanged, lets Open3D derive normals from the mirrored
     winding and exchanges the components back - lines 78-82, 107-115.  Here the orientation is explicit.)
  6. `mesh-origin.ply` (all components, no colour) and `mesh-color.ply` (largest component, uchar red/green/blue per vertex) under the runner's save_path.

Differences a user will notice: lattice coordinates are index / (N - 1) (the reference divides by N, extract_mesh.py:78, which shrinks the mesh by 1/N), and
mesh-origin.ply is NOT mirrored."""
import os
import numpy as np
import torch
from .utils.isosurface import marching_tetrahedra, write_ply

RAY_BACKOFF = 0.2                    # extract_mesh.py:121
LATTICE_CHUNK = 512 * 512 * 512      # extract_mesh.py:44: lattice points per slab


@torch.no_grad()
def occupancy_lattice(model, resolution, device, batch=4096 * 128):
    """int32 [N, N, N] (indexed [ix, iy, iz]) = trunc(max(log-density, 0)) at the lattice points linspace(0, 1, N)^3  (extract_mesh.py:41-70)"""
    N = int(resolution)
    step = max(min(LATTICE_CHUNK // (N * N), N), 1)
    assert N % step == 0, "the resolution must be a multiple of the slab thickness (extract_mesh.py:46)"
    axis = torch.linspace(0.0, 1.0, N, device=device)
    out = np.empty((N, N, N), dtype=np.int32)
    for k in range(0, N, step):
        x = axis[k:k + step]
        xyz = torch.stack(torch.meshgrid(x, axis, axis, indexing="ij"), -1).reshape(-1, 3)
        slab = torch.empty(xyz.shape[0], dtype=torch.int32, device=device)
        for i in range(0, xyz.shape[0], batch):
            pos = xyz[i:i + batch].contiguous()
            sigma = model(pos, torch.zeros_like(pos))[:, -1].float()
            slab[i:i + batch] = sigma.clamp_min(0.0).to(torch.int32)           # .int(): truncation toward zero
        out[k:k + step] = slab.view(step, N, N).cpu().numpy()
    return out


def smooth_occupancy(occupied, sigma=3.0):
    """a smooth signed field of a binary set, > 0 inside, surface at 0: signed Euclidean distance clipped to a band of 4 sigma, then a Gaussian of width sigma
    (the "gaussian" method of mcubes.smooth; its "constrained" method - an iterative solver PyMCubes picks for small volumes - is not reproduced)"""
    from scipy import ndimage
    occupied = np.asarray(occupied, dtype=bool)
    if not occupied.any() or occupied.all():
        return np.where(occupied, 1.0, -1.0)
    band = 4.0 * sigma
    inside = ndimage.distance_transform_edt(occupied)
    outside = ndimage.distance_transform_edt(~occupied)
    signed = np.clip(inside - 0.5, None, band) * occupied - np.clip(outside - 0.5, None, band) * (~occupied)
    return ndimage.gaussian_filter(signed, sigma=sigma, mode="nearest")


def largest_component(triangles, n_vertices):
    """the triangles of the largest EDGE-connected cluster (by triangle count, ties: the first) - Open3D cluster_connected_triangles + argmax (extract_mesh.py:91-93)"""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    t = np.asarray(triangles, dtype=np.int64)
    if len(t) == 0:
        return t
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    key = np.minimum(e[:, 0], e[:, 1]) * n_vertices + np.maximum(e[:, 0], e[:, 1])
    owner = np.tile(np.arange(len(t)), 3)
    order = np.argsort(key, kind="stable")
    key, owner = key[order], owner[order]
    same = key[1:] == key[:-1]                                                  # consecutive owners of one edge are neighbours (non-manifold edges chain up)
    a, b = owner[:-1][same], owner[1:][same]
    graph = coo_matrix((np.ones(len(a), dtype=np.int8), (a, b)), shape=(len(t), len(t)))
    _, label = connected_components(graph, directed=False)
    return t[label == np.argmax(np.bincount(label))]


def drop_unreferenced(vertices, triangles):
    """(vertices that some triangle uses, triangles re-indexed) - Open3D remove_unreferenced_vertices (extract_mesh.py:95)"""
    used = np.zeros(len(vertices), dtype=bool)
    used[triangles.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return vertices[used], remap[triangles]


def vertex_normals(vertices, triangles):
    """unit vertex normals = normalised sum of the unit normals of the adjacent triangles (Open3D compute_vertex_normals with normalisation)"""
    tri = vertices[triangles]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
    out = np.zeros_like(vertices, dtype=np.float64)
    for c in range(3):
        np.add.at(out, triangles[:, c], n)
    return out / np.maximum(np.linalg.norm(out, axis=-1, keepdims=True), 1e-30)


@torch.no_grad()
def vertex_colors(runner, vertices, outward):
    """uint8 [nv, 3]: one rendered ray per vertex, from 0.2 outside the surface along the inward normal (extract_mesh.py:107-144)"""
    ds = runner.dataset["train"]
    dev = ds.device
    d = torch.as_tensor(-outward, dtype=torch.float32, device=dev).contiguous()
    o = torch.as_tensor(vertices, dtype=torch.float32, device=dev) - d * RAY_BACKOFF
    o = ((o - 0.5) * float(ds.aabb_scale) + 0.5).contiguous()
    ids = torch.zeros((o.shape[0],), dtype=torch.int32, device=dev)
    rgb, alpha = runner._render_rays(ids, o, d, runner.render_chunk)
    rgb = rgb + torch.tensor(runner.background_color, dtype=torch.float32, device=dev) * (1 - alpha)
    return (rgb * 255 + 0.5).clamp(0, 255).to(torch.uint8).cpu().numpy()


def extract_mesh(runner, resolution=512, smooth=False, save_dir=None, log=print):
    """the whole of tools/extract_mesh.py after `runner.load_ckpt`; returns (vertices f32 [nv, 3] in the unit cube, triangles i32 [nt, 3], colours u8 [nv, 3])"""
    save_dir = save_dir or runner.save_path
    os.makedirs(save_dir, exist_ok=True)
    N = int(resolution)
    occ = occupancy_lattice(runner.model, N, runner.dataset["train"].device)
    if smooth:
        verts, tris = marching_tetrahedra(smooth_occupancy(occ > 0), 0.0)
    else:
        verts, tris = marching_tetrahedra(occ, 0.5)
    if len(tris) == 0:
        raise RuntimeError(f"no surface: {int((occ > 0).sum())} of {N ** 3} lattice points are occupied")
    verts = verts / (N - 1)
    write_ply(os.path.join(save_dir, "mesh-origin.ply"), verts, tris)
    log("mesh origin generated mesh-origin.ply (%d vertices, %d triangles)" % (len(verts), len(tris)))
    tris = largest_component(tris, len(verts))
    verts, tris = drop_unreferenced(verts, tris)
    normals = vertex_normals(verts, tris)                    # marching_tetrahedra orients triangles from occupied to empty: these point out of the object
    colors = vertex_colors(runner, verts, normals)
    write_ply(os.path.join(save_dir, "mesh-color.ply"), verts, tris, colors=colors)
    log("mesh color generated mesh-color.ply (%d vertices, %d triangles)" % (len(verts), len(tris)))
    return verts.astype(np.float32), tris.astype(np.int32), colors
