"""Iso-surface extraction on a regular lattice and a PLY writer - what the reference gets from PyMCubes (`mcubes.marching_cubes`, renderer.py:32) and trimesh
(`trimesh.Trimesh(...).export`, neus_runner.py:311-312; tools/extract_mesh.py), neither of which is installed here.

Marching TETRAHEDRA instead of marching cubes: every lattice cube is cut into six tetrahedra around its main diagonal and every tetrahedron contributes zero, one or
two triangles - no 256-entry case table, no ambiguous faces, the same surface up to the triangulation.  Only cubes that straddle the threshold are visited (numpy,
vectorised over cubes), vertices on shared edges are merged, triangles are oriented so that their normal points from values above the threshold to values below
(for u = -sdf: out of the object)."""
import numpy as np

# cube corners: bit 0 = +x, bit 1 = +y, bit 2 = +z
_CORNERS = np.array([[(c >> 0) & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.int64)
# six tetrahedra around the diagonal corner 0 - corner 7, walking the ring of the other six corners (each consecutive pair shares a cube edge or face diagonal)
_RING = [1, 3, 2, 6, 4, 5]
_TETS = np.array([[0, 7, _RING[i], _RING[(i + 1) % 6]] for i in range(6)], dtype=np.int64)


def marching_tetrahedra(u, threshold=0.0):
    """u: float [X, Y, Z] samples on a lattice.  Returns (vertices f64 [nv, 3] in lattice-index coordinates, triangles i64 [nt, 3])."""
    u = np.asarray(u, dtype=np.float64)
    X, Y, Z = u.shape
    if min(X, Y, Z) < 2:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    above = u > threshold
    # cubes whose eight corners are not all on one side
    cnt = np.zeros((X - 1, Y - 1, Z - 1), dtype=np.int8)
    for dx, dy, dz in _CORNERS:
        cnt += above[dx:X - 1 + dx, dy:Y - 1 + dy, dz:Z - 1 + dz]
    base = np.argwhere((cnt > 0) & (cnt < 8))                                  # [nc, 3]
    if len(base) == 0:
        return np.zeros((0, 3)), np.zeros((0, 3), dtype=np.int64)
    corner_idx = base[:, None, :] + _CORNERS[None, :, :]                        # [nc, 8, 3]
    corner_val = u[corner_idx[..., 0], corner_idx[..., 1], corner_idx[..., 2]]  # [nc, 8]
    flat_id = (corner_idx[..., 0] * Y + corner_idx[..., 1]) * Z + corner_idx[..., 2]

    def edge(p_id, p_val, q_id, q_val):
        t = (threshold - p_val) / (q_val - p_val)
        return p_id, q_id, t

    tris = []
    for tet in _TETS:
        ids, vals = flat_id[:, tet], corner_val[:, tet]                         # [nc, 4]
        ins = vals > threshold
        n_in = ins.sum(1)
        for k in (1, 3):                                                         # one vertex alone on its side: one triangle
            sel = n_in == k
            if not sel.any():
                continue
            i_s, v_s, in_s = ids[sel], vals[sel], ins[sel]
            lone = np.argmax(in_s if k == 1 else ~in_s, axis=1)
            others = np.array([[j for j in range(4) if j != l] for l in range(4)])[lone]      # [m, 3]
            r = np.arange(len(lone))
            a_id, a_val = i_s[r, lone], v_s[r, lone]
            es = [edge(a_id, a_val, i_s[r, others[:, j]], v_s[r, others[:, j]]) for j in range(3)]
            tris.append((es[0], es[1], es[2], a_id if k == 1 else None, a_id if k == 3 else None))
        sel = n_in == 2                                                          # two and two: a quadrilateral, two triangles
        if sel.any():
            i_s, v_s, in_s = ids[sel], vals[sel], ins[sel]
            order = np.argsort(~in_s, axis=1, kind="stable")                     # the two vertices above the threshold first
            r = np.arange(len(order))[:, None]
            i_o, v_o = i_s[r, order], v_s[r, order]
            a, b, c, d = (i_o[:, j] for j in range(4))
            va, vb, vc, vd = (v_o[:, j] for j in range(4))
            ac, ad, bd, bc = edge(a, va, c, vc), edge(a, va, d, vd), edge(b, vb, d, vd), edge(b, vb, c, vc)
            tris.append((ac, ad, bd, a, None))
            tris.append((ac, bd, bc, a, None))

    def unflatten(i):
        return np.stack([i // (Y * Z), (i // Z) % Y, i % Z], -1).astype(np.float64)

    # every triangle corner is a point on a lattice edge: (endpoint ids, interpolation parameter)
    all_pts, all_key, inside_ref, outside_ref = [], [], [], []
    for e0, e1, e2, in_id, out_id in tris:
        for p_id, q_id, t in (e0, e1, e2):
            pts = unflatten(p_id) + t[:, None] * (unflatten(q_id) - unflatten(p_id))
            lo, hi = np.minimum(p_id, q_id), np.maximum(p_id, q_id)             # an edge is shared by several tetrahedra: one vertex per lattice edge
            all_pts.append(pts)
            all_key.append(lo * (X * Y * Z) + hi)
        inside_ref.append(unflatten(in_id) if in_id is not None else None)
        outside_ref.append(unflatten(out_id) if out_id is not None else None)
    n_per = [len(t[0][0]) for t in tris]
    pts = np.concatenate(all_pts)
    key = np.concatenate(all_key)
    uniq, first, inverse = np.unique(key, return_index=True, return_inverse=True)
    vertices = pts[first]
    faces = []
    cursor = 0
    for (e0, e1, e2, in_id, out_id), m, ref_in, ref_out in zip(tris, n_per, inside_ref, outside_ref):
        f = np.stack([inverse[cursor:cursor + m], inverse[cursor + m:cursor + 2 * m], inverse[cursor + 2 * m:cursor + 3 * m]], -1)
        cursor += 3 * m
        tri = vertices[f]
        normal = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        centre = tri.mean(1)
        toward_out = (centre - ref_in) if ref_in is not None else (ref_out - centre)     # from the side above the threshold to the side below
        flip = (normal * toward_out).sum(-1) < 0
        f[flip] = f[flip][:, ::-1]
        faces.append(f)
    faces = np.concatenate(faces)
    faces = faces[(faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])]      # a value exactly on the threshold collapses an edge
    return vertices, faces.astype(np.int64)


def write_ply(path, vertices, triangles, colors=None):
    """binary little-endian PLY (what trimesh's export writes for a .ply path); `colors` uint8 [nv, 3] adds the uchar red / green / blue vertex properties that
    tools/extract_mesh.py:145-156 writes through plyfile"""
    triangles = np.asarray(triangles, dtype="<i4")
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    props = "property float x\nproperty float y\nproperty float z\n"
    if colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
        props += "property uchar red\nproperty uchar green\nproperty uchar blue\n"
    vert = np.empty(len(vertices), dtype=fields)
    v32 = np.asarray(vertices, dtype="<f4").reshape(-1, 3)
    vert["x"], vert["y"], vert["z"] = v32[:, 0], v32[:, 1], v32[:, 2]
    if colors is not None:
        c8 = np.asarray(colors, dtype=np.uint8).reshape(-1, 3)
        assert len(c8) == len(v32), "one colour per vertex"
        vert["red"], vert["green"], vert["blue"] = c8[:, 0], c8[:, 1], c8[:, 2]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex %d\n%selement face %d\nproperty list uchar int vertex_indices\nend_header\n"
              % (len(vert), props, len(triangles)))
    rec = np.empty(len(triangles), dtype=[("n", "u1"), ("v", "<i4", 3)])
    rec["n"] = 3
    rec["v"] = triangles
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(vert.tobytes())
        f.write(rec.tobytes())


def read_ply(path):
    """reads back what write_ply wrote: (vertices f32 [nv, 3], triangles i32 [nt, 3], colours u8 [nv, 3] or None)"""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply" and f.readline().strip() == b"format binary_little_endian 1.0"
        n_vert = n_face = 0
        vert_props = []
        section = None
        while True:
            line = f.readline().decode("ascii").strip()
            if line == "end_header":
                break
            words = line.split()
            if words[0] == "element":
                section = words[1]
                if section == "vertex":
                    n_vert = int(words[2])
                elif section == "face":
                    n_face = int(words[2])
            elif words[0] == "property" and section == "vertex":
                vert_props.append((words[2], {"float": "<f4", "uchar": "u1"}[words[1]]))
        vert = np.frombuffer(f.read(n_vert * np.dtype(vert_props).itemsize), dtype=vert_props)
        face = np.frombuffer(f.read(n_face * 13), dtype=[("n", "u1"), ("v", "<i4", 3)])
    assert (face["n"] == 3).all()
    xyz = np.stack([vert["x"], vert["y"], vert["z"]], -1)
    rgb = np.stack([vert["red"], vert["green"], vert["blue"]], -1) if "red" in vert.dtype.names else None
    return xyz, face["v"].copy(), rgb
