"""A procedural scene in the IDR / NeuS "DTU" layout (image/*.png, mask/*.png, cameras_sphere.npz with world_mat_i / scale_mat_i) for the NeuS tests and tools - the DTU
scans the reference's configs point at (dataset/dtu_scan24) cannot be downloaded here.  Two overlapping spheres, analytically ray-traced: Lambertian shading of a
position-dependent albedo over a dark textured backdrop, so that colour, silhouette (mask) and the SDF's zero set all have a known ground truth:
`scene_sdf(points)` is the exact signed distance in the normalised (unit-sphere) frame."""
import os
import numpy as np

# the object in the NORMALISED frame (inside the unit sphere): union of two spheres
# (deliberately NOT the sphere of radius 0.5 around the origin that IDR's geometric initialisation starts the SDF network from)
_SPHERES = [(np.array([-0.12, 0.05, 0.0]), 0.36), (np.array([0.3, 0.1, 0.15]), 0.26)]
_SCALE, _SHIFT = 2.0, np.array([0.3, -0.2, 0.1])        # scale_mat: normalised -> world


def scene_sdf(p):
    """exact signed distance of the union of the spheres (normalised frame), p [..., 3]"""
    return np.min(np.stack([np.linalg.norm(p - c, axis=-1) - r for c, r in _SPHERES], 0), 0)


def _look_at(eye, target=np.zeros(3)):
    fwd = target - eye
    fwd /= np.linalg.norm(fwd)
    up = np.array([0.0, 0.0, 1.0]) if abs(fwd[2]) < 0.95 else np.array([0.0, 1.0, 0.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.stack([right, down, fwd], 0)             # rows: camera x (right), y (down), z (forward) in world coordinates = world -> camera rotation


def _trace(o, d):
    """nearest hit of rays (normalised frame) with the spheres: (hit mask, depth, normal)"""
    best = np.full(o.shape[:-1], np.inf)
    normal = np.zeros_like(o)
    for c, r in _SPHERES:
        oc = o - c
        b = (oc * d).sum(-1)
        disc = b * b - ((oc * oc).sum(-1) - r * r)
        t = -b - np.sqrt(np.maximum(disc, 0.0))
        ok = (disc > 0) & (t > 0) & (t < best)
        best = np.where(ok, t, best)
        n = (o + t[..., None] * d - c) / r
        normal = np.where(ok[..., None], n, normal)
    return np.isfinite(best), best, normal


def make_scene(root, n_images=12, W=96, H=72, seed=0, focal=None):
    """writes the data set under `root`; returns a dict with the ground truth (K, poses in the normalised frame, images as float RGB, masks)"""
    from PIL import Image
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "image"), exist_ok=True)
    os.makedirs(os.path.join(root, "mask"), exist_ok=True)
    focal = focal or 1.4 * W
    K = np.array([[focal, 0.0, (W - 1) / 2.0 + 0.7], [0.0, focal * 1.01, (H - 1) / 2.0 - 0.4], [0.0, 0.0, 1.0]])
    scale_mat = np.eye(4)
    scale_mat[:3, :3] *= _SCALE
    scale_mat[:3, 3] = _SHIFT
    cams, truth = {}, {"K": K, "poses": [], "images": [], "masks": []}
    light = np.array([0.4, -0.5, 0.75])
    light /= np.linalg.norm(light)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pix = np.stack([xs, ys, np.ones_like(xs)], -1).astype(np.float64)
    for i in range(n_images):
        phi = 2 * np.pi * (i + 0.3 * rng.random()) / n_images
        theta = np.deg2rad(55 + 50 * rng.random())
        eye_n = 2.6 * np.array([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)])           # camera centre, normalised frame (outside the unit sphere)
        R = _look_at(eye_n)
        # world frame: X_w = s X_n + shift; a camera at s eye_n + shift looking with the same rotation sees the same image
        eye_w = _SCALE * eye_n + _SHIFT
        world_mat = np.eye(4)
        world_mat[:3, :3] = K @ R
        world_mat[:3, 3] = -K @ R @ eye_w
        cams["world_mat_%d" % i] = world_mat.astype(np.float32)
        cams["scale_mat_%d" % i] = scale_mat.astype(np.float32)
        d_cam = pix @ np.linalg.inv(K).T
        d = d_cam @ R                                    # camera -> world rotation = R^T; row-vector form: d_cam @ R
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        o = np.broadcast_to(eye_n, d.shape)
        hit, depth, normal = _trace(o, d)
        p = o + np.where(hit, depth, 0.0)[..., None] * d
        albedo = 0.55 + 0.4 * np.sin(6.0 * p + np.array([0.0, 2.0, 4.0]))
        shade = 0.25 + 0.75 * np.clip((normal * light).sum(-1), 0.0, 1.0)
        backdrop = 0.08 + 0.05 * np.sin(9.0 * d[..., :1] + 3.0 * d[..., 1:2]) * np.ones(3)
        rgb = np.where(hit[..., None], albedo * shade[..., None], backdrop).clip(0.0, 1.0)
        img8 = (rgb * 255.0 + 0.5).astype(np.uint8)
        Image.fromarray(img8).save(os.path.join(root, "image", "%03d.png" % i))
        Image.fromarray(np.repeat((hit * 255).astype(np.uint8)[..., None], 3, -1)).save(os.path.join(root, "mask", "%03d.png" % i))
        pose = np.eye(4)
        pose[:3, :3] = R.T
        pose[:3, 3] = eye_n
        truth["poses"].append(pose)
        truth["images"].append(img8)
        truth["masks"].append(hit)
    np.savez(os.path.join(root, "cameras_sphere.npz"), **cams)
    return truth
