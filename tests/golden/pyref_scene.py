"""Inputs shared by tests/golden/make_golden_pyref.py (which runs the REFERENCE's Python modules on them) and tests/test_pyref_golden.py (which runs jnerf_amd's): small
network shapes, fixed points / rays / pixel indices, and a tiny NeRF-synthetic-layout data set on disk.  Everything is a pure function of constants and seeded numpy
generators, so both sides see identical bits."""
import json
import os
import numpy as np

NEUS_ENCODERS = dict(nerf_pos_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=4), nerf_dir_encoder=dict(type="FrequencyEncoder", multires=2, input_dims=3),
                     sdf_encoder=dict(type="FrequencyEncoder", multires=4, input_dims=3), rendering_encoder=dict(type="FrequencyEncoder", multires=2, input_dims=3))
NEUS_MODEL = dict(nerf_network=dict(D=3, W=24, output_ch=4, skips=[1], use_viewdirs=True),
                  sdf_network=dict(d_out=17, d_hidden=32, n_layers=4, skip_in=[2], bias=0.5, scale=1.0, geometric_init=True, weight_norm=True),
                  variance_network=dict(init_val=0.3),
                  rendering_network=dict(d_feature=16, mode="idr", d_out=3, d_hidden=24, n_layers=2, weight_norm=True, squeeze_out=True))
NEUS_RENDERER = dict(n_samples=8, n_importance=8, n_outside=4, up_sample_steps=2, perturb=1.0)
# tag -> (perturb, n_outside, cos_anneal_ratio, white background)
NEUS_RENDER_CASES = {"plain": (0.0, 0, 1.0, False), "perturbed_outside": (1.0, 4, 0.3, False), "white": (0.0, 0, 0.0, True)}
NEUS_COLOR_PROBE = np.array([[0.7, -0.4, 1.1]], dtype=np.float32)


def neus_points(n=24):
    rng = np.random.default_rng(21)
    return (rng.normal(size=(n, 3)) * 0.5).astype(np.float32)


def neus_dirs(n):
    rng = np.random.default_rng(22)
    d = rng.normal(size=(n, 3))
    return (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)


def neus_rays(n=6):
    """rays towards the unit sphere from outside it, with near / far from the sphere as neus_dataset.near_far_from_sphere computes them"""
    rng = np.random.default_rng(23)
    o = rng.normal(size=(n, 3))
    o = 2.5 * o / np.linalg.norm(o, axis=-1, keepdims=True)
    target = rng.normal(size=(n, 3)) * 0.25
    d = target - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    mid = -(o * d).sum(-1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32), (mid - 1.0).astype(np.float32), (mid + 1.0).astype(np.float32)


def ema_delta(step, i, shape):
    return (np.random.default_rng(100 + 10 * step + i).normal(size=tuple(shape)) * 0.1).astype(np.float32)


# ------------------------------------------------------------------ a NeRF-synthetic-layout data set (transforms_{train,val,test}.json + RGBA PNGs)
NERF_DATASET_ARGS = dict(aabb_scale=2)
NERF_W, NERF_H = 12, 10
NOVEL_POSE = np.array([[0.6, -0.48, 0.64, 2.1], [0.8, 0.36, -0.48, -1.3], [0.0, 0.8, 0.6, 1.7]], dtype=np.float32)


def _pose(rng):
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    q *= np.sign(np.linalg.det(q))
    m = np.eye(4)
    m[:3, :3] = q
    m[:3, 3] = rng.normal(size=3) * 2.0
    return m


def write_nerf_dataset(root):
    from PIL import Image
    rng = np.random.default_rng(31)
    for split, n in (("train", 3), ("val", 11), ("test", 2)):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in range(n):
            img = rng.integers(0, 256, size=(NERF_H, NERF_W, 4), dtype=np.uint8)
            Image.fromarray(img, "RGBA").save(os.path.join(root, split, "r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": _pose(rng).tolist()})
        meta = {"camera_angle_x": 0.6911112070083618, "frames": frames}
        if split == "test":
            meta.update(cx=NERF_W / 2 + 0.75, cy=NERF_H / 2 - 0.5, k1=0.01)
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump(meta, f)


def pixel_indices(n_images, H, W):
    rng = np.random.default_rng(41)
    idx = rng.integers(0, n_images * H * W, size=40)
    idx[:4] = [0, W - 1, H * W - 1, n_images * H * W - 1]
    return idx.astype(np.int64)


# ------------------------------------------------------------------ NGPNetworks (ngp_network.py): weights and encodings
def ngp_weights():
    """the five nn.Linear weights (out, in) of the density and colour networks; multiples of 2^-9 below 0.4 - exact in fp16, which FMLP's packing casts the last layer to"""
    rng = np.random.default_rng(51)
    shapes = [(64, 32), (16, 64), (64, 32), (64, 64), (3, 64)]
    return [(rng.integers(-200, 201, size=s) / 512.0).astype(np.float32) for s in shapes]


def ngp_inputs(n=40):
    rng = np.random.default_rng(52)
    feat = (rng.normal(size=(n, 32)) * 0.5).astype(np.float32)
    sh = (rng.normal(size=(n, 16)) * 0.5).astype(np.float32)
    dout = rng.normal(size=(n, 4)).astype(np.float32)
    return feat, sh, dout


# ------------------------------------------------------------------ DensityGridSampler orchestration (density_grid_sampler.py)
SAMPLER_CFG = dict(n_rays_per_batch=4096, cone_angle_constant=0.00390625, fp16=False, near_distance=0.2, n_training_steps=16, target_batch_size=1 << 18,
                   background_color=[0.0, 0.0, 0.0], m_training_step=0)
SAMPLER_CASES = {"lego": dict(aabb_scale=1, const_dt=True, block=5000000), "fox": dict(aabb_scale=4, const_dt=False, block=1500000),
                 "wide": dict(aabb_scale=32, const_dt=False, block=5000000)}
SAMPLER_STEPS = [0, 16, 240, 256, 272]
SAMPLER_MEASURED = [16 * 50000, 16 * 262144, 16 * 700000, 0, 16 * 3000, 123457]

LEVEL_TABLE_AABBS = [1, 2, 4, 8, 16, 32, 128]
NEUS_SCENE = dict(n_images=3, W=16, H=12, seed=3)


# ------------------------------------------------------------------ NeuSRunner.train end to end (runner/neus_runner.py)
NEUS_RUN_CASES = {"mask": dict(mask_weight=0.1, use_white_bkgd=False, n_outside=0), "womask": dict(mask_weight=0.0, use_white_bkgd=True, n_outside=4)}


def neus_run_cfg(root, mask_weight, use_white_bkgd, n_outside):
    return dict(dataset=dict(type="NeuSDataset", dataset_dir=root, render_cameras_name="cameras_sphere.npz", object_cameras_name="cameras_sphere.npz"),
                encoder=NEUS_ENCODERS, model=dict(type="NeuS", **NEUS_MODEL), render=dict(type="NeuSRenderer", **dict(NEUS_RENDERER, n_outside=n_outside)),
                optim=dict(type="PlainSGD", lr=0.02), base_exp_dir=os.path.join(root, "log"), learning_rate_alpha=0.05, end_iter=6, batch_size=16,
                validate_resolution_level=4, warm_up_end=2, anneal_end=4, use_white_bkgd=use_white_bkgd, save_freq=1000, val_freq=1000, val_mesh_freq=1000,
                report_freq=1000, igr_weight=0.1, mask_weight=mask_weight, fp16=False)

ORI_ENCODERS = dict(pos_encoder=dict(type="FrequencyEncoder", multires=4), dir_encoder=dict(type="FrequencyEncoder", multires=2))
ORI_MODEL = dict(D=4, W=32, skips=[2])


# ------------------------------------------------------------------ a small RENDERED NeRF-synthetic-layout scene for the end-to-end run of the reference's Runner
REFRUN = dict(W=40, H=40, n_train=6, n_val=3, n_test=1, camera_angle_x=0.8, steps=18, n_rays_per_batch=4096, target_batch_size=1 << 14)
_BALLS = [((0.0, 0.0, 0.0), 0.55, (0.9, 0.25, 0.2)), ((0.55, 0.35, 0.1), 0.3, (0.2, 0.8, 0.3)), ((-0.45, 0.3, -0.35), 0.28, (0.25, 0.35, 0.9))]


def _nerf_camera(theta, phi, radius=4.0):
    """camera-to-world in the NeRF / Blender convention (x right, y up, the camera looks along -z), on a sphere around the origin"""
    eye = radius * np.array([np.cos(phi) * np.cos(theta), np.cos(phi) * np.sin(theta), np.sin(phi)])
    back = eye / np.linalg.norm(eye)
    right = np.cross([0.0, 0.0, 1.0], back)
    right /= np.linalg.norm(right)
    up = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up, back, eye
    return m


def write_rendered_nerf_dataset(root, res=None):
    """three opaque shaded balls in front of a transparent background, RGBA PNGs + transforms_{train,val,test}.json (`res`: square images of that size instead of W x H)"""
    from PIL import Image
    W, H, fov = res or REFRUN["W"], res or REFRUN["H"], REFRUN["camera_angle_x"]
    focal = 0.5 * W / np.tan(0.5 * fov)
    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d_cam = np.stack([(xs + 0.5 - W / 2) / focal, -(ys + 0.5 - H / 2) / focal, -np.ones_like(xs, dtype=np.float64)], -1)
    light = np.array([0.5, 0.3, 0.8]) / np.linalg.norm([0.5, 0.3, 0.8])
    k = 0
    for split, n in (("train", REFRUN["n_train"]), ("val", REFRUN["n_val"]), ("test", REFRUN["n_test"])):
        os.makedirs(os.path.join(root, split), exist_ok=True)
        frames = []
        for i in range(n):
            m = _nerf_camera(2 * np.pi * (k * 0.37 % 1.0), np.deg2rad(15 + 40 * ((k * 0.61) % 1.0)))
            k += 1
            d = d_cam @ m[:3, :3].T
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            o = m[:3, 3]
            best = np.full((H, W), np.inf)
            rgb = np.zeros((H, W, 3))
            for c, r, col in _BALLS:
                oc = o - np.array(c)
                b = (d * oc).sum(-1)
                disc = b * b - (oc @ oc - r * r)
                t = -b - np.sqrt(np.maximum(disc, 0.0))
                hit = (disc > 0) & (t > 0) & (t < best)
                nrm = (o + t[..., None] * d - np.array(c)) / r
                shade = 0.3 + 0.7 * np.clip((nrm * light).sum(-1), 0.0, 1.0)
                rgb = np.where(hit[..., None], np.array(col) * shade[..., None], rgb)
                best = np.where(hit, t, best)
            alpha = np.isfinite(best)
            img = np.concatenate([rgb * alpha[..., None], alpha[..., None].astype(np.float64)], -1)
            Image.fromarray((img * 255 + 0.5).astype(np.uint8), "RGBA").save(os.path.join(root, split, "r_%d.png" % i))
            frames.append({"file_path": "./%s/r_%d" % (split, i), "transform_matrix": m.tolist()})
        with open(os.path.join(root, "transforms_%s.json" % split), "w") as f:
            json.dump({"camera_angle_x": fov, "frames": frames}, f)
REFRUN_SEEDS = dict(perm=4000, bg=5000, grid=31, mlp=5, probe=61)
# the runs of tests/golden/make_golden_refrun.py: name -> (fixture file, configuration on top of ngp_base.py)
REFRUN_CASES = {"lego": dict(file="golden_refrun_v1.npz", aabb_scale=None, const_dt=True, steps=REFRUN["steps"]),
                "cone": dict(file="golden_refrun_cone_v1.npz", aabb_scale=2, const_dt=False, steps=6),          # fox-style sampling: two cascades, cone stepping
                "render": dict(file="golden_refrun_render_v1.npz", aabb_scale=None, const_dt=True, steps=0, res=20)}     # no training: one occupancy refresh, then the inference path
# a camera-to-world pose in the NeRF convention for render_img_with_pose (it goes through matrix_nerf2ngp): on the camera sphere, between the training views
NOVEL_POSE_NGP = _nerf_camera(1.1, 0.5)[:3, :].astype(np.float32)
