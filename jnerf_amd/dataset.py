"""Datasets behind the reference's DATASETS registry.

NerfDataset          <- python/jnerf/dataset/dataset.py:17-262 (transforms JSON + images, nerf->ngp pose convention, per-batch rays)
SyntheticNerfDataset <- (ours) procedural scene with the same interface: cameras on a sphere, ground-truth RGBA rendered analytically on the
                        GPU at construction.  Used by bench.py / smoke() / tests — there is no network for datasets and /root/reference
                        (hence data/fox) does not exist on the GPU box.
Interface the runner/sampler/encoders read: next(ds) -> (img_ids i32[R], rays_o[R,3], rays_d[R,3], rgba[R,4]);
.n_images .resolution[W,H] .aabb_scale .aabb_range .metadata[n,11] .transforms_gpu[n,4,3] .focal_lengths[n,2] .image_data .batch_size
.generate_rays_total_test(img_ids,W,H) .generate_rays_with_pose(pose,W,H) .have_img"""
import json
import os
from math import pi, tan
import numpy as np
import torch
from . import ops
from .utils.config import get_cfg
from .utils.registry import DATASETS

NERF_SCALE = 0.33                       # dataset/dataset_util.py:11


def fov_to_focal_length(resolution, degrees):
    return 0.5 * resolution / tan(0.5 * degrees * pi / 180)


def read_image(path):
    from PIL import Image               # imageio / cv2 (dataset_util.py:30-36) are not needed: Pillow decodes the same files
    img = np.asarray(Image.open(path)).astype(np.float32)
    if img.ndim == 2:
        img = img[:, :, None]
    return img / 255.0


class _RayDatasetBase:
    """ray generation + batching shared by both datasets (dataset.py:57-66, 172-253)"""

    def _finalize(self, device, seed):
        self.device = torch.device(device)
        self.resolution = [self.W, self.H]
        self.transforms_gpu = torch.as_tensor(np.asarray(self.transforms_gpu, np.float32)).to(device).transpose(1, 2).contiguous()   # [n,4,3] = col-major 3x4 (dataset.py:165)
        self.focal_lengths = torch.as_tensor(np.asarray(self.focal_lengths, np.float32)).to(device).contiguous()
        self.metadata = torch.as_tensor(np.asarray(self.metadata, np.float32)).to(device).contiguous()
        self.aabb_range = (0.5 - self.aabb_scale / 2, 0.5 + self.aabb_scale / 2)                                    # dataset.py:154-157
        self.gen = torch.Generator(device=device)
        self.gen.manual_seed(seed)
        self.shuffle_index = None
        self.idx_now = 0

    def _reshuffle(self):
        self.shuffle_index = torch.randperm(self.n_images * self.H * self.W, device=self.device, generator=self.gen)
        self.idx_now = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.shuffle_index is None or self.idx_now + self.batch_size >= self.shuffle_index.shape[0]:
            self._reshuffle()
        index = self.shuffle_index[self.idx_now:self.idx_now + self.batch_size]
        self.idx_now += self.batch_size
        return self.generate_random_data(index, self.batch_size)

    def next_fused(self, bg, out=None):
        """(ours) next(ds) with the target compositing of runner.py:68 (rgb*a + bg*(1-a)) done inside the ray-generation kernel:
        -> (img_ids, rays_o, rays_d, target[R,3])"""
        if self.shuffle_index is None or self.idx_now + self.batch_size >= self.shuffle_index.shape[0]:
            self._reshuffle()
        index = self.shuffle_index[self.idx_now:self.idx_now + self.batch_size]
        self.idx_now += self.batch_size
        if index.is_cuda:
            index.record_stream(torch.cuda.current_stream())       # the permutation may have been drawn on another (side) stream
        return ops.generate_rays(index, self.W, self.H, self.focal_lengths, self.metadata, self.transforms_gpu, images=self.image_data, bg=bg, out=out)

    def generate_random_data(self, index, bs):
        img_ids, rays_o, rays_d, _ = ops.generate_rays(index, self.W, self.H, self.focal_lengths, self.metadata, self.transforms_gpu)
        rgb_tar = self.image_data.view(-1, 4)[index]
        return img_ids, rays_o, rays_d, rgb_tar

    def generate_rays_total_test(self, img_ids, W, H):
        """all rays of ONE image (img_ids is constant, runner.py:201-204); third value = pixel offsets"""
        img = int(img_ids[0].item()) if torch.is_tensor(img_ids) else int(img_ids)
        index = torch.arange(self.H * self.W, device=self.device, dtype=torch.int64) + img * self.H * self.W
        _, rays_o, rays_d, _ = ops.generate_rays(index, self.W, self.H, self.focal_lengths, self.metadata, self.transforms_gpu)
        return rays_o, rays_d, index - img * self.H * self.W

    def generate_rays_with_pose(self, pose, W, H):
        m = self.matrix_nerf2ngp(np.array(pose, np.float32)[:3, :].copy(), self.scale, self.offset)
        xf = torch.as_tensor(m).to(self.device).t().contiguous().view(1, 4, 3)
        index = torch.arange(self.H * self.W, device=self.device, dtype=torch.int64)
        _, rays_o, rays_d, _ = ops.generate_rays(index, self.W, self.H, self.focal_lengths[:1].contiguous(), self.metadata[:1].contiguous(), xf)
        return rays_o, rays_d

    def matrix_nerf2ngp(self, matrix, scale, offset):
        """dataset.py:255-262: negate per correct_pose, scale/offset the translation, cycle rows [1,2,0]"""
        matrix[:, 0] *= self.correct_pose[0]
        matrix[:, 1] *= self.correct_pose[1]
        matrix[:, 2] *= self.correct_pose[2]
        matrix[:, 3] = matrix[:, 3] * scale + offset
        return matrix[[1, 2, 0]]


@DATASETS.register_module()
class NerfDataset(_RayDatasetBase):
    def __init__(self, root_dir, batch_size, mode="train", H=0, W=0, correct_pose=[1, -1, -1], aabb_scale=None, scale=None, offset=None,
                 img_alpha=True, to_jt=True, have_img=True, preload_shuffle=True):
        self.root_dir, self.batch_size, self.mode = root_dir, batch_size, mode
        assert mode in ("train", "val", "test")
        self.H, self.W, self.correct_pose, self.aabb_scale = H, W, correct_pose, aabb_scale
        self.scale = NERF_SCALE if scale is None else scale
        self.offset = [0.5, 0.5, 0.5] if offset is None else offset
        self.img_alpha, self.have_img = img_alpha, have_img
        self.transforms_gpu, self.image_data, self.focal_lengths, self.n_images = [], [], [], 0
        cfg = get_cfg()
        self.load_data(cfg.device or "cuda", int(cfg.rank or 0))

    def load_data(self, device, rank):
        json_paths = []
        for root, _, files in os.walk(self.root_dir):
            for f in sorted(files):
                stem, ext = os.path.splitext(f)
                if ext == ".json" and (self.mode in stem or (self.mode == "train" and "val" in stem)):      # train includes val (dataset.py:77)
                    json_paths.append(os.path.join(root, f))
        json_data = None
        for p in json_paths:
            with open(p) as f:
                d = json.load(f)
            if json_data is None:
                json_data = d
            else:
                json_data["frames"] += d["frames"]
        assert json_data is not None, f"dataset is not found at {self.root_dir}"
        if "h" in json_data:
            self.H = int(json_data["h"])
        if "w" in json_data:
            self.W = int(json_data["w"])
        frames = json_data["frames"][::10] if self.mode == "val" else json_data["frames"]
        imgs = []
        for fr in frames:
            if self.have_img:
                path = os.path.join(self.root_dir, fr["file_path"])
                if not os.path.exists(path):
                    path += ".png"
                    if not os.path.exists(path):
                        continue                                                  # missing files are skipped (dataset.py:104-107)
                img = read_image(path)
                if self.H == 0 or self.W == 0:
                    self.H, self.W = int(img.shape[0]), int(img.shape[1])
                imgs.append(img)
            self.n_images += 1
            m = np.array(fr["transform_matrix"], np.float32)[:-1, :]
            self.transforms_gpu.append(self.matrix_nerf2ngp(m, self.scale, self.offset))
        meta = np.zeros(11, np.float32)
        meta[0:4] = [json_data.get(k, 0) for k in ("k1", "k2", "p1", "p2")]
        meta[4] = json_data.get("cx", self.W / 2) / self.W
        meta[5] = json_data.get("cy", self.H / 2) / self.H

        def fl(res, axis):
            if "fl_" + axis in json_data:
                return json_data["fl_" + axis]
            if "camera_angle_" + axis in json_data:
                return fov_to_focal_length(res, json_data["camera_angle_" + axis] * 180 / pi)
            return 0
        x_fl, y_fl = fl(self.W, "x"), fl(self.H, "y")
        if x_fl != 0:
            focal = [x_fl, y_fl if y_fl != 0 else x_fl]
        elif y_fl != 0:
            focal = [y_fl, y_fl]
        else:
            raise RuntimeError("Couldn't read fov.")
        meta[6:8] = focal
        self.metadata = np.repeat(meta[None], self.n_images, 0)
        self.focal_lengths = np.repeat(np.array([focal], np.float32), self.n_images, 0)
        if self.aabb_scale is None:
            self.aabb_scale = json_data.get("aabb_scale", 1)
        if self.have_img:
            data = torch.as_tensor(np.stack(imgs)).to(device)
            if self.img_alpha and data.shape[-1] == 3:
                data = torch.cat([data, torch.ones_like(data[..., :1])], -1)
        else:
            data = torch.zeros((self.n_images, self.H, self.W, 4), device=device)
        self.image_data = data.reshape(self.n_images, -1, 4).contiguous()
        self._finalize(device, seed=rank)


def camera_ring(n_images, radius, W, H, fov_deg, seed):
    """n pinhole cameras on a sphere around (0.5,0.5,0.5) looking at the centre; rows of the returned [n,3,4] are ngp-convention poses"""
    rng = np.random.default_rng(seed)
    out = np.zeros((n_images, 3, 4), np.float32)
    for i in range(n_images):
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        fwd = -v
        up = np.array([0.0, 0.0, 1.0]) if abs(v[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        out[i, :, 0], out[i, :, 1], out[i, :, 2], out[i, :, 3] = right, down, fwd, 0.5 + radius * v
    f = 0.5 * W / tan(0.5 * fov_deg * pi / 180)
    return out, f


def synthetic_field(p):
    """analytic density / colour of the procedural scene, p in world (ngp) coordinates [n,3] -> sigma [n], rgb [n,3]"""
    c = p - 0.5
    centres = torch.tensor([[0.0, 0.0, 0.0], [0.18, 0.1, -0.05], [-0.15, 0.12, 0.1], [0.02, -0.2, 0.12]], device=p.device)
    radii = torch.tensor([0.16, 0.09, 0.08, 0.07], device=p.device)
    cols = torch.tensor([[0.9, 0.3, 0.2], [0.2, 0.8, 0.3], [0.2, 0.3, 0.9], [0.9, 0.8, 0.2]], device=p.device)
    d = torch.linalg.norm(c[:, None, :] - centres[None], dim=-1)            # [n,4]
    s = torch.clamp((radii[None] - d) / 0.02, 0.0, 1.0)                       # soft shells
    sigma = 60.0 * s.sum(-1)
    w = s + 1e-6
    rgb = (w[..., None] * cols[None]).sum(1) / w.sum(-1, keepdim=True)
    stripes = 0.75 + 0.25 * torch.sin(40.0 * (c[:, 0:1] + c[:, 1:2] * 0.5 + c[:, 2:3] * 0.25))
    return sigma, torch.clamp(rgb * stripes, 0.0, 1.0)


def _hash01(ix, iy, iz, salt=0):
    """integer lattice -> [0, 1): a cheap integer hash (any fixed function would do - it only has to be the same for every ray that sees the cell)"""
    h = (ix * 73856093) ^ (iy * 19349663) ^ (iz * 83492791) ^ (salt * 2654435761)
    h = (h ^ (h >> 13)) * 1274126177
    h = h ^ (h >> 16)
    return (h & 0xFFFFFF).to(torch.float32) / float(1 << 24)


def bricks_field(p, d):
    """A lego-DIFFICULTY stand-in (NeRF-synthetic lego itself cannot be fetched here): a bulldozer-shaped hull (body, cabin, blade, two tracks with
    gaps, two thin pipes: ~2.5 % of the unit cube, ~5 % of the 128^3 occupancy cells) built from staggered bricks with mortar gaps and studs, hard surfaces, a palette
    colour per brick, a printed texture at about one pixel footprint, and view-dependent shading (diffuse + a specular lobe the degree-4 SH head cannot fully follow).
    p, d [n,3] (ngp world coordinates, unit directions) -> sigma [n], rgb [n,3]."""
    c = p - 0.5
    x, y, z = c[:, 0], c[:, 1], c[:, 2]

    def box(cx, cy, cz, hx, hy, hz):
        return ((x - cx).abs() < hx) & ((y - cy).abs() < hy) & ((z - cz).abs() < hz)
    body = box(0.0, 0.0, -0.025, 0.22, 0.12, 0.075)
    cabin = box(0.05, 0.0, 0.11, 0.08, 0.09, 0.06) & ~box(0.05, 0.0, 0.115, 0.07, 0.10, 0.035)          # a cabin with window openings through it
    blade = box(0.255, 0.0, -0.05, 0.015, 0.20, 0.07)
    track_gap = (torch.remainder(x + 0.25, 0.05) < 0.041)                                                 # track links with gaps between them
    tracks = (box(0.0, 0.15, -0.095, 0.25, 0.018, 0.045) | box(0.0, -0.15, -0.095, 0.25, 0.018, 0.045)) & track_gap
    hull = body | cabin | blade | tracks
    # staggered bricks: 0.05 x 0.05 x 0.03, every other layer shifted by half a brick; a 1.5e-3 mortar gap all round; ~12 % of the bricks missing
    bsx, bsy, bsz, gap = 0.05, 0.05, 0.03, 0.0015
    layer = torch.floor(z / bsz)
    xs = x + 0.5 * bsx * torch.remainder(layer, 2.0)
    ix, iy, iz = torch.floor(xs / bsx).to(torch.int64), torch.floor(y / bsy).to(torch.int64), layer.to(torch.int64)
    lx, ly, lz = xs - (ix.to(torch.float32) + 0.5) * bsx, y - (iy.to(torch.float32) + 0.5) * bsy, z - (iz.to(torch.float32) + 0.5) * bsz
    hsh = _hash01(ix, iy, iz)
    in_brick = (lx.abs() < 0.5 * bsx - gap) & (ly.abs() < 0.5 * bsy - gap) & (lz.abs() < 0.5 * bsz - gap)
    solid = hull & in_brick & (hsh > 0.12)
    # studs: 2 x 2 cylinders (radius 6e-3, height 5e-3) on top of every brick whose cell above is outside the hull - tested from the cell above
    below = _hash01(ix, iy, iz - 1) > 0.12
    sx, sy = torch.remainder(xs, 0.5 * bsx) - 0.25 * bsx, torch.remainder(y, 0.5 * bsy) - 0.25 * bsy
    stud = (~hull) & below & (lz < -0.5 * bsz + 0.005) & (sx * sx + sy * sy < 0.006 ** 2)
    zb = z - bsz                                                                                            # the stud belongs to the brick one layer down: that one must be in the hull
    stud = stud & (((x).abs() < 0.22) & (y.abs() < 0.12) & ((zb + 0.025).abs() < 0.075) | ((x - 0.05).abs() < 0.08) & (y.abs() < 0.09) & ((zb - 0.11).abs() < 0.06))
    # two thin pipes (radius 5e-3) from the cabin roof down to the blade
    def pipe(a, b):
        a = torch.tensor(a, device=p.device); b = torch.tensor(b, device=p.device)
        ab = b - a
        t = ((c - a) @ ab / (ab @ ab)).clamp(0.0, 1.0)
        return ((c - (a + t[:, None] * ab)) ** 2).sum(-1) < 0.005 ** 2
    pipes = pipe([0.05, 0.07, 0.17], [0.25, 0.16, 0.02]) | pipe([0.05, -0.07, 0.17], [0.25, -0.16, 0.02])
    occ = solid | stud | pipes
    sigma = 500.0 * occ.to(torch.float32)
    # ---- appearance: palette colour per brick, a printed texture (cells of 1/640: about one pixel footprint of the 800 x 800 views), face normal from the dominant
    # axis of the offset inside the brick, diffuse + specular under a fixed light
    palette = torch.tensor([[0.85, 0.12, 0.10], [0.95, 0.78, 0.10], [0.10, 0.35, 0.80], [0.12, 0.60, 0.25], [0.88, 0.88, 0.86], [0.15, 0.15, 0.17]], device=p.device)
    base = palette[(hsh * 5.999).to(torch.int64).clamp(0, 5)]
    base = torch.where(pipes[:, None], torch.tensor([0.6, 0.6, 0.65], device=p.device).expand_as(base), base)
    tex = 0.93 + 0.14 * _hash01(torch.floor(p[:, 0] * 640).to(torch.int64), torch.floor(p[:, 1] * 640).to(torch.int64), torch.floor(p[:, 2] * 640).to(torch.int64), salt=7)
    tex = tex * (0.9 + 0.1 * torch.sign(torch.sin(c[:, 0] * 400.0) * torch.sin(c[:, 1] * 400.0)))      # a fine checker print
    ax = torch.stack([lx.abs() / bsx, ly.abs() / bsy, lz.abs() / bsz], -1)
    dom = ax.argmax(-1)
    sgn = torch.sign(torch.stack([lx, ly, lz], -1).gather(1, dom[:, None])[:, 0])
    nrm = torch.nn.functional.one_hot(dom, 3).to(torch.float32) * sgn[:, None]
    light = torch.nn.functional.normalize(torch.tensor([0.4, 0.3, 0.86], device=p.device), dim=0)
    diffuse = 0.55 + 0.45 * (nrm @ light).clamp_min(0.0)
    half = torch.nn.functional.normalize(light[None] - d, dim=-1)
    spec = 0.45 * ((nrm * half).sum(-1).clamp_min(0.0)) ** 40
    rgb = (base * (tex * diffuse)[:, None] + spec[:, None]).clamp(0.0, 1.0)
    return sigma, rgb


@DATASETS.register_module()
class SyntheticNerfDataset(_RayDatasetBase):
    def __init__(self, batch_size, n_images=50, W=400, H=400, aabb_scale=4, radius=1.3, fov_deg=40.0, mode="train", seed=0, n_steps=None, scene="spheres", **_):
        self.scene = scene                   # "spheres": four soft striped spheres (rounds 1-2; trains to > 50 dB) | "bricks": the lego-difficulty stand-in (bricks_field)
        n_steps = n_steps or (640 if scene == "bricks" else 192)
        self.batch_size, self.mode, self.W, self.H, self.aabb_scale, self.n_images = batch_size, mode, W, H, aabb_scale, n_images
        self.scale, self.offset, self.correct_pose, self.have_img = NERF_SCALE, [0.5, 0.5, 0.5], [1, -1, -1], True
        cfg = get_cfg()
        device, rank = cfg.device or "cuda", int(cfg.rank or 0)
        poses, f = camera_ring(n_images, radius, W, H, fov_deg, seed + (1000 if mode != "train" else 0))
        self.transforms_gpu = poses
        self.focal_lengths = np.full((n_images, 2), f, np.float32)
        meta = np.zeros((n_images, 11), np.float32)
        meta[:, 4:6] = 0.5
        meta[:, 6:8] = f
        self.metadata = meta
        self._finalize(device, seed=rank)
        self.image_data = self._render_ground_truth(n_steps)

    @torch.no_grad()
    def _render_ground_truth(self, n_steps):
        out = torch.empty((self.n_images, self.H * self.W, 4), dtype=torch.float32, device=self.device)
        chunk = (1 << 16) if n_steps <= 256 else (1 << 13)
        t = torch.linspace(0.0, 1.0, n_steps, device=self.device)
        for i in range(self.n_images):
            index = torch.arange(self.H * self.W, device=self.device, dtype=torch.int64) + i * self.H * self.W
            _, o, d, _ = ops.generate_rays(index, self.W, self.H, self.focal_lengths, self.metadata, self.transforms_gpu)
            for s in range(0, o.shape[0], chunk):
                oo, dd = o[s:s + chunk], d[s:s + chunk]
                b = -((oo - 0.5) * dd).sum(-1)                                  # closest approach to the centre; scene radius < 0.4
                t0, t1 = (b - 0.4).clamp_min(0.0), b + 0.4
                ts = t0[:, None] + (t1 - t0)[:, None] * t[None]
                dt = ((t1 - t0) / (n_steps - 1))[:, None]
                p = oo[:, None, :] + ts[..., None] * dd[:, None, :]
                if self.scene == "bricks":
                    sigma, rgb = bricks_field(p.reshape(-1, 3), dd[:, None, :].expand_as(p).reshape(-1, 3))
                else:
                    sigma, rgb = synthetic_field(p.reshape(-1, 3))
                sigma, rgb = sigma.view(-1, n_steps), rgb.view(-1, n_steps, 3)
                alpha = 1.0 - torch.exp(-sigma * dt)
                T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha[:, :-1]], 1), 1)
                w = alpha * T
                a = w.sum(1, keepdim=True)
                c = (w[..., None] * rgb).sum(1) / a.clamp_min(1e-6)             # straight (un-premultiplied) colour, like a PNG with alpha
                out[i, s:s + chunk, :3] = c
                out[i, s:s + chunk, 3:] = a
        return out
