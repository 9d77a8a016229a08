"""NeuSDataset behind the DATASETS registry (python/jnerf/dataset/neus_dataset.py:40-181): the IDR / NeuS "DTU" layout - `image/*.png`, `mask/*.png` and a
`cameras_sphere.npz` with one world-to-image projection `world_mat_i` and one normalisation `scale_mat_i` per view (the object sits inside the unit sphere after
scale_mat^-1).  Images live on the device; rays are generated there.

The reference reads images with cv2 (BGR channel order) and splits P = K [R|t] with cv2.decomposeProjectionMatrix; cv2 is not installed here: Pillow reads the files
(channels flipped to the reference's BGR order, so colour networks and checkpoints stay interchangeable) and `decompose_projection` below does the RQ split."""
import os
from glob import glob
import numpy as np
import torch
from .utils.config import get_cfg
from .utils.registry import DATASETS


def decompose_projection(P):
    """P (3x4) = K [R | -R C]: K upper triangular with a positive diagonal, R a rotation, C the camera centre - what cv2.decomposeProjectionMatrix returns as
    (cameraMatrix, rotMatrix, transVect) with transVect = (C, 1) (neus_dataset.py:21-24).  RQ by QR of the row-reversed transpose."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:, :3]
    rev = np.eye(3)[::-1]
    q, r = np.linalg.qr((rev @ M).T)
    K = rev @ r.T @ rev
    R = rev @ q.T
    sign = np.diag(np.sign(np.diag(K)) + (np.diag(K) == 0))
    K, R = K @ sign, sign @ R                      # sign @ sign = I: the product is unchanged, the diagonal of K becomes positive
    C = -np.linalg.solve(M, P[:, 3])
    return K, R, np.concatenate([C, [1.0]])[:, None]


def load_K_Rt_from_P(filename, P=None):
    """neus_dataset.py:12-37 (IDR): 4x4 intrinsics and camera-to-world pose from a projection matrix (or from a text file holding one)"""
    if P is None:
        lines = open(filename).read().splitlines()
        if len(lines) == 4:
            lines = lines[1:]
        P = np.asarray([[float(v) for v in line.split(" ")[:4]] for line in lines], dtype=np.float32).squeeze()
    K, R, t = decompose_projection(P)
    K = K / K[2, 2]
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.transpose()
    pose[:3, 3] = (t[:3] / t[3])[:, 0]
    return intrinsics, pose


def _read_bgr(path):
    from PIL import Image
    a = np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a[..., ::-1])


@DATASETS.register_module()
class NeuSDataset:
    def __init__(self, dataset_dir, render_cameras_name, object_cameras_name):
        print("Load data: Begin")
        self.device = torch.device(get_cfg().device or "cuda")
        self.data_dir, self.render_cameras_name, self.object_cameras_name = dataset_dir, render_cameras_name, object_cameras_name
        self.camera_outside_sphere = True
        self.scale_mat_scale = 1.1
        camera_dict = np.load(os.path.join(self.data_dir, self.render_cameras_name))
        self.camera_dict = camera_dict
        self.images_lis = sorted(glob(os.path.join(self.data_dir, "image/*.png")))
        self.n_images = len(self.images_lis)
        assert self.n_images > 0, f"no images under {self.data_dir}/image"
        self.images = torch.from_numpy(np.stack([_read_bgr(p) for p in self.images_lis])).to(self.device).float() / 256.0     # [n, H, W, 3], the reference's /256
        self.masks_lis = sorted(glob(os.path.join(self.data_dir, "mask/*.png")))
        self.masks = torch.from_numpy(np.stack([_read_bgr(p) for p in self.masks_lis])).to(self.device).float() / 256.0
        self.world_mats_np = [camera_dict["world_mat_%d" % i].astype(np.float32) for i in range(self.n_images)]      # world -> image
        self.scale_mats_np = [camera_dict["scale_mat_%d" % i].astype(np.float32) for i in range(self.n_images)]      # unit sphere -> world
        intrinsics_all, pose_all = [], []
        for scale_mat, world_mat in zip(self.scale_mats_np, self.world_mats_np):
            intrinsics, pose = load_K_Rt_from_P(None, (world_mat @ scale_mat)[:3, :4])
            intrinsics_all.append(torch.from_numpy(intrinsics).float())
            pose_all.append(torch.from_numpy(pose).float())
        self.intrinsics_all = torch.stack(intrinsics_all).to(self.device)            # [n, 4, 4]
        self.intrinsics_all_inv = torch.linalg.inv(self.intrinsics_all)
        self.focal = self.intrinsics_all[0][0, 0]
        self.pose_all = torch.stack(pose_all).to(self.device)                        # [n, 4, 4] camera -> normalised world
        self.H, self.W = self.images.shape[1], self.images.shape[2]
        self.image_pixels = self.H * self.W
        # region of interest for mesh extraction: the object camera file's unit cube carried into this file's normalised frame (neus_dataset.py:85-92)
        corner_min, corner_max = np.array([-1.01, -1.01, -1.01, 1.0]), np.array([1.01, 1.01, 1.01, 1.0])
        object_scale_mat = np.load(os.path.join(self.data_dir, self.object_cameras_name))["scale_mat_0"]
        to_local = np.linalg.inv(self.scale_mats_np[0]) @ object_scale_mat
        self.object_bbox_min = (to_local @ corner_min[:, None])[:3, 0]
        self.object_bbox_max = (to_local @ corner_max[:, None])[:3, 0]
        print("Load data: End")

    def _pixel_grid(self, resolution_level):
        tx = torch.linspace(0, self.W - 1, self.W // resolution_level, device=self.device)
        ty = torch.linspace(0, self.H - 1, self.H // resolution_level, device=self.device)
        pixels_x, pixels_y = torch.meshgrid(tx, ty, indexing="ij")
        return torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], -1)      # [W', H', 3]

    def _rays_through(self, p, intrinsics_inv, rot, origin):
        p = torch.matmul(intrinsics_inv[None, None, :3, :3], p[:, :, :, None]).squeeze(-1)
        rays_v = p / p.square().sum(-1, keepdim=True).clamp_min(1e-6).sqrt()
        rays_v = torch.matmul(rot[None, None, :3, :3], rays_v[:, :, :, None]).squeeze(-1)
        rays_o = origin[None, None, :3].expand(rays_v.shape)
        return rays_o.transpose(0, 1), rays_v.transpose(0, 1)                        # [H', W', 3]

    def gen_rays_at(self, img_idx, resolution_level=1):
        """rays of one whole view in world space (neus_dataset.py:105-120)"""
        pose = self.pose_all[img_idx]
        return self._rays_through(self._pixel_grid(resolution_level), self.intrinsics_all_inv[img_idx], pose[:3, :3], pose[:3, 3])

    def gen_random_rays_at(self, img_idx, batch_size):
        """batch_size random pixels of one view: [rays_o, rays_v, colour, mask] rows (neus_dataset.py:122-138)"""
        img_idx = int(img_idx)
        pixels_x = torch.randint(0, self.W, [batch_size], device=self.device)
        pixels_y = torch.randint(0, self.H, [batch_size], device=self.device)
        color = self.images[img_idx][pixels_y, pixels_x]
        mask = self.masks[img_idx][pixels_y, pixels_x]
        point = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], -1).float()
        point = point @ self.intrinsics_all_inv[img_idx, :3, :3].T
        rays_v = point / point.square().sum(-1, keepdim=True).clamp_min(1e-6).sqrt()
        rays_v = rays_v @ self.pose_all[img_idx, :3, :3].T
        rays_o = self.pose_all[img_idx, None, :3, 3].expand(rays_v.shape)
        return torch.cat([rays_o, rays_v, color, mask[:, :1]], -1)                   # [batch_size, 10]

    def gen_rays_between(self, idx_0, idx_1, ratio, resolution_level=1):
        """rays of a view interpolated between two cameras: slerp of the rotations, lerp of the centres in camera space (neus_dataset.py:140-168)"""
        from scipy.spatial.transform import Rotation as Rot, Slerp
        pose_0 = np.linalg.inv(self.pose_all[idx_0].detach().cpu().numpy())
        pose_1 = np.linalg.inv(self.pose_all[idx_1].detach().cpu().numpy())
        slerp = Slerp([0, 1], Rot.from_matrix(np.stack([pose_0[:3, :3], pose_1[:3, :3]])))
        pose = np.eye(4, dtype=np.float32)
        pose[:3, :3] = slerp(ratio).as_matrix()
        pose[:3, 3] = ((1.0 - ratio) * pose_0 + ratio * pose_1)[:3, 3]
        pose = torch.from_numpy(np.linalg.inv(pose).astype(np.float32)).to(self.device)
        return self._rays_through(self._pixel_grid(resolution_level), self.intrinsics_all_inv[0], pose[:3, :3], pose[:3, 3])

    def near_far_from_sphere(self, rays_o, rays_d):
        """the ray's closest approach to the origin +- 1: the unit sphere's extent along the ray (neus_dataset.py:170-176)"""
        a = (rays_d ** 2).sum(-1, keepdim=True)
        b = 2.0 * (rays_o * rays_d).sum(-1, keepdim=True)
        mid = 0.5 * (-b) / a
        return mid - 1.0, mid + 1.0

    def image_at(self, idx, resolution_level):
        """the view as uint8 BGR at reduced resolution (cv.resize there; Pillow's bilinear here)"""
        from PIL import Image
        img = Image.open(self.images_lis[idx]).convert("RGB").resize((self.W // resolution_level, self.H // resolution_level), Image.BILINEAR)
        return np.asarray(img, dtype=np.uint8)[..., ::-1].clip(0, 255)
