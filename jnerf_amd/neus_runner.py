"""NeuSRunner (python/jnerf/runner/neus_runner.py:21-315): the training / validation driver of projects/neus - one image per iteration, `batch_size` random rays of it,
L1 colour loss + eikonal term (`igr_weight`) + optional mask loss, warm-up then cosine learning-rate schedule, cosine annealing of the opacity's ray/normal term,
periodic validation images / normal maps / depth maps and iso-surface meshes.

Same constructor, attributes and methods as the reference (`train`, `validate_image`, `validate_mesh`, `render_novel_image`, `save_checkpoint`, `load_checkpoint`,
`get_cos_anneal_ratio`, `update_learning_rate`, `get_image_perm`); checkpoints are `jt.save` containers (utils/jittor_pickle.py: `{'neus': state_dict, 'iter_step'}`,
neus_runner.py:151-161), images are written with Pillow and meshes with utils/isosurface.py where the reference uses cv2 / trimesh (not installed here)."""
import logging
import os
import numpy as np
import torch
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, DATASETS, OPTIMS, SAMPLERS
from .neus_network import safe_clip
from . import neus_network, neus_renderer, neus_dataset, optim  # noqa: F401  (registers NeuS / NeuSRenderer / NeuSDataset / Adam)


def _jet(x):
    """uint8 [..] -> BGR uint8 [.., 3] of the 'jet' colour map (cv.applyColorMap(., cv.COLORMAP_JET) there)"""
    v = x.astype(np.float32) / 255.0
    r = np.clip(1.5 - np.abs(4.0 * v - 3.0), 0, 1)
    g = np.clip(1.5 - np.abs(4.0 * v - 2.0), 0, 1)
    b = np.clip(1.5 - np.abs(4.0 * v - 1.0), 0, 1)
    return (np.stack([b, g, r], -1) * 255.0 + 0.5).astype(np.uint8)


def _imwrite_bgr(path, img):
    from PIL import Image
    img = np.asarray(img)
    img = img.clip(0, 255).astype(np.uint8)
    Image.fromarray(np.ascontiguousarray(img[..., ::-1]) if img.ndim == 3 else img).save(path)


class NeuSRunner:
    def __init__(self, mode="train", is_continue=False):
        self.cfg = cfg = get_cfg()
        self.base_exp_dir = cfg.base_exp_dir
        os.makedirs(self.base_exp_dir, exist_ok=True)
        self.iter_step = 0
        # training parameters (neus_runner.py:33-44)
        self.end_iter, self.save_freq, self.report_freq = cfg.end_iter, cfg.save_freq, cfg.report_freq
        self.val_freq, self.val_mesh_freq, self.batch_size = cfg.val_freq, cfg.val_mesh_freq, cfg.batch_size
        self.validate_resolution_level = cfg.validate_resolution_level
        self.learning_rate_alpha = cfg.learning_rate_alpha
        self.use_white_bkgd = cfg.use_white_bkgd
        self.warm_up_end, self.anneal_end = cfg.warm_up_end, cfg.anneal_end
        self.igr_weight, self.mask_weight = cfg.igr_weight, cfg.mask_weight
        self.is_continue, self.mode = is_continue, mode
        self.model_list, self.writer = [], None

        self.dataset = build_from_cfg(cfg.dataset, DATASETS)
        cfg.dataset_obj = self.dataset
        self.neus_network = build_from_cfg(cfg.model, NETWORKS)
        self.renderer = build_from_cfg(cfg.render, SAMPLERS)
        self.renderer.set_neus_network(self.neus_network)
        self.learning_rate = cfg.optim.lr
        self.optimizer = build_from_cfg(cfg.optim, OPTIMS, params=self.neus_network.parameters())
        self.device = self.dataset.device

        latest_model_name = None
        if is_continue:
            ckpt_dir = os.path.join(self.base_exp_dir, "checkpoints")
            names = sorted(n for n in os.listdir(ckpt_dir) if n.endswith("pkl") and int(n[5:-4]) <= self.end_iter)
            latest_model_name = names[-1] if names else None
        if latest_model_name is not None:
            logging.info("Find checkpoint: {}".format(latest_model_name))
            self.load_checkpoint(latest_model_name)

    # ------------------------------------------------------------------------------------------------------------------ training (neus_runner.py:76-137)
    def train_step(self, image_index):
        """one iteration on `batch_size` random rays of one view; returns the loss terms (the loop body of the reference's train())"""
        data = self.dataset.gen_random_rays_at(image_index, self.batch_size)
        rays_o, rays_d, true_rgb, mask = data[:, :3], data[:, 3:6], data[:, 6:9], data[:, 9:10]
        near, far = self.dataset.near_far_from_sphere(rays_o, rays_d)
        background_rgb = torch.ones([1, 3], device=data.device) if self.use_white_bkgd else None
        mask = (mask > 0.5).float() if self.mask_weight > 0.0 else torch.ones_like(mask)
        mask_sum = mask.sum() + 1e-5
        render_out = self.renderer.render(rays_o, rays_d, near, far, background_rgb=background_rgb, cos_anneal_ratio=self.get_cos_anneal_ratio())
        color_fine_loss = ((render_out["color_fine"] - true_rgb) * mask).abs().sum() / mask_sum
        eikonal_loss = render_out["gradient_error"]
        # (the reference feeds the clipped opacity sum to binary_cross_entropy_with_logits, neus_runner.py:108 - kept as written)
        mask_loss = torch.nn.functional.binary_cross_entropy_with_logits(safe_clip(render_out["weight_sum"], 1e-3, 1.0 - 1e-3), mask)
        loss = color_fine_loss + eikonal_loss * self.igr_weight + mask_loss * self.mask_weight
        self.optimizer.zero_grad()
        self.optimizer.backward(loss)
        self.optimizer.step()
        self.iter_step += 1
        return {"loss": loss.detach(), "color_loss": color_fine_loss.detach(), "eikonal_loss": eikonal_loss.detach(), "mask_loss": mask_loss.detach(),
                "s_val": render_out["s_val"].mean().detach()}

    def train(self):
        self.update_learning_rate()
        res_step = self.end_iter - self.iter_step
        image_perm = self.get_image_perm()
        for _ in range(res_step):
            out = self.train_step(image_perm[self.iter_step % len(image_perm)])
            if self.iter_step % self.report_freq == 0:
                print(self.base_exp_dir)
                print("iter:{:8>d} loss = {} lr={}".format(self.iter_step, float(out["loss"]), self.optimizer.param_groups[0]["lr"]))
            if self.iter_step % self.save_freq == 0:
                self.save_checkpoint()
            if self.iter_step % self.val_freq == 0:
                self.validate_image()
            if self.iter_step % self.val_mesh_freq == 0:
                self.validate_mesh()
            self.update_learning_rate()
            if self.iter_step % len(image_perm) == 0:
                image_perm = self.get_image_perm()

    def get_image_perm(self):
        return torch.randperm(self.dataset.n_images)

    def get_cos_anneal_ratio(self):
        return 1.0 if self.anneal_end == 0.0 else float(np.min([1.0, self.iter_step / self.anneal_end]))

    def update_learning_rate(self):
        """linear warm-up to `warm_up_end`, then a cosine from 1 down to `learning_rate_alpha` at `end_iter` (neus_runner.py:148-156)"""
        if self.iter_step < self.warm_up_end:
            learning_factor = self.iter_step / self.warm_up_end
        else:
            alpha = self.learning_rate_alpha
            progress = (self.iter_step - self.warm_up_end) / (self.end_iter - self.warm_up_end)
            learning_factor = (np.cos(np.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha
        for g in self.optimizer.param_groups:
            g["lr"] = self.learning_rate * learning_factor

    # ------------------------------------------------------------------------------------------------------------------ checkpoints (neus_runner.py:151-169)
    def load_checkpoint(self, checkpoint_name):
        from .utils import jittor_pickle
        checkpoint = jittor_pickle.to_torch(jittor_pickle.load(os.path.join(self.base_exp_dir, "checkpoints", checkpoint_name)))
        self.neus_network.load_state_dict(checkpoint["neus"])
        self.iter_step = int(checkpoint["iter_step"])
        logging.info("End")

    def save_checkpoint(self):
        from .utils import jittor_pickle
        from .optim import flush_all
        flush_all()
        checkpoint = {"neus": self.neus_network.state_dict(), "iter_step": self.iter_step}
        os.makedirs(os.path.join(self.base_exp_dir, "checkpoints"), exist_ok=True)
        jittor_pickle.dump(checkpoint, os.path.join(self.base_exp_dir, "checkpoints", "ckpt_{:0>6d}.pkl".format(self.iter_step)))

    # ------------------------------------------------------------------------------------------------------------------ validation (neus_runner.py:171-312)
    def _render_batches(self, rays_o, rays_d, want_geometry):
        """chunks of batch_size rays through renderer.render; colours, and (weights-composited) normals / depths per ray"""
        rgb, normals, depths = [], [], []
        for o, d in zip(rays_o.reshape(-1, 3).split(self.batch_size), rays_d.reshape(-1, 3).split(self.batch_size)):
            near, far = self.dataset.near_far_from_sphere(o, d)
            background_rgb = torch.ones([1, 3], device=o.device) if self.use_white_bkgd else None
            out = self.renderer.render(o, d, near, far, cos_anneal_ratio=self.get_cos_anneal_ratio(), background_rgb=background_rgb)
            rgb.append(out["color_fine"].detach().cpu().numpy())
            if want_geometry:
                n_samples = self.renderer.n_samples + self.renderer.n_importance
                w = out["weights"][:, :n_samples].detach()
                inside = out["inside_sphere"]
                normals.append((out["gradients"].detach() * w[:, :, None] * inside[..., None]).sum(1).cpu().numpy())
                depths.append((out["z_vals"].detach() * w * inside).sum(1).cpu().numpy())
            del out
        return rgb, normals, depths

    def validate_image(self, idx=-1, resolution_level=-1):
        if idx < 0:
            idx = np.random.randint(self.dataset.n_images)
        print("Validate: iter: {}, camera: {}".format(self.iter_step, idx))
        if resolution_level < 0:
            resolution_level = self.validate_resolution_level
        rays_o, rays_d = self.dataset.gen_rays_at(idx, resolution_level=resolution_level)
        H, W, _ = rays_o.shape
        rgb, normals, depths = self._render_batches(rays_o, rays_d, want_geometry=True)
        img_fine = (np.concatenate(rgb, 0).reshape([H, W, 3]) * 256).clip(0, 255)
        rot = np.linalg.inv(self.dataset.pose_all[idx, :3, :3].detach().cpu().numpy())          # world normals into the camera frame
        normal_img = (np.matmul(rot[None, :, :], np.concatenate(normals, 0)[:, :, None]).reshape([H, W, 3]) * 128 + 128).clip(0, 255)
        depth_img = _jet((np.concatenate(depths, 0).reshape([H, W]) * 255).astype(np.uint8))
        for sub in ("validations_fine", "normals", "depths"):
            os.makedirs(os.path.join(self.base_exp_dir, sub), exist_ok=True)
        name = "{:0>8d}_{}_{}.png".format(self.iter_step, 0, idx)
        _imwrite_bgr(os.path.join(self.base_exp_dir, "validations_fine", name), np.concatenate([img_fine, self.dataset.image_at(idx, resolution_level=resolution_level)]))
        _imwrite_bgr(os.path.join(self.base_exp_dir, "normals", name), normal_img)
        _imwrite_bgr(os.path.join(self.base_exp_dir, "depths", name), depth_img)
        return img_fine

    def render_novel_image(self, idx_0, idx_1, ratio, resolution_level):
        """a view interpolated between two cameras (neus_runner.py:268-294)"""
        rays_o, rays_d = self.dataset.gen_rays_between(idx_0, idx_1, ratio, resolution_level=resolution_level)
        H, W, _ = rays_o.shape
        rgb, _, _ = self._render_batches(rays_o, rays_d, want_geometry=False)
        return (np.concatenate(rgb, 0).reshape([H, W, 3]) * 256).clip(0, 255).astype(np.uint8)

    def validate_mesh(self, world_space=False, resolution=64, threshold=0.0):
        from .utils.isosurface import write_ply
        bound_min = torch.tensor(self.dataset.object_bbox_min, dtype=torch.float32, device=self.device)
        bound_max = torch.tensor(self.dataset.object_bbox_max, dtype=torch.float32, device=self.device)
        vertices, triangles = self.renderer.extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold)
        os.makedirs(os.path.join(self.base_exp_dir, f"meshes_{resolution}"), exist_ok=True)
        if world_space:
            vertices = vertices * self.dataset.scale_mats_np[0][0, 0] + self.dataset.scale_mats_np[0][:3, 3][None]
        path = os.path.join(self.base_exp_dir, f"meshes_{resolution}", "{:0>8d}.ply".format(self.iter_step))
        write_ply(path, vertices, triangles)
        logging.info("End")
        return vertices, triangles
