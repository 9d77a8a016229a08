"""NeuSRenderer behind the SAMPLERS registry (python/jnerf/models/samplers/neus_render/renderer.py:68-400): hierarchical sampling along each ray steered by the SDF
(`up_sample` / `cat_z_vals`), the SDF -> opacity conversion of NeuS (`render_core`: logistic CDF of the signed distance at both ends of a section), the NeRF++ background
on inverted-sphere coordinates (`render_core_outside`), and iso-surface extraction.

Fixed sample counts per ray (64 + 64 inside, 32 outside), so rays are rows of dense [batch, n] tensors - plain torch ops with autograd (the gradient of the SDF network
enters the colour and the eikonal term: double backward), as the reference is plain Jittor ops.  The SDF -> opacity -> weights -> colour chain of render_core has a
hand-written HIP implementation (`ngp_neus_composite_fwd/_bwd`, csrc/neus.hip, one wavefront per ray) that the renderer uses on the GPU (`fused_composite`); the torch
expression of the same formulas stays as the CPU form used by the CPU tests and as the statement the kernel is tested against."""
import numpy as np
import torch
from .neus_network import safe_clip, jt_norm
from .utils.registry import SAMPLERS


def extract_fields(bound_min, bound_max, resolution, query_func, block=64):
    """renderer.py:11-26: query_func on a resolution^3 lattice, 64^3 points at a time"""
    axes = [torch.linspace(float(bound_min[d]), float(bound_max[d]), resolution, device=bound_min.device).split(block) for d in range(3)]
    u = np.zeros([resolution] * 3, dtype=np.float32)
    with torch.no_grad():
        for xi, xs in enumerate(axes[0]):
            for yi, ys in enumerate(axes[1]):
                for zi, zs in enumerate(axes[2]):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs)).float().cpu().numpy()
                    u[xi * block: xi * block + len(xs), yi * block: yi * block + len(ys), zi * block: zi * block + len(zs)] = val
    return u


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func):
    """renderer.py:29-38 (mcubes.marching_cubes there; PyMCubes is not installed here: utils/isosurface.py's marching tetrahedra on the same lattice)"""
    from .utils.isosurface import marching_tetrahedra
    u = extract_fields(bound_min, bound_max, resolution, query_func)
    vertices, triangles = marching_tetrahedra(u, threshold)
    b_max, b_min = bound_max.detach().cpu().numpy(), bound_min.detach().cpu().numpy()
    vertices = vertices / (resolution - 1.0) * (b_max - b_min)[None, :] + b_min[None, :]
    return vertices, triangles


def sample_pdf(bins, weights, n_samples, det=False):
    """renderer.py:41-66 (NeRF's inverse-transform sampling of a piecewise-constant density over `bins`)"""
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cat([torch.zeros_like(pdf[..., :1]), torch.cumsum(pdf, -1)], -1)
    if det:
        u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, device=cdf.device).expand(*cdf.shape[:-1], n_samples)
    else:
        u = torch.rand(*cdf.shape[:-1], n_samples, device=cdf.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, right=True)
    below, above = (inds - 1).clamp_min(0), inds.clamp_max(cdf.shape[-1] - 1)
    cdf_lo, cdf_hi = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    bin_lo, bin_hi = torch.gather(bins, -1, below), torch.gather(bins, -1, above)
    denom = cdf_hi - cdf_lo
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return bin_lo + (u - cdf_lo) / denom * (bin_hi - bin_lo)


def _transmittance_weights(alpha):
    """w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-6) (renderer.py:98, 161-162, 248)"""
    ones = torch.ones_like(alpha[:, :1])
    return alpha * torch.cumprod(torch.cat([ones, 1.0 - alpha + 1e-6], -1), -1)[:, :-1]


def neus_alpha(sdf, true_cos, dists, inv_s, cos_anneal_ratio):
    """renderer.py:216-236, elementwise: section opacity from the SDF at the section midpoint and its directional derivative along the ray.  Returns (alpha before the
    [0,1] clip, prev_cdf - next_cdf, prev_cdf)."""
    iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + torch.relu(-true_cos) * cos_anneal_ratio)   # always non-positive
    next_sdf = sdf + iter_cos * dists * 0.5
    prev_sdf = sdf - iter_cos * dists * 0.5
    prev_cdf, next_cdf = torch.sigmoid(prev_sdf * inv_s), torch.sigmoid(next_sdf * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf
    return (p + 1e-5) / (c + 1e-5), p, c


@SAMPLERS.register_module()
class NeuSRenderer:
    def __init__(self, n_samples, n_importance, n_outside, up_sample_steps, perturb, fused_composite=None):
        self.nerf = self.sdf_network = self.deviation_network = self.color_network = None
        self.n_samples, self.n_importance, self.n_outside = n_samples, n_importance, n_outside
        self.up_sample_steps, self.perturb = up_sample_steps, perturb
        self.fused_composite = fused_composite        # None: the HIP compositing kernel whenever the tensors live on the GPU

    def set_neus_network(self, neus_network):
        self.nerf = neus_network.nerf_outside
        self.sdf_network = neus_network.sdf_network
        self.deviation_network = neus_network.deviation_network
        self.color_network = neus_network.color_network

    # ------------------------------------------------------------------------------------------------------------------ background (renderer.py:90-115)
    def render_core_outside(self, rays_o, rays_d, z_vals, sample_dist, nerf, background_rgb=None):
        batch_size, n_samples = z_vals.shape
        dists = torch.cat([z_vals[..., 1:] - z_vals[..., :-1], z_vals.new_full((batch_size, 1), sample_dist)], -1)
        mid_z_vals = z_vals + dists * 0.5
        pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z_vals[..., :, None]
        dis_to_center = safe_clip(jt_norm(pts, dim=-1, keepdim=True), 1.0, 1e5)
        pts = torch.cat([pts / dis_to_center, 1.0 / dis_to_center], -1)                      # inverted-sphere parametrisation (x/r, 1/r)
        dirs = rays_d[:, None, :].expand(batch_size, n_samples, 3)
        density, sampled_color = nerf(pts.reshape(-1, 3 + int(self.n_outside > 0)), dirs.reshape(-1, 3))
        sampled_color = torch.sigmoid(sampled_color).reshape(batch_size, n_samples, 3)
        alpha = 1.0 - torch.exp(-torch.nn.functional.softplus(density.reshape(batch_size, n_samples)) * dists)
        alpha = safe_clip(alpha, -1e6, 1e6)
        weights = _transmittance_weights(alpha)
        color = (weights[:, :, None] * sampled_color).sum(1)
        if background_rgb is not None:
            color = color + background_rgb * (1.0 - weights.sum(-1, keepdim=True))
        return {"color": color, "sampled_color": sampled_color, "alpha": alpha, "weights": weights, "density": density.reshape(batch_size, n_samples), "dists": dists}

    # ------------------------------------------------------------------------------------------------------------------ hierarchical sampling (renderer.py:117-181)
    def up_sample(self, rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
        """new sample depths from the opacity the current SDF samples imply at a FIXED sharpness inv_s"""
        batch_size, n_samples = z_vals.shape
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
        radius = jt_norm(pts, dim=-1)
        inside_sphere = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)
        sdf = sdf.reshape(batch_size, n_samples)
        prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
        prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
        mid_sdf = (prev_sdf + next_sdf) * 0.5
        cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)
        # the smaller of this section's slope and the previous section's: keeps the sampling robust where the SDF dips towards a surface and rises again (renderer.py:131-146)
        prev_cos_val = torch.cat([torch.zeros_like(cos_val[:, :1]), cos_val[:, :-1]], -1)
        cos_val = torch.minimum(prev_cos_val, cos_val).clamp(-1e3, 0.0) * inside_sphere
        dist = next_z - prev_z
        prev_cdf = torch.sigmoid((mid_sdf - cos_val * dist * 0.5) * inv_s)
        next_cdf = torch.sigmoid((mid_sdf + cos_val * dist * 0.5) * inv_s)
        alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)
        return sample_pdf(z_vals, _transmittance_weights(alpha), n_importance, det=True).detach()

    def cat_z_vals(self, rays_o, rays_d, z_vals, new_z_vals, sdf, last=False):
        batch_size, n_samples = z_vals.shape
        n_importance = new_z_vals.shape[1]
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z_vals[..., :, None]
        z_vals, index = torch.sort(torch.cat([z_vals, new_z_vals], -1), dim=-1)
        if not last:
            new_sdf = self.sdf_network.sdf(pts.reshape(-1, 3)).reshape(batch_size, n_importance)
            sdf = torch.gather(torch.cat([sdf, new_sdf], -1), -1, index)
        return z_vals, sdf

    # ------------------------------------------------------------------------------------------------------------------ the SDF render (renderer.py:183-281)
    def _use_fused(self, t):
        return t.is_cuda if self.fused_composite is None else bool(self.fused_composite)

    def render_core(self, rays_o, rays_d, z_vals, sample_dist, sdf_network, deviation_network, color_network, background_alpha=None, background_sampled_color=None,
                    background_rgb=None, cos_anneal_ratio=0.0):
        batch_size, n_samples = z_vals.shape
        dists = torch.cat([z_vals[..., 1:] - z_vals[..., :-1], z_vals.new_full((batch_size, 1), sample_dist)], -1)
        mid_z_vals = z_vals + dists * 0.5
        pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z_vals[..., :, None]).reshape(-1, 3)
        dirs = rays_d[:, None, :].expand(batch_size, n_samples, 3).reshape(-1, 3)

        sdf_nn_output = sdf_network(pts)
        sdf, feature_vector = sdf_nn_output[:, :1], sdf_nn_output[:, 1:]
        gradients = sdf_network.gradient(pts)
        sampled_color = color_network(pts, gradients, dirs, feature_vector).reshape(batch_size, n_samples, 3)
        inv_s = safe_clip(deviation_network(torch.zeros([1, 3], device=pts.device))[:, :1], 1e-6, 1e6)          # one learnable scalar
        true_cos = (dirs * gradients).sum(-1, keepdim=True)

        pts_norm = jt_norm(pts, dim=-1, keepdim=True).reshape(batch_size, n_samples)
        inside_sphere = (pts_norm < 1.0).float().detach()
        relax_inside_sphere = (pts_norm < 1.2).float().detach()

        if self._use_fused(sdf):
            from . import neus_ops
            color, weights, alpha, p, c = neus_ops.composite(
                sdf.reshape(batch_size, n_samples), true_cos.reshape(batch_size, n_samples), dists, inv_s.reshape(()), sampled_color, inside_sphere,
                background_alpha, background_sampled_color, float(cos_anneal_ratio))
            p, c = p.reshape(-1, 1), c.reshape(-1, 1)
        else:
            # "cos_anneal_ratio" grows from 0 to 1 over the first iterations: the annealed cosine keeps sections facing away from the camera from dying early (renderer.py:216-219)
            a, p, c = neus_alpha(sdf, true_cos, dists.reshape(-1, 1), inv_s.expand(batch_size * n_samples, 1), cos_anneal_ratio)
            alpha = safe_clip(a.reshape(batch_size, n_samples), 0.0, 1.0)
            if background_alpha is not None:                         # outside the unit sphere the background model takes over, and its far samples follow
                alpha = alpha * inside_sphere + background_alpha[:, :n_samples] * (1.0 - inside_sphere)
                alpha = torch.cat([alpha, background_alpha[:, n_samples:]], -1)
                sampled_color = sampled_color * inside_sphere[:, :, None] + background_sampled_color[:, :n_samples] * (1.0 - inside_sphere)[:, :, None]
                sampled_color = torch.cat([sampled_color, background_sampled_color[:, n_samples:]], 1)
            weights = _transmittance_weights(alpha)
            color = (sampled_color * weights[:, :, None]).sum(1)
        weights_sum = weights.sum(-1, keepdim=True)
        if background_rgb is not None:                               # fixed background, usually white or none
            color = color + background_rgb * (1.0 - weights_sum)

        # eikonal term: the SDF's gradient should have unit length (inside a slightly relaxed sphere)
        gradient_error = (jt_norm(gradients.reshape(batch_size, n_samples, 3), dim=-1) - 1.0) ** 2
        gradient_error = (relax_inside_sphere * gradient_error).sum() / (relax_inside_sphere.sum() + 1e-5)
        return {"color": color, "sdf": sdf, "dists": dists, "gradients": gradients.reshape(batch_size, n_samples, 3), "s_val": 1.0 / inv_s.expand(batch_size * n_samples, 1),
                "mid_z_vals": mid_z_vals, "p": p, "c": c, "alpha": alpha, "weights": weights, "cdf": c.reshape(batch_size, n_samples), "gradient_error": gradient_error,
                "inside_sphere": inside_sphere}

    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0):
        batch_size = len(rays_o)
        dev = rays_o.device
        sample_dist = 2.0 / self.n_samples                           # the region of interest is the unit sphere
        z_vals = near + (far - near) * torch.linspace(0.0, 1.0, self.n_samples, device=dev)[None, :]
        z_vals_outside = torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside, device=dev) if self.n_outside > 0 else None
        n_samples = self.n_samples
        perturb = perturb_overwrite if perturb_overwrite >= 0 else self.perturb
        if perturb > 0:
            z_vals = z_vals + (torch.rand([batch_size, 1], device=dev) - 0.5) * 2.0 / self.n_samples
            if self.n_outside > 0:                                   # stratified jitter of the inverse depths
                mids = 0.5 * (z_vals_outside[..., 1:] + z_vals_outside[..., :-1])
                upper, lower = torch.cat([mids, z_vals_outside[..., -1:]], -1), torch.cat([z_vals_outside[..., :1], mids], -1)
                z_vals_outside = lower[None, :] + (upper - lower)[None, :] * torch.rand([batch_size, z_vals_outside.shape[-1]], device=dev)
        if self.n_outside > 0:
            z_vals_outside = far / torch.flip(z_vals_outside, dims=[-1]) + 1.0 / self.n_samples

        background_alpha = background_sampled_color = None
        if self.n_importance > 0:
            with torch.no_grad():
                pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
                sdf = self.sdf_network.sdf(pts.reshape(-1, 3)).reshape(batch_size, self.n_samples)
                for i in range(self.up_sample_steps):                # sharpness 64, 128, 256, 512
                    new_z_vals = self.up_sample(rays_o, rays_d, z_vals, sdf, self.n_importance // self.up_sample_steps, 64 * 2 ** i)
                    z_vals, sdf = self.cat_z_vals(rays_o, rays_d, z_vals, new_z_vals, sdf, last=(i + 1 == self.up_sample_steps))
            n_samples = self.n_samples + self.n_importance

        if self.n_outside > 0:
            z_vals_feed, _ = torch.sort(torch.cat([z_vals, z_vals_outside], -1), dim=-1)
            ret_outside = self.render_core_outside(rays_o, rays_d, z_vals_feed, sample_dist, self.nerf)
            background_sampled_color, background_alpha = ret_outside["sampled_color"], ret_outside["alpha"]

        ret_fine = self.render_core(rays_o, rays_d, z_vals, sample_dist, self.sdf_network, self.deviation_network, self.color_network, background_rgb=background_rgb,
                                    background_alpha=background_alpha, background_sampled_color=background_sampled_color, cos_anneal_ratio=cos_anneal_ratio)
        weights = ret_fine["weights"]
        return {"color_fine": ret_fine["color"], "s_val": ret_fine["s_val"].reshape(batch_size, n_samples).mean(-1, keepdim=True), "cdf_fine": ret_fine["cdf"],
                "weight_sum": weights.sum(-1, keepdim=True), "weight_max": weights.max(-1, keepdim=True)[0], "sdf": ret_fine["sdf"], "gradients": ret_fine["gradients"],
                "alpha": ret_fine["alpha"], "z_vals": z_vals, "weights": weights, "gradient_error": ret_fine["gradient_error"], "inside_sphere": ret_fine["inside_sphere"]}

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0):
        return extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold, query_func=lambda pts: -self.sdf_network.sdf(pts))
