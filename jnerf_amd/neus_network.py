"""NeuS networks behind the NETWORKS registry (python/jnerf/models/networks/neus_network.py:9-276): the SDF network (IDR's geometric initialisation, softplus(beta=100),
skip connection at layer 4), the rendering (colour) network, the NeRF++ background network and the single learnable variance.

The reference's NeuS is plain Jittor ops on frequency encodings (neus_womask.py / neus_wmask.py) - here plain torch ops (rocBLAS GEMMs + autograd, double backward for
the eikonal term), like OriginNeRFNetworks "plumbing", with the reference's attribute names so state dicts carry the reference's keys (`sdf_network.lin0.weight`,
`deviation_network.variance`, `color_network.lin3.bias`, `nerf_outside.pts_linears.4.weight` ...).  What is ours: `encoder.sdf_encoder = dict(type='HashEncoder')`
(BASELINE.json config [4]: hash encoder + SDF-to-density render path, projects/neus/configs/neus_hash.py) puts the HIP multiresolution hash grid under the SDF network;
its position gradient and the second-order terms the eikonal loss needs come from `ngp_hash_encode_fwd_dydx` / `ngp_hash_encode_bwd_input` / `..._bwd_bwd_*`
(encoders.py)."""
import math
import torch
from torch import nn
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, ENCODERS
from . import encoders  # noqa: F401  (registers FrequencyEncoder / HashEncoder)
from .encoders import input_gradient_only


def safe_clip(x, lo, hi):
    """Jittor's Var.safe_clip as the NeuS code uses it: the VALUE is clamped, the gradient passes through unchanged (Jittor documents it as "different from clamp:
    gradient passes through"; Jittor is an external dependency of the reference, absent from /root/reference - restated, parity unpinned)."""
    return x + (x.clamp(lo, hi) - x).detach()


def jt_norm(x, dim=-1, keepdim=False, eps=1e-6):
    """jt.norm(x, p=2, dim, keepdim, eps) = sqrt(max(sum x^2, eps)) (Jittor's misc.norm: the eps is a floor under the square root, not an addend)"""
    return x.square().sum(dim, keepdim=keepdim).clamp_min(eps).sqrt()


def jt_linear(in_features, out_features):
    """nn.Linear with Jittor's default initialisation (jittor.nn.Linear: weight = init.invariant_uniform -> U(+-sqrt(g / fan_in)), g = 3 unless the config says otherwise - network.invariant_uniform_bound; bias = U(+-1 / sqrt(fan_in))) instead of
    torch's kaiming-uniform default, which is sqrt(3) times narrower in the weights.  Jittor is an external dependency of the reference: restated from its published
    source, parity unpinned (same note as network.py:invariant_uniform)."""
    lin = nn.Linear(in_features, out_features)
    with torch.no_grad():
        from .network import invariant_uniform_bound
        b = invariant_uniform_bound(in_features)
        lin.weight.uniform_(-b, b)
        lin.bias.uniform_(-1.0 / math.sqrt(in_features), 1.0 / math.sqrt(in_features))
    return lin


def _encoder(cfg_enc):
    """the reference builds an encoder only when multires > 0 (neus_network.py:31-35, 131-135, 188-194); a hash encoder has no multires and is always built"""
    if cfg_enc is None:
        return None
    if cfg_enc.get("type") == "FrequencyEncoder" and not cfg_enc.get("multires", 0) > 0:
        return None
    return build_from_cfg(cfg_enc, ENCODERS)


class SDFNetwork(nn.Module):
    """neus_network.py:10-108.  dims = [embedded input] + n_layers x d_hidden + [d_out]; layer l+1 in skip_in shrinks layer l's output by the width of the embedded
    input, which is concatenated back (scaled by 1/sqrt(2)) in front of layer l+1."""

    def __init__(self, d_out, d_hidden, n_layers, skip_in=(4,), bias=0.5, scale=1, geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        self.cfg = get_cfg()
        enc_cfg = self.cfg.encoder.sdf_encoder
        d_in = enc_cfg.get("input_dims", 3)
        self.embed_fn_fine = _encoder(enc_cfg)
        self.hash_input = enc_cfg.get("type") == "HashEncoder"
        multires = enc_cfg.get("multires", 0) if not self.hash_input else 0
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        if self.embed_fn_fine is not None:
            # (ours, hash variant) the raw position stays in front of the hash features - the geometric initialisation below needs it, exactly as it uses the
            # identity part of the frequency embedding
            dims[0] = self.embed_fn_fine.out_dim + (d_in if self.hash_input else 0)
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.d_in = d_in
        embedded = self.embed_fn_fine is not None
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = jt_linear(dims[l], out_dim)
            if geometric_init:                                   # neus_network.py:49-68: the network starts as the SDF of a sphere of radius `bias`
                with torch.no_grad():
                    if l == self.num_layers - 2:
                        sign = -1.0 if inside_outside else 1.0
                        lin.weight.normal_(sign * math.sqrt(math.pi) / math.sqrt(dims[l]), 1e-4)
                        lin.bias.fill_(-sign * bias)
                    elif embedded and l == 0:                    # only the three raw coordinates feed the first layer
                        lin.bias.zero_()
                        lin.weight.zero_()
                        lin.weight[:, :3].normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
                    elif embedded and l in self.skip_in:         # ... and the skip layer ignores the embedding's non-identity columns
                        lin.bias.zero_()
                        lin.weight.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
                        lin.weight[:, -(dims[0] - 3):].zero_()
                    else:
                        lin.bias.zero_()
                        lin.weight.normal_(0.0, math.sqrt(2) / math.sqrt(out_dim))
            # weight_norm is accepted and ignored, as in the reference (the call is commented out there, neus_network.py:70-71)
            setattr(self, "lin" + str(l), lin)
        self.activation = nn.Softplus(beta=100)
        self.to(self.cfg.device or "cuda")

    def _embed(self, x):
        if self.embed_fn_fine is None:
            return x
        if self.hash_input:
            # the hash grid covers [0,1]^3; NeuS' region of interest is the unit sphere around the origin
            feat = self.embed_fn_fine((x * 0.5 + 0.5).clamp(0.0, 1.0))
            return torch.cat([x, feat.to(x.dtype)], -1)
        return self.embed_fn_fine(x)

    def forward(self, inputs):
        inputs = self._embed(inputs * self.scale)
        x = inputs
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                x = torch.cat([x, inputs], 1) / math.sqrt(2)
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.activation(x)
        return torch.cat([x[:, :1] / self.scale, x[:, 1:]], -1)

    execute = forward

    def sdf(self, x):
        return self.forward(x)[:, :1]

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        """d sdf / d x, differentiable itself (the eikonal term and the colour network's normal input back-propagate through it; neus_network.py:99-108)"""
        with torch.enable_grad():
            x = x.detach().requires_grad_(True) if not x.requires_grad else x
            y = self.sdf(x)
            with input_gradient_only(self.embed_fn_fine):
                (g,) = torch.autograd.grad(y, x, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True)
        return g


class RenderingNetwork(nn.Module):
    """neus_network.py:112-174: colour from (point, view direction, normal, feature vector)"""

    def __init__(self, d_feature, mode, d_out, d_hidden, n_layers, weight_norm=True, squeeze_out=True):
        super().__init__()
        self.cfg = get_cfg()
        self.mode = mode
        self.squeeze_out = squeeze_out
        dims = [9 + d_feature] + [d_hidden] * n_layers + [d_out]
        self.embedview_fn = _encoder(self.cfg.encoder.rendering_encoder)
        if self.embedview_fn is not None:
            dims[0] += self.embedview_fn.out_dim - 3
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, "lin" + str(l), jt_linear(dims[l], dims[l + 1]))
        self.relu = nn.ReLU()
        self.to(self.cfg.device or "cuda")

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        parts = {"idr": [points, view_dirs, normals, feature_vectors], "no_view_dir": [points, normals, feature_vectors], "no_normal": [points, view_dirs, feature_vectors]}[self.mode]
        x = torch.cat(parts, -1)
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return torch.sigmoid(x) if self.squeeze_out else x

    execute = forward


class NeRF(nn.Module):
    """neus_network.py:178-253: the background (outside the unit sphere) radiance field on inverted-sphere coordinates (x/r, 1/r)"""

    def __init__(self, D=8, W=256, output_ch=4, skips=[4], use_viewdirs=False):
        super().__init__()
        self.cfg = get_cfg()
        self.D, self.W = D, W
        enc = self.cfg.encoder
        self.d_in, self.d_in_view = enc.nerf_pos_encoder.input_dims, enc.nerf_dir_encoder.input_dims
        self.embed_fn, self.embed_fn_view = _encoder(enc.nerf_pos_encoder), _encoder(enc.nerf_dir_encoder)
        self.input_ch = self.embed_fn.out_dim if self.embed_fn is not None else 3
        self.input_ch_view = self.embed_fn_view.out_dim if self.embed_fn_view is not None else 3
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        self.pts_linears = nn.ModuleList([jt_linear(self.input_ch, W)] + [jt_linear(W + self.input_ch, W) if i in self.skips else jt_linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([jt_linear(self.input_ch_view + W, W // 2)])
        if use_viewdirs:
            self.feature_linear = jt_linear(W, W)
            self.alpha_linear = jt_linear(W, 1)
            self.rgb_linear = jt_linear(W // 2, 3)
        else:
            self.output_linear = jt_linear(W, output_ch)
        self.to(self.cfg.device or "cuda")

    def forward(self, input_pts, input_views):
        if self.embed_fn is not None:
            input_pts = self.embed_fn(input_pts)
        if self.embed_fn_view is not None:
            input_views = self.embed_fn_view(input_views)
        h = input_pts
        for i, lin in enumerate(self.pts_linears):
            h = torch.relu(lin(h))
            if i in self.skips:
                h = torch.cat([input_pts, h], -1)
        assert self.use_viewdirs, "NeRF(use_viewdirs=False) has no forward in the reference either (neus_network.py:252-253)"
        alpha = self.alpha_linear(h)
        h = torch.cat([self.feature_linear(h), input_views], -1)
        for lin in self.views_linears:
            h = torch.relu(lin(h))
        return alpha, self.rgb_linear(h)

    execute = forward


class SingleVarianceNetwork(nn.Module):
    """neus_network.py:256-262: inv_s = exp(10 * variance), one learnable scalar"""

    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(float(init_val), device=get_cfg().device or "cuda"))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)

    execute = forward


@NETWORKS.register_module()
class NeuS(nn.Module):
    """neus_network.py:264-276: the four networks under the names NeuSRenderer.set_neus_network picks up"""

    def __init__(self, nerf_network, sdf_network, variance_network, rendering_network):
        super().__init__()
        self.nerf_outside = NeRF(**nerf_network)
        self.sdf_network = SDFNetwork(**sdf_network)
        self.deviation_network = SingleVarianceNetwork(**variance_network)
        self.color_network = RenderingNetwork(**rendering_network)
