"""ctypes front end + autograd node of the NeuS compositing kernels (csrc/neus.hip: ngp_neus_composite_fwd / _bwd) - the SDF -> opacity -> weights -> colour chain
of NeuSRenderer.render_core (renderer.py:216-252) as one launch per direction.  GPU tensors only: there is no CPU form here (the renderer's torch expression of the same
formulas is what the CPU tests run, and what tests/test_neus_gpu.py compares this node with)."""
import torch
from . import _lib as L
from .ops import _p, _stream, check


def _f32c(t):
    return None if t is None else t.detach().float().contiguous()


class _NeusComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, true_cos, dists, inv_s, color, inside, bg_alpha, bg_color, ratio):
        assert sdf.is_cuda, "ngp_neus_composite_*: device tensors only"
        B, n = sdf.shape
        n_total = n if bg_alpha is None else bg_alpha.shape[1]
        ins = [_f32c(t) for t in (sdf, true_cos, dists, inv_s.reshape(1), color, inside, bg_alpha, bg_color)]
        dev = sdf.device
        out_color = torch.empty((B, 3), dtype=torch.float32, device=dev)
        weights = torch.empty((B, n_total), dtype=torch.float32, device=dev)
        alpha = torch.empty((B, n_total), dtype=torch.float32, device=dev)
        p = torch.empty((B, n), dtype=torch.float32, device=dev)
        c = torch.empty((B, n), dtype=torch.float32, device=dev)
        check(L.lib().ngp_neus_composite_fwd(_stream(), B, n, n_total, *[_p(t) for t in ins], float(ratio), _p(out_color), _p(weights), _p(alpha), _p(p), _p(c)), "ngp_neus_composite_fwd")
        ctx.save_for_backward(*[t for t in ins if t is not None])
        ctx.has_bg, ctx.ratio, ctx.shape = bg_alpha is not None, float(ratio), (B, n, n_total)
        ctx.mark_non_differentiable(alpha, p, c)            # reported values (render_core's 'alpha', 'p', 'c' / 'cdf' entries); no loss term reads them
        return out_color, weights, alpha, p, c

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_color, g_weights, _ga, _gp, _gc):
        saved = list(ctx.saved_tensors)
        sdf, cosv, dists, inv_s, color, inside = saved[:6]
        bg_alpha, bg_color = (saved[6], saved[7]) if ctx.has_bg else (None, None)
        B, n, n_total = ctx.shape
        dev = sdf.device
        d_sdf, d_cos = torch.empty_like(sdf), torch.empty_like(cosv)
        d_color = torch.empty_like(color)
        d_part = torch.empty(B, dtype=torch.float32, device=dev)
        d_bg_alpha = torch.empty_like(bg_alpha) if ctx.has_bg else None
        d_bg_color = torch.empty_like(bg_color) if ctx.has_bg else None
        g_color = _f32c(g_color) if g_color is not None else torch.zeros((B, 3), dtype=torch.float32, device=dev)
        g_weights = _f32c(g_weights)
        check(L.lib().ngp_neus_composite_bwd(_stream(), B, n, n_total, _p(sdf), _p(cosv), _p(dists), _p(inv_s), _p(color), _p(inside), _p(bg_alpha), _p(bg_color), ctx.ratio,
                                             _p(g_color), _p(g_weights), _p(d_sdf), _p(d_cos), _p(d_part), _p(d_color), _p(d_bg_alpha), _p(d_bg_color)), "ngp_neus_composite_bwd")
        return d_sdf, d_cos, None, d_part.sum().reshape(()), d_color, None, d_bg_alpha, d_bg_color, None


def composite(sdf, true_cos, dists, inv_s, color, inside, bg_alpha, bg_color, cos_anneal_ratio):
    """sdf / true_cos / dists / inside [B, n]; inv_s 0-d; color [B, n, 3]; bg_alpha [B, n_total] / bg_color [B, n_total, 3] or None.
    Returns (colour [B,3], weights [B,n_total], alpha [B,n_total], p [B,n], c [B,n])."""
    return _NeusComposite.apply(sdf, true_cos, dists, inv_s, color, inside, bg_alpha, bg_color, float(cos_anneal_ratio))
