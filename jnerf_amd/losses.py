"""HuberLoss / MSELoss with the reference's semantics (python/jnerf/models/losses/huber_loss.py:6-14, mse_loss.py:6-15)."""
import math
import torch
from torch import nn
from . import ops
from .utils.registry import LOSSES


class _Huber(torch.autograd.Function):
    """value and elementwise derivative in one kernel (csrc/sampler.hip k_huber); the ~10 tiny elementwise launches of the generic
    formulation cost more than the maths"""

    @staticmethod
    def forward(ctx, x, target, delta):
        loss, grad = ops.huber(x, target, delta)
        ctx.save_for_backward(grad)
        return loss

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g * grad, None, None


@LOSSES.register_module()
class HuberLoss(nn.Module):
    """Unreduced: returns [R,3]; the optimiser back-propagates its SUM (Jittor's `optimizer.step(loss)` convention)."""

    def __init__(self, delta):
        super().__init__()
        self.delta = delta
        self.delta_quad = 0.5 * delta ** 2

    def forward(self, x, target):
        if x.is_cuda and x.dtype == torch.float32 and not target.requires_grad:
            return _Huber.apply(x, target, self.delta)
        rel = torch.abs(x - target)
        sqr = 0.5 / self.delta * rel * rel
        return torch.where(rel > self.delta, rel - 0.5 * self.delta, sqr)


def img2mse(x, y):
    return torch.mean((x - y) ** 2)


def mse2psnr(x):
    x = x if isinstance(x, float) else float(x)
    return -10.0 * math.log(x) / math.log(10.0)


@LOSSES.register_module()
class MSELoss(nn.Module):
    def forward(self, x, target):
        return img2mse(x, target)
