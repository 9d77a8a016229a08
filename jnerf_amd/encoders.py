"""Position / direction encoders behind the reference's ENCODERS registry names.

HashEncoder  <- python/jnerf/models/position_encoders/hash_encoder/hash_encoder.py:8-30 + grid_encode.py:11-190
SHEncoder    <- python/jnerf/models/position_encoders/sh_encoder/sh_encoder.py:10-56
FrequencyEncoder <- python/jnerf/models/position_encoders/freq_encoder/freq_encoder.py:11-52 (plain torch ops, "plumbing" configs only)

Parameters are fp32 masters; with cfg.fp16 the kernels gather from an fp16 shadow that the fused Adam+EMA sweep refreshes."""
import contextlib
import torch
from torch import nn
from . import ops
from .utils.config import get_cfg
from .utils.registry import ENCODERS


@contextlib.contextmanager
def input_gradient_only(enc):
    """While active, the backward of `enc` (a HashEncoder; anything else: no-op) serves `torch.autograd.grad(y, x)` - the SDF network's own input gradient
    (neus_network.py:99-108): only dL/dx is produced, dL/dy is NOT scattered into the table gradient (it is not a loss gradient).  The table gradient of the loss
    arrives later, when loss.backward() runs the same backward without the flag plus _HashEncodeBwdInput.backward for the second-order terms."""
    if not isinstance(enc, HashEncoder):
        yield
        return
    old, enc._input_grad_only = enc._input_grad_only, True
    try:
        yield
    finally:
        enc._input_grad_only = old


class _HashEncodeBwdInput(torch.autograd.Function):
    """dL/dx = contraction of dL/dy with kernel_grid's dy_dx, as a differentiable node: a network trained on its own input gradient (eikonal term) back-propagates
    through it - w.r.t. dL/dy (ngp_hash_encode_bwd_input_bwd_dy) and w.r.t. the table (ngp_hash_encode_bwd_input_bwd_grid, added into m_grid.grad like the
    first-order scatter).  The mixed second derivative w.r.t. x is not produced (sample positions carry no parameters)."""

    @staticmethod
    def forward(ctx, dy, x, grid, dy_dx, enc):
        ctx.enc = enc
        ctx.save_for_backward(dy, x, dy_dx)
        return ops.hash_encode_bwd_input(dy, dy_dx, ops.LAYOUT_AOS)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        dy, x, dy_dx = ctx.saved_tensors
        u = u.contiguous().float()
        enc = ctx.enc
        ops.hash_encode_bwd_input_bwd_grid(x.detach(), dy.detach(), u, enc.level_table, enc.grad_buffer())
        return ops.hash_encode_bwd_input_bwd_dy(u, dy_dx, dtype=dy.dtype), None, None, None, None


class _HashEncode(torch.autograd.Function):
    """GridEncode.execute / .grad (grid_encode.py:71-125, 137-190): table gradient by scatter; the reference returns no gradient w.r.t. the positions - here they get one
    when they ask for it (x.requires_grad), from kernel_grid's dy_dx branch."""

    @staticmethod
    def forward(ctx, x, grid, enc):
        table = enc.table_for_kernels()
        ctx.enc = enc
        ctx.want_dx = bool(ctx.needs_input_grad[0])
        if ctx.want_dx:
            # (ours) the positions ask for a gradient - a hash-grid SDF network differentiates the encoder w.r.t. its input: the forward also writes kernel_grid's
            # dy_dx output (HashEncode.h:205-251; the reference compiles that branch but passes nullptr, grid_encode.py:96) and backward contracts it with dL/dy
            out, dy_dx = ops.hash_encode_fwd_dydx(x.detach(), table, enc.level_table)
            ctx.save_for_backward(x, grid, dy_dx)
            return out
        ctx.save_for_backward(x)
        return ops.hash_encode_fwd(x, table, enc.level_table)

    @staticmethod
    def backward(ctx, dy):
        x = ctx.saved_tensors[0]
        enc = ctx.enc
        dy = dy.contiguous()
        if not enc._input_grad_only:
            enc.accumulate_grad(x.detach(), dy.detach(), ops.LAYOUT_AOS)
        if ctx.want_dx:
            _, grid, dy_dx = ctx.saved_tensors
            return _HashEncodeBwdInput.apply(dy, x, grid, dy_dx, enc), None, None
        return None, None, None          # the reference's contract (grid_encode.py:190: `return None, grid_grad`)


@ENCODERS.register_module()
class HashEncoder(nn.Module):
    def __init__(self, n_pos_dims=3, n_features_per_level=2, n_levels=16, base_resolution=16, log2_hashmap_size=19):
        super().__init__()
        self.cfg = get_cfg()
        self.using_fp16 = bool(self.cfg.fp16)
        aabb_scale = getattr(self.cfg.dataset_obj, "aabb_scale", None) or 1          # (NeuS data sets have no aabb_scale: the unit cube)
        # like the reference (hash_encoder.py:17-18) the geometry is fixed: L=16, F=2, T=2^19, base 16, whatever the ctor args say
        self.level_table, self.offsets, self.n_params = ops.level_table(aabb_scale)
        self.grad_type = "float16" if self.using_fp16 else "float32"
        dev = self.cfg.device or "cuda"
        self.m_grid = nn.Parameter(torch.empty(self.n_params, dtype=torch.float32, device=dev).uniform_(-1e-4, 1e-4))   # hash_encoder.py:22-23
        self.register_buffer("m_grid_half", self.m_grid.detach().half() if self.using_fp16 else None, persistent=False)
        self.shadow_dirty = False
        self._bwd_ws = None
        self._input_grad_only = False
        self.out_dim = 32
        self.out_dtype = torch.float16 if self.using_fp16 else torch.float32

    def table_for_kernels(self):
        from .optim import flush_all
        flush_all()
        if not self.using_fp16:
            return self.m_grid.detach()
        if self.shadow_dirty:
            self.m_grid_half.copy_(self.m_grid.detach())
            self.shadow_dirty = False
        return self.m_grid_half

    def grad_buffer(self):
        if self.m_grid.grad is None:
            self.m_grid.grad = torch.zeros_like(self.m_grid)
        return self.m_grid.grad

    def accumulate_grad(self, x, dy, layout, n_valid=None):
        """scatter-add dL/dy into m_grid.grad (fp32 accumulation; the buffer is zeroed by the optimiser sweep, not per call)"""
        need = ops.hash_bwd_workspace_bytes(self.level_table, x.shape[0], dy.dtype)
        if self._bwd_ws is None or self._bwd_ws.numel() < need:
            self._bwd_ws = torch.empty(need, dtype=torch.uint8, device=self.m_grid.device)
        ops.hash_encode_bwd(x, dy, self.level_table, self.n_params, grad=self.grad_buffer(), layout=layout, zero_first=False, n_valid=n_valid, workspace=self._bwd_ws)

    def forward(self, x):
        return _HashEncode.apply(x, self.m_grid, self)

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.shadow_dirty = True


@ENCODERS.register_module()
class SHEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.cfg = get_cfg()
        self.using_fp16 = bool(self.cfg.fp16)
        self.m_sh_degree = 4
        self.out_dim = 16

    def forward(self, x):
        return ops.sh_encode(x.detach(), torch.float16 if self.using_fp16 else torch.float32)   # no gradient (sh_encoder.py:55-56)


@ENCODERS.register_module()
class FrequencyEncoder(nn.Module):
    def __init__(self, multires, include_input=True, input_dims=3, log_sampling=True):
        super().__init__()
        self.include_input = include_input
        max_freq = multires - 1
        self.freq_bands = (2.0 ** torch.linspace(0.0, max_freq, steps=multires)) if log_sampling else torch.linspace(1.0, 2.0 ** max_freq, steps=multires)
        self.out_dim = (input_dims if include_input else 0) + input_dims * 2 * multires

    def forward(self, x):
        outs = [x] if self.include_input else []
        for f in self.freq_bands.tolist():
            outs += [torch.sin(x * f), torch.cos(x * f)]
        return torch.cat(outs, -1)
