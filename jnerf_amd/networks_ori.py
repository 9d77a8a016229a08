"""OriginNeRFNetworks behind the NETWORKS registry (python/jnerf/models/networks/ori_nerf_network.py:8-76): the 8 x 256 positional-encoding MLP of the original NeRF
with a skip connection at layer 4 and a 128-wide view branch.  BASELINE.json config [0] is "plumbing only": this is plain torch (rocBLAS GEMMs + autograd,
autocast when cfg.fp16) on top of the same FrequencyEncoder / DensityGridSampler / compositing kernels - not a hot-path component."""
import torch
from torch import nn
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, ENCODERS


@NETWORKS.register_module()
class OriginNeRFNetworks(nn.Module):
    def __init__(self, D=8, W=256, skips=[4]):
        super().__init__()
        self.D, self.W, self.skips = D, W, list(skips)
        self.cfg = get_cfg()
        self.using_fp16 = bool(self.cfg.fp16)
        dev = self.cfg.device or "cuda"
        self.pos_encoder = build_from_cfg(self.cfg.encoder.pos_encoder, ENCODERS)
        self.dir_encoder = build_from_cfg(self.cfg.encoder.dir_encoder, ENCODERS)
        pe, de = self.pos_encoder.out_dim, self.dir_encoder.out_dim
        # layer i+1 takes the skip concatenation [pos, h] produced after layer i (ori_nerf_network.py:20-21, 42-46)
        self.pts_linears = nn.ModuleList([nn.Linear(pe, W)] + [nn.Linear(W + pe, W) if i in self.skips else nn.Linear(W, W) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([nn.Linear(de + W, W // 2)])
        self.feature_linear = nn.Linear(W, W)
        self.alpha_linear = nn.Linear(W, 1)
        self.rgb_linear = nn.Linear(W // 2, 3)
        self.to(dev)
        self.fused = False

    def _trunk(self, pos):
        h = pos
        for i, l in enumerate(self.pts_linears):
            h = torch.relu(l(h))
            if i in self.skips:
                h = torch.cat([pos, h], -1)
        return h

    def forward(self, pos_input, dir_input):
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.using_fp16 and pos_input.is_cuda):
            d = self.dir_encoder(dir_input)
            h = self._trunk(self.pos_encoder(pos_input))
            alpha = self.alpha_linear(h)
            h = torch.cat([self.feature_linear(h), d.to(h.dtype)], -1)
            for l in self.views_linears:
                h = torch.relu(l(h))
            out = torch.cat([self.rgb_linear(h), alpha], -1)
        return out.half() if self.using_fp16 else out.float()

    def density(self, pos_input):
        with torch.autocast("cuda", dtype=torch.float16, enabled=self.using_fp16 and pos_input.is_cuda):
            alpha = self.alpha_linear(self._trunk(self.pos_encoder(pos_input)))
        return alpha.half() if self.using_fp16 else alpha.float()

    def set_fp16(self):
        pass    # autocast instead of casting the parameters: the masters stay fp32 for the optimiser
