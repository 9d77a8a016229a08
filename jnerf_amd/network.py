"""NGPNetworks / FMLP behind the reference's NETWORKS registry (python/jnerf/models/networks/ngp_network.py:8-96).

cfg.fp16 (projects/ngp/configs/ngp_fox.py:73): both MLPs + SH + concats run in ONE fp16-MFMA kernel with weights in LDS
(csrc/field_mlp.hip), fed level-major by the XCD-aware hash kernel; backward recomputes the forward in-kernel.
fp32 (ngp_base.py): the reference itself falls back to plain nn.Linear chains (ngp_network.py:57-67) — so do we (rocBLAS GEMMs via torch),
with the fp32 hash kernels."""
import math
import torch
from torch import nn
from . import ops
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, ENCODERS


def invariant_uniform(out_f, in_f, device):
    """Jittor's init.invariant_uniform (external): U(+-sqrt(3/fan_in)) — documented assumption, see DESIGN.md 'parity unpinned'."""
    b = math.sqrt(3.0 / in_f)
    return torch.empty((out_f, in_f), dtype=torch.float32, device=device).uniform_(-b, b)


class FMLP(nn.Module):
    """Weight pack of the fully-fused MLP (ngp_network.py:8-37, fully_fused_mlp.py:26-41): every layer (out,in) row-major, last layer
    zero-padded to 16 rows, concatenated into one flat parameter `con_weights`."""

    def __init__(self, weight_shapes, device="cuda"):
        super().__init__()
        assert len(weight_shapes) > 2
        self.weight_shapes = list(weight_shapes)
        self.output_shape1 = weight_shapes[-1]
        ws = []
        for i in range(len(weight_shapes) - 1):
            w = invariant_uniform(weight_shapes[i + 1], weight_shapes[i], device)
            if i == len(weight_shapes) - 2 and w.shape[0] < 16:
                w = torch.cat([w, torch.zeros((16 - w.shape[0], w.shape[1]), device=device)], 0)
            ws.append(w.reshape(-1))
        self.con_weights = nn.Parameter(torch.cat(ws))
        self.register_buffer("con_weights_half", self.con_weights.detach().half(), persistent=False)
        self.shadow_dirty = False

    def half_weights(self):
        from .optim import flush_all
        flush_all()
        if self.shadow_dirty:
            self.con_weights_half.copy_(self.con_weights.detach())
            self.shadow_dirty = False
        return self.con_weights_half

    def grad_buffer(self):
        if self.con_weights.grad is None:
            self.con_weights.grad = torch.zeros_like(self.con_weights)
        return self.con_weights.grad

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self.shadow_dirty = True

    def layers(self):
        """[(out,in) fp32 views] for the generic path"""
        out, off = [], 0
        s = self.weight_shapes
        for i in range(len(s) - 1):
            o = max(s[i + 1], 16) if i == len(s) - 2 else s[i + 1]
            out.append(self.con_weights[off:off + o * s[i]].view(o, s[i]))
            off += o * s[i]
        return out

    def forward(self, x):
        from .optim import flush_all
        flush_all()
        h = x.float()
        ls = self.layers()
        for i, w in enumerate(ls):
            h = torch.nn.functional.linear(h, w)
            if i < len(ls) - 1:
                h = torch.relu(h)
        return h[:, :self.output_shape1].to(x.dtype)


class _FusedField(torch.autograd.Function):
    """hash encode (level-major) -> fused SH + density MLP + colour MLP;  backward: fused dgrad/wgrad -> atomic scatter."""

    @staticmethod
    def forward(ctx, pos, dirs, grid, wd, wc, net, n_valid):
        enc = net.pos_encoder
        n = pos.shape[0]
        if pos.stride(0) != 3:
            pos = pos.contiguous()      # compact [n,3] copy (3 MB): the 16 level passes then stream 12 B/sample out of L2 instead of 28 B records
        feat = net._feat_buffer(n)
        ops.hash_encode_fwd(pos, enc.table_for_kernels(), enc.level_table, out=feat, layout=ops.LAYOUT_SOA, n_valid=n_valid)
        out = ops.field_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out_dtype=torch.float16, n_valid=n_valid, packed=net.packed_weights(refresh=True))
        ctx.net, ctx.n_valid = net, n_valid
        ctx.save_for_backward(pos, dirs, feat)
        return out

    @staticmethod
    def backward(ctx, dout):
        pos, dirs, feat = ctx.saved_tensors
        net, n_valid = ctx.net, ctx.n_valid
        enc = net.pos_encoder
        n = pos.shape[0]
        dfeat, slabs, wsum = net._bwd_buffers(n)
        ops.field_bwd(feat, dirs, None, None, dout.contiguous(), layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid,
                      packed=net.packed_weights(refresh=False))      # the fragments the forward of this step built (the weights have not changed since)
        ops.reduce_slabs(slabs, out=net._flat_weight_grad(), accumulate=True)     # both MLP packs' .grad are views of this one buffer
        enc.accumulate_grad(pos, dfeat, ops.LAYOUT_SOA, n_valid=n_valid)
        return None, None, None, None, None, None, None


@NETWORKS.register_module()
class NGPNetworks(nn.Module):
    def __init__(self, use_fully=True, density_hidden_layer=1, density_n_neurons=64, rgb_hidden_layer=2, rgb_n_neurons=64):
        super().__init__()
        self.use_fully = use_fully
        self.cfg = get_cfg()
        self.using_fp16 = bool(self.cfg.fp16)
        dev = self.cfg.device or "cuda"
        self.pos_encoder = build_from_cfg(self.cfg.encoder.pos_encoder, ENCODERS)
        self.dir_encoder = build_from_cfg(self.cfg.encoder.dir_encoder, ENCODERS)
        self.fused = bool(self.use_fully and self.using_fp16 and density_n_neurons == 64 and rgb_n_neurons == 64
                          and self.pos_encoder.out_dim == 32 and self.dir_encoder.out_dim == 16 and hasattr(self.pos_encoder, "level_table"))
        if self.fused:
            self.density_mlp = FMLP([self.pos_encoder.out_dim, density_n_neurons, 16], dev)
            self.rgb_mlp = FMLP([self.dir_encoder.out_dim + 16, rgb_n_neurons, rgb_n_neurons, 3], dev)
        else:
            if self.use_fully and not self.using_fp16:
                print("Warning: FFMLPs only support float16. Automatically use original MLPs instead.")       # ngp_network.py:58
            self.density_mlp = nn.Sequential(nn.Linear(self.pos_encoder.out_dim, density_n_neurons, bias=False), nn.ReLU(),
                                             nn.Linear(density_n_neurons, 16, bias=False)).to(dev)
            self.rgb_mlp = nn.Sequential(nn.Linear(self.dir_encoder.out_dim + 16, rgb_n_neurons, bias=False), nn.ReLU(),
                                         nn.Linear(rgb_n_neurons, rgb_n_neurons, bias=False), nn.ReLU(), nn.Linear(rgb_n_neurons, 3, bias=False)).to(dev)
            for m in list(self.density_mlp) + list(self.rgb_mlp):
                if isinstance(m, nn.Linear):
                    with torch.no_grad():
                        m.weight.copy_(invariant_uniform(m.out_features, m.in_features, dev))
        self._bufs = {}

    # ---- scratch owned by the module (no per-step allocation; SURVEY.md §8b "ops never allocate")
    def _feat_buffer(self, n):
        b = self._bufs.get("feat")
        if b is None or b.shape[1] != n:
            b = self._bufs["feat"] = torch.empty((16, n, 2), dtype=torch.float16, device=self.pos_encoder.m_grid.device)
        return b

    def _flat_weight_grad(self):
        """density_mlp.con_weights.grad and rgb_mlp.con_weights.grad as two views of ONE fp32[10240] buffer (slab layout of ngp_field_bwd)"""
        g = self._bufs.get("wgrad")
        dg, cg = self.density_mlp.con_weights.grad, self.rgb_mlp.con_weights.grad
        if g is None or dg is None or cg is None or dg.data_ptr() != g.data_ptr() or cg.data_ptr() != g[3072:].data_ptr():
            g = torch.zeros(10240, dtype=torch.float32, device=self.pos_encoder.m_grid.device)
            if dg is not None:
                g[:3072] += dg
            if cg is not None:
                g[3072:] += cg
            self.density_mlp.con_weights.grad, self.rgb_mlp.con_weights.grad = g[:3072], g[3072:]
            self._bufs["wgrad"] = g
        return g

    def _bwd_buffers(self, n):
        key = ("bwd", n)
        if key not in self._bufs:
            dev = self.pos_encoder.m_grid.device
            self._bufs = {k: v for k, v in self._bufs.items() if not (isinstance(k, tuple) and k[0] == "bwd")}
            self._bufs[key] = (torch.empty((16, n, 2), dtype=torch.float16, device=dev),
                               torch.empty((ops.field_bwd_slabs(n), 10240), dtype=torch.float32, device=dev),
                               torch.empty(10240, dtype=torch.float32, device=dev))
        return self._bufs[key]

    def packed_weights(self, refresh=True):
        """MFMA-ordered fragments of both weight packs (ngp_field_pack_weights); rebuilt from the current fp16 weights when `refresh`"""
        buf = getattr(self, "_packed", None)
        if buf is None or refresh:
            wd, wc = self.density_mlp.half_weights(), self.rgb_mlp.half_weights()
            if buf is None:
                buf = self._packed = torch.empty(ops.PACKED_WEIGHT_HALVES, dtype=torch.float16, device=wd.device)
            ops.field_pack_weights(wd, wc, out=buf)
        return buf

    def forward(self, pos_input, dir_input):
        if self.fused:
            sampler = self.cfg.sampler_obj
            n_valid = sampler.n_valid_for(pos_input) if sampler is not None and hasattr(sampler, "n_valid_for") else None
            if torch.is_grad_enabled():
                return _FusedField.apply(pos_input, dir_input, self.pos_encoder.m_grid, self.density_mlp.con_weights, self.rgb_mlp.con_weights, self, n_valid)
            enc = self.pos_encoder
            feat = ops.hash_encode_fwd(pos_input, enc.table_for_kernels(), enc.level_table, layout=ops.LAYOUT_SOA, n_valid=n_valid)
            return ops.field_fwd(feat, dir_input, self.density_mlp.half_weights(), self.rgb_mlp.half_weights(), layout=ops.LAYOUT_SOA, out_dtype=torch.float16, n_valid=n_valid)
        # generic path == the reference's execute_ (ngp_network.py:77-84)
        d = self.dir_encoder(dir_input)
        p = self.pos_encoder(pos_input)
        density = self.density_mlp(p.float())
        rgb = self.rgb_mlp(torch.cat([density, d.float()], -1))
        out = torch.cat([rgb, density[..., :1]], -1)
        return out.half() if self.using_fp16 else out

    def density(self, pos_input):
        if self.fused:
            enc = self.pos_encoder
            n = pos_input.shape[0]
            feat = ops.hash_encode_fwd(pos_input, enc.table_for_kernels(), enc.level_table, layout=ops.LAYOUT_SOA)
            return ops.density_fwd(feat, self.density_mlp.half_weights(), n, layout=ops.LAYOUT_SOA, out_dtype=torch.float16).view(n, 1)
        return self.density_mlp(self.pos_encoder(pos_input).float())[:, :1]

    def set_fp16(self):
        pass   # parameters stay fp32 masters; fp16 shadows are maintained by the optimiser sweep
