"""NGPNetworks / FMLP behind the reference's NETWORKS registry (python/jnerf/models/networks/ngp_network.py:8-96).

cfg.fp16 (projects/ngp/configs/ngp_fox.py:73): both MLPs + SH + concats run in ONE fp16-MFMA kernel with weights in LDS
(csrc/field_mlp.hip), fed level-major by the XCD-aware hash kernel; backward recomputes the forward in-kernel.
fp32 (ngp_base.py, the lego headline): the reference falls back to plain nn.Linear chains (ngp_network.py:57-67).  Here the module structure is the
reference's (nn.Sequential of bias-free nn.Linear: same parameter names and shapes in the state dict), but with `use_fully` the five weight matrices are
VIEWS of one flat fp32[10240] pack in the fused kernels' layout and the network runs as ONE fp32-MFMA kernel (csrc/field32.hip, v_mfma_f32_16x16x4_f32);
`use_fully = False` keeps the rocBLAS/autograd chain (the generic path the fused kernels are tested against)."""
import math
import sys
import torch
from torch import nn
from . import ops
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, ENCODERS


def invariant_uniform_bound(in_f):
    """Jittor's init.invariant_uniform lives inside Jittor (external, not installable here): U(+-sqrt(g / fan_in)).  g = 3 (variance 1 / fan_in) is SURVEY.md's reading and
    the default of rounds 1-3; a second reading of Jittor's init.py has bound = sqrt(1 / fan) - PyTorch's nn.Linear default - i.e. g = 1.  Nothing in /root/reference
    decides it (DESIGN.md 'parity unpinned'); `invariant_uniform_gain = 1.0` in the config selects the other reading so that the two can be trained side by side."""
    g = get_cfg().invariant_uniform_gain
    return math.sqrt((3.0 if g is None else float(g)) / in_f)


def invariant_uniform(out_f, in_f, device):
    b = invariant_uniform_bound(in_f)
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
    """hash encode (level-major) -> fused SH + density MLP + colour MLP;  backward: fused dgrad/wgrad -> scatter.  fp16 or fp32 kernels by net.fused_dtype."""

    @staticmethod
    def forward(ctx, pos, dirs, net, n_valid, *params):
        enc = net.pos_encoder
        n = pos.shape[0]
        if pos.stride(0) != 3:
            pos = pos.contiguous()      # compact [n,3] copy (3 MB): the 16 level passes then stream 12 B/sample out of L2 instead of 28 B records
        feat = net._feat_buffer(n)
        ops.hash_encode_fwd(pos, enc.table_for_kernels(), enc.level_table, out=feat, layout=ops.LAYOUT_SOA, n_valid=n_valid)
        if net.fused_dtype == torch.float16:
            out = ops.field_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out_dtype=torch.float16, n_valid=n_valid, packed=net.packed_weights(refresh=True))
        else:
            out = ops.field32_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, n_valid=n_valid, packed=net.packed_weights(refresh=True))
        ctx.net, ctx.n_valid = net, n_valid
        ctx.save_for_backward(pos, dirs, feat)
        return out

    @staticmethod
    def backward(ctx, dout):
        pos, dirs, feat = ctx.saved_tensors
        net, n_valid = ctx.net, ctx.n_valid
        enc = net.pos_encoder
        n = pos.shape[0]
        dfeat, slabs = net._bwd_buffers(n)
        packed = net.packed_weights(refresh=False)      # the fragments the forward of this step built (the weights have not changed since)
        if net.fused_dtype == torch.float16:
            ops.field_bwd(feat, dirs, None, None, dout.contiguous(), layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed)
        else:
            ops.field32_bwd(feat, dirs, None, None, dout.contiguous().float(), layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed)
        ops.reduce_slabs(slabs, out=net._flat_weight_grad(), accumulate=True)     # every MLP parameter's .grad is a view of this one buffer
        enc.accumulate_grad(pos, dfeat, ops.LAYOUT_SOA, n_valid=n_valid)
        return (None,) * len(ctx.needs_input_grad)      # the gradients were accumulated into the parameters' .grad buffers above


# offsets (floats) of the five weight matrices inside the flat pack the fused kernels read: wd = W0 [64,32] | W1 [16,64];  wc = V0 [64,32] | V1 [64,64] | V2 [16,64] (3 rows used)
_PACK32 = ((0, 64, 32), (2048, 16, 64), (3072, 64, 32), (5120, 64, 64), (9216, 3, 64))


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
        shapes_ok = bool(density_n_neurons == 64 and rgb_n_neurons == 64 and self.pos_encoder.out_dim == 32 and self.dir_encoder.out_dim == 16
                         and hasattr(self.pos_encoder, "level_table") and density_hidden_layer == 1 and rgb_hidden_layer == 2)
        self.fused = bool(self.use_fully and shapes_ok and torch.device(dev).type == "cuda")
        self.fused_dtype = torch.float16 if self.using_fp16 else torch.float32
        self._pack32 = None
        if self.fused and self.using_fp16:
            self.density_mlp = FMLP([self.pos_encoder.out_dim, density_n_neurons, 16], dev)
            self.rgb_mlp = FMLP([self.dir_encoder.out_dim + 16, rgb_n_neurons, rgb_n_neurons, 3], dev)
        else:
            if self.use_fully and not self.fused:
                print("Warning: the fused field kernels need the standard 32->64->16 / 32->64->64->3 shapes on a GPU. Automatically use original MLPs instead.", file=sys.stderr)   # ngp_network.py:58
            self.density_mlp = nn.Sequential(nn.Linear(self.pos_encoder.out_dim, density_n_neurons, bias=False), nn.ReLU(),
                                             nn.Linear(density_n_neurons, 16, bias=False)).to(dev)
            self.rgb_mlp = nn.Sequential(nn.Linear(self.dir_encoder.out_dim + 16, rgb_n_neurons, bias=False), nn.ReLU(),
                                         nn.Linear(rgb_n_neurons, rgb_n_neurons, bias=False), nn.ReLU(), nn.Linear(rgb_n_neurons, 3, bias=False)).to(dev)
            lins = self._linears()
            if self.fused:      # fp32 fused path: the five nn.Linear weights become views of ONE flat buffer in the kernels' pack layout (padding rows of the last layer stay zero)
                self._pack32 = torch.zeros(10240, dtype=torch.float32, device=dev)
                for m, (off, o, i) in zip(lins, _PACK32):
                    m.weight = nn.Parameter(self._pack32[off:off + o * i].view(o, i))
            for m in lins:
                with torch.no_grad():
                    m.weight.copy_(invariant_uniform(m.out_features, m.in_features, dev))
        self._bufs = {}
        self.grad_pack_listeners = []       # called with the flat weight-gradient buffer whenever _flat_weight_grad (re)creates it (Runner: Adam.register_grad_pack)

    def _linears(self):
        return [m for m in list(self.density_mlp) + list(self.rgb_mlp) if isinstance(m, nn.Linear)]

    def load_state_dict(self, *a, **k):
        self._weights_version = getattr(self, "_weights_version", 0) + 1        # weights changed outside the optimiser: cached MFMA fragments (fastpath.py) are stale
        return super().load_state_dict(*a, **k)

    def flat_param_views(self):
        """(flat fp32 pack, [parameters that are views of it, in pack order]) for the optimiser, or None: Adam keeps ONE flat m / v buffer for them so the fused sweep
        updates all five matrices (and the zero padding, which stays zero: g = m = v = 0) in one launch"""
        if self._pack32 is None:
            return None
        return self._pack32, [m.weight for m in self._linears()]

    # ---- scratch owned by the module (no per-step allocation; SURVEY.md §8b "ops never allocate")
    def _feat_buffer(self, n):
        b = self._bufs.get("feat")
        if b is None or b.shape[1] != n:
            b = self._bufs["feat"] = torch.empty((16, n, 2), dtype=self.fused_dtype, device=self.pos_encoder.m_grid.device)
        return b

    def mlp_params(self):
        return [self.density_mlp.con_weights, self.rgb_mlp.con_weights] if isinstance(self.density_mlp, FMLP) else [m.weight for m in self._linears()]

    def _flat_weight_grad(self):
        """every MLP parameter's .grad as a view of ONE fp32[10240] buffer (slab layout of ngp_field_bwd / ngp_field32_bwd)"""
        g = self._bufs.get("wgrad")
        if self._pack32 is None:
            spans = [(self.density_mlp.con_weights, 0, 3072), (self.rgb_mlp.con_weights, 3072, 7168)]
        else:
            spans = [(m.weight, off, o * i) for m, (off, o, i) in zip(self._linears(), _PACK32)]
        ok = g is not None and all(p.grad is not None and p.grad.data_ptr() == g[off:].data_ptr() for p, off, _ in spans)
        if not ok:
            g = torch.zeros(10240, dtype=torch.float32, device=self.pos_encoder.m_grid.device)
            for p, off, cnt in spans:
                if p.grad is not None:
                    g[off:off + cnt] += p.grad.reshape(-1)
                p.grad = g[off:off + cnt].view_as(p)
            self._bufs["wgrad"] = g
            for fn in self.grad_pack_listeners:
                fn(g)
        return g

    def _bwd_buffers(self, n):
        key = ("bwd", n)
        if key not in self._bufs:
            dev = self.pos_encoder.m_grid.device
            self._bufs = {k: v for k, v in self._bufs.items() if not (isinstance(k, tuple) and k[0] == "bwd")}
            ns = ops.field_bwd_slabs(n) if self.fused_dtype == torch.float16 else ops.field32_bwd_slabs(n)
            self._bufs[key] = (torch.empty((16, n, 2), dtype=self.fused_dtype, device=dev), torch.empty((ns, 10240), dtype=torch.float32, device=dev))
        return self._bufs[key]

    def weight_packs(self):
        """(wd, wc) in the dtype the fused kernels read (reading them completes a deferred all-reduce + sweep)"""
        if self._pack32 is None:
            return self.density_mlp.half_weights(), self.rgb_mlp.half_weights()
        from .optim import flush_all
        flush_all()
        return self._pack32[:3072], self._pack32[3072:]

    def packed_weights(self, refresh=True):
        """MFMA-ordered fragments of both weight packs (ngp_field_pack_weights / ngp_field32_pack_weights); rebuilt from the current weights when `refresh`"""
        buf = getattr(self, "_packed", None)
        if buf is None or refresh:
            wd, wc = self.weight_packs()
            if self.fused_dtype == torch.float16:
                if buf is None:
                    buf = self._packed = torch.empty(ops.PACKED_WEIGHT_HALVES, dtype=torch.float16, device=wd.device)
                ops.field_pack_weights(wd, wc, out=buf)
            else:
                if buf is None:
                    buf = self._packed = torch.empty(ops.PACKED32_WEIGHT_FLOATS, dtype=torch.float32, device=wd.device)
                ops.field32_pack_weights(wd, wc, out=buf)
        return buf

    def forward(self, pos_input, dir_input):
        if self.fused:
            sampler = self.cfg.sampler_obj
            n_valid = sampler.n_valid_for(pos_input) if sampler is not None and hasattr(sampler, "n_valid_for") else None
            if torch.is_grad_enabled():
                return _FusedField.apply(pos_input, dir_input, self, n_valid, self.pos_encoder.m_grid, *self.mlp_params())
            enc = self.pos_encoder
            feat = ops.hash_encode_fwd(pos_input, enc.table_for_kernels(), enc.level_table, layout=ops.LAYOUT_SOA, n_valid=n_valid)
            wd, wc = self.weight_packs()
            if self.fused_dtype == torch.float16:
                return ops.field_fwd(feat, dir_input, wd, wc, layout=ops.LAYOUT_SOA, out_dtype=torch.float16, n_valid=n_valid)
            return ops.field32_fwd(feat, dir_input, wd, wc, layout=ops.LAYOUT_SOA, n_valid=n_valid)
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
            wd, _ = self.weight_packs()
            if self.fused_dtype == torch.float16:
                return ops.density_fwd(feat, wd, n, layout=ops.LAYOUT_SOA, out_dtype=torch.float16).view(n, 1)
            return ops.density32_fwd(feat, wd, n, layout=ops.LAYOUT_SOA).view(n, 1)
        return self.density_mlp(self.pos_encoder(pos_input).float())[:, :1]

    def set_fp16(self):
        pass   # parameters stay fp32 masters; fp16 shadows are maintained by the optimiser sweep
