"""Autograd-free sequencing of one training iteration for the standard Instant-NGP stack (HashEncoder + SHEncoder + fused NGPNetworks +
DensityGridSampler + HuberLoss + Adam/ExpDecay/EMA).  Launches exactly the kernels the module path launches (the `torch.autograd.Function`s of
encoders.py / network.py / sampler.py / losses.py), in the same order, on the same buffers — it only removes the per-launch Python
overhead of nn.Module.__call__ / autograd graph construction, which at ~1 ms per iteration had become the bottleneck.
Runner.train_step uses it automatically; `fast_path = False` in the config (or any non-standard component) selects the module path."""
import torch
from . import ops


class FusedTrainStep:
    @staticmethod
    def applicable(runner):
        from .network import NGPNetworks
        from .sampler import DensityGridSampler
        from .losses import HuberLoss
        from .optim import Adam, ExpDecay, EMA
        m, o, e = runner.model, runner.optimizer, runner.ema_optimizer
        return (runner.cfg.fast_path is not False and isinstance(m, NGPNetworks) and m.fused and isinstance(runner.sampler, DensityGridSampler)
                and isinstance(runner.loss_func, HuberLoss) and isinstance(o, ExpDecay) and isinstance(o._nested_optimizer, Adam)
                and isinstance(e, EMA) and e._adam is o._nested_optimizer)

    def __init__(self, runner):
        self.r = runner
        m = runner.model
        self.enc, self.dm, self.cm = m.pos_encoder, m.density_mlp, m.rgb_mlp
        self.s = runner.sampler
        dev = self.enc.m_grid.device
        n = self.s.target_batch_size
        self.n = n
        self.out = torch.empty((n, 4), dtype=torch.float16, device=dev)
        self.dout = torch.empty((n, 4), dtype=torch.float16, device=dev)
        self._per_rays = {}

    def _ray_bufs(self, nr, dev):
        b = self._per_rays.get(nr)
        if b is None:
            if len(self._per_rays) > 8:
                self._per_rays.clear()
            b = self._per_rays[nr] = tuple(torch.empty((nr, 3), dtype=torch.float32, device=dev) for _ in range(3))
        return b

    def __call__(self, b):
        r, s, enc, m = self.r, self.s, self.enc, self.r.model
        coords, numsteps, numsteps_c, n_valid = s._coords, s._rays_numsteps, s._rays_numsteps_compacted, s._n_valid
        n = self.n
        nr = numsteps.shape[0]
        rgb, loss, lgrad = self._ray_bufs(nr, coords.device)
        packed = m.packed_weights(refresh=True)                            # (reading the weights also completes a deferred all-reduce + sweep)
        table = enc.table_for_kernels()
        dirs = coords[:, 4:]
        pos = s._pos_train                                                # compact [n,3] positions (the marcher writes them next to the 28-byte records)
        feat = m._feat_buffer(n)
        ops.hash_encode_fwd(pos, table, enc.level_table, out=feat, layout=ops.LAYOUT_SOA, n_valid=n_valid)
        ops.field_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out=self.out, n_valid=n_valid, packed=packed)
        ops.composite_fwd_huber(self.out, coords, numsteps, numsteps_c, b["bg"], b["target"], r.loss_func.delta, s.NERF_CASCADES, out=rgb, loss=loss, grad=lgrad)
        ops.composite_bwd(self.out, coords, numsteps_c, lgrad, rgb, s.density_grid_mean, s.NERF_CASCADES, dout=self.dout, zero_first=False)
        dfeat, slabs, _ = m._bwd_buffers(n)
        ops.field_bwd(feat, dirs, None, None, self.dout, layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed)
        ops.reduce_slabs(slabs, out=m._flat_weight_grad(), accumulate=True)
        enc.accumulate_grad(pos, dfeat, ops.LAYOUT_SOA, n_valid=n_valid)
        r.optimizer.step(None)            # ExpDecay lr schedule -> Adam.step without a loss: all-reduce (data parallel) + bookkeeping
        r.ema_optimizer.ema_step()        # fused Adam + EMA sweep (deferred to the next parameter read under data parallelism)
        return loss
