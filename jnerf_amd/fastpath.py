"""Autograd-free sequencing of one training iteration for the standard Instant-NGP stack (HashEncoder + SHEncoder + fused NGPNetworks +
DensityGridSampler + HuberLoss + Adam/ExpDecay/EMA).  Launches exactly the kernels the module path launches (the `torch.autograd.Function`s of
encoders.py / network.py / sampler.py / losses.py), in the same order, on the same buffers — it only removes the per-launch Python
overhead of nn.Module.__call__ / autograd graph construction, which at ~1 ms per iteration had become the bottleneck.
Runner.train_step uses it automatically; `fast_path = False` in the config (or any non-standard component) selects the module path."""
import ctypes as C
import torch
import torch.distributed as dist
from . import _lib as L
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
        self.enc = m.pos_encoder
        self.s = runner.sampler
        dev = self.enc.m_grid.device
        n = self.s.target_batch_size
        self.n = n
        self.half = m.fused_dtype == torch.float16            # fp16 table shadow + fp16-MFMA network (ngp_fox.py) | fp32 table + fp32-MFMA network (ngp_base.py)
        self.out = torch.empty((n, 4), dtype=m.fused_dtype, device=dev)
        self.dout = torch.empty((n, 4), dtype=m.fused_dtype, device=dev)
        self._per_rays = {}
        # the whole sequence is ONE call into the library (ngp_train_step) - also under data parallelism: with an RCCL process group the exchange step
        # (reduce-scatter -> sharded sweep -> all-gather, csrc/dp_comm.hip) runs inside that call; with any other backend (gloo in the tests) the call is split
        # in two phases around torch.distributed's all-reduce.  `native_step = False` keeps the per-stage calls of __call__ below.
        self.timed_stage = None         # name from _lib.STAGES: the library brackets that stage of every native step with HIP events (bench.py)
        self.native = runner.cfg.native_step is not False
        self._args, self._grad_sig = None, None
        self._dp_plan, self._grad_wire = None, None
        self._frags_token = None        # (optimiser step, model weights version) for which the library's fused sweep left fresh MFMA fragments in m._packed

    def _ray_bufs(self, nr, dev):
        b = self._per_rays.get(nr)
        if b is None:
            if len(self._per_rays) > 8:
                self._per_rays.clear()
            b = self._per_rays[nr] = tuple(torch.empty((nr, 3), dtype=torch.float32, device=dev) for _ in range(3))
        return b

    def _native_args(self, dev):
        """the argument block of ngp_train_step: everything that does not change from step to step is filled in once"""
        r, s, enc, m = self.r, self.s, self.enc, self.r.model
        adam, ema = r.optimizer._nested_optimizer, r.ema_optimizer
        a = L.NgpTrainStep()
        n = self.n
        P = lambda t: None if t is None else t.data_ptr()
        a.n, a.cascades, a.run_optimizer = n, s.NERF_CASCADES, 1
        a.density_grid_mean = P(s.density_grid_mean)
        self._level_table = ops._tbl(enc.level_table)                  # keeps the host array alive
        a.level_table_host = self._level_table
        a.table_grad, a.n_params = P(enc.grad_buffer()), enc.n_params
        need = ops.hash_bwd_workspace_bytes(enc.level_table, n, m.fused_dtype)           # sized for the path this precision takes (fp32: ~0.6 GB, fp16: ~1.7 GB)
        if enc._bwd_ws is None or enc._bwd_ws.numel() < need:
            enc._bwd_ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.hash_workspace, a.hash_workspace_bytes = P(enc._bwd_ws), enc._bwd_ws.numel()
        dfeat, slabs = m._bwd_buffers(n)
        a.feat, a.dfeat, a.out, a.dout = P(m._feat_buffer(n)), P(dfeat), P(self.out), P(self.dout)
        a.wgrad_slabs, a.n_slabs, a.wgrad_flat = P(slabs), slabs.shape[0], P(m._flat_weight_grad())
        a.dtype = L.F16 if self.half else L.F32
        a.grad_overwrite = 1                          # single GPU: backward overwrites the gradient buffers, the sweep does not zero them
        if getattr(m, "_packed", None) is None:
            m.packed_weights(refresh=True)
        a.packed_weights = P(m._packed)
        a.huber_delta = r.loss_func.delta
        pg, eg = adam.param_groups[0], ema.param_groups[0]
        # parameter tensors the sweep visits: every parameter on its own, except those that are views of one flat pack (fp32 network) - the pack is ONE tensor
        flat_ids = set(adam._flat[3]) if adam._flat else set()
        ents = []
        for i, p in enumerate(pg["params"]):
            if id(p) in flat_ids:
                continue
            assert p.grad is not None and p.grad.dtype == torch.float32
            if p is enc.m_grid:
                self._table_index = i
            ents.append((p.data, p.grad, pg["m"][i], pg["values"][i], eg["values"][i], adam._half.get(id(p)), p.numel()))
        if adam._flat:
            pack, fm, fv, _ = adam._flat
            first = next(i for i, p in enumerate(pg["params"]) if id(p) in flat_ids)
            assert eg["values"][first].data_ptr() == pg["params"][first].data_ptr(), "flat sweep needs the aliased EMA (EMA.attach)"
            ents.append((pack, m._flat_weight_grad(), fm, fv, pack, None, pack.numel()))
        a.n_opt = len(ents)
        assert a.n_opt <= 4
        for i, (p_, g_, m_, v_, e_, h_, cnt) in enumerate(ents):
            a.p[i], a.g[i], a.m[i], a.v[i], a.ema[i], a.p_half[i], a.numel[i] = P(p_), P(g_), P(m_), P(v_), P(e_), P(h_), cnt
        a.beta0, a.beta1, a.eps, a.ema_decay = adam.betas[0], adam.betas[1], adam.eps, ema.decay
        self._keep = (dfeat, slabs)
        # ---- data parallel: the exchange step inside the call (RCCL), or the two-phase split around the host's own collective
        from . import dp
        a.phase, a.comm, a.dp = L.PHASE_ALL, None, None
        self._dp_host_collective = False
        if dp.active():
            comm = dp.library_comm_or_fallback()
            self.dp_exchange = "in-library RCCL" if comm is not None else "torch.distributed all-reduce between the two phases of the step"
            if comm is None:
                self._dp_host_collective = True           # gloo (tests) or a failed library communicator: BACKWARD -> dist.all_reduce -> SWEEP, see _call_native
                if r.cfg.dp_host_sharded:
                    # (tests) the SHARDED sweep without RCCL: the sweep phase updates this rank's shard only and the host gathers the shards - the same plan and shard
                    # arithmetic as the in-library exchange, executed with more than one rank on a box that has one GPU
                    table_g = enc.grad_buffer().data_ptr()
                    a.dp_table = next(i for i in range(a.n_opt) if a.g[i] == table_g)
                    self._dp_plan = dp.plan(enc.level_table, enc.n_params, n_buckets=2 if r.cfg.dp_overlap else 1)
                    self._host_sharded_dp = C.addressof(self._dp_plan)
                    gather_master = not self.half
                    self._host_gather = [pg["params"][self._table_index].data] if gather_master else [adam._half[id(pg["params"][self._table_index])]]
                    adam.register_sharded(self._dp_plan, [pg["m"][self._table_index], pg["values"][self._table_index]] + ([] if gather_master else [pg["params"][self._table_index].data]))
            else:
                cfg = r.cfg
                table_g = enc.grad_buffer().data_ptr()
                a.dp_table = next(i for i in range(a.n_opt) if a.g[i] == table_g)
                a.dp_overlap = 1 if cfg.dp_overlap else 0
                self._dp_plan = dp.plan(enc.level_table, enc.n_params, n_buckets=2 if cfg.dp_overlap else 1)
                a.comm, a.dp = comm, C.addressof(self._dp_plan)
                # fp16 mode: the table gradient travels as scaled fp16 by default (the reference's gradients are fp16 to begin with).  fp32 mode (r4): opt-in,
                # `dp_grad_dtype = "fp16"` - halves the 48.8 MB reduce-scatter; every rank's gradient element is rounded once to 11 significant bits (values scaled by
                # 2^14 into fp16's normal range), i.e. 2^-12 relative per addend instead of fp32's 2^-24: the same wire the fp16 configuration always uses, NOT bit-identical
                # to the single-GPU fp32 run (the default fp32 wire is, tests/test_train_gpu.py::test_rccl_exchange_step_adds_no_arithmetic)
                if (self.half and cfg.dp_grad_dtype != "fp32") or (not self.half and cfg.dp_grad_dtype == "fp16"):
                    if self._grad_wire is None:
                        self._grad_wire = torch.empty(enc.n_params, dtype=torch.float16, device=dev)
                    a.grad_wire, a.wire_scale = self._grad_wire.data_ptr(), adam.DP_HALF_SCALE
                # fp32 mode: the kernels read the master itself -> gathered every step.  fp16 mode: they read the shadow; the fp32 master (and, in both modes, the
                # Adam moments) stay valid on their owner's shard only, until Adam.sync_sharded_state() collects them (checkpoints, state_dict, tests)
                a.dp_gather_master = 0 if self.half else 1
                sharded = [pg["m"][self._table_index], pg["values"][self._table_index]] + ([] if a.dp_gather_master else [pg["params"][self._table_index].data])
                adam.register_sharded(self._dp_plan, sharded)
        return a

    def _call_native(self, b):
        r, s, enc, m = self.r, self.s, self.enc, self.r.model
        coords, numsteps, numsteps_c = s._coords, s._rays_numsteps, s._rays_numsteps_compacted
        nr = numsteps.shape[0]
        rgb, loss, lgrad = self._ray_bufs(nr, coords.device)
        wd, wc = m.weight_packs()
        table = enc.table_for_kernels()
        sig = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in r.optimizer._nested_optimizer.param_groups[0]["params"])
        if self._args is None or sig != self._grad_sig:           # (a .grad that was re-allocated behind our back - zero_grad(set_to_none), user code - invalidates the cached pointers)
            self._args = self._native_args(coords.device)
            self._grad_sig = tuple(p.grad.data_ptr() if p.grad is not None else 0 for p in r.optimizer._nested_optimizer.param_groups[0]["params"])
        a = self._args
        ed, adam, ema = r.optimizer, r.optimizer._nested_optimizer, r.ema_optimizer
        a.frags_fresh = 1 if self._frags_token == (adam.n_step, getattr(m, "_weights_version", 0), m._packed.data_ptr()) else 0
        ed.advance_schedule()                       # == ExpDecay.step / Adam.step / EMA.ema_step bookkeeping; the sweep itself is launched by the library
        adam.n_step += 1; ed.steps += 1; ema.steps += 1
        a.n_rays, a.step, a.lr = nr, adam.n_step, adam.lr
        a.timed_stage = -1 if self.timed_stage is None else L.STAGES[self.timed_stage]
        a.coords, a.pos, a.numsteps, a.numsteps_compacted, a.n_valid = coords.data_ptr(), s._pos_train.data_ptr(), numsteps.data_ptr(), numsteps_c.data_ptr(), s._n_valid.data_ptr()
        a.bg, a.target = b["bg"].data_ptr(), b["target"].data_ptr()
        fl = b.get("flag")             # (flag tensor, value, status tensor): the batch was marched on a sampling stream and handed over by device flag (Runner.train_step)
        if fl is not None:
            a.wait_flag, a.wait_value, a.wait_status = fl[0].data_ptr(), fl[1] & 0xFFFFFFFF, fl[2].data_ptr()
        else:
            a.wait_flag, a.wait_value, a.wait_status = None, 0, None
        a.table, a.wd, a.wc = table.data_ptr(), wd.data_ptr(), wc.data_ptr()
        a.rgb, a.loss, a.loss_grad = rgb.data_ptr(), loss.data_ptr(), lgrad.data_ptr()
        if self._dp_host_collective:
            # a process group without RCCL (gloo): the library runs the iteration in two phases and the host sums the two gradient buffers in between
            a.phase = L.PHASE_BACKWARD
            L.check(L.lib().ngp_train_step(ops._stream(), C.byref(a)), "ngp_train_step(backward)")
            for g in (enc.grad_buffer(), m._flat_weight_grad()):
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
            a.phase = L.PHASE_SWEEP
            sharded = getattr(self, "_host_sharded_dp", None)
            a.dp = sharded
            L.check(L.lib().ngp_train_step(ops._stream(), C.byref(a)), "ngp_train_step(sweep)")
            a.dp = None
            if sharded:
                from . import dp
                dp.allgather_shards_host(self._dp_plan, self._host_gather)
                adam.mark_sharded_dirty()
            self._frags_token = (adam.n_step, getattr(m, "_weights_version", 0), m._packed.data_ptr())
            return loss
        L.check(L.lib().ngp_train_step(ops._stream(), C.byref(a)), "ngp_train_step")
        self._frags_token = (adam.n_step, getattr(m, "_weights_version", 0), m._packed.data_ptr())
        if a.comm:
            adam.mark_sharded_dirty()
        return loss

    def stage_timings(self, max_n=1 << 16):
        """milliseconds of the bracketed stage for every native step since the last call (synchronises)"""
        buf = (C.c_float * max_n)()
        n = L.lib().ngp_train_step_timings(buf, max_n)
        if n < 0:
            L.check(n, "ngp_train_step_timings")
        return [buf[i] for i in range(n)]

    def __call__(self, b):
        if self.native:
            return self._call_native(b)
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
        if self.half:
            ops.field_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out=self.out, n_valid=n_valid, packed=packed)
        else:
            ops.field32_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out=self.out, n_valid=n_valid, packed=packed)
        ops.composite_fwd_huber(self.out, coords, numsteps, numsteps_c, b["bg"], b["target"], r.loss_func.delta, s.NERF_CASCADES, out=rgb, loss=loss, grad=lgrad)
        ops.composite_bwd(self.out, coords, numsteps_c, lgrad, rgb, s.density_grid_mean, s.NERF_CASCADES, dout=self.dout, zero_first=False)
        dfeat, slabs = m._bwd_buffers(n)
        if self.half:
            ops.field_bwd(feat, dirs, None, None, self.dout, layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed)
        else:
            ops.field32_bwd(feat, dirs, None, None, self.dout, layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed)
        ops.reduce_slabs(slabs, out=m._flat_weight_grad(), accumulate=True)
        enc.accumulate_grad(pos, dfeat, ops.LAYOUT_SOA, n_valid=n_valid)
        r.optimizer.step(None)            # ExpDecay lr schedule -> Adam.step without a loss: all-reduce (data parallel) + bookkeeping
        r.ema_optimizer.ema_step()        # fused Adam + EMA sweep (deferred to the next parameter read under data parallelism)
        return loss
