"""Adam / ExpDecay / EMA behind the reference's OPTIMS registry (python/jnerf/optims/{adam,expdecay,ema}.py).

API kept: optimizer.step(loss) (back-propagates the SUM of an unreduced loss like Jittor), optimizer._nested_optimizer,
ema_optimizer.ema_step(), .param_groups[*]['params'|'values'|'m'], .state_dict().
MI355X execution: Adam + EMA + gradient zeroing + fp16-shadow refresh are ONE streaming kernel per parameter tensor
(csrc/optim.hip): Adam.step() defers the sweep until EMA.ema_step() when an EMA is attached (Runner builds both, runner.py:35-37),
so parameters are read and written once per step instead of ~12 times."""
import weakref
import torch
import torch.distributed as dist
from . import ops
from .utils.registry import OPTIMS

_LIVE = weakref.WeakSet()


def flush_all():
    """complete every deferred gradient all-reduce + parameter sweep (called by the modules right before parameters are read)"""
    for o in list(_LIVE):
        o.flush()


def sync_all_sharded():
    """data parallel with the sharded sweep (fastpath.py): collect the state that lives on its owner's shard only - Adam moments, in fp16 mode also the fp32
    master of the hash table - on every rank.  Called before anything reads that state as a whole: checkpoints, state_dict(), replica comparisons."""
    for o in list(_LIVE):
        o.sync_sharded_state()


@OPTIMS.register_module()
class Adam:
    # data parallel, fp16 gradients on the wire: the fp32 gradient carries the compositor's 128 / n_rays loss scale (values of 1e-7 .. 1e-3), i.e. it sits in and below
    # fp16's subnormal range; it travels multiplied by 2^14 (sums over <= 8 ranks stay far below 65504) and the sweep divides it out again
    DP_HALF_SCALE = 16384.0

    def __init__(self, params, lr=1e-1, eps=1e-15, betas=(0.9, 0.99), **kwargs):
        if kwargs:      # jt.nn.Adam also takes weight_decay; the fused sweep does not implement it, and silently training without it would be worse than failing
            raise TypeError(f"Adam: unsupported arguments {sorted(kwargs)} (supported: lr, eps, betas)")
        self.lr, self.eps, self.betas = lr, eps, tuple(betas)
        params = [p for p in params]
        self.param_groups = [{"params": params, "values": [torch.zeros_like(p) for p in params], "m": [torch.zeros_like(p) for p in params]}]
        self.n_step = 0
        self._ema = None
        self._pending = False
        self._half = {}
        self._comm_stream = None
        self._comm_pending = False
        self._comm_half = {}            # id(param) -> fp16 communication buffer (large fp32 gradients travel as fp16, see allreduce_grads)
        self._eff_grad = {}             # id(param) -> the reduced fp16 buffer the next sweep reads instead of p.grad
        self._deferred_ema = None
        self._flat = None               # (flat parameter pack, flat m, flat v, ids of the parameters that are views of it), see use_flat_state
        self._grad_packs = set()        # data_ptr of flat gradient buffers whose views tile them (plus zero padding): safe to all-reduce as ONE collective
        self._sharded = None            # (NgpDpPlan, [tensors valid on the owner's shard only]) while the native data-parallel step runs the sharded sweep
        self._sharded_dirty = False
        _LIVE.add(self)

    @property
    def defaults(self):
        return {"lr": self.lr, "eps": self.eps, "betas": self.betas, "n_step": self.n_step}

    def attach_half_shadows(self, model):
        """fp16 copies the gather/MFMA kernels read; refreshed by the sweep"""
        for m in model.modules():
            if getattr(m, "m_grid_half", None) is not None:
                self._half[id(m.m_grid)] = m.m_grid_half
            if getattr(m, "con_weights_half", None) is not None:
                self._half[id(m.con_weights)] = m.con_weights_half

    def use_flat_state(self, flat_pack, params):
        """`params` are views into the contiguous fp32 buffer `flat_pack` (NGPNetworks' fp32 weight pack): their first / second moments become views of ONE flat
        buffer each, so the native step sweeps the whole pack - padding included, which stays zero because g = m = v = 0 there - in a single launch"""
        fm, fv = torch.zeros_like(flat_pack), torch.zeros_like(flat_pack)
        pg, base = self.param_groups[0], flat_pack.data_ptr()
        for p in params:
            i = next(k for k, q in enumerate(pg["params"]) if q is p)
            off = (p.data_ptr() - base) // p.element_size()
            pg["m"][i] = fm[off:off + p.numel()].view_as(p)
            pg["values"][i] = fv[off:off + p.numel()].view_as(p)
        self._flat = (flat_pack, fm, fv, [id(p) for p in params])

    def register_grad_pack(self, flat_grad):
        """`flat_grad` is a contiguous buffer every element of which is either the gradient of one of this optimiser's parameters or zero padding (NGPNetworks'
        fp32[10240] weight-gradient pack): allreduce_grads then sends it as one collective instead of one per view"""
        self._grad_packs.add(flat_grad.data_ptr())

    def register_sharded(self, plan, tensors):
        """the native data-parallel step (csrc/train_step.hip) sweeps only this rank's shard of the hash table: `tensors` (Adam moments; in fp16 mode the fp32
        master too) are valid on the owner's shard only until sync_sharded_state() all-gathers them"""
        self._sharded = (plan, list(tensors))

    def mark_sharded_dirty(self):
        self._sharded_dirty = self._sharded is not None

    def sync_sharded_state(self):
        if self._sharded is None or not self._sharded_dirty:
            return
        from . import dp
        dp.allgather_shards(*self._sharded)
        self._sharded_dirty = False

    def zero_grad(self):
        for p in self.param_groups[0]["params"]:
            if p.grad is not None:
                p.grad.zero_()

    def backward(self, loss):
        # d(sum(loss))/d(loss) == 1: start autograd from an expanded scalar instead of launching a reduction whose value nobody reads
        if loss.dim() == 0:
            loss.backward()
        else:
            loss.backward(gradient=torch.ones((), dtype=loss.dtype, device=loss.device).expand_as(loss))

    @staticmethod
    def _world():
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1

    @staticmethod
    def _dp_active():
        """gradients go through the collectives: more than one rank, or `dp_force_collectives = True` in the config (a single-rank process group then runs the
        complete data-parallel sequence - fp16 conversion, RCCL all-reduce on the comm stream, deferred sweep - which is how the RCCL path is tested on a 1-GPU box)"""
        from . import dp
        return dp.active()

    def allreduce_grads(self):
        """Ray-batch data parallelism: SUM the hash-table gradient (49.9 MB fp32) and the two MLP gradients over ranks — RCCL over xGMI on
        the GPUs, gloo in the CPU unit tests.  On the GPU the collectives are issued on a side stream right after backward and are only
        waited for when the parameters are next READ (flush()): the next iteration's ray generation and occupancy-grid marching — which
        do not depend on the parameters — run underneath the all-reduce."""
        if not self._dp_active():
            return
        grads = [p.grad for p in self.param_groups[0]["params"] if p.grad is not None]
        # gradients that are views tiling one flat buffer (the two MLP packs, network.py:_flat_weight_grad) travel as ONE collective
        by_base = {}
        for g in grads:
            by_base.setdefault(id(g._base) if g._base is not None else id(g), []).append(g)
        merged = []
        for group in by_base.values():
            base = group[0]._base
            if base is not None and len(group) > 1 and base.is_contiguous() and base.data_ptr() in self._grad_packs:   # a REGISTERED pack only: its zero padding travels along, one collective instead of five
                merged.append(base)
            else:
                merged.extend(group)
        grads = merged
        if grads and grads[0].is_cuda:
            from .utils.config import get_cfg
            half_ok = get_cfg().dp_grad_dtype != "fp32" and bool(get_cfg().fp16)      # `dp_grad_dtype = "fp32"` in the config keeps full-width collectives
            main = torch.cuda.current_stream()
            send = []
            for g in grads:
                owner = next((p for p in self.param_groups[0]["params"] if p.grad is g), None)
                if half_ok and owner is not None and g.dtype == torch.float32 and g.numel() >= (1 << 20) and g.numel() % 8 == 0:
                    # the hash-table gradient: 52 MB fp32 -> 26 MB fp16 over xGMI (the reference's gradients are fp16 in the first place);
                    # one streaming pass converts and zeroes the fp32 buffer, the sweep then reads the reduced fp16 buffer
                    hb = self._comm_half.get(id(owner))
                    if hb is None or hb.numel() != g.numel():
                        hb = self._comm_half[id(owner)] = torch.empty(g.numel(), dtype=torch.float16, device=g.device)
                    ops.grad_to_half(g.view(-1), hb, zero_src=True, scale=self.DP_HALF_SCALE)
                    self._eff_grad[id(owner)] = hb
                    send.append(hb)
                else:
                    send.append(g)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(main)
            from . import dp
            with torch.cuda.stream(self._comm_stream):
                if dp.library_comm() is not None:       # RCCL from inside the library: one group = one launch (ngp_allreduce_grads)
                    dp.allreduce_grads(send)
                else:
                    for g in send:
                        dist.all_reduce(g, op=dist.ReduceOp.SUM)
            self._comm_pending = True
        else:
            for g in grads:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)

    def flush(self):
        if self._comm_pending:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
            self._comm_pending = False
        if self._deferred_ema is not None:
            ema, self._deferred_ema = self._deferred_ema, None
            self._sweep(ema if ema is not False else None)

    def step(self, loss=None):
        if loss is not None:
            self.backward(loss)
        self.allreduce_grads()
        self.n_step += 1
        if self._ema is not None:
            self._pending = True                    # the fused sweep runs in EMA.ema_step()
        elif self._comm_pending:
            self._deferred_ema = False              # sweep (without EMA) once the all-reduce has landed
        else:
            self._sweep(None)

    @torch.no_grad()
    def _sweep(self, ema):
        pg = self.param_groups[0]
        lr = pg.get("lr", self.lr)          # Jittor's optimisers read a per-group learning rate first (`pg.get("lr", self.lr)`): NeuSRunner.update_learning_rate writes it there
        for i, p in enumerate(pg["params"]):
            if p.grad is None:
                continue
            e = ema.param_groups[0]["values"][i] if ema is not None else None
            # the fused kernel streams 16-byte vectors; a 1- or 3-element bias (OriginNeRFNetworks' alpha / rgb heads) takes the same update in plain torch ops
            if p.is_cuda and p.numel() % 4 == 0 and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0 and p.is_contiguous():
                g_eff = self._eff_grad.pop(id(p), None)            # reduced fp16 gradient of the data-parallel path (p.grad was zeroed by the conversion pass)
                ops.adam_ema_step(p.data, g_eff.view_as(p.grad) if g_eff is not None else p.grad, pg["m"][i], pg["values"][i], e, self._half.get(id(p)), lr, self.n_step, self.betas[0], self.betas[1], self.eps,
                                  ema.decay if ema is not None else 0.0, zero_grad=True, grad_mul=(1.0 / self.DP_HALF_SCALE) if g_eff is not None else 1.0)
            else:                                   # CPU tensors (gloo unit tests of the data-parallel logic) and tiny / unaligned tensors: same math in torch
                b0, b1 = self.betas
                g = p.grad
                m, v = pg["m"][i], pg["values"][i]
                m.mul_(b0).add_(g, alpha=1 - b0)
                v.mul_(b1).addcmul_(g, g, value=1 - b1)
                step_size = lr * (1 - b1 ** self.n_step) ** 0.5 / (1 - b0 ** self.n_step)
                # fused-EMA mode stores the EMA IN the parameter (EMA.attach aliases values[i] = p.data): the blend must read the value from BEFORE the Adam
                # update, exactly like the kernel's `E = P` (without the snapshot the EMA of these tensors was a silent no-op)
                e_old = p.data.clone() if (ema is not None and e.data_ptr() == p.data_ptr()) else e
                p.data.sub_(m * step_size / (v.sqrt() + self.eps))
                if ema is not None:
                    d, k = ema.decay, self.n_step
                    p.data.copy_(((1 - d) * p.data + d * e_old * (1 - d ** (k - 1))) / (1 - d ** k))
                    e.copy_(p.data)
                h = self._half.get(id(p))
                if h is not None:
                    h.copy_(p.data)
                g.zero_()
        self._pending = False

    def state_dict(self):
        self.flush()
        self.sync_sharded_state()
        return {"defaults": {"lr": self.lr, "eps": self.eps, "betas": self.betas, "n_step": self.n_step,
                             "param_groups": [{"values": [t.detach().cpu() for t in self.param_groups[0]["values"]], "m": [t.detach().cpu() for t in self.param_groups[0]["m"]]}]}}

    def load_state_dict(self, sd):
        d = sd["defaults"]
        self.lr, self.eps, self.betas, self.n_step = d["lr"], d["eps"], tuple(d["betas"]), d["n_step"]
        for k in ("values", "m"):
            for dst, src in zip(self.param_groups[0][k], d["param_groups"][0][k]):
                dst.copy_(src)


@OPTIMS.register_module()
class ExpDecay:
    """optims/expdecay.py:7-30"""

    def __init__(self, nested_optimizer, decay_start, decay_interval, decay_base, decay_end=None):
        self.base_lr = nested_optimizer.lr
        self._nested_optimizer = nested_optimizer
        self.decay_start, self.decay_interval, self.decay_base = decay_start, decay_interval, decay_base
        self.decay_end = 10000000 if decay_end is None else decay_end
        self.steps = 0
        self.m_learning_rate_factor = 1

    def advance_schedule(self):
        """the learning-rate part of step() (expdecay.py:20-23)"""
        if self.steps >= self.decay_start and (self.steps - self.decay_start) % self.decay_interval == 0 and self.steps <= self.decay_end:
            self.m_learning_rate_factor *= self.decay_base
        self._nested_optimizer.lr = self.base_lr * self.m_learning_rate_factor

    def step(self, loss=None):
        self.advance_schedule()
        self._nested_optimizer.step(loss)
        self.steps += 1

    def zero_grad(self):
        return self._nested_optimizer.zero_grad()

    def backward(self, loss, retain_graph=False):
        return self._nested_optimizer.backward(loss)

    def state_dict(self):
        return {"steps": self.steps, "m_learning_rate_factor": self.m_learning_rate_factor}

    def load_state_dict(self, sd):
        self.steps, self.m_learning_rate_factor = sd["steps"], sd["m_learning_rate_factor"]


@OPTIMS.register_module()
class EMA:
    """optims/ema.py:8-37 — the trained parameters ARE their debiased EMA (ema_step overwrites them)."""

    def __init__(self, params, decay):
        params = [p for p in params]
        self.decay = decay
        self.steps = 0
        self.param_groups = [{"params": params, "values": [p.detach().clone() for p in params]}]
        self._adam = None

    def attach(self, adam):
        """fuse with the Adam sweep (Runner wires this; without it ema_step runs standalone).  ema_step ends with v <- p (ema.py:37), so at every step
        boundary the stored EMA equals the parameter: in fused mode `values` ALIAS the parameters and the sweep neither reads nor writes a
        separate EMA buffer (8 B/parameter less traffic)."""
        adam = getattr(adam, "_nested_optimizer", adam)
        self._adam = adam
        adam._ema = self
        if self.steps == 0 and all(p.is_cuda for p in self.param_groups[0]["params"]):
            self.param_groups[0]["values"] = [p.data for p in self.param_groups[0]["params"]]

    @torch.no_grad()
    def ema_step(self, loss=None):
        assert loss is None
        self.steps += 1
        if self._adam is not None and self._adam._pending:
            assert self._adam.n_step == self.steps, "EMA and Adam must be stepped in lock-step to be fused"
            if self._adam._comm_pending:
                self._adam._pending = False
                self._adam._deferred_ema = self     # sweep at the next parameter read, after the all-reduce
            else:
                self._adam._sweep(self)
            return
        old = 1 - self.decay ** (self.steps - 1)
        new = 1 / (1 - self.decay ** self.steps)
        for p, v in zip(self.param_groups[0]["params"], self.param_groups[0]["values"]):
            p.data.copy_(((1 - self.decay) * p.data + self.decay * v * old) * new)
            v.copy_(p.data)
        if self._adam is not None:
            for p in self.param_groups[0]["params"]:
                h = self._adam._half.get(id(p))
                if h is not None:
                    h.copy_(p.data)

    def state_dict(self):
        return {"defaults": {"decay": self.decay, "steps": self.steps, "param_groups": [{"values": [t.detach().cpu() for t in self.param_groups[0]["values"]]}]}}

    def load_state_dict(self, sd):
        d = sd["defaults"]
        self.decay, self.steps = d["decay"], d["steps"]
        for dst, src in zip(self.param_groups[0]["values"], d["param_groups"][0]["values"]):
            dst.copy_(src)
