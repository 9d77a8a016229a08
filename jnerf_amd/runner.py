"""Runner — the caller of the hot path (python/jnerf/runner/runner.py:14-264), re-hosted on torch: same construction order
through the registries and the global cfg, same train/test/render_img/save_ckpt/load_ckpt methods and checkpoint keys."""
import os
import sys
import numpy as np
import torch
from .utils.config import get_cfg
from .utils.registry import build_from_cfg, NETWORKS, DATASETS, OPTIMS, SAMPLERS, LOSSES
from .losses import img2mse, mse2psnr
from . import encoders, network, networks_ori, sampler, optim, losses, dataset  # noqa: F401  (register modules)


class Runner:
    def __init__(self):
        self.cfg = get_cfg()
        cfg = self.cfg
        if cfg.device is None:
            cfg.device = "cuda"
        if cfg.log_dir and not os.path.exists(cfg.log_dir):
            os.makedirs(cfg.log_dir, exist_ok=True)
        self.exp_name = cfg.exp_name
        self.dataset = {}
        self.dataset["train"] = build_from_cfg(cfg.dataset.train, DATASETS)
        cfg.dataset_obj = self.dataset["train"]
        self.dataset["val"] = build_from_cfg(cfg.dataset.val, DATASETS) if cfg.dataset.val else self.dataset["train"]
        self.dataset["test"] = None
        self.model = build_from_cfg(cfg.model, NETWORKS)
        cfg.model_obj = self.model
        self.sampler = build_from_cfg(cfg.sampler, SAMPLERS)
        cfg.sampler_obj = self.sampler
        params = list(self.model.parameters())
        self._sync_initial_parameters(params)
        self.optimizer = build_from_cfg(cfg.optim, OPTIMS, params=params)
        self.optimizer.attach_half_shadows(self.model)
        flat = self.model.flat_param_views() if hasattr(self.model, "flat_param_views") else None
        if flat is not None:
            self.optimizer.use_flat_state(*flat)
        if getattr(self.model, "fused", False) and hasattr(self.model, "grad_pack_listeners"):
            # the MLP gradients are views tiling one flat buffer: a data-parallel all-reduce may send that buffer as one collective (and only such registered packs)
            self.model.grad_pack_listeners.append(self.optimizer.register_grad_pack)
            self.optimizer.register_grad_pack(self.model._flat_weight_grad())
        self.optimizer = build_from_cfg(cfg.expdecay, OPTIMS, nested_optimizer=self.optimizer)
        self.ema_optimizer = build_from_cfg(cfg.ema, OPTIMS, params=params)
        self.ema_optimizer.attach(self.optimizer)                       # Adam + EMA become one fused sweep
        self.loss_func = build_from_cfg(cfg.loss, LOSSES)
        self.background_color = cfg.background_color
        self.tot_train_steps = cfg.tot_train_steps
        self.n_rays_per_batch = cfg.n_rays_per_batch
        self.save_path = os.path.join(cfg.log_dir or "./logs", self.exp_name or "exp")
        self.ckpt_path = cfg.ckpt_path if cfg.ckpt_path else os.path.join(self.save_path, "params.pkl")
        self.start = 0
        if cfg.load_ckpt:
            self.load_ckpt(self.ckpt_path)
        self.alpha_image = cfg.alpha_image
        cfg.m_training_step = 0
        self.val_freq = 4096
        self.pipeline = cfg.pipeline_sampling is not False      # `pipeline_sampling = False` in the config restores the strictly sequential loop
        self._queue, self._sides, self._done_steps, self._fast, self._rays_event = {}, [], set(), None, None
        self.pipeline_depth = int(cfg.pipeline_depth or 2)          # batches marched ahead of the one being trained on
        self.done_period = int(cfg.pipeline_done_period or 4)       # the training stream records a `done` checkpoint every this many steps (train_step)
        n_sets = len(getattr(self.sampler, "_sets", ())) or 3
        assert n_sets >= self.pipeline_depth + self.done_period - 1, (
            f"pipeline_buffer_sets = {n_sets} is too small for pipeline_depth = {self.pipeline_depth} and pipeline_done_period = {self.done_period}")
        self._train_stream = None
        # rays per inference pass (the reference: n_rays_per_batch = 4096).  16384: a pass that asks for more than the sampler's fixed 4096*1024-sample capacity has to be
        # redone in 4096-ray passes, and 32768-ray passes did that on the rows through the object of an 800 x 800 view (27 ms per view instead of 18)
        self.render_chunk = int(cfg.render_chunk or 16384)
        self.W, self.H = self.dataset["train"].resolution

    def _sync_initial_parameters(self, params):
        """data parallel: every rank starts from rank 0's random initialisation (each rank seeds its own RNG for ray batches and backgrounds);
        identical summed gradients + the deterministic sweep then keep the replicas bit-identical"""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        with torch.no_grad():
            for p in params:
                dist.broadcast(p.data, src=0)
        if torch.cuda.is_available():
            # (r4) once, before the first kernel reads a parameter: host-staged backends (gloo) land the broadcast through copy streams of their own, and the training
            # loop runs on yet another stream (training_stream) - nothing is left to stream-ordering subtleties here
            torch.cuda.synchronize()
        for m in self.model.modules():
            if hasattr(m, "shadow_dirty"):
                m.shadow_dirty = True                       # fp16 shadows are rebuilt from the synchronised masters at the next read

    def training_stream(self):
        """Context manager: run the training loop on a stream of its own.  The default (null) stream pays ~10 us more per iteration at the step boundary on
        ROCm (tools/probe_boundary.py: 0.678 -> 0.667 ms per ngp_base.py iteration).  Exiting synchronises the stream, so results are visible to code on any
        stream afterwards.  `train_on_default_stream = True` in the config turns it into a no-op."""
        import contextlib
        if not torch.cuda.is_available() or self.cfg.train_on_default_stream:
            return contextlib.nullcontext()
        if self._train_stream is None:
            # (probe knob `train_stream_priority`: -1 = the higher of the device's two levels, so that freed wave slots / LDS go to the training stream's workgroups before
            # the sampling streams' - measured in round 2 (worse) and again in round 6)
            self._train_stream = torch.cuda.Stream(priority=int(self.cfg.train_stream_priority or 0))

        @contextlib.contextmanager
        def ctx():
            self._train_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._train_stream):
                try:
                    yield
                finally:
                    self._train_stream.synchronize()
        return ctx()

    def drain(self):
        """wait for a batch that was marched ahead on the side stream (call before dropping the Runner or touching its buffers from elsewhere)"""
        for st in self._sides:
            st.synchronize()
        self._queue.clear()
        if getattr(self, "_flags", None) is not None and int(self._flags[-1].item()) != 0:
            raise RuntimeError("a batch hand-over timed out on the GPU (ngp_flag_wait waited 2 s for a sampling stream): the results of this run are not valid")
        if hasattr(self.sampler, "finish_batch_rays_update"):
            self.sampler.finish_batch_rays_update()

    def finish(self):
        """end of a training run (Runner.train; collective under data parallelism): drain + the FINAL poll of the split kernels' range flag - synchronous, agreed on by all ranks"""
        self.drain()
        self._poll_field32_range(final=True)
        self._collective_polls = False                      # from here on (Runner.test on rank 0 alone) a poll after a rendered image acts locally again

    def __del__(self):
        try:
            self.drain()
        except Exception:
            pass

    # ---- one training iteration == the body of Runner.train (runner.py:64-76)
    def _make_batch(self, step, bufs=None):
        """ray generation, random background, target compositing (runner.py:65-68) and occupancy-grid sampling (runner.py:70) of ONE batch, on the
        current stream.  `bufs` = persistent ray buffers of the batch's buffer set (pipelined batches): nothing is allocated, so nothing has to be handed
        between the streams' allocator pools."""
        self.cfg.m_training_step = step
        ds = self.dataset["train"]
        if hasattr(self.sampler, "finish_batch_rays_update"):
            self.sampler.finish_batch_rays_update()       # the adaptive ray count decided after the previous 16-batch window (read back lazily)
        cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
        if cur is not None and self._rays_event is not None:
            cur.wait_event(self._rays_event)    # batches are generated on alternating streams: the dataset's permutation / cursor state is handed on in order
        if hasattr(ds, "next_fused"):       # same values as the three lines of runner.py:65-68, one kernel
            if bufs is not None:
                bg = torch.rand((ds.batch_size, 3), out=bufs["bg"][:ds.batch_size])
                img_ids, rays_o, rays_d, rgb_target = ds.next_fused(bg, out=(bufs["img"], bufs["o"], bufs["d"], bufs["target"]))
            else:
                bg = torch.rand((ds.batch_size, 3), device=ds.device)
                img_ids, rays_o, rays_d, rgb_target = ds.next_fused(bg)
        else:
            img_ids, rays_o, rays_d, rgb_target = next(ds)
            bg = torch.rand((rgb_target.shape[0], 3), device=rgb_target.device)
            rgb_target = rgb_target[..., :3] * rgb_target[..., 3:] + bg * (1 - rgb_target[..., 3:])
        if cur is not None:
            if self._rays_event is None:
                self._rays_event = torch.cuda.Event()
            self._rays_event.record(cur)        # (before the march: only the short ray-generation part is serialised between the streams)
        pos, dirs = self.sampler.sample(img_ids, rays_o, rays_d, is_training=True)
        return {"step": step, "bg": bg, "target": rgb_target, "pos": pos, "dirs": dirs, "state": self.sampler.export_batch_state(),
                "keep": (img_ids, rays_o, rays_d)}

    def _poll_field32_range(self, final=False, local=False):
        """(r4) The fp32 configuration's default field kernels work on split fp16 operands with fixed prescales (csrc/field_split.hip): features beyond ~255 or
        activations beyond ~4094 do not fit.  The kernels flag operands that come within a factor four of that (ngp_field32_range_check); polled where the host waits
        anyway - every 16th step, after every rendered image.  Near the limit: this process continues on the exact-product fp32-MFMA kernels (nothing has overflowed
        yet, results unchanged up to fp32 rounding).  Beyond it: the launches since the last poll produced infinities - an error, never a silent one."""
        m = self.model
        if not (torch.cuda.is_available() and getattr(m, "fused", False) and getattr(m, "fused_dtype", None) == torch.float32):
            return
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        exact = getattr(self, "_field32_exact", False)
        if exact and (local or not multi):  # (r6) with peers a collective poll is joined whatever this rank's own state: a rank that stopped polling would leave them in the all-reduce
            return
        from . import ops
        flag = 0 if exact else ops.field32_range_check(reset=True, synchronize=bool(final))
        if local and multi and getattr(self, "_collective_polls", False):
            # (r6) a rendered image INSIDE the training loop (val_img runs on every rank): the read above also consumed bits raised by training launches, and
            # acting on them here - switching kernels or raising on this rank alone - is the divergence _agree_on_range_flag exists to prevent.  The bits join the carry
            # and the next collective poll decides for everyone.  (Runner.test after training - rank 0 alone, no collective in reach - keeps the local behaviour.)
            self._range_flag_carry = getattr(self, "_range_flag_carry", 0) | int(flag)
            return
        if not local:
            flag = self._agree_on_range_flag(flag, final)
            if exact:
                return
        if flag & 2:
            raise RuntimeError("fp32 field network: an operand left the range of the split-operand kernels (|feature| > 255 or |activation| > 4094) - the results since the "
                               "last check contain infinities.  Re-run with NGP_FIELD32_FWD=mfma32 NGP_FIELD32_BWD=2 (exact-product kernels, no operand range).")
        if flag & 1:
            print("[jnerf_amd] fp32 field network: operands within 4x of the split-operand kernels' range; continuing on the exact-product fp32-MFMA kernels", flush=True)
            ops.field32_select(True)
            self._field32_exact = True

    def _agree_on_range_flag(self, flag, final=False):
        """Data parallel (r5): every rank polls its OWN device's flag, and a rank that raised (or switched kernels) alone would leave its peers blocked in the
        next collective.  The flags are MAX-all-reduced before anyone acts.  In the loop the reduction is asynchronous - issued at this poll, consumed at the next one, 16
        steps later (a blocking read-back here would drain the pipeline in every refresh window; bit 0 fires at a quarter of the range, so 16 steps of slack are safe, and an
        error is still an error 16 steps later) - and the final poll (drain) reduces synchronously, so no flag is ever dropped."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return flag
        on_gpu = dist.get_backend() == "nccl"
        carry = getattr(self, "_range_flag_carry", 0) | int(flag)           # bits this rank has seen and not yet acted on
        pend = getattr(self, "_range_flag_pending", None)
        agreed = 0
        if pend is not None:                                                 # the reduction issued at the previous poll
            work, host, ev = pend
            if work is not None:
                work.wait()
            if ev is not None:
                ev.synchronize()
            agreed = int(host.item())
            self._range_flag_pending = None
        if final:
            t = torch.tensor([carry], dtype=torch.int32, device=self.sampler.device if on_gpu else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)                         # (values 0..3: MAX keeps bit 1 whenever any rank holds it - the error wins)
            self._range_flag_carry = 0
            return agreed | int(t.item())
        if on_gpu:
            # like update_batch_rays (sampler.py): on a stream of its own, result copied to pinned memory, read at the next poll - the training stream is never waited for
            if getattr(self, "_flag_stream", None) is None:
                self._flag_stream = torch.cuda.Stream()
                self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            with torch.cuda.stream(self._flag_stream):
                t = torch.tensor([carry], dtype=torch.int32, device=self.sampler.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                self._flag_host.copy_(t, non_blocking=True)
                t.record_stream(self._flag_stream)
                ev = torch.cuda.Event(); ev.record(self._flag_stream)
            self._range_flag_pending = (None, self._flag_host, ev)
        else:
            t = torch.tensor([carry], dtype=torch.int32)
            self._range_flag_pending = (dist.all_reduce(t, op=dist.ReduceOp.MAX, async_op=True), t, None)
        self._range_flag_carry = 0                                           # handed to the reduction in flight
        return agreed

    def train_step(self, i):
        """Software-pipelined: the ray generation + marching of batches i+1 .. i+depth only read the dataset and the occupancy bitfield, so they are
        issued on side streams while batch i goes through the network / backward / optimiser.  The marcher is a dependent-latency kernel (its
        duration is the longest ray's serial chain, ~0.4 ms, while the chip is almost idle), as long as a whole training step: with `pipeline_depth`
        = 2 two marches overlap each other on two side streams and the step rate is set by the main stream alone.  Batches are never marched across
        an occupancy-grid refresh (every update_den_freq-th step needs the new weights, and the batches after it the refreshed grid)."""
        cfg = self.cfg
        main = torch.cuda.current_stream() if torch.cuda.is_available() else None
        b = self._queue.pop(i, None)
        if self._fast is None:
            from .fastpath import FusedTrainStep
            self._fast = FusedTrainStep(self) if FusedTrainStep.applicable(self) else False
        if b is not None:
            # hand-over of a batch marched on a sampling stream.  Native step: by device flag - the sampling stream ended the batch with ngp_flag_signal and the library's
            # call starts with ngp_flag_wait (csrc/train_step.hip).  Built on round 2's reading of the timeline (an event hand-over costing the training stream ~29 us per
            # iteration); measured in round 3 it changes nothing (gpurun_out/r3d_*: 1471 vs 1477 it/s) - the boundary gap is not the event - so it is OPT-IN
            # (`flag_handover = True` in the config); by default, and on the module path, the batch's event is waited for.
            if self._fast and self._fast.native and b.get("flag") is not None and cfg.flag_handover is True:
                pass
            else:
                b["flag"] = None
                main.wait_event(b["ready"])
            self.sampler.import_batch_state(b["state"])
        else:
            b = self._make_batch(i)                      # on the main stream: first step, refresh steps, pipeline off
            if self._sides and self.sampler.grid_updated_in_last_sample:
                self._grid_event.record(main)            # side streams must not read the bitfield before this refresh has finished
                self._grid_valid = True
        cfg.m_training_step = i
        self._collective_polls = True                    # until finish(): see _poll_field32_range(local=True)
        if i % 16 == 0 and i:
            self._poll_field32_range()
        if self._fast:
            loss = self._fast(b)                         # same kernels, same order, no autograd / nn.Module overhead (fastpath.py)
        else:
            network_outputs = self.model(b["pos"], b["dirs"])
            rgb = self.sampler.rays2rgb(network_outputs, b["bg"])
            loss = self.loss_func(rgb, b["target"])
            self.optimizer.step(loss)
            self.ema_optimizer.ema_step()
        if not (self.pipeline and main is not None):
            return loss
        # ---- batches i+1 .. i+depth on the side streams.  Issued AFTER step i's own launches: every 16th prefetch ends in update_batch_rays' host
        # read-back, and the main stream should have step i queued while the host waits for it.
        n_sets, P = len(self.sampler._sets), self.done_period
        if not self._sides:
            # (probe knobs, tools/gpu_r3_g.sh: HIP multiplexes streams onto a few hardware queues in creation order, and which streams end up together moves the
            # iteration time by +-4 % - `pipeline_dummy_streams` unused streams created first shift that mapping, `pipeline_side_streams` = 1 marches on one stream)
            self._dummy_streams = [torch.cuda.Stream() for _ in range(int(cfg.pipeline_dummy_streams or 0))]
            # (r6) ONE sampling stream by default: rounds 2-5 marched on two (two marchers may then run at once, and beside any kernel of the step).  With the step at ~0.52 ms a
            # batch's ray generation + marching (~0.2 ms) still fits the step on one stream, consecutive marchers never overlap each other, and the training stream's kernels
            # keep more of the chip: +5.1 % it/s in three A/B pairs of one call (profiles/r06f_ab_lines.txt: 1929 / 1929 / 1938 vs 1834 / 1837 / 1841).
            n_side = int(cfg.pipeline_side_streams or os.environ.get("NGP_PIPELINE_SIDE_STREAMS") or 1)
            self._sides = [torch.cuda.Stream(priority=int(cfg.pipeline_side_priority or 0)) for _ in range(n_side)]
            self._ready = [torch.cuda.Event() for _ in range(n_sets)]            # persistent events, re-recorded (no create/destroy per step)
            self._done = [torch.cuda.Event() for _ in range(n_sets // P + 2)]
            self._grid_event, self._grid_valid = torch.cuda.Event(), False
            self._flags = torch.zeros(n_sets + 1, dtype=torch.int32, device=self.sampler.device)      # one hand-over flag per buffer set + the wait kernels' status word
            # Persistent ray buffers, one per buffer set, sized for the largest ray count update_batch_rays can choose.  Round 2 / early round 3 allocated bg / target /
            # rays per batch on the sampling stream and `record_stream`ed them for the training stream: when the batch was released, the caching allocator put one
            # event record PER TENSOR into the training stream's queue - five marker packets between the sweep and the next step's first kernel, ~6 us each: that was
            # the "unexplained" 30 us at the step boundary (HIP API trace, gpurun_out/r3l_api_timeline.txt: 10 hipEventRecord + 7 hipEventQuery per step).  The sets are
            # recycled under the same `done` checkpoints as the sampler's.
            cap, dev = int(self.sampler.target_batch_size), self.sampler.device
            ds = self.dataset["train"]
            self._ray_bufs = None
            if hasattr(ds, "next_fused") and cfg.pipeline_persistent_rays is not False and os.environ.get("NGP_PIPELINE_ALLOC_RAYS") != "1":
                self._ray_bufs = [dict(bg=torch.empty((cap, 3), device=dev), target=torch.empty((cap, 3), device=dev), o=torch.empty((cap, 3), device=dev),
                                       d=torch.empty((cap, 3), device=dev), img=torch.empty(cap, dtype=torch.int32, device=dev)) for _ in range(n_sets)]
            if self.sampler.grid_updated_in_last_sample:
                self._grid_event.record(main); self._grid_valid = True
        # `done` checkpoint of the training stream, every P-th step only: an event record is a marker packet in the stream's queue and costs ~15 us of dead time
        # on MI355X (rocprof timeline: 30 us between the sweep and the next step's first kernel with one record + one wait per step)
        if i % P == P - 1:
            self._done[(i // P) % len(self._done)].record(main)
            self._done_steps.add(i)
            self._done_steps = {c for c in self._done_steps if c > i - n_sets - 2 * P}
        cur_state = None
        for k in range(i + 1, i + 1 + self.pipeline_depth):
            if k >= self.tot_train_steps or k % self.sampler.update_den_freq == 0:
                break                                    # never across a refresh
            if k in self._queue:
                continue
            side = self._sides[k % len(self._sides)]
            if k - n_sets >= 0:                          # the buffer set batch k writes was last read by step k - n_sets: wait for the first checkpoint at or after it
                c = (k - n_sets) // P * P + P - 1            # (<= i: n_sets >= pipeline_depth + P - 1, checked in __init__)
                if c in self._done_steps:
                    side.wait_event(self._done[(c // P) % len(self._done)])
            if self._grid_valid:
                side.wait_event(self._grid_event)
            if cur_state is None:
                cur_state = self.sampler.export_batch_state()
            with torch.cuda.stream(side):
                nb = self._make_batch(k, self._ray_bufs[k % n_sets] if self._ray_bufs is not None else None)
                if cfg.flag_handover is True:
                    from . import ops as _ops
                    _ops.flag_signal(self._flags[k % n_sets:k % n_sets + 1], k + 1)     # behind the batch's kernels on this stream: the training stream's ngp_flag_wait
                    nb["flag"] = (self._flags[k % n_sets:k % n_sets + 1], k + 1, self._flags[n_sets:])
                nb["ready"] = self._ready[k % n_sets]
                nb["ready"].record(side)
            if self._ray_bufs is None:
                for t in (nb["bg"], nb["target"]) + tuple(nb["keep"]):
                    if torch.is_tensor(t):
                        t.record_stream(main)
            self._queue[k] = nb
        if cur_state is not None:
            self.sampler.import_batch_state(cur_state)
            cfg.m_training_step = i
        return loss

    def train(self):
        os.makedirs(self.save_path, exist_ok=True)
        loss = None
        with self.training_stream():
            for i in range(self.start, self.tot_train_steps):
                loss = self.train_step(i)
                if i > 0 and i % self.val_freq == 0:
                    psnr = mse2psnr(self.val_img(i))
                    print("STEP={} | LOSS={} | VAL PSNR={}".format(i, loss.mean().item(), psnr))
            self.finish()
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        from .optim import sync_all_sharded
        sync_all_sharded()                               # (collective: every rank) state that lives on its owner's shard under the sharded sweep
        if not multi or dist.get_rank() == 0:           # replicas are bit-identical: one writer (every rank writing the same files would race)
            self.save_ckpt(os.path.join(self.save_path, "params.pkl"))
        if multi:
            dist.barrier()                               # only the checkpoint write sits inside the collective window ...
        if not multi or dist.get_rank() == 0:
            self.test()                                  # ... the full test render (minutes with OriginNeRFNetworks) runs outside it: no rank waits in a barrier that can time out

    def test(self, load_ckpt=False):
        if load_ckpt:
            assert os.path.exists(self.ckpt_path), "ckpt file does not exist: " + self.ckpt_path
            self.load_ckpt(self.ckpt_path)
        if self.dataset["test"] is None:
            self.dataset["test"] = build_from_cfg(self.cfg.dataset.test, DATASETS)
        os.makedirs(os.path.join(self.save_path, "test"), exist_ok=True)
        mse_list = self.render_test(save_path=os.path.join(self.save_path, "test"))
        if self.dataset["test"].have_img:
            tot = sum(mse2psnr(m) for m in mse_list)
            print("TOTAL TEST PSNR===={}".format(tot / len(mse_list)))
            return tot / len(mse_list)

    def render(self, load_ckpt=True, save_path=None, nframe=80):
        """runner.py:101-121: the demo video along camera_path.path_spherical().  The reference writes demo.mp4 through cv2; without cv2 (this image) the frames are
        written as a PNG sequence into `<save_path minus .mp4>_frames/` instead.  Returns the path written."""
        if load_ckpt:
            assert os.path.exists(self.ckpt_path), "ckpt file does not exist: " + self.ckpt_path
            self.load_ckpt(self.ckpt_path)
        if save_path is None or save_path == "":
            save_path = os.path.join(self.save_path, "demo.mp4")
        else:
            assert save_path.endswith(".mp4"), "suffix of save_path need to be .mp4"
        from .camera_path import path_spherical
        print("rendering video with specified camera path")
        os.makedirs(os.path.dirname(save_path) or ".", exist_ok=True)
        try:
            import cv2
        except ImportError:
            cv2 = None
        poses = path_spherical(nframe=nframe)
        if cv2 is not None:
            W, H = int(self.W), int(self.H)
            writer = cv2.VideoWriter(save_path, cv2.VideoWriter_fourcc(*"mp4v"), 28, (W, H))
            for pose in poses:
                img = (self.render_img_with_pose(pose) * 255 + 0.5).clip(0, 255).astype("uint8")
                writer.write(cv2.cvtColor(img, cv2.COLOR_BGR2RGB))
            writer.release()
            return save_path
        out = save_path[:-4] + "_frames"
        os.makedirs(out, exist_ok=True)
        print("cv2 is not installed: writing the frames to " + out, file=sys.stderr)
        for k, pose in enumerate(poses):
            self.save_img(os.path.join(out, f"{k:04d}.png"), self.render_img_with_pose(pose))
        return out

    def save_ckpt(self, path):
        from .optim import flush_all
        flush_all()
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            from .optim import sync_all_sharded
            sync_all_sharded()      # (multi-rank runs: train() has done it on every rank - it is a collective, and save_ckpt runs on rank 0 only)
        elif getattr(self.optimizer._nested_optimizer, "_sharded_dirty", False):
            #  Adam.state_dict() would all-gather the sharded moments - a collective - from this one rank and hang.  Every rank has to sync first.
            raise RuntimeError("save_ckpt on a multi-rank run while the sharded optimiser state is stale: call optim.sync_all_sharded() on EVERY rank first (Runner.train does), "
                               "then save on rank 0")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        ck = {"global_step": self.cfg.m_training_step, "model": self.model.state_dict(), "sampler": self.sampler.state_dict(),
              "optimizer": self.optimizer.state_dict(), "nested_optimizer": self.optimizer._nested_optimizer.state_dict(),
              "ema_optimizer": self.ema_optimizer.state_dict(),
              "extra": {"rng_state": self.sampler.rng_state.copy(), "n_rays_per_batch": self.sampler.n_rays_per_batch}}
        if self.cfg.ckpt_format == "jittor":     # the reference's container: EXACTLY the six keys of runner.py:124-131 as numpy arrays (jt.load reads it); our "extra" entry (generator
            from .utils import jittor_pickle       # state, adaptive ray count) has no counterpart there - a run resumed from this container restarts those two like the reference does
            ck.pop("extra")
            jittor_pickle.dump(ck, path)
        else:
            torch.save(ck, path)

    def load_ckpt(self, path):
        print("Loading ckpt from:", path)
        from .utils import jittor_pickle
        with open(path, "rb") as f:
            head = f.read()
        # the container is chosen by what the file IS, not by the config: jt.save's trailer | a zip (torch.save) | with `ckpt_format = 'jittor'` a bare pickle of numpy arrays
        is_zip = head[:4] == b"PK\x03\x04"
        if head.endswith(jittor_pickle.MAGIC) or (self.cfg.ckpt_format == "jittor" and not is_zip):
            # the reference's own container (jt.save: a pickle of numpy arrays + sha1 + magic, utils/jittor_pickle.py) - read without Jittor
            ckpt = jittor_pickle.to_torch(jittor_pickle.loads(head, path))
        else:
            try:
                ckpt = torch.load(path, map_location="cpu", weights_only=False)
            except Exception as e:
                raise RuntimeError(f"{path} is neither a torch.save container nor a jt.save file (no HCAJSLHD trailer; set `ckpt_format = 'jittor'` in the config to "
                                   f"read a bare pickle of numpy arrays).  Original error: {e!r}") from e
        if not (isinstance(ckpt, dict) and "model" in ckpt and "global_step" in ckpt):
            raise RuntimeError(f"{path}: unexpected checkpoint layout (keys {list(ckpt)[:8] if isinstance(ckpt, dict) else type(ckpt)}); expected the keys of runner/runner.py:123-135")
        self.start = ckpt["global_step"]
        ref_file = "extra" not in ckpt          # written by the reference's Runner: its module tree may hold buffers ours does not register (and vice versa)
        for name, mod in (("model", self.model), ("sampler", self.sampler)):
            res = mod.load_state_dict(ckpt[name], strict=not ref_file)
            missing, unexpected = (list(getattr(res, "missing_keys", ())), list(getattr(res, "unexpected_keys", ()))) if res is not None else ([], [])
            if missing or unexpected:       # only possible with strict=False (a file written by the reference's Runner): said, not swallowed
                print(f"[jnerf_amd] load_ckpt {name}: keys the file lacks {missing[:8]}{' ...' if len(missing) > 8 else ''}; keys it has that this build does not {unexpected[:8]}{' ...' if len(unexpected) > 8 else ''}")
        self.optimizer.load_state_dict(ckpt["optimizer"])
        self.optimizer._nested_optimizer.load_state_dict(ckpt["nested_optimizer"])
        self.ema_optimizer.load_state_dict(ckpt["ema_optimizer"])
        if "extra" in ckpt:     # the reference does not persist these; we do, so resume is exact
            self.sampler.rng_state[:] = ckpt["extra"]["rng_state"]
            self.sampler.n_rays_per_batch = ckpt["extra"]["n_rays_per_batch"]
            self.dataset["train"].batch_size = self.sampler.n_rays_per_batch

    def val_img(self, it):
        import torch.distributed as dist
        writer = not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
        with torch.no_grad():
            img, _, img_tar = self.render_img(dataset_mode="val")
            if writer:                                   # every rank renders (replicas are identical), one writes the files
                self.save_img(self.save_path + f"/img{it}.png", img)
                self.save_img(self.save_path + f"/target{it}.png", img_tar)
            return float(np.mean((img - img_tar) ** 2))

    def render_test(self, save_img=True, save_path=None):
        save_path = save_path or self.save_path
        mse_list = []
        for img_i in range(self.dataset["test"].n_images):
            with torch.no_grad():
                img, alpha, img_tar = self.render_img(dataset_mode="test", img_id=img_i)
                if save_img:
                    self.save_img(save_path + f"/{self.exp_name}_r_{img_i}.png", img, alpha if self.alpha_image else None)
                mse_list.append(float(np.mean((img - img_tar) ** 2)))
        return mse_list

    def save_img(self, path, img, alpha=None):
        from PIL import Image
        if alpha is not None:
            img = np.concatenate([img, alpha], axis=-1)
        Image.fromarray((img * 255 + 0.5).clip(0, 255).astype("uint8")).save(path)

    @torch.no_grad()
    def _render_rays(self, img_ids, rays_o_total, rays_d_total, chunk):
        """sampler.sample -> model -> rays2rgb(inference) over ray chunks (runner.py:211-226), results assembled on the device.
        `chunk` rays per pass; the reference uses n_rays_per_batch = 4096 and two .numpy() read-backs per chunk."""
        n = rays_o_total.shape[0]
        dev = rays_o_total.device
        imgs = torch.empty((n, 3), device=dev)
        alphas = torch.empty((n, 1), device=dev)
        counts = torch.zeros(2, dtype=torch.int64, device=dev)      # [samples rendered, chunks whose requested samples exceeded the capacity]
        if self._native_render_ok():
            self._render_rays_native(rays_o_total, rays_d_total, chunk, imgs, alphas, counts)
            host = counts.tolist()
            self._poll_field32_range(local=True)
            self.n_samples_rendered = int(host[0])
            if host[1] and chunk > self.sampler.max_samples // self.sampler.MAX_STEP:
                return self._render_rays(img_ids, rays_o_total, rays_d_total, self.sampler.max_samples // self.sampler.MAX_STEP)
            return imgs, alphas
        self.sampler.sync_free_inference = True
        try:
            for pixel in range(0, n, chunk):
                rays_o, rays_d = rays_o_total[pixel:pixel + chunk], rays_d_total[pixel:pixel + chunk]
                pos, dirs = self.sampler.sample(img_ids, rays_o, rays_d)
                cnt = self.sampler._inference_counter
                counts[0] += cnt[3]
                counts[1] += (cnt[1].to(torch.int64) & 0xffffffff) > self.sampler.max_samples
                network_outputs = self.model(pos, dirs)
                rgb, alpha = self.sampler.rays2rgb(network_outputs, inference=True)
                imgs[pixel:pixel + chunk] = rgb
                alphas[pixel:pixel + chunk] = alpha
        finally:
            self.sampler.sync_free_inference = False
        host = counts.tolist()                              # ONE read-back per image: samples rendered + overflow flag
        self.n_samples_rendered = int(host[0])
        if host[1] and chunk > self.sampler.max_samples // self.sampler.MAX_STEP:
            # a chunk asked for more than the sampler's fixed 4096*1024-sample capacity (ray_sampler.py:15): its trailing rays came back empty.  Re-render with
            # the reference's chunk size (capacity / MAX_STEP = 4096 rays can never overflow) instead of returning black pixels.
            return self._render_rays(img_ids, rays_o_total, rays_d_total, self.sampler.max_samples // self.sampler.MAX_STEP)
        return imgs, alphas

    def _native_render_ok(self):
        """the standard stack on the GPU (hash encoder + SH + fused field network in either precision + occupancy-grid sampler): one ngp_render_chunk call per chunk"""
        m, s = self.model, self.sampler
        return bool(torch.cuda.is_available() and getattr(m, "fused", False) and hasattr(s, "density_grid_bitfield") and s.density_grid_bitfield.is_cuda
                    and self.cfg.native_render is not False)

    def _render_rays_native(self, rays_o_total, rays_d_total, chunk, imgs, alphas, counts):
        """the chunk loop of _render_rays through ngp_render_chunk (csrc/train_step.hip): the same kernels in the same order as sampler.sample -> model ->
        rays2rgb(inference), but one FFI crossing per chunk and no tensor plumbing in between (the Python loop took 34 ms per 800 x 800 view for 17 ms of kernels)"""
        import ctypes as C
        from . import _lib as L, ops
        from .optim import flush_all
        flush_all()
        m, s, enc = self.model, self.sampler, self.model.pos_encoder
        dev = rays_o_total.device
        cap = int(s.max_samples)
        # several buffer sets on as many streams: chunk i+1 is marched (a latency-bound traversal) while chunk i is in the gather / MLP kernels
        n_sets = max(1, min(4, int(self.cfg.render_streams or 3)))           # (800 x 800 lego-like view: 1 stream 1222, 2: 1397, 3: 1488, 4: 1504 Msamples/s)
        bufs = getattr(self, "_render_bufs", None)
        if bufs is None or len(bufs) != n_sets or bufs[0]["chunk"] < chunk or bufs[0]["dtype"] != m.fused_dtype:
            bufs = self._render_bufs = [dict(chunk=chunk, dtype=m.fused_dtype, coords=s._inference_coords() if k == 0 else torch.empty((cap, 7), dtype=torch.float32, device=dev),
                                             pos=torch.empty((cap, 3), dtype=torch.float32, device=dev),
                                             numsteps=torch.empty((chunk, 2), dtype=torch.int32, device=dev), numsteps_c=torch.empty((chunk, 2), dtype=torch.int32, device=dev),
                                             counters=torch.zeros(4, dtype=torch.int32, device=dev), scratch=torch.empty(ops.march_scratch_elems(chunk), dtype=torch.int32, device=dev),
                                             feat=torch.empty((16, cap, 2), dtype=m.fused_dtype, device=dev), out=torch.empty((cap, 4), dtype=m.fused_dtype, device=dev),
                                             stream=torch.cuda.Stream() if n_sets > 1 else None) for k in range(n_sets)]
        packed = m.packed_weights(refresh=True)
        table = enc.table_for_kernels()
        rays_o_total, rays_d_total = rays_o_total.contiguous(), rays_d_total.contiguous()
        args = []
        for b in bufs:
            a = L.NgpRenderChunk()
            a.cap, a.max_samples, a.const_dt, a.cascades = cap, cap, int(bool(s.const_dt)), int(s.NERF_CASCADES)
            a.dtype = L.F16 if m.fused_dtype == torch.float16 else L.F32
            a.aabb0, a.aabb1, a.near_distance, a.cone_angle = float(s.aabb_range[0]), float(s.aabb_range[1]), float(s.near_distance), float(s.cone_angle_constant)
            a.bitfield, a.rng_state_host = s.density_grid_bitfield.data_ptr(), s.rng_state.ctypes.data
            a.coords, a.pos, a.numsteps, a.numsteps_compacted = b["coords"].data_ptr(), b["pos"].data_ptr(), b["numsteps"].data_ptr(), b["numsteps_c"].data_ptr()
            a.counters, a.scratch = b["counters"].data_ptr(), b["scratch"].data_ptr()
            a.table, a.level_table_host, a.packed_weights = table.data_ptr(), enc.level_table.ctypes.data, packed.data_ptr()
            a.feat, a.out, a.totals = b["feat"].data_ptr(), b["out"].data_ptr(), counts.data_ptr()
            ob = s.occupancy_bounds() if hasattr(s, "occupancy_bounds") else None
            a.occ_bounds = ob.data_ptr() if ob is not None else None
            args.append(a)
        n = rays_o_total.shape[0]
        main = torch.cuda.current_stream()
        for b in bufs:
            if b["stream"] is not None:
                b["stream"].wait_stream(main)               # rays, weights, image buffers are ready on the caller's stream
        for j, pixel in enumerate(range(0, n, chunk)):
            a, b = args[j % n_sets], bufs[j % n_sets]
            k = min(chunk, n - pixel)
            a.n_rays = k
            a.rays_o, a.rays_d = rays_o_total.data_ptr() + pixel * 12, rays_d_total.data_ptr() + pixel * 12
            a.rgb_out, a.alpha_out = imgs.data_ptr() + pixel * 12, alphas.data_ptr() + pixel * 4
            stream = C.c_void_p(b["stream"].cuda_stream) if b["stream"] is not None else ops._stream()
            L.check(L.lib().ngp_render_chunk(stream, C.byref(a)), "ngp_render_chunk")
        for b in bufs:
            if b["stream"] is not None:
                main.wait_stream(b["stream"])
        for t in (rays_o_total, rays_d_total, imgs, alphas, counts):
            for b in bufs:
                if b["stream"] is not None:
                    t.record_stream(b["stream"])

    @torch.no_grad()
    def render_img(self, dataset_mode="train", img_id=None):
        """runner.py:197-236"""
        ds = self.dataset[dataset_mode]
        W, H = int(self.W), int(self.H)
        if img_id is None:
            img_id = int(np.random.randint(0, ds.n_images))
        img_ids = torch.full((H * W,), img_id, dtype=torch.int32, device=ds.device)
        rays_o_total, rays_d_total, _ = ds.generate_rays_total_test(img_ids, W, H)
        imgs, alphas = self._render_rays(img_ids, rays_o_total, rays_d_total, self.render_chunk)
        imgs, alphas = imgs.view(H, W, 3), alphas.view(H, W, 1)
        tar = ds.image_data[img_id].view(H, W, 4)
        bgc = torch.tensor(self.background_color, dtype=torch.float32, device=ds.device)
        tar = tar[..., :3] * tar[..., 3:] + bgc * (1 - tar[..., 3:])
        if not self.alpha_image:
            imgs = imgs + bgc * (1 - alphas)
            return imgs.cpu().numpy(), None, tar.cpu().numpy()
        return imgs.cpu().numpy(), alphas.cpu().numpy(), tar.cpu().numpy()

    @torch.no_grad()
    def render_img_with_pose(self, pose):
        """runner.py:238-264"""
        ds = self.dataset["train"]
        W, H = int(self.W), int(self.H)
        fake_ids = torch.zeros((H * W,), dtype=torch.int32, device=ds.device)
        rays_o_total, rays_d_total = ds.generate_rays_with_pose(pose, W, H)
        img, alpha = self._render_rays(fake_ids, rays_o_total, rays_d_total, self.render_chunk)
        img, alpha = img.view(H, W, 3), alpha.view(H, W, 1)
        if not self.alpha_image:
            img = img + torch.tensor(self.background_color, dtype=torch.float32, device=ds.device) * (1 - alpha)
        return img.cpu().numpy()
