"""DensityGridSampler behind the reference's SAMPLERS registry
(python/jnerf/models/samplers/density_grid_sampler/density_grid_sampler.py:18-271).

Same constructor arguments, attributes (n_rays_per_batch, density_grid, density_grid_bitfield, density_grid_mean, _coords,
_rays_numsteps[_compacted]) and methods (sample, rays2rgb, update_density_grid, update_batch_rays).  Differences, all MI355X-motivated:
  * training sampling is ONE pass (march + compaction, deterministic ray-ordered slots); the reference's dead first forward
    (density_grid_sampler.py:153-159, SURVEY.md App.B-1) is dropped — its result never influenced the output;
  * counters stay on the device: the only host read-back on the training loop is the scalar every 16th step (update_batch_rays);
  * rows >= counters[3] of the compacted buffer are skipped by every consumer instead of being zero-filled and pushed through the net."""
from math import ceil, log2
import numpy as np
import torch
from torch import nn
from . import ops
from .utils.config import get_cfg
from .utils.registry import SAMPLERS


class _Composite(torch.autograd.Function):
    """CalcRgb.execute / .grad (calc_rgb.py:31-108)"""

    @staticmethod
    def forward(ctx, net_out, bg, s):
        rgb = ops.composite_fwd(net_out, s._coords, s._rays_numsteps, s._rays_numsteps_compacted, bg, s.NERF_CASCADES)
        ctx.s = s
        ctx.save_for_backward(net_out, rgb)
        ctx.coords, ctx.nsc = s._coords, s._rays_numsteps_compacted
        return rgb

    @staticmethod
    def backward(ctx, grad_rgb):
        net_out, rgb = ctx.saved_tensors
        s = ctx.s
        dout = ops.composite_bwd(net_out, ctx.coords, ctx.nsc, grad_rgb.contiguous(), rgb, s.density_grid_mean, s.NERF_CASCADES, dout=s._dout_buffer(net_out),
                                 zero_first=s._n_valid is None or not getattr(s.model, "fused", False))
        # rows < n_valid are all written by their rays.  The fused field network never reads the rows beyond (device-side n_valid); the generic
        # nn.Linear path back-propagates every row of the fixed-capacity buffer, so there the stale tail must be zero (the reference's buffer is zero-padded)
        return dout, None, None


@SAMPLERS.register_module()
class DensityGridSampler(nn.Module):
    def __init__(self, update_den_freq=16, update_block_size=5000000):
        super().__init__()
        self.cfg = get_cfg()
        self.model = self.cfg.model_obj
        self.dataset = self.cfg.dataset_obj
        self.update_den_freq = update_den_freq
        self.update_block_size = update_block_size
        self.n_rays_per_batch = self.cfg.n_rays_per_batch
        self.cone_angle_constant = self.cfg.cone_angle_constant
        self.using_fp16 = bool(self.cfg.fp16)
        self.near_distance = self.cfg.near_distance
        self.n_training_steps = self.cfg.n_training_steps
        self.target_batch_size = self.cfg.target_batch_size
        self.const_dt = bool(self.cfg.const_dt)
        self.NERF_CASCADES = 5
        self.NERF_GRIDSIZE = 128
        self.NERF_MIN_OPTICAL_THICKNESS = 0.01
        self.MAX_STEP = 1024
        self.background_color = self.cfg.background_color
        self.aabb_range = tuple(float(v) for v in self.dataset.aabb_range)
        max_aabb_scale = 1 << (self.NERF_CASCADES - 1)
        if self.dataset.aabb_scale > max_aabb_scale:
            self.NERF_CASCADES = ceil(log2(self.dataset.aabb_scale)) + 1
        self.max_cascade = 0
        while (1 << self.max_cascade) < self.dataset.aabb_scale:
            self.max_cascade += 1
        dev = self.cfg.device or "cuda"
        self.device = torch.device(dev)
        G3 = self.NERF_GRIDSIZE ** 3
        self.density_grid_decay = 0.95
        self.density_n_elements = self.NERF_CASCADES * G3
        self.register_buffer("density_grid", torch.zeros(self.density_n_elements, dtype=torch.float32, device=dev))
        self.density_grid_tmp = torch.zeros(self.density_n_elements, dtype=torch.float32, device=dev)
        self.register_buffer("density_grid_bitfield", torch.zeros(self.density_n_elements // 8, dtype=torch.uint8, device=dev))
        self.register_buffer("density_grid_mean", torch.zeros(1, dtype=torch.float32, device=dev))
        self.register_buffer("density_grid_ema_step", torch.zeros(1, dtype=torch.int32, device=dev))
        # (ours) bounding boxes of the occupied cells, refreshed with the bitfield: the marcher drops rays that cannot meet an occupied cell and stops behind the last box -
        # identical samples (`march_occupancy_bounds = False` in the config turns it off)
        self._occ_bounds = torch.zeros(ops.OCC_BOUNDS_INTS, dtype=torch.int32, device=dev) if torch.device(dev).type == "cuda" else None
        self._occ_bounds_valid = False
        self.max_samples = 4096 * self.MAX_STEP                        # ray_sampler.py:15 — fixed even after the ray count grows
        # the reference's global pcg32{1337} (ops/code_ops/global_vars.py:13-16); multi-GPU ranks take disjoint sub-streams
        from .rng import pcg32_seed
        self.rng_state = pcg32_seed(1337)
        rank = int(self.cfg.rank or 0)
        if rank:
            from .rng import pcg32_advance
            pcg32_advance(self.rng_state, rank << 40)
        self.measured_batch_size = torch.zeros(1, dtype=torch.int32, device=dev)
        cap_r = 1 << 18
        # buffer sets: the Runner marches up to two batches ahead on side streams while batch i is still being trained on (software pipelining), and lets a set be
        # rewritten only after a `done` checkpoint of the training stream, recorded every 4th step (runner.py) - hence 8 sets (12 MB each)
        n_sets = int(self.cfg.pipeline_buffer_sets or 8)
        self._sets = [dict(numsteps=torch.empty((cap_r, 2), dtype=torch.int32, device=dev), numsteps_c=torch.empty((cap_r, 2), dtype=torch.int32, device=dev),
                           counters=torch.zeros(4, dtype=torch.int32, device=dev), coords=torch.zeros((self.target_batch_size, 7), dtype=torch.float32, device=dev),
                           pos=torch.zeros((self.target_batch_size, 3), dtype=torch.float32, device=dev)) for _ in range(n_sets)]
        self._march_scratch = {}                            # count-pass output + t-cache of one march call: one per STREAM (main / side 0 / side 1): calls on one stream are ordered, and
                                                            # a main-stream march (first step, refresh steps, resume) no longer shares a buffer with the side stream of its parity
        self._set_idx = 0
        self._numsteps_buf, self._numsteps_c_buf = self._sets[0]["numsteps"], self._sets[0]["numsteps_c"]
        self._counters = self._sets[0]["counters"]
        self._scratch = None
        self._coords_train = self._sets[0]["coords"]
        self._dout = None
        self._coords = None
        self._n_valid = None
        self._order_event = None
        self._pending_rays_update, self._measured_host = None, None
        self.grid_updated_in_last_sample = False
        self.sync_free_inference = False        # Runner.render_img switches it on (large ray chunks, no .item() per chunk)

    # ------------------------------------------------------------------ hot path
    def n_valid_for(self, pos):
        """device-side sample count when `pos` is a view of one of this sampler's fixed-capacity buffers, else None"""
        if self._n_valid is not None and self._coords is not None and pos.data_ptr() == self._coords.data_ptr():
            return self._n_valid
        return None

    def _dout_buffer(self, like):
        if self._dout is None or self._dout.shape != like.shape or self._dout.dtype != like.dtype:
            self._dout = torch.empty_like(like)
        return self._dout

    def sample(self, img_ids, rays_o, rays_d, rgb_target=None, is_training=False):
        self.grid_updated_in_last_sample = False
        if is_training and self.cfg.m_training_step % self.update_den_freq == 0:
            self.update_density_grid()
            self.grid_updated_in_last_sample = True
        rays_o, rays_d = rays_o.contiguous(), rays_d.contiguous()
        n = rays_o.shape[0]
        if not is_training:
            if self.sync_free_inference:
                # (ours) no host read-back: fixed-capacity buffers + device-side sample count, same one-pass marcher as training with cap = capacity
                need = ops.march_scratch_elems(n)
                if self._scratch is None or self._scratch.numel() < need:
                    self._scratch = torch.empty(need, dtype=torch.int32, device=self.device)
                coords = self._inference_coords()
                if getattr(self, "_inf_bufs", None) is None:
                    self._inf_bufs = (torch.empty((1 << 18, 2), dtype=torch.int32, device=self.device), torch.empty((1 << 18, 2), dtype=torch.int32, device=self.device),
                                      torch.zeros(4, dtype=torch.int32, device=self.device))
                numsteps, numsteps_c = self._inf_bufs[0][:n], self._inf_bufs[1][:n]
                self._counters = self._inf_bufs[2]
                ops.march_rays_compacted(rays_o, rays_d, self.density_grid_bitfield, self.aabb_range, self.rng_state, self.max_samples, self.max_samples,
                                         self.cone_angle_constant, self.near_distance, self.const_dt, self.NERF_CASCADES,
                                         coords_out=coords, numsteps=numsteps, numsteps_c=numsteps_c, counters=self._counters, scratch=self._scratch, occ_bounds=self.occupancy_bounds())
                self._coords = coords
                self._rays_numsteps = numsteps_c
                self._n_valid = self._counters[3:4]
                self._inference_counter = self._counters
                return coords[:, :3], coords[:, 4:]
            coords, numsteps, counters, _ = ops.march_rays(rays_o, rays_d, self.density_grid_bitfield, self.aabb_range, self.rng_state, self.max_samples,
                                                           self.cone_angle_constant, self.near_distance, self.const_dt, self.NERF_CASCADES,
                                                           coords=self._inference_coords(), zero_coords=False)
            samples = int(counters[1].item())                          # ray_sampler.py:70 (inference only)
            samples = min(samples, self.max_samples)
            self._coords = coords[:samples]
            self._rays_numsteps = numsteps
            self._n_valid = None
            return self._coords[:, :3], self._coords[:, 4:]
        self._set_idx = (self._set_idx + 1) % len(self._sets)
        bs = self._sets[self._set_idx]
        self._coords_train, self._counters = bs["coords"], bs["counters"]
        numsteps, numsteps_c = bs["numsteps"][:n], bs["numsteps_c"][:n]
        need = ops.march_scratch_elems(n)
        key = torch.cuda.current_stream().cuda_stream if self.device.type == "cuda" else 0
        if key not in self._march_scratch or self._march_scratch[key].numel() < need:
            # sized once for the largest ray count update_batch_rays can choose (<= target_batch_size): no re-allocation while the streams use it
            self._march_scratch[key] = torch.empty(max(need, ops.march_scratch_elems(min(self.target_batch_size, 1 << 18))), dtype=torch.int32, device=self.device)
        scratch = self._march_scratch[key]
        ops.march_rays_compacted(rays_o, rays_d, self.density_grid_bitfield, self.aabb_range, self.rng_state, self.max_samples, self.target_batch_size,
                                 self.cone_angle_constant, self.near_distance, self.const_dt, self.NERF_CASCADES,
                                 coords_out=self._coords_train, numsteps=numsteps, numsteps_c=numsteps_c, counters=self._counters, scratch=scratch, pos_out=bs["pos"],
                                 occ_bounds=self.occupancy_bounds())
        self._pos_train = bs["pos"]                                    # compact [n,3] copy of coords[:, :3], written by the marcher's write pass
        # two side streams may be marching at once: their read-modify-writes of the running sample count are ordered by an event chain
        # (the wait sits AFTER this batch's march kernels in stream order, so the marches themselves still overlap)
        on_gpu = self.measured_batch_size.is_cuda
        if on_gpu and self._order_event is not None:
            torch.cuda.current_stream().wait_event(self._order_event)
        self.measured_batch_size += self._counters[2:3]                # density_grid_sampler.py:155
        if self.cfg.m_training_step % self.update_den_freq == (self.update_den_freq - 1):
            self.update_batch_rays()
        if on_gpu:
            if self._order_event is None:
                self._order_event = torch.cuda.Event()
            self._order_event.record()
        self._coords = self._coords_train
        self._rays_numsteps, self._rays_numsteps_compacted = numsteps, numsteps_c
        self._n_valid = self._counters[3:4]
        return self._coords[:, :3], self._coords[:, 4:]

    # ---- batch state hand-over for the pipelined training loop (Runner): everything rays2rgb / the network need about ONE sampled batch
    def export_batch_state(self):
        return (self._coords, self._rays_numsteps, getattr(self, "_rays_numsteps_compacted", None), self._n_valid, self._coords_train, self._counters,
                getattr(self, "_pos_train", None))

    def import_batch_state(self, st):
        self._coords, self._rays_numsteps, self._rays_numsteps_compacted, self._n_valid, self._coords_train, self._counters, self._pos_train = st

    def _inference_coords(self):
        if getattr(self, "_coords_inf", None) is None:
            self._coords_inf = torch.empty((self.max_samples, 7), dtype=torch.float32, device=self.device)
        return self._coords_inf

    def rays2rgb(self, network_outputs, training_background_color=None, inference=False):
        assert network_outputs.shape[0] == self._coords.shape[0]
        if inference:
            return ops.composite_inference(network_outputs.contiguous(), self._coords, self._rays_numsteps, self.NERF_CASCADES)
        bg = training_background_color
        if bg is None:
            bg = torch.tensor(self.background_color, dtype=torch.float32, device=self.device).expand(self._rays_numsteps.shape[0], 3).contiguous()
        return _Composite.apply(network_outputs, bg.contiguous(), self)

    # ------------------------------------------------------------------ occupancy grid (density_grid_sampler.py:204-264)
    @torch.no_grad()
    def update_density_grid_nerf(self, decay, n_uniform, n_nonuniform):
        if self.cfg.m_training_step == 0:
            ops.grid_mark_untrained(self.density_n_elements, self.dataset.focal_lengths, self.dataset.transforms_gpu,
                                    self.dataset.resolution[0], self.dataset.resolution[1], grid=self.density_grid)
        self.density_grid_tmp.zero_()
        n_total = n_uniform + n_nonuniform
        pos = torch.empty((n_total, 3), dtype=torch.float32, device=self.device)
        idx = torch.empty(n_total, dtype=torch.int32, device=self.device)
        mo = self.cfg.grid_samples_morton_order is not False       # (ours) same samples, Morton-ordered in memory: the 16-level gather of model.density() walks the tables coherently
        ops.grid_generate_samples(n_uniform, self.rng_state, self.density_grid_ema_step, self.aabb_range, self.density_grid, self.max_cascade + 1, -0.01,
                                  pos=pos[:n_uniform], idx=idx[:n_uniform], morton_order=mo)
        if n_nonuniform:
            ops.grid_generate_samples(n_nonuniform, self.rng_state, self.density_grid_ema_step, self.aabb_range, self.density_grid, self.max_cascade + 1,
                                      self.NERF_MIN_OPTICAL_THICKNESS, pos=pos[n_uniform:], idx=idx[n_uniform:], morton_order=mo)
        else:   # the reference still advances the global rng for the empty second call (generate_grid_samples…py:44)
            from .rng import pcg32_advance
            pcg32_advance(self.rng_state, 1 << 32)
        for i in range(0, n_total, self.update_block_size):
            d = self.model.density(pos[i:i + self.update_block_size])
            ops.grid_splat_max(idx[i:i + self.update_block_size], d.reshape(-1).contiguous(), self.density_grid_tmp)
        ops.grid_ema(self.density_grid, self.density_grid_tmp, self.density_grid_decay)
        self.density_grid_ema_step += 1
        ops.grid_update_bitfield(self.density_grid, self.NERF_CASCADES, mean=self.density_grid_mean, bitfield=self.density_grid_bitfield)
        if self._occ_bounds is not None and self.cfg.march_occupancy_bounds is not False:
            ops.grid_occupied_bounds(self.density_grid_bitfield, self.NERF_CASCADES, out=self._occ_bounds)
            self._occ_bounds_valid = True

    def occupancy_bounds(self):
        """the bounds tensor while it describes the current bitfield (it is refreshed with it; a bitfield loaded from a checkpoint invalidates it), else None"""
        return self._occ_bounds if self._occ_bounds_valid else None

    def load_state_dict(self, *a, **k):
        self._occ_bounds_valid = False
        return super().load_state_dict(*a, **k)

    def update_density_grid(self):
        alpha = pow(self.density_grid_decay, self.n_training_steps / 16)
        n_cascades = self.max_cascade + 1
        G3 = self.NERF_GRIDSIZE ** 3
        if self.cfg.m_training_step < 256:
            self.update_density_grid_nerf(alpha, G3 * n_cascades, 0)
        else:
            self.update_density_grid_nerf(alpha, G3 * n_cascades // 4, G3 * n_cascades // 4)

    def update_batch_rays(self):
        """density_grid_sampler.py:266-271, split in two: the running sample count is copied to pinned host memory asynchronously here (in stream order
        after the 16th batch's march) and turned into the new ray count by finish_batch_rays_update() when the NEXT batch is about to be generated -
        the only consumer.  A blocking read-back at this point would stall the host two batches ahead of the training stream and drain the pipeline."""
        measured = self.measured_batch_size
        if self.cfg.world_size and self.cfg.world_size > 1:
            import torch.distributed as dist
            m = measured.float()
            dist.all_reduce(m)                                          # every rank must pick the same ray count
            src, scale = m, 1.0 / self.cfg.world_size
        else:
            src, scale = measured, 1.0
        if measured.is_cuda:
            if self._measured_host is None:
                self._measured_host = {}
            host = self._measured_host.get(src.dtype)
            if host is None:
                host = self._measured_host[src.dtype] = torch.empty(1, dtype=src.dtype, pin_memory=True)
            host.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending_rays_update = (host, ev, scale)
        else:
            self._pending_rays_update = (src.clone(), None, scale)
        self.measured_batch_size.zero_()

    def finish_batch_rays_update(self):
        if self._pending_rays_update is None:
            return
        host, ev, scale = self._pending_rays_update
        self._pending_rays_update = None
        if ev is not None:
            ev.synchronize()
        measured_val = float(host.item()) * scale
        measured_batch_size = max(measured_val / 16, 1)                 # density_grid_sampler.py:266-271
        rays_per_batch = int(self.n_rays_per_batch * self.target_batch_size / measured_batch_size)
        self.n_rays_per_batch = int(min((int(rays_per_batch) + 127) // 128 * 128, self.target_batch_size))
        self.dataset.batch_size = self.n_rays_per_batch
        self.n_ray_count_updates = getattr(self, "n_ray_count_updates", 0) + 1
