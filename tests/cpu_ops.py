"""TEST INFRASTRUCTURE ONLY: `jnerf_amd.ops` entry points re-bound to the C oracle for CPU tensors, so that the HOST side of this build - Runner's module path, NerfDataset,
DensityGridSampler, NGPNetworks on HashEncoder / SHEncoder, the autograd bridges, HuberLoss, Adam / ExpDecay / EMA - can be executed end to end without a GPU
(tests/test_refrun_golden.py replays the reference's own training run through it).  The product never takes this route: jnerf_amd.ops binds libngp_hip.so only and fails
loudly without it; this module is imported by tests alone and patches the functions in place for the duration of a `with oracle_backed_ops():` block.

Every replacement keeps the signature and the buffer conventions of the function it stands in for (pre-allocated outputs are written in place, fixed-capacity buffers
plus device-side counters stay as they are) and calls the oracle function of the same name."""
import contextlib
import types
import numpy as np
import torch


def _np(t):
    return t.detach().contiguous().numpy()


def _into(dst, arr):
    dst.reshape(-1)[:arr.size].copy_(torch.from_numpy(np.ascontiguousarray(arr)).reshape(-1).to(dst.dtype))
    return dst


@contextlib.contextmanager
def oracle_backed_ops():
    from oracle import oracle as O
    from jnerf_amd import ops
    saved = {}

    def bind(name, fn):
        saved[name] = getattr(ops, name)
        setattr(ops, name, fn)

    def generate_rays(pixel_index, W, H, focal, metadata, xforms, images=None, bg=None, out=None):
        assert out is None
        ids, o, d = O.generate_rays(_np(pixel_index).astype(np.int64), W, H, _np(focal), np.ascontiguousarray(_np(metadata)[:, 4:6]), _np(xforms))
        target = None
        if images is not None:                                   # the fused target compositing of runner.py:66-67
            rgba = images.reshape(-1, 4)[pixel_index]
            target = (rgba[:, :3] * rgba[:, 3:] + bg * (1 - rgba[:, 3:])).contiguous()
        return torch.from_numpy(ids.astype(np.int32)), torch.from_numpy(o), torch.from_numpy(d), target

    def march_rays_compacted(rays_o, rays_d, bitfield, aabb, rng_state, max_samples, cap, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5,
                             coords_out=None, numsteps=None, numsteps_c=None, counters=None, scratch=None, pos_out=None, occ_bounds=None):
        rng = types.SimpleNamespace(st=rng_state)                # advanced in place, like the library's host-side state
        co, ns, cnt, _ = O.march_rays(_np(rays_o), _np(rays_d), _np(bitfield), aabb, rng, max_samples, cone_angle, near, const_dt, cascades)
        M = int(min(cnt[1], max_samples))
        cc, nsc, counter = O.compact_coords(co[:M], ns, cap)
        k = int(min(int(nsc[:, 0].sum()), cap))
        n = rays_o.shape[0]
        coords_out = torch.zeros((cap, 7)) if coords_out is None else coords_out
        numsteps = torch.empty((n, 2), dtype=torch.int32) if numsteps is None else numsteps
        numsteps_c = torch.empty((n, 2), dtype=torch.int32) if numsteps_c is None else numsteps_c
        counters = torch.empty(4, dtype=torch.int32) if counters is None else counters
        _into(coords_out, cc)
        _into(numsteps, ns.view(np.int32))
        _into(numsteps_c, nsc.view(np.int32))
        _into(counters, np.asarray([cnt[0], cnt[1], counter[0], k], np.uint32).view(np.int32))      # [rays, samples marched, samples kept (unclamped), valid rows]
        if pos_out is not None:
            _into(pos_out, cc[:, :3])
        return coords_out, numsteps, numsteps_c, counters

    def march_rays(rays_o, rays_d, bitfield, aabb, rng_state, max_samples, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5, coords=None, zero_coords=True):
        rng = types.SimpleNamespace(st=rng_state)
        co, ns, cnt, idx = O.march_rays(_np(rays_o), _np(rays_d), _np(bitfield), aabb, rng, max_samples, cone_angle, near, const_dt, cascades)
        coords = torch.empty((max_samples, 7)) if coords is None else coords
        _into(coords, co)
        return coords, torch.from_numpy(ns.view(np.int32).copy()), torch.from_numpy(cnt.view(np.int32).copy()), torch.from_numpy(idx)

    def hash_encode_fwd(pos, table, level_tbl, out=None, layout=ops.LAYOUT_AOS, n_valid=None):
        assert out is None and layout == ops.LAYOUT_AOS and n_valid is None
        return torch.from_numpy(O.hash_encode_fwd(_np(pos), _np(table), level_tbl))

    def hash_encode_bwd(pos, dLdy, level_tbl, n_params, grad=None, grad_dtype=None, layout=ops.LAYOUT_AOS, zero_first=True, n_valid=None, workspace=None):
        assert layout == ops.LAYOUT_AOS and n_valid is None
        g = torch.from_numpy(O.hash_encode_bwd(_np(pos), _np(dLdy), level_tbl, n_params))
        if grad is None:
            return g
        if zero_first:
            grad.zero_()
        return grad.add_(g)

    def sh_encode(d, dtype=torch.float32):
        assert dtype == torch.float32
        return torch.from_numpy(O.sh_encode(_np(d), np.float32))

    def composite_fwd(net, coords, numsteps, numsteps_c, bg, cascades=5, out=None):
        rgb = torch.from_numpy(O.composite_fwd(_np(net), _np(coords), _np(numsteps).view(np.uint32), _np(numsteps_c).view(np.uint32), _np(bg), cascades))
        return rgb if out is None else out.copy_(rgb)

    def composite_bwd(net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades=5, dout=None, zero_first=True):
        d = torch.from_numpy(O.composite_bwd(_np(net), _np(coords), _np(numsteps_c).view(np.uint32), _np(loss_grad), _np(rgb_ray), float(density_grid_mean.reshape(-1)[0]), cascades))
        return d if dout is None else dout.copy_(d)

    def composite_inference(net, coords, numsteps, cascades=5):
        rgb, alpha = O.composite_inference(_np(net), _np(coords), _np(numsteps).view(np.uint32), cascades)
        return torch.from_numpy(rgb), torch.from_numpy(alpha)

    def grid_mark_untrained(n_elements, focal, xforms, W, H, grid=None):
        g = torch.from_numpy(O.grid_mark_untrained(n_elements, _np(focal), _np(xforms), W, H))
        return g if grid is None else grid.copy_(g)

    def grid_generate_samples(n, rng_state, ema_step, aabb, grid, n_cascades, thresh, pos=None, idx=None, morton_order=False):
        rng = types.SimpleNamespace(st=rng_state)
        p, i = O.grid_generate_samples(n, rng, int(ema_step.reshape(-1)[0]), aabb, _np(grid), n_cascades, thresh)      # (memory order of the samples: the generator's own)
        pos = torch.empty((n, 3)) if pos is None else pos
        idx = torch.empty(n, dtype=torch.int32) if idx is None else idx
        _into(pos, p)
        _into(idx, i.view(np.int32))
        return pos, idx

    def grid_splat_max(indices, density, grid_tmp):
        tmp = _np(grid_tmp)
        O.grid_splat_max(_np(indices).view(np.uint32), _np(density).astype(np.float32), tmp)
        return grid_tmp.copy_(torch.from_numpy(tmp))

    def grid_ema(grid, grid_tmp, decay=0.95):
        g = _np(grid)
        O.grid_ema(g, _np(grid_tmp), decay)
        return grid.copy_(torch.from_numpy(g))

    def grid_update_bitfield(grid, cascades=5, mean=None, bitfield=None):
        bits, m = O.grid_update_bitfield(_np(grid), cascades)
        mean = torch.empty(1) if mean is None else mean
        bitfield = torch.zeros(128 ** 3 * cascades // 8, dtype=torch.uint8) if bitfield is None else bitfield
        _into(mean, m)
        _into(bitfield, bits)
        return bitfield, mean

    for name, fn in dict(generate_rays=generate_rays, march_rays_compacted=march_rays_compacted, march_rays=march_rays, march_scratch_elems=lambda n: n + 1024,
                         hash_encode_fwd=hash_encode_fwd, hash_encode_bwd=hash_encode_bwd, hash_bwd_workspace_bytes=lambda tbl, n, dtype=None, grad_dtype=None: 16, sh_encode=sh_encode,
                         composite_fwd=composite_fwd, composite_bwd=composite_bwd, composite_inference=composite_inference, grid_mark_untrained=grid_mark_untrained,
                         grid_generate_samples=grid_generate_samples, grid_splat_max=grid_splat_max, grid_ema=grid_ema, grid_update_bitfield=grid_update_bitfield).items():
        bind(name, fn)
    try:
        yield
    finally:
        for name, fn in saved.items():
            setattr(ops, name, fn)
