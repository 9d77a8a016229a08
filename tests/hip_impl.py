"""numpy-in / numpy-out adapter over jnerf_amd.ops (the C ABI), signature-compatible with oracle.oracle so the same
parity checks run against either.  GPU only."""
import numpy as np
import torch
from jnerf_amd import ops

DEV = "cuda"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def u32(t):
    return N(t).view(np.uint32)


def level_table(aabb_scale):
    return ops.level_table(aabb_scale)


def hash_encode_fwd(x, grid, table, layout=ops.LAYOUT_AOS):
    out = ops.hash_encode_fwd(T(x.astype(np.float32)), T(grid), table, layout=layout)
    return N(out)


def hash_encode_fwd_dydx(x, grid, table):
    out, dydx = ops.hash_encode_fwd_dydx(T(x.astype(np.float32)), T(grid), table)
    return N(out), N(dydx)


def hash_encode_bwd_input(dy, dydx):
    return N(ops.hash_encode_bwd_input(T(dy), T(dydx)))


def hash_encode_bwd(x, dy, table, n_params, grad_dtype=None, layout=ops.LAYOUT_AOS):
    g = ops.hash_encode_bwd(T(x.astype(np.float32)), T(dy), table, n_params, grad_dtype=grad_dtype, layout=layout)
    return N(g)


def sh_encode(d, dtype=np.float32):
    return N(ops.sh_encode(T(d.astype(np.float32)), torch.float16 if dtype == np.float16 else torch.float32))


class _Rng:
    def __init__(self, st):
        self.st = st


def march_rays(rays_o, rays_d, bitfield, aabb, rng, max_samples, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5):
    c, ns, cnt, ri = ops.march_rays(T(rays_o), T(rays_d), T(bitfield), aabb, rng.st, max_samples, cone_angle, near, const_dt, cascades)
    return N(c), u32(ns), u32(cnt), N(ri)


def march_rays_compacted(rays_o, rays_d, bitfield, aabb, rng, max_samples, cap, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5):
    c, ns, nsc, cnt = ops.march_rays_compacted(T(rays_o), T(rays_d), T(bitfield), aabb, rng.st, max_samples, cap, cone_angle, near, const_dt, cascades)
    return N(c), u32(ns), u32(nsc), u32(cnt)


def march_rays_compacted_pos(rays_o, rays_d, bitfield, aabb, rng, max_samples, cap, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5):
    pos = torch.full((cap, 3), -7.0, dtype=torch.float32, device=DEV)
    c, ns, nsc, cnt = ops.march_rays_compacted(T(rays_o), T(rays_d), T(bitfield), aabb, rng.st, max_samples, cap, cone_angle, near, const_dt, cascades, pos_out=pos)
    return N(c), u32(cnt), N(pos)


def composite_fwd_huber(net, coords, numsteps, numsteps_c, bg, target, delta, cascades=5):
    r, l, g = ops.composite_fwd_huber(T(net), T(coords), T(numsteps.view(np.int32)), T(numsteps_c.view(np.int32)), T(bg), T(target), delta, cascades)
    return N(r), N(l), N(g)


def compact_coords(coords_in, numsteps_in, cap):
    c, ns, cnt = ops.compact_coords(T(coords_in), T(numsteps_in.view(np.int32)), cap)
    return N(c), u32(ns), u32(cnt)


def composite_fwd(net, coords, numsteps, numsteps_c, bg, cascades=5):
    return N(ops.composite_fwd(T(net), T(coords), T(numsteps.view(np.int32)), T(numsteps_c.view(np.int32)), T(bg), cascades))


def composite_bwd(net, coords, numsteps_c, loss_grad, rgb_ray, mean, cascades=5):
    m = torch.tensor([mean], dtype=torch.float32, device=DEV)
    return N(ops.composite_bwd(T(net), T(coords), T(numsteps_c.view(np.int32)), T(loss_grad), T(rgb_ray), m, cascades))


def composite_inference(net, coords, numsteps, cascades=5):
    r, a = ops.composite_inference(T(net), T(coords), T(numsteps.view(np.int32)), cascades)
    return N(r), N(a)


def grid_mark_untrained(n_elements, focal, xforms, W, H):
    return N(ops.grid_mark_untrained(n_elements, T(focal), T(xforms), W, H))


def grid_generate_samples(n, rng, step, aabb, grid, n_cascades, thresh):
    st = torch.tensor([step], dtype=torch.int32, device=DEV)
    p, i = ops.grid_generate_samples(n, rng.st, st, aabb, T(grid), n_cascades, thresh)
    return N(p), u32(i)


def grid_splat_max(indices, mlp_out, grid_tmp):
    return N(ops.grid_splat_max(T(indices.view(np.int32)), T(mlp_out), T(grid_tmp)))


def grid_ema(grid, grid_tmp, decay=0.95):
    return N(ops.grid_ema(T(grid), T(grid_tmp), decay))


def grid_update_bitfield(grid, cascades=5):
    b, m = ops.grid_update_bitfield(T(grid), cascades)
    return N(b), N(m)


def huber(x, target, delta=0.1):
    l, g = ops.huber(T(x), T(target), delta)
    return N(l), N(g)


def adam_ema_step(p, g, m, v, ema, lr, step, b0=0.9, b1=0.99, eps=1e-15, ema_decay=0.95, half=False):
    tp, tg, tm, tv = T(p), T(g), T(m), T(v)
    te = T(ema) if ema is not None else None
    th = torch.empty(p.shape, dtype=torch.float16, device=DEV) if half else None
    ops.adam_ema_step(tp, tg, tm, tv, te, th, lr, step, b0, b1, eps, ema_decay, zero_grad=True)
    return N(tp), N(tm), N(tv), (N(te) if te is not None else None), (N(th) if half else None), N(tg)


def generate_rays(index, W, H, focal, meta, xforms):
    img, o, d, _ = ops.generate_rays(T(index.astype(np.int64)), W, H, T(focal), T(meta), T(xforms))
    return N(img), N(o), N(d)
