"""TEST INFRASTRUCTURE ONLY — numpy/ctypes front-end of oracle/libngp_oracle.so (oracle/ngp_oracle.c).

Every function takes/returns numpy arrays; T ∈ {float32, float16} is chosen by the dtype of the table /
network-output argument.  See ngp_oracle.c for the reference file:line each routine restates.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libngp_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_h2f.restype = C.c_float
        _LIB.orc_f2h.restype = C.c_uint16
        _LIB.orc_f2h.argtypes = [C.c_float]
        _LIB.orc_pcg32_next_float.restype = C.c_float
        _LIB.orc_pcg32_next_uint.restype = C.c_uint32
        _LIB.orc_level_table.restype = C.c_uint32
        _LIB.orc_level_table.argtypes = [C.c_double, C.c_void_p, C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _is_half(a):
    assert a.dtype in (np.float32, np.float16), a.dtype
    return int(a.dtype == np.float16)


# ---------------------------------------------------------------- pcg32
class PCG32:
    """pcg32{seed} of ops/op_include/pcg32/pcg32.h; default stream (initseq=1), seed 1337 = the reference's global rng."""

    def __init__(self, seed=1337, initseq=1):
        self.st = np.zeros(2, np.uint64)
        lib().orc_pcg32_seed(C.c_uint64(seed), C.c_uint64(initseq), _p(self.st))

    def next_uint(self):
        return int(lib().orc_pcg32_next_uint(_p(self.st)))

    def next_float(self):
        return float(lib().orc_pcg32_next_float(_p(self.st)))

    def advance(self, delta=1 << 32):
        lib().orc_pcg32_advance(_p(self.st), C.c_int64(delta))

    def copy(self):
        r = PCG32.__new__(PCG32)
        r.st = self.st.copy()
        return r


# ---------------------------------------------------------------- hash grid
def level_table(aabb_scale):
    """-> (table u32[16,4] = offset,size,res,scale_bits ; offsets u32[17] ; n_params)"""
    table = np.zeros((16, 4), np.uint32)
    offsets = np.zeros(17, np.uint32)
    n_params = lib().orc_level_table(float(aabb_scale), _p(table), _p(offsets))
    return table, offsets, int(n_params)


def hash_encode_fwd(x, grid, table):
    x = _c(x, np.float32)
    n = x.shape[0]
    out = np.zeros((n, 32), grid.dtype)
    lib().orc_hash_encode_fwd(C.c_uint32(n), _p(x), _p(grid), _p(table), _p(out), _is_half(grid))
    return out


def hash_encode_fwd_dydx(x, grid, table):
    """-> (out [n,32] T, dydx [n,3,32] f32): HashEncode.h:117-251 with the dy_dx branch enabled"""
    x = _c(x, np.float32)
    n = x.shape[0]
    out = np.zeros((n, 32), grid.dtype)
    dydx = np.zeros((n, 3, 32), np.float32)
    lib().orc_hash_encode_fwd_dydx(C.c_uint32(n), _p(x), _p(grid), _p(table), _p(out), _p(dydx), _is_half(grid))
    return out, dydx


def hash_encode_bwd_input(dy, dydx):
    dy = np.ascontiguousarray(dy)
    n = dy.shape[0]
    dLdx = np.zeros((n, 3), np.float32)
    lib().orc_hash_encode_bwd_input(C.c_uint32(n), _p(dy), _p(_c(dydx, np.float32)), _p(dLdx), _is_half(dy))
    return dLdx


def hash_encode_bwd(x, dy, table, n_params):
    x = _c(x, np.float32)
    dy = np.ascontiguousarray(dy)
    grad = np.zeros(n_params, dy.dtype)
    lib().orc_hash_encode_bwd(C.c_uint32(x.shape[0]), _p(x), _p(dy), _p(table), _p(grad), C.c_uint64(n_params), _is_half(dy))
    return grad


def sh_encode(d, dtype=np.float32):
    d = _c(d, np.float32)
    out = np.zeros((d.shape[0], 16), dtype)
    lib().orc_sh_encode(C.c_uint32(d.shape[0]), _p(d), _p(out), _is_half(out))
    return out


# ---------------------------------------------------------------- field MLPs (fp32)
def field_fwd(feat, sh, wd, wc, save=False):
    feat, sh, wd, wc = _c(feat, np.float32), _c(sh, np.float32), _c(wd, np.float32), _c(wc, np.float32)
    n = feat.shape[0]
    out = np.zeros((n, 4), np.float32)
    if save:
        h, den, g0, g1 = (np.zeros((n, 64), np.float32), np.zeros((n, 16), np.float32), np.zeros((n, 64), np.float32), np.zeros((n, 64), np.float32))
    else:
        h = den = g0 = g1 = None
    lib().orc_field_fwd(C.c_uint32(n), _p(feat), _p(sh), _p(wd), _p(wc), _p(out), _p(h), _p(den), _p(g0), _p(g1))
    return (out, h, den, g0, g1) if save else out


def density_fwd(feat, wd):
    feat, wd = _c(feat, np.float32), _c(wd, np.float32)
    out = np.zeros(feat.shape[0], np.float32)
    lib().orc_density_fwd(C.c_uint32(feat.shape[0]), _p(feat), _p(wd), _p(out))
    return out


def field_bwd(feat, sh, wd, wc, dout):
    feat, sh, wd, wc, dout = (_c(a, np.float32) for a in (feat, sh, wd, wc, dout))
    n = feat.shape[0]
    dfeat = np.zeros((n, 32), np.float32)
    dwd = np.zeros(3072, np.float32)
    dwc = np.zeros(7168, np.float32)
    lib().orc_field_bwd(C.c_uint32(n), _p(feat), _p(sh), _p(wd), _p(wc), _p(dout), _p(dfeat), _p(dwd), _p(dwc))
    return dfeat, dwd, dwc


# ---------------------------------------------------------------- sampler
def march_rays(rays_o, rays_d, bitfield, aabb, rng, max_samples, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5):
    """-> coords[max_samples,7], numsteps[n,2], counters[2], ray_indices[n]; rng (PCG32) is advanced by 2^32 like ray_sampler.py:61."""
    rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
    n = rays_o.shape[0]
    coords = np.zeros((max_samples, 7), np.float32)
    numsteps = np.zeros((n, 2), np.uint32)
    counters = np.zeros(2, np.uint32)
    ray_idx = np.zeros(n, np.int32)
    bitfield = _c(bitfield, np.uint8)
    lib().orc_march_rays(C.c_uint32(n), C.c_float(aabb[0]), C.c_float(aabb[1]), C.c_uint32(max_samples), _p(rays_o), _p(rays_d), _p(bitfield),
                         C.c_float(cone_angle), C.c_float(near), int(const_dt), int(cascades), _p(rng.st), _p(counters), _p(ray_idx), _p(numsteps), _p(coords))
    return coords, numsteps, counters, ray_idx


def compact_coords(coords_in, numsteps_in, cap):
    coords_in, numsteps_in = _c(coords_in, np.float32), _c(numsteps_in, np.uint32)
    n = numsteps_in.shape[0]
    coords_out = np.zeros((cap, 7), np.float32)
    numsteps_out = np.zeros((n, 2), np.uint32)
    counter = np.zeros(1, np.uint32)
    lib().orc_compact_coords(C.c_uint32(n), C.c_uint32(cap), _p(coords_in), _p(numsteps_in), _p(coords_out), _p(numsteps_out), _p(counter))
    return coords_out, numsteps_out, counter


def composite_fwd(net, coords, numsteps, numsteps_c, bg, cascades=5):
    net = np.ascontiguousarray(net)
    coords, numsteps, numsteps_c, bg = _c(coords, np.float32), _c(numsteps, np.uint32), _c(numsteps_c, np.uint32), _c(bg, np.float32)
    n = numsteps.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    lib().orc_composite_fwd(C.c_uint32(n), _p(net), _p(coords), _p(numsteps), _p(numsteps_c), _p(bg), int(cascades), _p(rgb), _is_half(net))
    return rgb


def composite_bwd(net, coords, numsteps_c, loss_grad, rgb_ray, density_grid_mean, cascades=5):
    net = np.ascontiguousarray(net)
    coords, numsteps_c, loss_grad, rgb_ray = _c(coords, np.float32), _c(numsteps_c, np.uint32), _c(loss_grad, np.float32), _c(rgb_ray, np.float32)
    n = numsteps_c.shape[0]
    dout = np.zeros_like(net)
    lib().orc_composite_bwd(C.c_uint32(n), C.c_uint32(net.shape[0]), _p(net), _p(coords), _p(numsteps_c), _p(loss_grad), _p(rgb_ray),
                            C.c_float(density_grid_mean), int(cascades), _p(dout), _is_half(net))
    return dout


def composite_inference(net, coords, numsteps, cascades=5):
    net = np.ascontiguousarray(net)
    coords, numsteps = _c(coords, np.float32), _c(numsteps, np.uint32)
    n = numsteps.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    alpha = np.zeros((n, 1), np.float32)
    lib().orc_composite_inference(C.c_uint32(n), _p(net), _p(coords), _p(numsteps), int(cascades), _p(rgb), _p(alpha), _is_half(net))
    return rgb, alpha


# ---------------------------------------------------------------- density grid
def grid_mark_untrained(n_elements, focal, xforms, W, H):
    focal, xforms = _c(focal, np.float32), _c(xforms, np.float32)
    grid = np.zeros(n_elements, np.float32)
    lib().orc_grid_mark_untrained(C.c_uint32(n_elements), _p(grid), C.c_uint32(focal.shape[0]), _p(focal), _p(xforms), int(W), int(H))
    return grid


def grid_generate_samples(n, rng, step, aabb, grid, n_cascades, thresh):
    grid = _c(grid, np.float32)
    pos = np.zeros((n, 3), np.float32)
    idx = np.zeros(n, np.uint32)
    lib().orc_grid_generate_samples(C.c_uint32(n), _p(rng.st), C.c_uint32(step), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(grid), _p(pos), _p(idx),
                                    C.c_uint32(n_cascades), C.c_float(thresh))
    return pos, idx


def grid_splat_max(indices, mlp_out, grid_tmp):
    indices = _c(indices, np.uint32)
    mlp_out = np.ascontiguousarray(mlp_out)
    lib().orc_grid_splat_max(C.c_uint32(indices.shape[0]), _p(indices), _p(mlp_out), _p(grid_tmp), _is_half(mlp_out))
    return grid_tmp


def grid_ema(grid, grid_tmp, decay=0.95):
    lib().orc_grid_ema(C.c_uint32(grid.shape[0]), C.c_float(decay), _p(grid), _p(grid_tmp))
    return grid


def grid_update_bitfield(grid, cascades=5):
    grid = _c(grid, np.float32)
    mean = np.zeros(1, np.float32)
    bitfield = np.zeros(128 ** 3 * cascades // 8, np.uint8)
    lib().orc_grid_update_bitfield(_p(grid), int(cascades), _p(mean), _p(bitfield))
    return bitfield, mean


# ---------------------------------------------------------------- loss / optimiser / rays
def huber(x, target, delta=0.1):
    x, target = _c(x, np.float32), _c(target, np.float32)
    loss, grad = np.zeros_like(x), np.zeros_like(x)
    lib().orc_huber(C.c_uint32(x.size), _p(x), _p(target), C.c_float(delta), _p(loss), _p(grad))
    return loss, grad


def adam_ema_step(p, g, m, v, ema, lr, step, b0=0.9, b1=0.99, eps=1e-15, ema_decay=0.95):
    """in place on p, m, v, ema (float32); step is 1-based."""
    lib().orc_adam_ema_step(C.c_uint64(p.size), _p(p), _p(_c(g, np.float32)), _p(m), _p(v), _p(ema), C.c_float(lr), C.c_float(b0), C.c_float(b1),
                            C.c_float(eps), C.c_uint32(step), C.c_float(ema_decay))


def generate_rays(index, W, H, focal, pp, xforms):
    index = _c(index, np.int64)
    focal, pp, xforms = _c(focal, np.float32), _c(pp, np.float32), _c(xforms, np.float32)
    n = index.shape[0]
    img = np.zeros(n, np.int32)
    o = np.zeros((n, 3), np.float32)
    d = np.zeros((n, 3), np.float32)
    lib().orc_generate_rays(C.c_uint32(n), _p(index), int(W), int(H), _p(focal), _p(pp), _p(xforms), _p(img), _p(o), _p(d))
    return img, o, d
