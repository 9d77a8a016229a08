"""TEST INFRASTRUCTURE ONLY — ctypes front-end of oracle/_ref/*.so: the reference's OWN kernel headers
(/root/reference/python/jnerf/**/op_header/*.h) compiled for the host through oracle/ref_shim/ (serial launcher).
Exists only where /root/reference was present at build time (this container) or where the prebuilt .so files travelled.
Used to (a) validate oracle/ngp_oracle.c and (b) mint tests/golden/*.npz (tests/golden/make_golden.py)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "_ref")
_LIBS = {}
NAMES = ["hash", "sh", "march_constdt", "march_cone", "compact", "calc_rgb", "grid_mark", "grid_gen", "grid_splat", "grid_ema", "grid_bitfield", "pcg32"]
# NERF_CASCADES is a constant of the reference's generated prelude (density_grid_sampler.py:56-60, 96-116): one build per value. 5 is the default, 7 what aabb_scale 64 selects.
NAMES_C7 = ["march_constdt_c7", "march_cone_c7", "compact_c7", "calc_rgb_c7", "grid_gen_c7", "grid_bitfield_c7"]


def available():
    return all(os.path.exists(os.path.join(_DIR, f"libref_{n}.so")) for n in NAMES + NAMES_C7)


def _cs(cascades):
    assert cascades in (5, 7), "oracle/ref_shim/Makefile builds NERF_CASCADES = 5 and 7"
    return "" if cascades == 5 else "_c7"


def build(reference="/root/reference"):
    if not os.path.isdir(reference):
        return False
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "ref_shim"), f"REF={reference}"])
    return True


def _l(name):
    if name not in _LIBS:
        _LIBS[name] = C.CDLL(os.path.join(_DIR, f"libref_{name}.so"))
    return _LIBS[name]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _sfx(a):
    return "f16" if a.dtype == np.float16 else "f32"


def per_level_scale(aabb_scale):
    from math import exp, log
    return exp(log(2048.0 * aabb_scale / 16) / 15)   # grid_encode.py:20


class PCG32:
    def __init__(self, seed=1337, initseq=1):
        self.st = np.zeros(2, np.uint64)
        _l("pcg32").ref_pcg32_seed(C.c_uint64(seed), C.c_uint64(initseq), _p(self.st))

    def next_uint(self):
        f = _l("pcg32").ref_pcg32_next_uint
        f.restype = C.c_uint32
        return int(f(_p(self.st)))

    def next_float(self):
        f = _l("pcg32").ref_pcg32_next_float
        f.restype = C.c_float
        return float(f(_p(self.st)))

    def advance(self, delta=1 << 32):
        _l("pcg32").ref_pcg32_advance(_p(self.st), C.c_int64(delta))


def hash_fwd(x, grid, offsets, aabb_scale):
    x = _c(x, np.float32)
    out = np.zeros((x.shape[0], 32), grid.dtype)
    getattr(_l("hash"), "ref_hash_fwd_" + _sfx(grid))(C.c_uint32(x.shape[0]), _p(x), _p(grid), _p(_c(offsets, np.uint32)), C.c_double(per_level_scale(aabb_scale)), _p(out))
    return out


def hash_fwd_dydx(x, grid, offsets, aabb_scale):
    """kernel_grid with its dy_dx output enabled (HashEncode.h:205-251): -> (out [n,32] T, dydx [n,3,32] f32)"""
    x = _c(x, np.float32)
    out = np.zeros((x.shape[0], 32), grid.dtype)
    dydx = np.zeros((x.shape[0], 3, 32), np.float32)
    getattr(_l("hash"), "ref_hash_fwd_dydx_" + _sfx(grid))(C.c_uint32(x.shape[0]), _p(x), _p(grid), _p(_c(offsets, np.uint32)), C.c_double(per_level_scale(aabb_scale)), _p(out), _p(dydx))
    return out, dydx


def hash_bwd(x, dy, offsets, aabb_scale, n_params):
    x = _c(x, np.float32)
    dy = np.ascontiguousarray(dy)
    grad = np.zeros(n_params, dy.dtype)
    getattr(_l("hash"), "ref_hash_bwd_" + _sfx(dy))(C.c_uint32(x.shape[0]), _p(x), _p(dy), _p(_c(offsets, np.uint32)), C.c_double(per_level_scale(aabb_scale)), _p(grad), C.c_uint64(n_params))
    return grad


def sh(d, dtype=np.float32):
    d = _c(d, np.float32)
    out = np.zeros((d.shape[0], 16), dtype)
    getattr(_l("sh"), "ref_sh_" + _sfx(out))(C.c_uint32(d.shape[0]), _p(d), _p(out))
    return out


def march(rays_o, rays_d, bitfield, aabb, rng_state, max_samples, metadata, img_ids, xforms, cone_angle=1.0 / 256, near=0.2, const_dt=True, cascades=5):
    rays_o, rays_d = _c(rays_o, np.float32), _c(rays_d, np.float32)
    n = rays_o.shape[0]
    coords = np.zeros((max_samples, 7), np.float32)
    numsteps = np.zeros((n, 2), np.uint32)
    counters = np.zeros(2, np.uint32)
    ray_idx = np.zeros(n, np.int32)
    name = "march_constdt" if const_dt else "march_cone"
    getattr(_l(name + _cs(cascades)), "ref_" + name)(C.c_uint32(n), C.c_float(aabb[0]), C.c_float(aabb[1]), C.c_uint32(max_samples), _p(rays_o), _p(rays_d),
                                     _p(_c(bitfield, np.uint8)), C.c_float(cone_angle), _p(_c(metadata, np.float32)), _p(_c(img_ids, np.uint32)),
                                     _p(counters), _p(ray_idx), _p(numsteps), _p(coords), _p(_c(xforms, np.float32)), C.c_float(near), _p(rng_state))
    return coords, numsteps, counters, ray_idx


def compact(net, coords_in, numsteps_in, cap, aabb, cascades=5):
    net = np.ascontiguousarray(net)
    n = numsteps_in.shape[0]
    coords_out = np.zeros((cap, 7), np.float32)
    numsteps_out = np.zeros((n, 2), np.uint32)
    counter = np.zeros(1, np.uint32)
    rays_counter = np.zeros(1, np.uint32)
    getattr(_l("compact" + _cs(cascades)), "ref_compact_" + _sfx(net))(C.c_uint32(n), C.c_float(aabb[0]), C.c_float(aabb[1]), C.c_uint32(cap), _p(net), _p(_c(coords_in, np.float32)),
                                                       _p(coords_out), _p(_c(numsteps_in, np.uint32)), _p(counter), _p(numsteps_out), _p(rays_counter))
    return coords_out, numsteps_out, counter


def rgb_fwd(net, coords, numsteps, numsteps_c, bg, aabb, cascades=5):
    net = np.ascontiguousarray(net)
    n = numsteps.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    getattr(_l("calc_rgb" + _cs(cascades)), "ref_rgb_fwd_" + _sfx(net))(C.c_uint32(n), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(net), _p(_c(coords, np.float32)), _p(_c(numsteps, np.uint32)),
                                                         _p(rgb), _p(_c(numsteps_c, np.uint32)), _p(_c(bg, np.float32)))
    return rgb


def rgb_bwd(net, coords, numsteps_c, loss_grad, rgb_ray, mean, aabb, cascades=5):
    net = np.ascontiguousarray(net)
    n = numsteps_c.shape[0]
    dout = np.zeros_like(net)
    m = np.array([mean], np.float32)
    getattr(_l("calc_rgb" + _cs(cascades)), "ref_rgb_bwd_" + _sfx(net))(C.c_uint32(n), C.c_uint32(net.shape[0]), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(dout), _p(net),
                                                         _p(_c(numsteps_c, np.uint32)), _p(_c(coords, np.float32)), _p(_c(loss_grad, np.float32)), _p(_c(rgb_ray, np.float32)), _p(m))
    return dout


def rgb_inference(net, coords, numsteps, aabb, cascades=5):
    net = np.ascontiguousarray(net)
    n = numsteps.shape[0]
    rgb = np.zeros((n, 3), np.float32)
    alpha = np.zeros((n, 1), np.float32)
    getattr(_l("calc_rgb" + _cs(cascades)), "ref_rgb_inf_" + _sfx(net))(C.c_uint32(n), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(net), _p(_c(coords, np.float32)), _p(_c(numsteps, np.uint32)), _p(rgb), _p(alpha))
    return rgb, alpha


def grid_mark(n_elements, focal, xforms, W, H):
    focal = _c(focal, np.float32)
    grid = np.zeros(n_elements, np.float32)
    _l("grid_mark").ref_grid_mark(C.c_uint32(n_elements), _p(grid), C.c_uint32(focal.shape[0]), _p(focal), _p(_c(xforms, np.float32)), int(W), int(H))
    return grid


def grid_gen(n, rng_state, step, aabb, grid, n_cascades, thresh, cascades=5):
    pos = np.zeros((n, 3), np.float32)
    idx = np.zeros(n, np.uint32)
    _l("grid_gen" + _cs(cascades)).ref_grid_gen(C.c_uint32(n), _p(rng_state), C.c_uint32(step), C.c_float(aabb[0]), C.c_float(aabb[1]), _p(_c(grid, np.float32)), _p(pos), _p(idx),
                                C.c_uint32(n_cascades), C.c_float(thresh))
    return pos, idx


def grid_splat(indices, mlp_out, grid_tmp):
    mlp_out = np.ascontiguousarray(mlp_out)
    getattr(_l("grid_splat"), "ref_grid_splat_" + _sfx(mlp_out))(C.c_uint32(indices.shape[0]), _p(_c(indices, np.uint32)), _p(mlp_out), _p(grid_tmp))
    return grid_tmp


def grid_ema(grid, grid_tmp, decay=0.95):
    _l("grid_ema").ref_grid_ema(C.c_uint32(grid.shape[0]), C.c_float(decay), _p(grid), _p(grid_tmp))
    return grid


def grid_bitfield(grid, cascades=5):
    mean = np.zeros(1, np.float32)
    bitfield = np.zeros(128 ** 3 * cascades // 8, np.uint8)
    _l("grid_bitfield" + _cs(cascades)).ref_grid_bitfield(_p(_c(grid, np.float32)), _p(mean), _p(bitfield))
    return bitfield, mean
