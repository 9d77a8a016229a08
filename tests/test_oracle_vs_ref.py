"""Pins oracle/ngp_oracle.c (plain-C restatement) against oracle/_ref (the reference's OWN kernel headers compiled for
the host).  Runs wherever oracle/_ref/*.so exists (this container; the files also travel to the GPU box)."""
import numpy as np
import pytest
from oracle import oracle as O, ref as R
import synth

pytestmark = pytest.mark.skipif(not R.available(), reason="oracle/_ref not built (needs /root/reference)")


def test_pcg32_stream():
    a, b = O.PCG32(1337), R.PCG32(1337)
    assert [a.next_uint() for _ in range(64)] == [b.next_uint() for _ in range(64)]
    a.advance(8 * 1234567); b.advance(8 * 1234567)
    assert a.next_float() == b.next_float()
    a.advance(); b.advance()
    assert (a.st == b.st).all()


# aabb_scale >= 32: the finest levels' resolution exceeds 2^15, so grid_index's uint32 stride wraps to 0 before the z term (HashEncode.h:82-91) - the level is
# still hashed (its size is capped at 2^19), but the wrapped stride decides the dense-or-hashed test
AABB_SCALES = [1, 4, 8, 32, 64, 128]


@pytest.mark.parametrize("aabb_scale", AABB_SCALES)
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_hash_encode(aabb_scale, dtype):
    table, offsets, n_params = O.level_table(aabb_scale)
    rng = np.random.default_rng(0)
    n = 2048
    x = synth.uniform_positions(n)
    x[:8] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 0.5, 0.5], [0.999999, 0.3, 0.7], [1e-7, 1, 0], [0.25, 0.75, 1], [1, 1, 0]]
    grid = rng.uniform(-1, 1, n_params).astype(dtype)
    a = O.hash_encode_fwd(x, grid, table)
    b = R.hash_fwd(x, grid, offsets, aabb_scale)
    # same indices, same weights => identical up to the 1-ulp freedom of exp2f in the per-level scale
    assert np.allclose(a.astype(np.float32), b.astype(np.float32), rtol=0, atol=2e-3 if dtype == np.float16 else 2e-5)
    assert (a == b).mean() > 0.98
    dy = (rng.normal(size=(n, 32)) * 1e-2).astype(dtype)
    ga = O.hash_encode_bwd(x, dy, table, n_params)
    gb = R.hash_bwd(x, dy, offsets, aabb_scale, n_params)
    assert ((ga != 0) == (gb != 0)).mean() > 0.9999
    assert np.allclose(ga.astype(np.float32), gb.astype(np.float32), rtol=0, atol=2e-3 if dtype == np.float16 else 1e-6)


@pytest.mark.parametrize("aabb_scale", AABB_SCALES)
@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_hash_encode_dydx(aabb_scale, dtype):
    """the dy_dx branch of the reference's kernel_grid (HashEncode.h:205-251), compiled into oracle/_ref with the output pointer set: the restatement is bit-identical"""
    table, offsets, n_params = O.level_table(aabb_scale)
    x = synth.uniform_positions(2048, seed=9)
    x[:8] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 0.5, 0.5], [0.999999, 0.3, 0.7], [1e-7, 1, 0], [0.25, 0.75, 1], [1, 1, 0]]
    grid = np.random.default_rng(2).uniform(-1, 1, n_params).astype(dtype)
    oa, da = O.hash_encode_fwd_dydx(x, grid, table)
    ob, db = R.hash_fwd_dydx(x, grid, offsets, aabb_scale)
    assert np.array_equal(oa, O.hash_encode_fwd(x, grid, table)) and (oa == ob).mean() > 0.98
    assert np.array_equal(da, db)
    assert np.abs(db).max() > 1.0                      # derivatives scale with the level resolution: not a vacuous comparison


@pytest.mark.parametrize("dtype", [np.float32, np.float16])
def test_sh(dtype):
    d = synth.unit_dirs01(4096)
    assert (O.sh_encode(d, dtype) == R.sh(d, dtype)).all()


@pytest.mark.parametrize("const_dt,aabb", [(True, (0.0, 1.0)), (False, (-1.5, 2.5)), (False, (0.0, 1.0))])
def test_march_compact_composite(const_dt, aabb):
    xf, focal, meta = synth.camera_ring(8, radius=1.3)
    img, o, d, _ = synth.rays_from_cameras(xf, focal, meta, 64, 48, 512)
    d[0] = [0, 0, 1]; o[0] = [0.5, 0.5, -1.0]          # axis-aligned ray (inf idir components)
    o[1] = [5, 5, 5]; d[1] = [1, 0, 0]                  # misses the box
    bits = synth.shell_bitfield()
    cap = 512 * 1024
    ra, rb = O.PCG32(1337), R.PCG32(1337)
    ca, na, cnta, ia = O.march_rays(o, d, bits, aabb, ra, cap, const_dt=const_dt)
    cb, nb, cntb, ib = R.march(o, d, bits, aabb, rb.st, cap, meta, img, xf, const_dt=const_dt)
    assert (ra.st == rb.st).all()
    assert (na == nb).all() and (cnta == cntb).all() and (ia == ib).all()
    assert cnta[1] > 1000
    assert (ca == cb).all()                              # bit-exact records
    # overflow behaviour: capacity smaller than the demand
    small = int(cnta[1]) // 2
    ca2, na2, cnt2, _ = O.march_rays(o, d, bits, aabb, O.PCG32(1337), small, const_dt=const_dt)
    cb2, nb2, cntb2, _ = R.march(o, d, bits, aabb, R.PCG32(1337).st, small, meta, img, xf, const_dt=const_dt)
    assert (na2 == nb2).all() and (ca2 == cb2).all() and (cnt2 == cntb2).all() and (na2[:, 0] == 0).any()
    # compaction
    M = int(cnta[1])
    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float16):
        net = rng.normal(size=(M, 4)).astype(dtype)
        for ccap in (M + 100, M // 3):
            xa = O.compact_coords(ca[:M], na, ccap)
            xb = R.compact(net, cb[:M], nb, ccap, aabb)
            for u, v in zip(xa, xb):
                assert (u == v).all()
        cc, nc, _ = O.compact_coords(ca[:M], na, M // 3)
        netc = rng.normal(size=(M // 3, 4)).astype(dtype)
        bg = rng.random((512, 3), dtype=np.float32)
        fa = O.composite_fwd(netc, cc, na, nc, bg)
        fb = R.rgb_fwd(netc, cc, na, nc, bg, aabb)
        assert np.array_equal(fa, fb)
        G = rng.normal(size=(512, 3)).astype(np.float32)
        for mean in (0.001, 0.5):
            da = O.composite_bwd(netc, cc, nc, G, fa, mean)
            db = R.rgb_bwd(netc, cc, nc, G, fb, mean, aabb)
            assert np.array_equal(da, db)
        ia_, aa = O.composite_inference(net, ca[:M], na)
        ib_, ab = R.rgb_inference(net, cb[:M], nb, aabb)
        assert np.array_equal(ia_, ib_) and np.array_equal(aa, ab)


def test_march_compact_composite_seven_cascades():
    """NERF_CASCADES = 7 (density_grid_sampler.py:56-60: aabb_scale 64 raises the constant the generated prelude carries), box (-31.5, 32.5), cone stepping: the
    marcher's mip selection / MAX_CONE_STEPSIZE, compaction and all three compositing kernels against the reference built with that constant"""
    casc, aabb = 7, (-31.5, 32.5)
    xf, focal, meta = synth.camera_ring(8, radius=20.0)
    img, o, d, _ = synth.rays_from_cameras(xf, focal, meta, 64, 48, 384)
    d[0] = [0, 0, 1]; o[0] = [0.5, 0.5, -30.0]         # axis-aligned ray through every cascade
    o[1] = [50, 50, 50]; d[1] = [1, 0, 0]               # misses the box
    o[2] = [0.5, 0.5, 0.5]; d[2] = [0.6, 0.0, 0.8]      # starts inside the innermost cascade
    bits = synth.shell_bitfield(casc, radius=0.3) | synth.shell_bitfield(casc, radius=12.0, thickness=1.5) | synth.shell_bitfield(casc, radius=26.0, thickness=3.0)
    cap = 384 * 1024
    ra, rb = O.PCG32(1337), R.PCG32(1337)
    ca, na, cnta, ia = O.march_rays(o, d, bits, aabb, ra, cap, const_dt=False, cascades=casc)
    cb, nb, cntb, ib = R.march(o, d, bits, aabb, rb.st, cap, meta, img, xf, const_dt=False, cascades=casc)
    assert (ra.st == rb.st).all()
    assert (na == nb).all() and (cnta == cntb).all() and (ia == ib).all()
    M = int(cnta[1])
    assert M > 5000 and (ca == cb).all()
    # the outer cascades really are exercised: samples whose step exceeds what five cascades allow (MAX_CONE_STEPSIZE with NERF_CASCADES = 5 is sqrt(3)/8)
    assert (ca[:M, 3] > 1.73205080757 / 1024 * 16 * 1024 / 128 * 1.0001).any()
    # ... and differ from a five-cascade march of the same rays (otherwise the constant would not matter here)
    c5, n5, cnt5, _ = O.march_rays(o, d, bits[: 5 * 128 ** 3 // 8], aabb, O.PCG32(1337), cap, const_dt=False, cascades=5)
    assert int(cnt5[1]) != M
    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float16):
        net = rng.normal(size=(M, 4)).astype(dtype)
        for ccap in (M + 100, M // 3):
            xa = O.compact_coords(ca[:M], na, ccap)
            xb = R.compact(net, cb[:M], nb, ccap, aabb, cascades=casc)
            for u, v in zip(xa, xb):
                assert (u == v).all()
        cc, nc, _ = O.compact_coords(ca[:M], na, M // 3)
        netc = rng.normal(size=(M // 3, 4)).astype(dtype)
        bg = rng.random((384, 3), dtype=np.float32)
        fa = O.composite_fwd(netc, cc, na, nc, bg, cascades=casc)
        fb = R.rgb_fwd(netc, cc, na, nc, bg, aabb, cascades=casc)
        assert np.array_equal(fa, fb)
        G = rng.normal(size=(384, 3)).astype(np.float32)
        for mean in (0.001, 0.5):
            da = O.composite_bwd(netc, cc, nc, G, fa, mean, cascades=casc)
            db = R.rgb_bwd(netc, cc, nc, G, fb, mean, aabb, cascades=casc)
            assert np.array_equal(da, db)
        ia_, aa = O.composite_inference(net, ca[:M], na, cascades=casc)
        ib_, ab = R.rgb_inference(net, cb[:M], nb, aabb, cascades=casc)
        assert np.array_equal(ia_, ib_) and np.array_equal(aa, ab)


def test_density_grid_ops_seven_cascades():
    """the two density-grid kernels that read NERF_CASCADES (generate_grid_samples' cascade walk, update_bitfield's pooling chain) at 7"""
    casc = 7
    n_el = casc * 128 ** 3
    rng = np.random.default_rng(17)
    grid = (rng.random(n_el, dtype=np.float32) * 0.05).astype(np.float32)
    grid[rng.random(n_el) < 0.3] = -1.0
    ra, rb = O.PCG32(1337), R.PCG32(1337)
    pa, ia = O.grid_generate_samples(100000, ra, 5, (-31.5, 32.5), grid, casc, 0.01)
    pb, ib = R.grid_gen(100000, rb.st, 5, (-31.5, 32.5), grid, casc, 0.01, cascades=casc)
    assert np.array_equal(pa, pb) and np.array_equal(ia, ib) and (ra.st == rb.st).all() and int(ia.max()) >= 6 * 128 ** 3
    ba, ma = O.grid_update_bitfield(grid, casc)
    bb, mb = R.grid_bitfield(grid, casc)
    assert np.array_equal(ba, bb) and ma == mb and bb[6 * 128 ** 3 // 8:].any()


def test_density_grid_ops():
    xf, focal, meta = synth.camera_ring(6, radius=1.1)
    n_el = 5 * 128 ** 3
    ga = O.grid_mark_untrained(n_el, focal, xf, 64, 48)
    gb = R.grid_mark(n_el, focal, xf, 64, 48)
    assert np.array_equal(ga, gb) and (ga < 0).any() and (ga == 0).any()
    rng = np.random.default_rng(7)
    grid = np.where(ga < 0, ga, rng.random(n_el, dtype=np.float32) * 0.05).astype(np.float32)
    for n_casc, thresh, step in ((1, -0.01, 0), (3, 0.01, 7)):
        ra, rb = O.PCG32(1337), R.PCG32(1337)
        pa, ia = O.grid_generate_samples(100000, ra, step, (-1.5, 2.5), grid, n_casc, thresh)
        pb, ib = R.grid_gen(100000, rb.st, step, (-1.5, 2.5), grid, n_casc, thresh)
        assert np.array_equal(pa, pb) and np.array_equal(ia, ib) and (ra.st == rb.st).all()
    for dtype in (np.float32, np.float16):
        mlp = (rng.normal(size=100000) * 3).astype(dtype)
        ta = O.grid_splat_max(ia, mlp, np.zeros(n_el, np.float32))
        tb = R.grid_splat(ib, mlp, np.zeros(n_el, np.float32))
        assert np.array_equal(ta, tb)
    ea = O.grid_ema(grid.copy(), ta)
    eb = R.grid_ema(grid.copy(), tb)
    assert np.array_equal(ea, eb)
    ba, ma = O.grid_update_bitfield(ea)
    bb, mb = R.grid_bitfield(eb)
    assert ma[0] == mb[0] and np.array_equal(ba, bb) and ba.any()
