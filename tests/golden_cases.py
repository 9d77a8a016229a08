"""Checks of an implementation (oracle.oracle on CPU, tests/hip_impl on the GPU) against tests/golden/golden_v1.npz.
`exact=True` demands the bit-exactness the plain-C oracle achieves; the HIP path is held to bit-exactness for all integer /
index / sample-record work and to stated float tolerances where device math (exp, fp16 accumulation order, atomics order) differs."""
import hashlib
import os
import numpy as np
import synth

_G = None
CASES = ["case_pcg32", "case_hash", "case_hash_dydx", "case_sh", "case_march_lego", "case_march_fox", "case_grid", "case_hash_wide", "case_march_seven_cascades", "case_grid_seven_cascades"]


def load():
    global _G
    if _G is None:
        _G = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz")))
    return _G


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def close(a, b, atol, rtol=0.0, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - rtol * np.abs(b)
    assert a.shape == b.shape and np.isfinite(a).all() and err.max() <= atol, f"{what}: max err {np.abs(a - b).max():.3e} (atol {atol}, rtol {rtol})"


def case_pcg32(I, g, exact):
    from oracle import oracle as O   # PCG32 host helper (the HIP path uses it only through march/generate kernels)
    r = O.PCG32(1337)
    assert np.array_equal(np.array([r.next_uint() for _ in range(32)], np.uint32), g["pcg_uints"])
    r.advance(8 * 4095)
    assert np.array_equal(np.array([r.next_float() for _ in range(4)], np.float32), g["pcg_after_adv"])
    r.advance()
    assert np.array_equal(r.st, g["pcg_state_end"])


def case_hash(I, g, exact):
    x, dy = g["hash_x"], g["hash_dy"]
    for s in (1, 4):
        table, offsets, n_params = I.level_table(s)
        assert np.array_equal(offsets, g[f"hash_offsets_s{s}"])
        for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
            grid = synth.table(n_params, dt, amp=2.0)
            out = I.hash_encode_fwd(x, grid, table)
            ref = g[f"hash_fwd_s{s}_{nm}"]
            # the reference evaluates the level scale with exp2f; ours is the correctly rounded value => <=1 ulp of scale
            close(out, ref, atol=4e-3 if dt == np.float16 else 3e-5, what=f"hash fwd s{s} {nm}")
            grad = I.hash_encode_bwd(x[:128], dy[:128].astype(dt), table, n_params)
            idx, val = g[f"hash_bwd_idx_s{s}_{nm}"], g[f"hash_bwd_val_s{s}_{nm}"]
            nz = np.flatnonzero(grad)
            assert len(np.setxor1d(nz, idx)) <= max(4, len(idx) // 2000), "scatter touched different table entries"
            close(grad[idx], val, atol=2e-4 if dt == np.float16 else 2e-7, rtol=2e-2 if dt == np.float16 else 1e-4, what=f"hash bwd s{s} {nm}")


_GD = None


def case_hash_dydx(I, g, exact):
    """kernel_grid with its dy_dx output enabled (HashEncode.h:205-251) against golden_dydx_v1.npz (minted from oracle/_ref by tests/golden/make_golden_dydx.py): the derivative
    rows are sums of four fp32 products in a fixed order - bit-exact for every implementation"""
    global _GD
    if _GD is None:
        _GD = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_dydx_v1.npz")))
    x = _GD["x"]
    for s in (1, 4):
        table, offsets, n_params = I.level_table(s)
        for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
            grid = synth.table(n_params, dt, amp=2.0)
            out, dydx = I.hash_encode_fwd_dydx(x, grid, table)
            close(out, _GD[f"out_s{s}_{nm}"], atol=4e-3 if dt == np.float16 else 3e-5, what=f"hash fwd (dydx call) s{s} {nm}")
            assert dydx.shape == (x.shape[0], 3, 32) and dydx.dtype == np.float32
            assert np.array_equal(dydx, _GD[f"dydx_s{s}_{nm}"]), f"dy_dx s{s} {nm}: max |diff| {np.abs(dydx - _GD[f'dydx_s{s}_{nm}']).max():.3e}"
            # and it IS the derivative: central differences of the forward along each axis, in the interior of a cell (fp32 table, coarse levels: cells much wider than the step)
            if dt == np.float32 and s == 1:
                h = np.float32(2e-4)
                xi = x[6:70].copy()
                for d in range(3):
                    e = np.zeros(3, np.float32); e[d] = h
                    fp, fm = I.hash_encode_fwd(xi + e, grid, table).astype(np.float64), I.hash_encode_fwd(xi - e, grid, table).astype(np.float64)
                    fd = (fp - fm) / (2.0 * float(h))
                    an = dydx[6:70, d, :8].astype(np.float64)                 # levels 0..3 (resolution <= 43: a 2e-4 step rarely crosses a cell face)
                    ok = np.abs(fd[:, :8] - an) <= 2e-2 * np.abs(an).max() + 1e-3
                    assert ok.mean() > 0.97, (d, ok.mean())


def case_sh(I, g, exact):
    for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
        out = I.sh_encode(g["sh_d"], dt)
        if exact:
            assert np.array_equal(out, g[f"sh_{nm}"])
        else:
            close(out, g[f"sh_{nm}"], atol=1e-6 if dt == np.float32 else 2e-3, what="sh")


def _march(I, g, nm, const_dt, aabb, exact):
    from oracle import oracle as O
    o, d, bits = g["march_o"], g["march_d"], g["march_bits"]
    rng = O.PCG32(1337)
    coords, ns, cnt, ridx = I.march_rays(o, d, bits, aabb, rng, 128 * 1024, const_dt=const_dt)
    M = int(cnt[1])
    assert np.array_equal(ns, g[f"march_{nm}_numsteps"]) and np.array_equal(cnt, g[f"march_{nm}_counters"])
    assert np.array_equal(rng.st, g[f"march_{nm}_rng_end"])
    assert np.array_equal(coords[:M], g[f"march_{nm}_coords"]), "sample records differ"       # bit-exact, HIP included
    assert np.array_equal(ridx, g[f"march_{nm}_rayidx"])
    assert not coords[M:].any()
    cap = M * 2 // 3
    cc, nc, ccnt = I.compact_coords(coords[:M], ns, cap)
    assert np.array_equal(nc, g[f"compact_{nm}_numsteps"]) and np.array_equal(ccnt, g[f"compact_{nm}_counter"])
    net, bg, G, netfull = g[f"rgb_{nm}_net"], g[f"rgb_{nm}_bg"], g[f"rgb_{nm}_G"], g[f"rgb_{nm}_netfull"]
    for dt, dn in ((np.float32, "f32"), (np.float16, "f16")):
        f = I.composite_fwd(net.astype(dt), cc, ns, nc, bg)
        b = I.composite_bwd(net.astype(dt), cc, nc, G, g[f"rgb_{nm}_fwd_{dn}"], 0.001)
        ri, ra = I.composite_inference(netfull.astype(dt), coords[:M], ns)
        if exact:
            assert np.array_equal(f, g[f"rgb_{nm}_fwd_{dn}"]) and np.array_equal(b, g[f"rgb_{nm}_bwd_{dn}"])
            assert np.array_equal(ri, g[f"rgb_{nm}_inf_{dn}"]) and np.array_equal(ra, g[f"rgb_{nm}_alpha_{dn}"])
        else:   # device __expf / expf differ from glibc in the last ulps
            close(f, g[f"rgb_{nm}_fwd_{dn}"], atol=2e-5, rtol=1e-5, what="composite fwd")
            close(b, g[f"rgb_{nm}_bwd_{dn}"], atol=2e-6 if dt == np.float32 else 2e-4, rtol=1e-4 if dt == np.float32 else 4e-3, what="composite bwd")
            close(ri, g[f"rgb_{nm}_inf_{dn}"], atol=2e-5, rtol=1e-5, what="composite inference")
            close(ra, g[f"rgb_{nm}_alpha_{dn}"], atol=2e-5, what="alpha")


def case_march_lego(I, g, exact):
    _march(I, g, "lego", True, (0.0, 1.0), exact)


def case_march_fox(I, g, exact):
    _march(I, g, "fox", False, (-1.5, 2.5), exact)


def case_grid(I, g, exact):
    from oracle import oracle as O
    xf6, focal6, _ = synth.camera_ring(6, radius=1.1)
    n_el = 5 * 128 ** 3
    grid0 = I.grid_mark_untrained(n_el, focal6, xf6, 64, 48)
    assert np.array_equal(sha(grid0), g["grid_mark_sha"]) and (grid0 < 0).sum() == g["grid_mark_nneg"][0]
    grid = np.where(grid0 < 0, grid0, synth.table(n_el, np.float32, amp=0.1) + 0.05).astype(np.float32)
    rng = O.PCG32(1337)
    pos, idx = I.grid_generate_samples(4096, rng, 3, (-1.5, 2.5), grid, 3, 0.01)
    assert np.array_equal(idx, g["grid_gen_idx"]) and np.array_equal(pos, g["grid_gen_pos"]) and np.array_equal(rng.st, g["grid_gen_rng_end"])
    tmp = I.grid_splat_max(idx, g["grid_mlp"], np.zeros(n_el, np.float32))
    ema = I.grid_ema(grid.copy(), tmp)
    bf, mean = I.grid_update_bitfield(ema)
    if exact:
        assert np.array_equal(sha(ema), g["grid_ema_sha"]) and np.array_equal(sha(bf), g["grid_bitfield_sha"]) and mean[0] == g["grid_mean"][0]
    else:   # __expf in the splat and the parallel reduction order of the mean
        ema_ref = O.grid_ema(grid.copy(), O.grid_splat_max(idx, g["grid_mlp"], np.zeros(n_el, np.float32)))
        close(ema, ema_ref, atol=1e-7, rtol=1e-5, what="grid ema")
        # the reference (and the serial oracle) add 2M terms in sequence in fp32; the wave-parallel sum is the more accurate one
        close(mean, g["grid_mean"], atol=0, rtol=1e-3, what="grid mean")
        exact_mean = np.maximum(ema[:128 ** 3].astype(np.float64), 0).sum() / 128 ** 3
        close(mean, [exact_mean], atol=0, rtol=1e-5, what="grid mean vs fp64")
        bf_ref, _ = O.grid_update_bitfield(ema)
        assert np.array_equal(bf, bf_ref)


_GW = None


def load_wide():
    """tests/golden/golden_wide_v1.npz (minted from oracle/_ref by tests/golden/make_golden_wide.py): aabb_scale 8 .. 128 and NERF_CASCADES = 7"""
    global _GW
    if _GW is None:
        _GW = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_wide_v1.npz")))
    return _GW


def case_hash_wide(I, g, exact):
    """hash encode at aabb_scale 8 / 32 / 64 / 128 (HashEncode.h:68-94: from 32 on the uint32 stride of grid_index wraps on the finest levels)"""
    g = load_wide()
    x, dy = g["hash_x"], g["hash_dy"]
    for s in (8, 32, 64, 128):
        table, offsets, n_params = I.level_table(s)
        assert np.array_equal(offsets, g[f"hash_offsets_s{s}"])
        for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
            grid = synth.table(n_params, dt, amp=2.0)
            out = I.hash_encode_fwd(x, grid, table)
            close(out, g[f"hash_fwd_s{s}_{nm}"], atol=4e-3 if dt == np.float16 else 3e-5, what=f"hash fwd s{s} {nm}")
            grad = I.hash_encode_bwd(x[:64], dy[:64].astype(dt), table, n_params)
            idx, val = g[f"hash_bwd_idx_s{s}_{nm}"], g[f"hash_bwd_val_s{s}_{nm}"]
            nz = np.flatnonzero(grad)
            assert len(np.setxor1d(nz, idx)) <= max(4, len(idx) // 2000), f"s{s} {nm}: scatter touched different table entries"
            close(grad[idx], val, atol=2e-4 if dt == np.float16 else 2e-7, rtol=2e-2 if dt == np.float16 else 1e-4, what=f"hash bwd s{s} {nm}")
        _, dydx = I.hash_encode_fwd_dydx(x[:64], synth.table(n_params, np.float32, amp=2.0), table)
        assert np.array_equal(dydx, g[f"hash_dydx_s{s}"]), f"dy_dx s{s}"


def case_march_seven_cascades(I, g, exact):
    """NERF_CASCADES = 7, box (-31.5, 32.5), cone stepping (density_grid_sampler.py:56-60; ray_sampler.h, compacted_coord.h, calc_rgb.h built with that constant)"""
    from oracle import oracle as O
    sys_path_golden()
    import make_golden_wide as MW
    g = load_wide()
    xf, focal, meta, img, o, d, bits = MW.c7_scene()
    assert np.array_equal(o, g["c7_o"]) and np.array_equal(d, g["c7_d"]) and np.array_equal(sha(bits), g["c7_bits_sha"]), "the scene generator changed: re-mint the fixture"
    rng = O.PCG32(1337)
    coords, ns, cnt, ridx = I.march_rays(o, d, bits, MW.AABB, rng, MW.N_RAYS * 1024, const_dt=False, cascades=MW.CASC)
    M = int(cnt[1])
    assert np.array_equal(ns, g["c7_numsteps"]) and np.array_equal(cnt, g["c7_counters"]) and np.array_equal(rng.st, g["c7_rng_end"])
    assert np.array_equal(coords[:M], g["c7_coords"]), "sample records differ"
    assert np.array_equal(ridx, g["c7_rayidx"]) and not coords[M:].any()
    cap = M * 2 // 3
    cc, nc, ccnt = I.compact_coords(coords[:M], ns, cap)
    assert np.array_equal(nc, g["c7_compact_numsteps"]) and np.array_equal(ccnt, g["c7_compact_counter"])
    net, bg, G, netfull = g["c7_net"], g["c7_bg"], g["c7_G"], g["c7_netfull"]
    for dt, dn in ((np.float32, "f32"), (np.float16, "f16")):
        f = I.composite_fwd(net.astype(dt), cc, ns, nc, bg, cascades=MW.CASC)
        b = I.composite_bwd(net.astype(dt), cc, nc, G, g[f"c7_fwd_{dn}"], 0.001, cascades=MW.CASC)
        ri, ra = I.composite_inference(netfull.astype(dt), coords[:M], ns, cascades=MW.CASC)
        if exact:
            assert np.array_equal(f, g[f"c7_fwd_{dn}"]) and np.array_equal(b, g[f"c7_bwd_{dn}"])
            assert np.array_equal(ri, g[f"c7_inf_{dn}"]) and np.array_equal(ra, g[f"c7_alpha_{dn}"])
        else:
            close(f, g[f"c7_fwd_{dn}"], atol=2e-5, rtol=1e-5, what="composite fwd")
            close(b, g[f"c7_bwd_{dn}"], atol=2e-6 if dt == np.float32 else 2e-4, rtol=1e-4 if dt == np.float32 else 4e-3, what="composite bwd")
            close(ri, g[f"c7_inf_{dn}"], atol=2e-5, rtol=1e-5, what="composite inference")
            close(ra, g[f"c7_alpha_{dn}"], atol=2e-5, what="alpha")


def case_grid_seven_cascades(I, g, exact):
    from oracle import oracle as O
    sys_path_golden()
    import make_golden_wide as MW
    g = load_wide()
    n_el = MW.CASC * 128 ** 3
    grid = MW.c7_grid()
    rng = O.PCG32(1337)
    pos, idx = I.grid_generate_samples(4096, rng, 3, MW.AABB, grid, MW.CASC, 0.01)
    assert np.array_equal(idx, g["c7_grid_gen_idx"]) and np.array_equal(pos, g["c7_grid_gen_pos"]) and np.array_equal(rng.st, g["c7_grid_gen_rng_end"])
    bf, mean = I.grid_update_bitfield(grid, MW.CASC)
    close(mean, g["c7_grid_mean"], atol=0, rtol=0 if exact else 1e-5, what="grid mean")
    assert mean[0] > 0.011                                # the threshold is min(0.01, mean) = 0.01 whatever the mean's last bit: the bitfield is exact for every implementation
    assert np.array_equal(sha(bf), g["c7_grid_bitfield_sha"])


def sys_path_golden():
    import sys
    p = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    if p not in sys.path:
        sys.path.insert(0, p)
