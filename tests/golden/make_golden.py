"""Mints tests/golden/golden_v1.npz from oracle/_ref — the reference's OWN kernel headers
(/root/reference/python/jnerf/**/op_header/*.h) compiled for the host by oracle/ref_shim/Makefile.
Run in the build container (needs /root/reference):   python tests/golden/make_golden.py
The fixture holds inputs AND outputs, so checking against it needs neither /root/reference nor oracle/_ref."""
import hashlib
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import synth  # noqa: E402
from oracle import ref as R, oracle as O  # noqa: E402


def sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def main():
    assert R.build() and R.available()
    g = {}
    # ---- pcg32 (ops/op_include/pcg32/pcg32.h)
    r = R.PCG32(1337)
    g["pcg_uints"] = np.array([r.next_uint() for _ in range(32)], np.uint32)
    r.advance(8 * 4095); g["pcg_after_adv"] = np.array([r.next_float() for _ in range(4)], np.float32)
    r.advance(); g["pcg_state_end"] = r.st.copy()
    # ---- hash grid (HashEncode.h)
    rng = np.random.default_rng(11)
    x = synth.uniform_positions(512, seed=21)
    x[:6] = [[0, 0, 0], [1, 1, 1], [1, 0, 0.5], [0.5, 0.5, 0.5], [0.999999, 0.3, 0.7], [1e-7, 1, 0]]
    g["hash_x"] = x
    dy = (rng.normal(size=(512, 32)) * 1e-2).astype(np.float32)
    g["hash_dy"] = dy
    for s in (1, 4):
        table, offsets, n_params = O.level_table(s)
        g[f"hash_offsets_s{s}"] = offsets
        for dt, nm in ((np.float32, "f32"), (np.float16, "f16")):
            grid = synth.table(n_params, dt, amp=2.0)
            g[f"hash_fwd_s{s}_{nm}"] = R.hash_fwd(x, grid, offsets, s)
            grad = R.hash_bwd(x[:128], dy[:128].astype(dt), offsets, s, n_params)
            nz = np.flatnonzero(grad).astype(np.uint32)
            g[f"hash_bwd_idx_s{s}_{nm}"] = nz
            g[f"hash_bwd_val_s{s}_{nm}"] = grad[nz]
    # ---- SH (SphericalEncode.h)
    d = synth.unit_dirs01(512, seed=31)
    g["sh_d"] = d
    g["sh_f32"] = R.sh(d, np.float32)
    g["sh_f16"] = R.sh(d, np.float16)
    # ---- marcher / compaction / compositing (ray_sampler.h, compacted_coord.h, calc_rgb.h)
    xf, focal, meta = synth.camera_ring(8, radius=1.3)
    img, o, dd, _ = synth.rays_from_cameras(xf, focal, meta, 64, 48, 128)
    dd[0] = [0, 0, 1]; o[0] = [0.5, 0.5, -1.0]
    o[1] = [5, 5, 5]; dd[1] = [1, 0, 0]
    bits = synth.shell_bitfield()
    g["march_o"], g["march_d"], g["march_bits"] = o, dd, np.packbits(np.unpackbits(bits))  # identity; stored compressed
    for const_dt, aabb, nm in ((True, (0.0, 1.0), "lego"), (False, (-1.5, 2.5), "fox")):
        rs = R.PCG32(1337)
        coords, ns, cnt, ridx = R.march(o, dd, bits, aabb, rs.st, 128 * 1024, meta, img, xf, const_dt=const_dt)
        M = int(cnt[1])
        g[f"march_{nm}_numsteps"], g[f"march_{nm}_counters"], g[f"march_{nm}_coords"], g[f"march_{nm}_rayidx"] = ns, cnt, coords[:M], ridx
        g[f"march_{nm}_rng_end"] = rs.st.copy()
        cap = M * 2 // 3
        net_full = rng.normal(size=(M, 4)).astype(np.float32)
        cc, nc, ccnt = R.compact(net_full, coords[:M], ns, cap, aabb)
        g[f"compact_{nm}_numsteps"], g[f"compact_{nm}_counter"] = nc, ccnt
        net = rng.normal(size=(cap, 4)).astype(np.float32)
        bg = rng.random((128, 3), dtype=np.float32)
        G = rng.normal(size=(128, 3)).astype(np.float32)
        g[f"rgb_{nm}_net"], g[f"rgb_{nm}_bg"], g[f"rgb_{nm}_G"] = net, bg, G
        for dt, dn in ((np.float32, "f32"), (np.float16, "f16")):
            f = R.rgb_fwd(net.astype(dt), cc, ns, nc, bg, aabb)
            g[f"rgb_{nm}_fwd_{dn}"] = f
            g[f"rgb_{nm}_bwd_{dn}"] = R.rgb_bwd(net.astype(dt), cc, nc, G, f, 0.001, aabb)
            ri, ra = R.rgb_inference(net_full.astype(dt), coords[:M], ns, aabb)
            g[f"rgb_{nm}_inf_{dn}"], g[f"rgb_{nm}_alpha_{dn}"] = ri, ra
        g[f"rgb_{nm}_netfull"] = net_full
    # ---- density grid (mark_untrained…, generate_grid_samples…, splat…, ema…, update_bitfield.h)
    xf6, focal6, _ = synth.camera_ring(6, radius=1.1)
    n_el = 5 * 128 ** 3
    grid0 = R.grid_mark(n_el, focal6, xf6, 64, 48)
    g["grid_mark_sha"], g["grid_mark_nneg"] = sha(grid0), np.array([(grid0 < 0).sum()], np.int64)
    grid = np.where(grid0 < 0, grid0, synth.table(n_el, np.float32, amp=0.1) + 0.05).astype(np.float32)
    rs = R.PCG32(1337)
    pos, idx = R.grid_gen(4096, rs.st, 3, (-1.5, 2.5), grid, 3, 0.01)
    g["grid_gen_pos"], g["grid_gen_idx"], g["grid_gen_rng_end"] = pos, idx, rs.st.copy()
    mlp = (rng.normal(size=4096) * 3).astype(np.float32)
    g["grid_mlp"] = mlp
    tmp = R.grid_splat(idx, mlp, np.zeros(n_el, np.float32))
    ema = R.grid_ema(grid.copy(), tmp)
    g["grid_ema_sha"] = sha(ema)
    bf, mean = R.grid_bitfield(ema)
    g["grid_bitfield_sha"], g["grid_mean"] = sha(bf), mean
    out = os.path.join(HERE, "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB,", len(g), "arrays")


if __name__ == "__main__":
    main()
