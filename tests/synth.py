"""Seeded synthetic inputs shared by the parity tests, tests/golden/make_golden.py, smoke() and bench.py.
Everything is numpy + PCG32-free (np.random.default_rng) so it is reproducible anywhere."""
import numpy as np


def morton3D(x, y, z):
    def expand(v):
        v = v.astype(np.uint32)
        v = (v * np.uint32(0x00010001)) & np.uint32(0xFF0000FF)
        v = (v * np.uint32(0x00000101)) & np.uint32(0x0F00F00F)
        v = (v * np.uint32(0x00000011)) & np.uint32(0xC30C30C3)
        v = (v * np.uint32(0x00000005)) & np.uint32(0x49249249)
        return v
    return expand(x) | (expand(y) << np.uint32(1)) | (expand(z) << np.uint32(2))


def camera_ring(n_images, radius=1.3, W=64, H=48, fov_deg=50.0, seed=0):
    """n_images pinhole cameras on a sphere around (0.5,0.5,0.5) looking at the centre, in the ngp convention the
    sampler consumes: xforms [n,4,3] = column-major 3x4 (cols: right, down, forward, origin), focal [n,2] (pixels),
    metadata [n,11] = {k1,k2,p1,p2, cx/W, cy/H, fx, fy, light_dir[3]} (dataset.py:122-153)."""
    rng = np.random.default_rng(seed)
    xf = np.zeros((n_images, 4, 3), np.float32)
    for i in range(n_images):
        v = rng.normal(size=3)
        v /= np.linalg.norm(v)
        o = 0.5 + radius * v
        fwd = -v
        up = np.array([0.0, 0.0, 1.0]) if abs(v[2]) < 0.9 else np.array([1.0, 0.0, 0.0])
        right = np.cross(fwd, up); right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        xf[i, 0], xf[i, 1], xf[i, 2], xf[i, 3] = right, down, fwd, o
    f = 0.5 * W / np.tan(0.5 * np.deg2rad(fov_deg))
    focal = np.full((n_images, 2), f, np.float32)
    meta = np.zeros((n_images, 11), np.float32)
    meta[:, 4:6] = 0.5
    meta[:, 6:8] = focal
    return xf, focal, meta


def rays_from_cameras(xf, focal, meta, W, H, n_rays, seed=1):
    rng = np.random.default_rng(seed)
    n_img = xf.shape[0]
    idx = rng.integers(0, n_img * W * H, size=n_rays)
    img = (idx // (W * H)).astype(np.int32)
    off = idx % (W * H)
    x = ((off % W) + 0.5) / W
    y = ((off // W) + 0.5) / H
    dc = np.stack([(x - meta[img, 4]) * W / focal[img, 0], (y - meta[img, 5]) * H / focal[img, 1], np.ones_like(x)], -1)
    M = xf[img]  # [n,4,3]; cols 0..2 rotation
    d = M[:, 0] * dc[:, 0:1] + M[:, 1] * dc[:, 1:2] + M[:, 2] * dc[:, 2:3]
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return img, M[:, 3].astype(np.float32).copy(), d.astype(np.float32), idx


def shell_bitfield(cascades=5, radius=0.3, thickness=0.08, full_levels=False, seed=0):
    """Occupancy bitfield (morton order, 5 cascades x 128^3 bits): a spherical shell around the centre at every cascade,
    max-pooled upwards the way update_bitfield does conceptually (coarser cells that contain the shell are set)."""
    g = 128
    bits = np.zeros(cascades * g ** 3 // 8, np.uint8)
    ii = np.arange(g)
    X, Y, Z = np.meshgrid(ii, ii, ii, indexing="ij")
    m = morton3D(X.ravel(), Y.ravel(), Z.ravel())
    for c in range(cascades):
        scale = 2.0 ** c
        p = ((np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1) + 0.5) / g - 0.5) * scale + 0.5
        r = np.linalg.norm(p - 0.5, axis=-1)
        occ = np.abs(r - radius) < (thickness * 0.5 + 0.9 * scale / g)
        if full_levels:
            occ[:] = True
        cell = m[occ] + np.uint32(c * g ** 3)
        np.bitwise_or.at(bits, cell // 8, (np.uint8(1) << (cell % 8).astype(np.uint8)))
    return bits


def uniform_positions(n, seed=2):
    return np.random.default_rng(seed).random((n, 3), dtype=np.float32)


def unit_dirs01(n, seed=3):
    v = np.random.default_rng(seed).normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return ((v + 1) * 0.5).astype(np.float32)


def mlp_weights(seed=4):
    """packed (out,in) row-major weights, U(+-sqrt(3/fan_in)); density 3072, rgb 7168 with rows >= 3 of the last layer zero."""
    rng = np.random.default_rng(seed)
    def u(o, i):
        b = np.sqrt(3.0 / i)
        return rng.uniform(-b, b, size=(o, i)).astype(np.float32)
    wd = np.concatenate([u(64, 32).ravel(), u(16, 64).ravel()])
    v2 = np.zeros((16, 64), np.float32); v2[:3] = u(3, 64)
    wc = np.concatenate([u(64, 32).ravel(), u(64, 64).ravel(), v2.ravel()])
    return wd, wc


def table(n_params, dtype=np.float32, amp=1.0):
    """Closed-form pseudo-random hash table in (-amp/2, amp/2): integer-only generator, so it is reproducible bit-for-bit anywhere."""
    i = np.arange(n_params, dtype=np.uint64)
    h = (i * np.uint64(2654435761) + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    return (((h >> np.uint64(8)).astype(np.float64) / float(1 << 24) - 0.5) * amp).astype(dtype)
