"""-m gpu: the hash-grid backward's region path (round 4) at a batch LARGER than anything the training configurations use - more than 512 record regions per level, so the
accumulate kernel's gather walks its segment table in two blocks (csrc/hash_encode.hip: gather_flat's `w0` loop) and the run kernel fills 293 regions.

The one-block case (n <= 2^19 samples) is what every other test and the bench exercise.  First hardware run: the driver's round-4 end-of-round suite (GPUTEST_r04.json: XPASS) - the
xfail marker it carried until then is gone, a regression now fails the suite."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_hash_bwd_region_path_beyond_512_regions():
    from jnerf_amd import ops
    n = 600_000                                                     # 586 edge-record regions of 1024 samples, 293 run-record regions of 2048
    table, offsets, n_params = O.level_table(1)
    rng = np.random.default_rng(4)
    # ray-like positions: runs of consecutive samples along straight segments (the run kernel combines them), inside the unit cube
    n_rays = n // 40
    o = rng.random((n_rays, 3), dtype=np.float32) * 0.6 + 0.2
    d = rng.standard_normal((n_rays, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    t = (np.arange(40, dtype=np.float32) * np.float32(8.457e-4))[None, :, None]
    x = np.clip((o[:, None, :] + d[:, None, :] * t).reshape(-1, 3), 0.0, 0.999).astype(np.float32)[:n]
    dy = (rng.standard_normal((n, 32)) * 1e-3).astype(np.float32)
    ref = O.hash_encode_bwd(x, dy, table, n_params)
    tx, tdy = torch.from_numpy(x).cuda(), torch.from_numpy(np.ascontiguousarray(dy.reshape(n, 16, 2).transpose(1, 0, 2))).cuda()      # level-major pairs
    ws = torch.empty(ops.hash_bwd_workspace_bytes(table, n), dtype=torch.uint8, device="cuda")
    g = torch.full((n_params,), float("nan"), dtype=torch.float32, device="cuda")
    ops.hash_encode_bwd(tx, tdy, table, n_params, grad=g, layout=ops.LAYOUT_SOA, zero_first=True, workspace=ws)
    g2 = torch.full((n_params,), float("nan"), dtype=torch.float32, device="cuda")
    ops.hash_encode_bwd(tx, tdy, table, n_params, grad=g2, layout=ops.LAYOUT_SOA, zero_first=True, workspace=ws)
    assert torch.equal(g, g2)                                       # bit-reproducible
    out = g.cpu().numpy()
    assert np.isfinite(out).all()
    for l in range(16):                                             # the bound test_full_size_hash_fwd_bwd_vs_oracle_on_ray_coherent_samples holds the stage to, per level
        lo, hi = int(offsets[l]) * 2, int(offsets[l + 1]) * 2
        scale = np.abs(ref[lo:hi]).max()
        err = np.abs(out[lo:hi] - ref[lo:hi]).max()
        assert err <= 2e-5 * scale, (l, float(err), float(scale))   # (2e-5: the oracle's own serial fp32 sums are over 2.3x as many terms here)
