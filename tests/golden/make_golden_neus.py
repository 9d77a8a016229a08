"""Mints tests/golden/golden_neus_v1.npz: inputs and expected outputs for the NeuS kernels (csrc/neus.hip) and the second-order hash-encoder kernels
(ngp_hash_encode_bwd_input_bwd_dy / _bwd_grid).  Neither has a counterpart that could be RUN from the reference (its NeuS is Jittor Python, its dy_dx branch is disabled and
has no second-order code at all), so the expected values come from this repository's checkers - PARITY UNPINNED, stated in DESIGN.md section 2:
  * compositing: oracle/neus_oracle.py (per-ray / per-section numpy loops of renderer.py:216-252, fp64) for colour / weights / alpha; the gradients by fp64 torch autograd of
    the renderer's own torch expression of the same formulas (jnerf_amd/neus_renderer.py: neus_alpha, _transmittance_weights; safe_clip = straight-through clamp);
  * second-order hash terms: fp64 torch autograd through the pure-torch hash encoding of tests/test_neus_gpu.py (_hash_encode_ref), which tests/test_neus_cpu.py pins to the C
    oracle (itself bit-exact against the reference's kernel_grid incl. its dy_dx branch).
Run anywhere (CPU only):   python tests/golden/make_golden_neus.py"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import neus_oracle as NO, oracle as O  # noqa: E402
import tests.test_neus_gpu as TG  # noqa: E402


def main():
    g = {}
    rng = np.random.default_rng(2024)
    for tag, (B, n, n_out) in {"bg": (11, 128, 32), "nobg": (6, 70, None), "long": (3, 300, 100)}.items():
        ratio = {"bg": 0.4, "nobg": 1.0, "long": 0.0}[tag]
        ref = TG._inputs(rng, B, n, n_out)
        inv_s = torch.tensor(20.0 + 60.0 * rng.random(), dtype=torch.float64, requires_grad=True)
        leaves = [k for k in ("sdf", "cos", "color", "bg_alpha", "bg_color") if k in ref]
        for k in leaves:
            ref[k].requires_grad_(True)
        col, w, a, p, c = TG._torch_chain(ref, inv_s, ratio)
        gc, gw = torch.tensor(rng.normal(size=(B, 3))), torch.tensor(rng.normal(size=tuple(w.shape)) * 0.3)
        ((col * gc).sum() + (w * gw).sum()).backward()
        npin = {k: v.detach().numpy() for k, v in ref.items()}
        oc, ow, oa = NO.composite(npin["sdf"], npin["cos"], npin["dists"], float(inv_s.detach()), npin["color"], npin["inside"], npin.get("bg_alpha"), npin.get("bg_color"), ratio)
        assert np.allclose(oc, col.detach().numpy(), atol=1e-12) and np.allclose(ow, w.detach().numpy(), atol=1e-12)      # the two checkers agree with each other
        for k, v in npin.items():
            g[f"comp_{tag}_in_{k}"] = v.astype(np.float32)
        g[f"comp_{tag}_inv_s"] = np.float32(float(inv_s.detach()))
        g[f"comp_{tag}_ratio"] = np.float32(ratio)
        g[f"comp_{tag}_g_color"], g[f"comp_{tag}_g_weights"] = gc.numpy().astype(np.float32), gw.numpy().astype(np.float32)
        # expected values are recomputed from the fp32-ROUNDED inputs the kernel will see
        ref32 = {k: torch.tensor(v.astype(np.float32).astype(np.float64), requires_grad=(k in leaves)) for k, v in npin.items()}
        inv32 = torch.tensor(float(np.float32(float(inv_s.detach()))), dtype=torch.float64, requires_grad=True)
        col, w, a, p, c = TG._torch_chain(ref32, inv32, float(np.float32(ratio)))
        gc32, gw32 = torch.tensor(g[f"comp_{tag}_g_color"].astype(np.float64)), torch.tensor(g[f"comp_{tag}_g_weights"].astype(np.float64))
        ((col * gc32).sum() + (w * gw32).sum()).backward()
        for name, val in (("color", col), ("weights", w), ("alpha", a), ("p", p), ("c", c)):
            g[f"comp_{tag}_out_{name}"] = val.detach().numpy()
        for k in leaves:
            g[f"comp_{tag}_grad_{k}"] = ref32[k].grad.numpy()
        g[f"comp_{tag}_grad_inv_s"] = np.float64(float(inv32.grad))
    # second-order hash terms
    for aabb in (1, 4):
        lt, _, n_params = O.level_table(aabb)
        r2 = np.random.default_rng(100 + aabb)
        n = 128
        x = (r2.random((n, 3)) * 0.98 + 0.01).astype(np.float32)
        seed_table = 7 + aabb
        v = r2.normal(size=(n, 32)).astype(np.float32)
        u = r2.normal(size=(n, 3)).astype(np.float32)
        table = (np.random.default_rng(seed_table).normal(size=n_params) * 0.1).astype(np.float32)
        x64 = torch.tensor(x.astype(np.float64), requires_grad=True)
        t64 = torch.tensor(table.astype(np.float64), requires_grad=True)
        v64 = torch.tensor(v.astype(np.float64), requires_grad=True)
        y = TG._hash_encode_ref(x64, t64, lt)
        (gx,) = torch.autograd.grad(y, x64, v64, create_graph=True)
        (gx * torch.tensor(u.astype(np.float64))).sum().backward()
        tg = t64.grad.numpy()
        nz = np.flatnonzero(tg)
        g[f"hash2_s{aabb}_x"], g[f"hash2_s{aabb}_v"], g[f"hash2_s{aabb}_u"] = x, v, u
        g[f"hash2_s{aabb}_table_seed"] = np.int64(seed_table)            # the 12-13 M-entry table is regenerated from its seed, not stored
        g[f"hash2_s{aabb}_dLdx"] = gx.detach().numpy()
        g[f"hash2_s{aabb}_ddy"] = v64.grad.numpy()
        g[f"hash2_s{aabb}_grid_idx"], g[f"hash2_s{aabb}_grid_val"] = nz.astype(np.int64), tg[nz]     # sparse: ~n * 256 touched entries
    out = os.path.join(HERE, "golden_neus_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out) // 1024, "KiB;", len(g), "arrays")


if __name__ == "__main__":
    main()
