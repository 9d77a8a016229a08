"""Does an MFMA-bound field kernel overlap with an HBM/L2-bound hash / sweep kernel when the two run on different streams?  ("split the batch in two
halves on two streams so MFMA-bound field kernels overlap the L2/HBM-bound hash kernels".)  Measured before anything is built: every kernel alone at n and n/2 samples,
then pairs (field kernel on stream 1, memory kernel on stream 2, both at n/2) against the serial sum.  If 2 x T(pair at n/2) is well below T(A at n) + T(B at n) the
half-batch pipeline is worth building; if the pair costs its sum, it is not.  Synthetic ray-coherent positions (8192 rays x 32 samples).  Run through gpurun."""
import os
import sys
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops

DEV = "cuda"


def make(n, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rays = n // 32
    o = torch.rand(rays, 1, 3, generator=g) * 0.2 + 0.1
    d = torch.nn.functional.normalize(torch.rand(rays, 1, 3, generator=g) + 0.2, dim=-1)
    t = (torch.arange(32).float()[None, :, None] + torch.rand(rays, 1, 1, generator=g)) * (0.9 / 32 / 1.0)
    pos = (o + d * t).clamp(0.001, 0.999).reshape(-1, 3).contiguous().to(DEV)
    dirs = d.expand(rays, 32, 3).reshape(-1, 3).contiguous().to(DEV) * 0.5 + 0.5
    return pos, dirs


def main():
    n = 1 << 18
    lt, _, n_params = ops.level_table(1)
    table = (torch.rand(n_params, device=DEV) * 2e-4 - 1e-4)
    wd = (torch.rand(3072, device=DEV) - 0.5) * 0.3
    wc = (torch.rand(7168, device=DEV) - 0.5) * 0.3
    packed = ops.field32_pack_weights(wd, wc)
    grad, m, v = torch.zeros_like(table), torch.zeros_like(table), torch.zeros_like(table)
    sets = {}
    for size in (n, n // 2):
        pos, dirs = make(size)
        feat = ops.hash_encode_fwd(pos, table, lt, layout=ops.LAYOUT_SOA)
        dout = torch.randn(size, 4, device=DEV) * 1e-3
        dfeat = torch.zeros_like(feat)
        slabs = torch.empty((ops.field32_bwd_slabs(size), 10240), dtype=torch.float32, device=DEV)
        out = torch.empty((size, 4), dtype=torch.float32, device=DEV)
        ws = torch.empty(ops.hash_bwd_workspace_bytes(lt, size), dtype=torch.uint8, device=DEV)
        feat2 = torch.empty_like(feat)
        sets[size] = dict(pos=pos, dirs=dirs, feat=feat, dout=dout, dfeat=dfeat, slabs=slabs, out=out, ws=ws, feat2=feat2)

    def kernels(size):
        s = sets[size]
        return {
            "field32_bwd": lambda: ops.field32_bwd(s["feat"], s["dirs"], None, None, s["dout"], layout=ops.LAYOUT_SOA, dfeat=s["dfeat"], slabs=s["slabs"], packed=packed),
            "field32_fwd": lambda: ops.field32_fwd(s["feat"], s["dirs"], None, None, layout=ops.LAYOUT_SOA, out=s["out"], packed=packed),
            "hash_fwd": lambda: ops.hash_encode_fwd(s["pos"], table, lt, out=s["feat2"], layout=ops.LAYOUT_SOA),
            "hash_bwd": lambda: ops.hash_encode_bwd(s["pos"], s["dfeat"], lt, n_params, grad=grad, layout=ops.LAYOUT_SOA, zero_first=True, workspace=s["ws"]),
            "adam_sweep": lambda: ops.adam_ema_step(table, grad, m, v, None, None, 1e-9, 5, zero_grad=False),
        }

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 40

    def wall(fa, fb=None):
        for _ in range(3):
            with torch.cuda.stream(s1):
                fa()
            if fb:
                with torch.cuda.stream(s2):
                    fb()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            with torch.cuda.stream(s1):
                fa()
            if fb:
                with torch.cuda.stream(s2):
                    fb()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    full, half = kernels(n), kernels(n // 2)
    alone = {}
    for name in full:
        alone[name] = (wall(full[name]), wall(half[name]))
        print(f"{name:14s} alone: n {alone[name][0]:7.1f} us   n/2 {alone[name][1]:7.1f} us", flush=True)
    print()
    print(f"{'pair (both at n/2, two streams)':44s} {'pair':>8s} {'sum n/2':>8s} {'2 x pair':>9s} {'A(n)+B(n)':>10s}  gain")
    for a in ("field32_bwd", "field32_fwd"):
        for b in ("hash_fwd", "hash_bwd", "adam_sweep"):
            fb = half[b] if b != "adam_sweep" else full[b]              # the sweep is not split: it runs once per step beside one half's field kernel
            pair = wall(half[a], fb)
            sum_half = alone[a][1] + (alone[b][1] if b != "adam_sweep" else alone[b][0])
            serial = alone[a][0] + alone[b][0]
            twice = 2 * pair if b != "adam_sweep" else pair + alone[a][1]
            print(f"{a + ' || ' + b:44s} {pair:8.1f} {sum_half:8.1f} {twice:9.1f} {serial:10.1f}  {100 * (1 - twice / serial):5.1f} %", flush=True)


if __name__ == "__main__":
    main()
