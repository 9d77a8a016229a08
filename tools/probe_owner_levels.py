"""Per-level time of the dense-level (owner-computes) part of the hash scatter on a lego-like batch: NGP_PROBE_SKIP_BINS=1 leaves only k_hash_bwd_owner +
k_reduce_dense, NGP_PROBE_LEVEL_MASK selects the levels.  Run through gpurun."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from jnerf_amd import ops
import synth
from oracle import oracle as O
import hip_impl as H

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
xf, focal, meta = synth.camera_ring(16, radius=1.3)
_, o, d, _ = synth.rays_from_cameras(xf, focal, meta, 200, 150, 24000, seed=21)
coords, ns, nsc, cnt = H.march_rays_compacted(o, d, synth.shell_bitfield(), (0.0, 1.0), O.PCG32(1337), 4096 * 1024, 1 << 18, const_dt=True)
k = int(cnt[3]); x = torch.from_numpy(np.ascontiguousarray(coords[:k, :3])).cuda()
table, offsets, n_params = ops.level_table(1)
dy = (torch.randn((16, k, 2), device="cuda") * 1e-3)
dy = dy.half() if dt == "f16" else dy
ws = torch.empty(ops.hash_bwd_workspace_bytes(table, k), dtype=torch.uint8, device="cuda")
g = torch.zeros(n_params, device="cuda")
def run():
    ops.hash_encode_bwd(x, dy, table, n_params, grad=g, layout=ops.LAYOUT_SOA, zero_first=True, workspace=ws)
def timeit(reps=10):
    run(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): run()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("samples", k, dt)
print("everything (bins + dense, two streams): %.1f us" % timeit())
os.environ["NGP_PROBE_SKIP_BINS"] = "1"
print("dense levels only: %.1f us" % timeit())
for l in range(16):
    if int(table[l, 1]) < (1 << 19):
        os.environ["NGP_PROBE_LEVEL_MASK"] = str(1 << l)
        print("  level %d (res %d, %d entries): %.1f us" % (l, table[l, 2], table[l, 1], timeit()))
os.environ["NGP_PROBE_LEVEL_MASK"] = "0"
print("  no level (launch + abs-max + reduce only): %.1f us" % timeit())

os.environ.pop("NGP_PROBE_LEVEL_MASK", None)
print("chunk sweeps (dense levels only, all levels together):")
for spec in ("10,10,5,5,5", "32,32,32,32,32", "32,32,16,8,4", "32,32,32,8,4", "32,16,8,4,2", "16,16,8,6,6", "32,32,16,10,8", "20,20,10,8,6", "32,32,32,16,8"):
    os.environ["NGP_PROBE_DENSE_CHUNKS"] = spec
    print("  chunks %-16s %.1f us" % (spec, timeit()))
