"""VERDICT r4 item 2, second half: WHICH kernel stops being reproducible when another process shares the GPU?  Trains a small lego-configuration model for a few steps,
freezes one batch, then repeats every stage of the iteration on the SAME inputs `reps` times and counts the repetitions whose output differs from the first one - hash
forward, field forward, compositing (fused), field backward, slab reduction, hash backward (table gradient), marcher, Adam sweep (state restored before every repetition).
Run it alone and beside an unrelated process (tools/probe_repro_matrix.sh's hog); tools/_g.sh does both.   usage: python tools/probe_kernels_under_hog.py [reps]"""
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
torch.manual_seed(0)
ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=8, W=96, H=96, target_batch_size=1 << 18, n_rays_per_batch=4096, pipeline_sampling=False, scene="bricks")
r = Runner()
for i in range(40):
    r.train_step(i)
r.drain()
torch.cuda.synchronize()
s, m, enc = r.sampler, r.model, r.model.pos_encoder
n = s.target_batch_size
coords, numsteps, numsteps_c, n_valid, pos = s._coords, s._rays_numsteps, s._rays_numsteps_compacted, s._n_valid, s._pos_train
nr = numsteps.shape[0]
dirs = coords[:, 4:]
table = enc.table_for_kernels()
packed = m.packed_weights(refresh=True)
feat = m._feat_buffer(n)
out = torch.empty((n, 4), device="cuda"); dout = torch.empty((n, 4), device="cuda")
bg = torch.rand((nr, 3), device="cuda"); target = torch.rand((nr, 3), device="cuda")
rgb = torch.empty((nr, 3), device="cuda"); loss = torch.empty_like(rgb); lgrad = torch.empty_like(rgb)
dfeat, slabs = m._bwd_buffers(n)
wflat = torch.empty(10240, device="cuda")
grad = torch.empty(enc.n_params, device="cuda")
ws = torch.empty(ops.hash_bwd_workspace_bytes(enc.level_table, n, torch.float32), dtype=torch.uint8, device="cuda")
ds = r.dataset["train"]
ro = torch.rand((4096, 3), device="cuda") * 0.2 + torch.tensor([0.4, 0.4, -0.8], device="cuda")
rd = torch.nn.functional.normalize(torch.tensor([0.0, 0.0, 1.0], device="cuda") + (torch.rand((4096, 3), device="cuda") - 0.5) * 0.3, dim=-1)
mc = torch.empty((n, 7), device="cuda"); mns = torch.empty((4096, 2), dtype=torch.int32, device="cuda"); mnsc = torch.empty_like(mns); mcnt = torch.zeros(4, dtype=torch.int32, device="cuda")
mscr = torch.empty(ops.march_scratch_elems(4096), dtype=torch.int32, device="cuda"); mpos = torch.empty((n, 3), device="cuda")
rng0 = s.rng_state.copy()
P = enc.m_grid.detach().clone(); M = torch.rand_like(P) * 1e-3; V = torch.rand_like(P) * 1e-6


def st_hash_fwd():
    ops.hash_encode_fwd(pos, table, enc.level_table, out=feat, layout=ops.LAYOUT_SOA, n_valid=n_valid); return [feat]
def st_field_fwd():
    ops.field32_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, out=out, n_valid=n_valid, packed=packed); return [out]
def st_composite():
    ops.composite_train(out, coords, numsteps, numsteps_c, bg, target, 0.1, s.density_grid_mean, s.NERF_CASCADES, out=rgb, loss=loss, grad=lgrad, dout=dout, n_elems=n); return [rgb, lgrad, dout]
def st_field_bwd():
    ops.field32_bwd(feat, dirs, None, None, dout, layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, n_valid=n_valid, packed=packed); return [dfeat, slabs]
def st_reduce():
    ops.reduce_slabs(slabs, out=wflat, accumulate=False); return [wflat]
def st_hash_bwd():
    ops.hash_encode_bwd(pos, dfeat, enc.level_table, enc.n_params, grad=grad, layout=ops.LAYOUT_SOA, zero_first=True, n_valid=n_valid, workspace=ws); return [grad]
def st_march():
    rng = rng0.copy()
    ops.march_rays_compacted(ro, rd, s.density_grid_bitfield, s.aabb_range, rng, s.max_samples, n, s.cone_angle_constant, s.near_distance, s.const_dt, s.NERF_CASCADES,
                             coords_out=mc, numsteps=mns, numsteps_c=mnsc, counters=mcnt, scratch=mscr, pos_out=mpos, occ_bounds=s.occupancy_bounds())
    k = int(mcnt[3].item())
    return [mns, mnsc, mcnt, mc[:k]]
def st_adam():
    p, mm, vv = P.clone(), M.clone(), V.clone()
    ops.adam_ema_step(p, grad, mm, vv, p, None, 0.01, 7, zero_grad=False)
    return [p, mm, vv]


stages = [("k_hash_fwd", st_hash_fwd), ("k_field32_fwd_split", st_field_fwd), ("k_composite_train", st_composite), ("k_field32_bwd_split", st_field_bwd), ("k_reduce_slabs", st_reduce),
          ("hash backward (runs2 + pairs + accumulate2)", st_hash_bwd), ("marcher (wave count + scans + write)", st_march), ("k_adam_ema", st_adam)]
only = os.environ.get("PROBE_STAGES")              # comma-separated substrings of the stage names: run only those
if only:
    stages = [st for st in stages if any(k in st[0] for k in only.split(","))]
t0 = time.time()
for name, fn in stages:
    try:
        ref = [t.clone() for t in fn()]
        torch.cuda.synchronize()
        bad, worst = 0, 0
        for _ in range(reps):
            cur = fn()
            k = sum(int((a.view(torch.int32) != b.view(torch.int32)).sum()) if a.dtype == torch.float32 else int((a != b).sum()) for a, b in zip(cur, ref))
            if k:
                bad += 1; worst = max(worst, k)
        print(f"{name}: {bad} of {reps} repetitions differ from the first (worst: {worst} elements of {sum(t.numel() for t in ref)})", flush=True)
    except Exception as e:
        print(f"{name}: probe failed: {e!r}"[:300], flush=True)
print(f"({time.time() - t0:.0f} s)")
