"""Where does the host spend its ~0.65 ms per training step?  Wall-clock time inside the ctypes call ngp_train_step (blocks only if the training stream's queue is full) and inside
the side-stream batch preparation (_make_batch: ray generation + marcher launches + event records), per step.   python tools/probe_host_block.py"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner
from jnerf_amd import _lib as L

ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=100, W=800, H=800, device="cuda:0")
r = Runner()
lib = L.lib()
t_native, t_batch = [], []
orig = lib.ngp_train_step
def timed_native(*a):
    t0 = time.perf_counter(); rc = orig(*a); t_native.append(time.perf_counter() - t0); return rc
class Wrap:
    def __getattr__(self, k):
        return timed_native if k == "ngp_train_step" else getattr(lib, k)
L._lib = Wrap()
mb = r._make_batch
def timed_batch(step):
    t0 = time.perf_counter(); b = mb(step); t_batch.append(time.perf_counter() - t0); return b
r._make_batch = timed_batch
with r.training_stream():
    for i in range(600):
        r.train_step(i)
    torch.cuda.synchronize(); t_native.clear(); t_batch.clear()
    t0 = time.perf_counter(); per = []
    for i in range(600, 920):
        ts = time.perf_counter(); r.train_step(i); per.append(time.perf_counter() - ts)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
q = lambda v: " ".join(f"{np.percentile(np.array(v) * 1e6, p):7.0f}" for p in (10, 50, 90, 99))
print(f"320 steps: host issued in {t_issue / 320 * 1e3:.3f} ms/step, GPU done at {t_all / 320 * 1e3:.3f} ms/step")
print(f"train_step total   us p10/50/90/99: {q(per)}   mean {np.mean(per) * 1e6:.0f}")
print(f"ngp_train_step call us p10/50/90/99: {q(t_native)}   mean {np.mean(t_native) * 1e6:.0f}  (n={len(t_native)})")
print(f"_make_batch         us p10/50/90/99: {q(t_batch)}   mean {np.mean(t_batch) * 1e6:.0f}  (n={len(t_batch)})")
