"""Times the hash-grid backward (ngp_hash_encode_bwd_ws) on a REAL training batch: trains the bench configuration for a few hundred steps, then replays the
last batch's positions and dL/dfeatures through the encoder's accumulate_grad under the probe switches of csrc/hash_encode.hip.  Run through gpurun."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0")
    r = Runner()
    for i in range(steps):
        r.train_step(i)
    r.drain(); torch.cuda.synchronize()
    f = r._fast
    enc = f.enc
    n = f.n
    dfeat, _, _ = r.model._bwd_buffers(n)
    n_valid = f.s._n_valid
    print("n_valid", int(n_valid.item()) if n_valid.numel() == 1 else n_valid)
    fn = lambda: enc.accumulate_grad(f.s._pos_train, dfeat, ops.LAYOUT_SOA, n_valid=n_valid)
    rows = []
    def run(name, **env):
        for k, v in env.items():
            os.environ[k] = v
        rows.append((name, timeit(fn)))
        for k in env:
            os.environ.pop(k, None)
        print(f"{rows[-1][0]:60s} {rows[-1][1]:8.1f} us", flush=True)
    run("full backward (production: every level through the bins, res <= 300 with run combining)")
    run("abs-max pass only", NGP_PROBE_SKIP_BINS="1")
    for res in (0, 64, 128, 200, 450, 4096):
        run(f"run combining for levels with res <= {res}", NGP_HASH_BWD_RUN_RES=str(res))
    for st in (1024, 2048, 4096, 6144, 8192):
        run(f"LDS staging of {st} run records per workgroup", NGP_HASH_BWD_RUN_STAGE=str(st))
    run("owner-computes scan for everything (no bins)", NGP_HASH_BWD_NO_BINS="1")


if __name__ == "__main__":
    main()
