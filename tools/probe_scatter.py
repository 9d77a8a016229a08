"""Hash-grid backward of the ngp_base.py (lego, fp32) configuration on a REAL training batch (of the product build, or of a -DNGP_PROBE_* timing build of csrc/hash_encode.hip): per-kernel
HIP-event times (csrc/prof.hip); the product path twice (bit-reproducibility).  Run through gpurun.
usage: python tools/probe_scatter.py [steps] [scene] [lego|fox]      (fox: the fp16 / aabb_scale 4 / cone-stepping configuration)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
from jnerf_amd.presets import ngp_cfg
from jnerf_amd.runner import Runner


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    scene = sys.argv[2] if len(sys.argv) > 2 else "bricks"
    fox = len(sys.argv) > 3 and sys.argv[3] == "fox"
    torch.manual_seed(0)
    if fox:
        ngp_cfg(fp16=True, aabb_scale=4, const_dt=False, n_images=50, W=400, H=400, device="cuda:0", scene=scene)
    else:
        ngp_cfg(fp16=False, aabb_scale=1, const_dt=True, n_images=100, W=800, H=800, device="cuda:0", scene=scene)
    r = Runner()
    with r.training_stream():
        for i in range(steps):
            r.train_step(i)
        r.drain()
    torch.cuda.synchronize()
    f = r._fast
    enc = f.enc
    n = f.n
    dfeat, _ = r.model._bwd_buffers(n)
    n_valid = f.s._n_valid
    nv = int(n_valid.item())
    print("config", "fox" if fox else "lego", "scene", scene, "n", n, "n_valid", nv, flush=True)
    pos = f.s._pos_train
    table = enc.level_table
    ws = torch.empty(ops.hash_bwd_workspace_bytes(table, n, dfeat.dtype), dtype=torch.uint8, device="cuda")
    print("workspace MB", ws.numel() / 2 ** 20)

    def run_once(g):
        ops.hash_encode_bwd(pos, dfeat, table, enc.n_params, grad=g, layout=ops.LAYOUT_SOA, zero_first=True, workspace=ws, n_valid=n_valid)

    def variant(name, env, ref=None, reps=20):
        for k, v in env.items():
            os.environ[k] = v
        g = torch.full((enc.n_params,), float("nan"), dtype=torch.float32, device="cuda")
        run_once(g); torch.cuda.synchronize()
        g2 = torch.full((enc.n_params,), float("nan"), dtype=torch.float32, device="cuda")
        run_once(g2); torch.cuda.synchronize()
        repro = bool(torch.equal(g, g2))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            run_once(g2)
        b.record(); torch.cuda.synchronize()
        total = a.elapsed_time(b) / reps * 1e3
        ops.prof_enable("*")
        for _ in range(reps):
            run_once(g2)
        torch.cuda.synchronize()
        ops.prof_enable("")
        pk = {k.split("<")[0].strip("( "): sum(v) / len(v) * 1e3 for k, v in ops.prof_read().items() if v}
        for k in env:
            os.environ.pop(k, None)
        diff = ""
        if ref is not None:
            d = (g - ref).abs()
            rel = d / (ref.abs() + 1e-7 * ref.abs().max())
            diff = f" | vs ref: max abs {d.max().item():.3e} (max |ref| {ref.abs().max().item():.3e}), max rel {rel.max().item():.3e}, nan {int(torch.isnan(g).sum())}"
        print(f"{name:58s} {total:7.1f} us  repro={repro} " + " ".join(f"{k}={v:.1f}" for k, v in sorted(pk.items())) + diff, flush=True)
        return g

    ref = variant("product path (fp32: record regions | fp16: per-corner lists)", {})
    # (r6) the timing probes - parts of a kernel skipped, results wrong by design - are compile-time builds now, not environment switches of the product binary:
    #   EXTRA=-DNGP_PROBE_SCATTER bash jnerf_amd/csrc/build.sh     record kernels without their record stores
    #   EXTRA=-DNGP_PROBE_ACC=1 (2)                                accumulate: records loaded but not processed (not loaded)
    # run this script once per build; PROBE_BUILD_LABEL names the build in the output
    print("build:", os.environ.get("PROBE_BUILD_LABEL", "product"))
    variant("product path again", {}, ref)

if __name__ == "__main__":
    main()
