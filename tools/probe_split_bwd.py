"""Times ngp_field32_bwd on a synthetic 2^18-sample batch under the variant switch of the environment (NGP_FIELD32_BWD) - and, for a library built with EXTRA=-DNGP_PROBE_SPLIT=1..4 bash jnerf_amd/csrc/build.sh (parts of the split backward compiled out, results wrong: r6 moved the probes out of the product binary), with that part missing: where the
split-operand backward's time goes.  One process per setting (the switches are read once); see tools/gpu_r3_u.sh.  Run through gpurun."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops


def main():
    n = 1 << 18
    torch.manual_seed(0)
    feat = (torch.randn(16, n, 2, device="cuda") * 0.3).contiguous()
    dirs = torch.rand(n, 3, device="cuda")
    wd = (torch.rand(3072, device="cuda") - 0.5) * 0.6
    wc = (torch.rand(7168, device="cuda") - 0.5) * 0.5
    dout = torch.randn(n, 4, device="cuda") * 1e-4
    packed = ops.field32_pack_weights(wd, wc)
    dfeat = torch.zeros_like(feat)
    slabs = torch.empty((ops.field32_bwd_slabs(n), 10240), dtype=torch.float32, device="cuda")
    fn = lambda: ops.field32_bwd(feat, dirs, None, None, dout, layout=ops.LAYOUT_SOA, dfeat=dfeat, slabs=slabs, packed=packed)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"NGP_FIELD32_BWD={os.environ.get('NGP_FIELD32_BWD', '-')} build={os.environ.get('PROBE_BUILD_LABEL', 'product')}: {a.elapsed_time(b) / 30 * 1e3:7.1f} us")


if __name__ == "__main__":
    main()
