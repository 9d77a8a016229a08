"""Is the split-operand forward deterministic?  Same inputs, repeated launches, interleaved with the weight-pack kernels; run through gpurun."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops
n = 1 << 18
torch.manual_seed(0)
feat = (torch.randn(16, n, 2, device="cuda") * 0.3).contiguous()
dirs = torch.rand(n, 3, device="cuda")
wd = (torch.rand(3072, device="cuda") - 0.5) * 0.6
wc = (torch.rand(7168, device="cuda") - 0.5) * 0.5
ref = None
bad = 0
for it in range(40):
    packed = ops.field32_pack_weights(wd, wc)
    out = ops.field32_fwd(feat, dirs, None, None, layout=ops.LAYOUT_SOA, packed=packed)
    den = ops.density32_fwd(feat, None, n, layout=ops.LAYOUT_SOA, packed=packed)
    raw = ops.field32_fwd(feat, dirs, wd, wc, layout=ops.LAYOUT_SOA)
    cur = (out.clone(), den.clone(), raw.clone(), packed.clone())
    if ref is None:
        ref = cur
    else:
        for k, (a, b) in enumerate(zip(ref, cur)):
            if not torch.equal(a, b):
                bad += 1
                print("iteration", it, "output", k, "differs:", int((a != b).sum()), "elements, max", float((a.float() - b.float()).abs().max()))
print("field32_fwd / density32_fwd / raw-weights path / packed buffer: %d mismatches over 40 repetitions" % bad, "NGP_FIELD32_FWD =", os.environ.get("NGP_FIELD32_FWD", "split (default)"))
