"""GPU micro-benchmark of the hash-grid scatter: per-level time, atomic scope (agent vs workgroup/L2-local) and width (2 x f32 vs packed f16),
uniform-random vs ray-coherent positions.  Run through gpurun; prints a table (kept under profiles/)."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import ops, _lib


def main():
    lib = _lib.lib()
    f = lib.ngp_x_probe_hash_bwd
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int]
    n = 1 << 18
    table, offsets, n_params = ops.level_table(4)
    tb = np.ascontiguousarray(table).ctypes.data_as(C.c_void_p)
    torch.manual_seed(0)
    uni = torch.rand((n, 3), device="cuda")
    # ray-coherent: 4096 rays x 64 consecutive steps of 1/1024
    o = torch.rand((4096, 1, 3), device="cuda") * 0.5 + 0.25
    d = torch.nn.functional.normalize(torch.randn((4096, 1, 3), device="cuda"), dim=-1)
    t = torch.arange(64, device="cuda").view(1, 64, 1) * (1.0 / 1024)
    ray = ((o + d * t).clamp(0, 1)).reshape(-1, 3).contiguous()
    # surface-concentrated, like a trained scene: 37k rays x 7 consecutive steps on a thin shell inside [0.4,0.6]^3
    nr = n // 7 + 1
    dirs = torch.nn.functional.normalize(torch.randn((nr, 1, 3), device="cuda"), dim=-1)
    base = 0.5 + dirs * 0.08
    tt = torch.arange(7, device="cuda").view(1, 7, 1) * 4e-4
    conc = (base + torch.nn.functional.normalize(torch.randn((nr, 1, 3), device="cuda"), dim=-1) * tt).reshape(-1, 3)[:n].contiguous()
    dy = (torch.randn((16, n, 2), device="cuda") * 1e-3).half()
    gf32 = torch.zeros(n_params, device="cuda")
    gf16 = torch.zeros(n_params, device="cuda", dtype=torch.float16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def timeit(fn, reps=5):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    print("level res size | uniform: agent_f32 wg_f32 agent_pk16 wg_pk16 | ray-coherent: agent_f32 wg_f32 agent_pk16 wg_pk16   (us per level, n=2^18)")
    tot = np.zeros(8)
    for l in range(16):
        row = []
        for pos in (uni, ray):
            for pk, g in ((0, gf32), (1, gf16)):
                for scope in (0, 1):
                    row.append(timeit(lambda: f(s, n, C.c_void_p(pos.data_ptr()), C.c_void_p(dy.data_ptr()), tb, C.c_void_p(g.data_ptr()), l, scope, pk)))
        # order: (uni,f32,agent) (uni,f32,wg) (uni,pk,agent) (uni,pk,wg) (ray,...)
        tot += np.array(row)
        print(f"{l:2d} {table[l,2]:5d} {table[l,1]:7d} | " + " ".join(f"{v:8.1f}" for v in row[:4]) + " | " + " ".join(f"{v:8.1f}" for v in row[4:]))
    print("sum              | " + " ".join(f"{v:8.1f}" for v in tot[:4]) + " | " + " ".join(f"{v:8.1f}" for v in tot[4:]))
    # forward for comparison
    g16 = (torch.rand(n_params, device="cuda") - 0.5).half()
    for name, pos in (("uniform", uni), ("ray", ray), ("concentrated", conc)):
        us = timeit(lambda: ops.hash_encode_fwd(pos, g16, table, layout=ops.LAYOUT_SOA))
        print(f"fwd fp16 SoA {name}: {us:.1f} us")
        gb = torch.zeros(n_params, device="cuda")
        us = timeit(lambda: ops.hash_encode_bwd(pos, dy, table, n_params, grad=gb, layout=ops.LAYOUT_SOA, zero_first=False))
        print(f"bwd fp16->f32 SoA {name} (product kernel, XCD map): {us:.1f} us")


def per_level():
    """time the owner-computes scatter one level at a time (NGP_PROBE_LEVEL_MASK is read at every call)"""
    n = 1 << 18
    table, offsets, n_params = ops.level_table(4)
    torch.manual_seed(0)
    nr = n // 7 + 1
    dirs = torch.nn.functional.normalize(torch.randn((nr, 1, 3), device="cuda"), dim=-1)
    tt = torch.arange(7, device="cuda").view(1, 7, 1) * 4e-4
    conc = (0.5 + dirs * 0.08 + torch.nn.functional.normalize(torch.randn((nr, 1, 3), device="cuda"), dim=-1) * tt).reshape(-1, 3)[:n].contiguous()
    uni = torch.rand((n, 3), device="cuda")
    dy = (torch.randn((16, n, 2), device="cuda") * 1e-3).half()
    gb = torch.zeros(n_params, device="cuda")
    ws = torch.empty(ops.hash_bwd_workspace_bytes(table, n, dy.dtype), dtype=torch.uint8, device="cuda")
    # (r5) the owner-computes scan these rows used to time level by level is gone: without a workspace the call takes the reference's scheme (global float atomics)
    for name, pos, w in (("uniform, atomics", uni, None), ("concentrated, atomics", conc, None), ("uniform, workspace", uni, ws), ("concentrated, workspace", conc, ws)):
        fn = lambda: ops.hash_encode_bwd(pos, dy, table, n_params, grad=gb, layout=ops.LAYOUT_SOA, zero_first=False, workspace=w)
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        print(f"hash backward [{name}]: {a.elapsed_time(b) / 5 * 1e3:.0f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "levels":
        per_level()
        sys.exit(0)
    main()
