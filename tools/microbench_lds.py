"""LDS atomic throughput vs same-address conflict degree on MI355X (feeds the design of the owner-computes scatter)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import _lib
lib = _lib.lib()
f = lib.ngp_x_probe_lds_atomic
f.restype = C.c_int
f.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
out = torch.zeros(1024, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
blocks, iters = 256, 4096
for use_int in (0, 1, 2, 3):
    for distinct in (64, 32, 16, 8, 4, 2, 1):
        f(s, blocks, iters, distinct, C.c_void_p(out.data_ptr()), use_int); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(s, blocks, iters, distinct, C.c_void_p(out.data_ptr()), use_int); b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b)
        lane_ops = blocks * 1024 * iters
        cyc_per_wave_instr = ms * 1e-3 * 2.4e9 / (iters * 16)   # per CU: 16 waves x iters instructions
        print(f"{['f32', 'u32', 'u64', 'pk_f16'][use_int]} distinct={distinct:2d} ({64 // distinct:2d}-way): {ms:8.3f} ms  {lane_ops / ms / 1e6:9.1f} G lane-ops/s chip  ~{cyc_per_wave_instr:7.1f} cycles per wave-instruction per CU")
