"""Read / write / copy rooflines of the chip and the cost of the record kernels' append pattern (csrc/probes.hip k_probe_stream).  Run through gpurun.
The append rows emulate k_bin_pairs' copy-out: W workgroups, each writing one fragment to every one of S lists (S = levels x bins [x sub-lists])."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jnerf_amd import _lib as L


def main():
    lib = L.lib()
    fn = lib.ngp_x_probe_stream
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p]
    dev = torch.device("cuda:0")
    nbytes = 1 << 30
    a = torch.zeros(nbytes // 4, dtype=torch.int32, device=dev)
    b = torch.zeros(nbytes // 4 * 2, dtype=torch.int32, device=dev)
    sink = torch.zeros(4, dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def run(mode, blocks, threads, n16, streams=0, frag=0, spacing=0, reps=10):
        args = (st, mode, blocks, threads, n16, a.data_ptr(), b.data_ptr(), streams, frag, spacing, sink.data_ptr())
        fn(*args); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(*args)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    for mb in (100, 400):
        n16 = mb * (1 << 20) // 16
        for mode, name in ((0, "read"), (1, "write"), (2, "copy")):
            for blocks, threads in ((2048, 256), (1024, 1024)):
                us = run(mode, blocks, threads, n16)
                moved = mb * (2 if mode == 2 else 1)
                print(f"{name:6s} {mb:4d} MB  grid {blocks}x{threads}: {us:8.1f} us  {moved * 1.048576 / us:6.2f} TB/s", flush=True)
    # append pattern: 100 MB in total = W workgroups x S lists x frag bytes
    total = 100 << 20
    for S, label in ((768, "6 levels x 128 bins"), (6144, "x 8 sub-lists"), (384, "6 x 64 bins")):
        for frag in (256, 512, 1024, 2048, 4096):
            W = total // (S * frag)
            for spacing_mul in (1, 4):
                spacing = W * frag * spacing_mul                    # lists back to back | 4x apart (the real layout keeps 4x head-room)
                if S * spacing > b.numel() * 4:
                    continue
                us = run(3, W, 1024, 0, S, frag, spacing)
                print(f"append {label:22s} S={S:5d} frag={frag:5d} B  W={W:5d}  spacing x{spacing_mul}: {us:8.1f} us  {total / 1e6 / us:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
