"""GPU-side cost of event records / cross-stream waits between two dependent kernels (run under `rocprofv3 --kernel-trace`, read with the snippet below):
a long kernel keeps the GPU busy while the host enqueues  A, n x op, B  so that the gap A.end -> B.start is pure GPU-side packet processing."""
import sys, torch
x = torch.zeros(1 << 10, device="cuda")
big = torch.randn(8192, 8192, device="cuda")
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
done = torch.cuda.Event()
with torch.cuda.stream(side):
    done.record(side)
torch.cuda.synchronize()
for kind in ("record", "wait_done_event", "record+wait"):
    for n in (0, 1, 2, 4, 8):
        torch.cuda.synchronize()
        big @ big                                  # ~50 ms of GPU work: everything below is queued behind it
        x.add_(1.0)                                # kernel A
        evs = [torch.cuda.Event() for _ in range(n)]
        for e in evs:
            if kind in ("record", "record+wait"):
                e.record(main)
            if kind in ("wait_done_event", "record+wait"):
                main.wait_event(done)
        x.mul_(2.0)                                # kernel B
torch.cuda.synchronize()
