"""Does a pinned host->device `copy_(non_blocking=True)` return before earlier work on its stream has finished?
Enqueue ~50 ms of matmuls, then time the copy call on the host for a few sizes, on the busy stream and on a side
stream that first waits on an event recorded behind the matmuls."""
import time
import torch

dev = torch.device("cuda:0")
a = torch.randn(8192, 8192, device=dev)
side = torch.cuda.Stream(dev)


def busy():
    b = a
    for _ in range(12):
        b = b @ a * 1e-2
    return b


for nbytes in (448, 64 << 10, 1536 << 10, 9 << 20):
    src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    for mode in ("same-stream", "side-stream-after-event"):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        busy()
        t1 = time.perf_counter()
        if mode == "same-stream":
            dst.copy_(src, non_blocking=True)
        else:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dst.copy_(src, non_blocking=True)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%9d B  %-24s enqueue matmuls %.2f ms | copy call %.3f ms | drain %.2f ms" % (
            nbytes, mode, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3), flush=True)
