"""What one dependent kernel node of a replayed hipGraph costs on this box: chains of N small launches on one stream,
captured once and replayed; per-node time = replay time / N.  Sizes: 1 element (pure launch floor), 2 MB and 32 MB
element-wise passes (the sizes of the split-K partial-sum passes of the small conv layers)."""
import time, torch
dev = torch.device("cuda:0")
N = 1000
def chain(numel, fork=1):
    xs = [torch.zeros(numel, device=dev) for _ in range(fork)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for x in xs: x.add_(1.0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(N):
                xs[i % fork].add_(1.0)
        for _ in range(3): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        R = 10
        for _ in range(R): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / R / N * 1e6
for numel, label in ((1, "1 element"), (512 * 1024, "2 MB r+w"), (8 * 1024 * 1024, "32 MB r+w")):
    print("%-10s : %.2f us per dependent node" % (label, chain(numel)))
