"""Is the guard allocator itself sound?  Pure torch arithmetic on guarded memory against the CPU, with heavy allocate / free churn."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
mode = sys.argv[1]
if mode != "none":
    import conftest
    conftest.install_guard_allocator(mode)
import torch
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
bad = 0
for it in range(300):
    n = int(torch.randint(1, 70000, (1,), generator=g))
    a = torch.randn(n, generator=g)
    b = torch.randn(n, generator=g)
    ad, bd = a.to(dev), b.to(dev)
    r = (ad * bd + ad).relu()
    s = r.sum()
    cnt = int(torch.isnan(r).sum())
    ref = (a * b + a).relu()
    ok = torch.allclose(r.cpu(), ref) and abs(float(s) - float(ref.double().sum())) < 1e-3 * max(1.0, abs(float(ref.double().sum()))) and cnt == 0
    if not ok:
        bad += 1
        print("MISMATCH it %d n %d cnt %d sum %r ref %r" % (it, n, cnt, float(s), float(ref.sum())), flush=True)
    if it % 3 == 0:
        m = torch.randn(37, 53, generator=g)
        md = m.to(dev)
        mm = md @ md.t()
        if not torch.allclose(mm.cpu(), m @ m.t(), atol=1e-4):
            bad += 1
            print("MATMUL MISMATCH it %d" % it, flush=True)
print("SANITY bad=%d" % bad, flush=True)
