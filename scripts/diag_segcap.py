"""SegmentedCapture mechanics without a process group: islands in forward and inside a backward pass (engine thread)."""
import os, sys, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
faulthandler.enable(); faulthandler.dump_traceback_later(40, exit=True)
import torch
from dpig_amd import autograd as A
dev = torch.device("cuda:0")
log = []
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = x * 2
        A.eager_island(lambda: log.append("fwd island"))
        return y + 1
    @staticmethod
    def backward(ctx, g):
        a = g * 2
        if os.environ.get("BWD_ISLAND", "1") == "1":
            A.eager_island(lambda: log.append("bwd island"))
        return a + 0
st = torch.cuda.Stream(device=dev)
torch.cuda.set_stream(st)                 # everything on the stream the capture will use (as the trainers' warm-up does)
x = torch.ones(8, device=dev, requires_grad=True)
out = torch.zeros(8, device=dev)
def step():
    y = F.apply(x)
    (gx,) = torch.autograd.grad(y.sum(), x)
    out.copy_(y + gx)
    return out
step(); torch.cuda.synchronize(); print("eager", out[:2].tolist(), log); log.clear()
seg = A.SegmentedCapture(dev, stream=st)
print("capturing ..."); sys.stdout.flush()
seg.capture(step)
print("captured: segments", seg.segments, log); sys.stdout.flush(); log.clear()
out.zero_()
seg.replay(); torch.cuda.synchronize()
print("replayed", out[:2].tolist(), log)
