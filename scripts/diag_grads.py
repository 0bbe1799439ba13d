import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_model_gpu as T
dev = torch.device("cuda:0")
tr, gb, P, ob, OM = T._setup(dev)
import dpig_amd.tflib as lib
gnames = OM.g_var_names(P)
gl, aux = OM.stage1_g_loss(P, ob, hidden_num=T.HID, z_num=T.ZNUM)
gg = dict(zip(gnames, torch.autograd.grad(gl, [P.p[n] for n in gnames], allow_unused=True)))
# fp32 oracle as noise floor
P32 = OM.ParamStore(seed=11, dtype=torch.float32)
ob32 = OM.batch_to_torch({k: v.numpy() for k, v in ob.items()}, dtype=torch.float32)
gl32, _ = OM.stage1_g_loss(P32, ob32, hidden_num=T.HID, z_num=T.ZNUM)
gg32 = dict(zip(gnames, torch.autograd.grad(gl32, [P32.p[n] for n in gnames], allow_unused=True)))
out = tr.g_optim(gb)
print("loss", out["g_loss"].item(), gl.item(), gl32.item())
for n in gnames:
    if gg[n] is None: continue
    r = T._rel(lib._params[n]._dpig_grad, gg[n]); r32 = T._rel(gg32[n], gg[n])
    flag = " <<<" if r > 3 * r32 + 1e-5 else ""
    print("%-48s hip %.2e  cpu32 %.2e%s" % (n, r, r32, flag))
