"""would Winograd F(4x4, 3x3) hold the golden bar?  A CPU study, no kernel: the oracle's full-width Market stage-I graph
(conv_hidden_num 128, bs 2: tests/golden/stage1_market_b2.npz) is evaluated in fp32 with its 3x3 stride-1 convs replaced by fp32 emulations of
the minimal-filtering forms -- transforms, the position GEMMs and the output transform all rounded to fp32 as a kernel would -- and every golden
activation is compared with the fp64 oracle's value (max |err| / max |ref|, the measure of tests/test_golden_gpu.py; bar 1e-4 for 'f32w').

    direct   every conv by fp32 direct summation (torch-CPU)                      = what the 'f32' kernels compute
    F2       F(2x2, 3x3) wherever the product's kernels have the form             = the headline's 'f32w' mode
    F4       F(4x4, 3x3) on maps >= 24 x 24 with H, W multiples of 4 (dec3 / dec4 / ROI b0-b1 / encoder levels 0-1), F(2x2, 3x3) on the rest
    F4all    F(4x4, 3x3) wherever H and W are multiples of 4

    python scripts/f43_numerics.py  ->  profiles/r06_f43_numerics.txt   (~10 min on 8 cores)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dpig_amd import synthetic  # noqa: E402
from oracle import models as OM  # noqa: E402
from oracle import ops as OO  # noqa: E402

F32 = torch.float32
MATS = {
    2: (torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64),
        torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64),
        torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)),
    4: (torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]],
                     dtype=torch.float64),
        torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]],
                     dtype=torch.float64),
        torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)),
}


def wino_conv(x, w, b, m):
    """conv3x3 SAME stride 1 by F(m x m, 3 x 3), every stage in fp32 (the filter image is made from the fp32 filter in fp32, as the product does)."""
    BT, G, AT = (t.to(F32) for t in MATS[m])
    a = m + 2
    N, H, W, C = x.shape
    K = w.shape[3]
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2), (1, 1, 1, 1)).permute(0, 2, 3, 1)
    U = torch.einsum("ar,rsck->asck", G, w)
    U = torch.einsum("bs,asck->abck", G, U).contiguous()
    d = xp.unfold(1, a, m).unfold(2, a, m)                     # N, th, tw, C, a, a
    V = torch.einsum("ai,ntwcij->ntwcaj", BT, d)
    V = torch.einsum("bj,ntwcaj->ntwabc", BT, V).contiguous()
    M = torch.einsum("ntwabc,abck->ntwabk", V, U)              # the 16 / 36 position GEMMs (fp32 products and sums)
    Y = torch.einsum("ia,ntwabk->ntwibk", AT, M)
    Y = torch.einsum("jb,ntwibk->ntwijk", AT, Y)               # N, th, tw, m, m, K
    y = Y.permute(0, 1, 3, 2, 4, 5).reshape(N, H, W, K)
    return y if b is None else y + b


def make_conv(policy):
    direct = OO.conv2d_same
    count = {"F2": 0, "F4": 0, "direct": 0}

    def conv(x, w, b=None, stride=1):
        kh, kw, C, K = w.shape
        N, H, W, _ = x.shape
        form = kh == 3 and kw == 3 and stride == 1 and C % 64 == 0 and K % 64 == 0 and H % 2 == 0 and W % 2 == 0
        if policy != "direct" and form:
            big = H % 4 == 0 and W % 4 == 0 and (policy == "F4all" or (policy == "F4" and min(H, W) >= 24))
            if big:
                count["F4"] += 1
                return wino_conv(x, w, b, 4)
            count["F2"] += 1
            return wino_conv(x, w, b, 2)
        count["direct"] += 1
        return direct(x, w, b, stride)
    return conv, count


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    gold = np.load(os.path.join(ROOT, "tests", "golden", "stage1_market_b2.npz"))
    bseed, pseed, _, B = [int(v) for v in gold["meta"]]
    ob = OM.batch_to_torch(synthetic.make_batch(B, seed=bseed), dtype=F32)
    P = OM.ParamStore(seed=pseed, dtype=F32)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import subsample
    rows, keys = {}, None
    orig = OO.conv2d_same
    for policy in ("direct", "F2", "F4", "F4all"):
        t0 = time.time()
        conv, count = make_conv(policy)
        OO.conv2d_same = conv
        try:
            taps = {}
            with torch.no_grad():
                embs, G = OM.stage1_forward(P, ob, taps=taps)
                d_fake = OM.dcgan_discriminator(P, G, "dcgan")
        finally:
            OO.conv2d_same = orig
        errs = {}
        for k, v in taps.items():
            ref = gold["tap/" + k]
            errs[k] = float(np.abs(subsample(v) - ref).max() / float(gold["tapstat/" + k][1]))
        errs["D(G)"] = float(np.abs(d_fake.numpy().astype(np.float64) - gold["d_fake"]).max() / max(np.abs(gold["d_fake"]).max(), 1e-12))
        rows[policy] = (errs, dict(count), time.time() - t0)
        keys = list(errs)
        print(policy, count, "%.0f s" % (time.time() - t0), flush=True)
    out = ["# F(4x4,3x3) numerics on the full-width golden model (scripts/f43_numerics.py; fp32 emulation on the CPU, errors = max|err| / max|ref| against the fp64 oracle)",
           "# layers by form: " + "; ".join("%s: %s" % (p, rows[p][1]) for p in rows),
           "%-10s " % "activation" + " ".join("%10s" % p for p in rows)]
    for k in keys:
        out.append("%-10s " % k + " ".join("%10.2e" % rows[p][0][k] for p in rows))
    out.append("%-10s " % "max" + " ".join("%10.2e" % max(rows[p][0].values()) for p in rows))
    txt = "\n".join(out)
    print(txt)
    open(os.path.join(ROOT, "profiles", "r06_f43_numerics.txt"), "w").write(txt + "\n")


if __name__ == "__main__":
    main()
