"""GPU parity of the HIP conv family (through the C ABI) against the CPU oracle (fp64).

Tolerance: the kernels are exact-fp32 fmaf chains (v_mfma_f32_32x32x2_f32); against an fp64
oracle the error is fp32 round-off ~1e-7*sum|a*b|.  We require max|err| <= 2e-5 * max|ref|
(north_star bar: 1e-3 relative per activation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-5

# (N, H, W, C, K, k, stride)
SHAPES = [
    (2, 16, 8, 32, 128, 3, 1),     # decoder-style 3x3 s1
    (2, 16, 8, 64, 128, 3, 2),     # encoder down conv (TF pad (0,1))
    (1, 12, 12, 128, 256, 3, 1),   # ragged M (144 rows), 2 n-tiles
    (2, 9, 7, 36, 40, 3, 1),       # odd sizes, partial tiles in every dim
    (2, 9, 7, 36, 40, 3, 2),       # odd input with stride 2 (pad (1,1))
    (2, 16, 8, 3, 64, 5, 2),       # D.1: Cin=3 scalar path, 5x5 s2 (pad (1,2))
    (2, 8, 4, 64, 128, 5, 2),      # D.2-style
    (2, 16, 8, 256, 3, 3, 1),      # G.out: Cout=3
    (3, 6, 6, 96, 132, 1, 1),      # 1x1
    (1, 8, 4, 370, 128, 3, 1),     # G.stem: Cin=370 (not a multiple of 4)
    (2, 16, 8, 18, 128, 3, 1),     # pose conv of the collapsed G.stem: flattened-(tap,ci) wgrad rows
    (2, 16, 8, 3, 128, 3, 1),      # E.stem
    (2, 8, 8, 64, 1, 1, 1),        # N = 1 (narrow tile)
    (2, 8, 8, 20, 24, 3, 2),       # thin both ways, stride 2
    (2, 5, 40, 64, 3, 3, 1),       # Cout=3 vector-ALU path: two strips per row, the second one partial
    (1, 3, 70, 20, 3, 3, 1),       # Cout=3, 5 active lanes, three strips
    (1, 9, 70, 3, 32, 3, 1),       # Cin=3 vector-ALU path, 8 lanes per pixel, odd width
    (2, 11, 37, 3, 16, 5, 2),      # Cin=3 5x5 s2, odd sizes (pad (1,2) / (2,2)), 4 lanes per pixel
]


def _rand(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1)


def _close(got, ref, tol=TOL):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("split_k", [0, 3])
def test_conv_fwd(dev, shape, split_k):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2) * 0.2
    b = _rand((K,), 3)
    ref = O.conv2d_same(x, w, b, s)
    got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), stride=s, split_k=split_k)
    _close(got, ref)


@pytest.mark.parametrize("act", [1, 2])
def test_conv_fwd_fused_epilogue(dev, act):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = 2, 16, 8, 64, 64, 3, 1
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2) * 0.2
    b = _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    pre = O.conv2d_same(x, w, b, s) + res
    ref = O.relu(pre) if act == 1 else O.leaky_relu(pre, 0.2)
    got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), stride=s, act=act, alpha=0.2,
                       residual=res.float().to(dev))
    _close(got, ref)


def test_conv_fwd_channel_slices(dev):
    """Inputs/outputs living in channel slices of wider buffers (free concat)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 8, 4, 64, 128
    xbig = _rand((N, Hh, W, 96), 1)
    w = _rand((3, 3, C, K), 2) * 0.2
    ref = O.conv2d_same(xbig[..., 32:96], w, None, 1)
    xg = xbig.float().to(dev)
    ybig = torch.full((N, Hh, W, 192), 7.0, device=dev)
    H.conv2d_fwd(xg[..., 32:96], w.float().to(dev), None, out=ybig[..., 64:192])
    _close(ybig[..., 64:192], ref)
    assert (ybig[..., :64] == 7.0).all()


def test_conv_upsample_fused(dev):
    """nearest-2x upsample followed by 1x1 conv + bias + relu == low-res conv, replicated."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 8, 4, 96, 64
    x = _rand((N, Hh, W, C), 1)
    w = _rand((1, 1, C, K), 2) * 0.3
    b = _rand((K,), 3)
    ref = O.relu(O.conv2d_same(O.upsample2x(x), w, b, 1))
    got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), act=1, upsample2x=True)
    _close(got, ref)
    # backward pieces
    dy = _rand(tuple(ref.shape), 5)
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    y = O.conv2d_same(O.upsample2x(xr), wr, None, 1)
    y.backward(dy)
    dx = H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), upsample2x=True)
    dw = H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (1, 1, C, K), upsample2x=True)
    _close(dx, xr.grad)
    _close(dw, wr.grad)


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_dgrad_wgrad(dev, shape):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1).requires_grad_(True)
    w = (_rand((k, k, C, K), 2) * 0.2).requires_grad_(True)
    y = O.conv2d_same(x, w, None, s)
    dy = _rand(tuple(y.shape), 3)
    y.backward(dy)
    dx = H.conv2d_dgrad(dy.float().to(dev), w.detach().float().to(dev), (N, Hh, W, C), stride=s)
    dw = H.conv2d_wgrad(x.detach().float().to(dev), dy.float().to(dev), (k, k, C, K), stride=s)
    _close(dx, x.grad)
    _close(dw, w.grad)


def test_conv_dgrad_fused_mask_accum(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = 2, 8, 8, 64, 128, 3, 2
    x = _rand((N, Hh, W, C), 1).requires_grad_(True)
    w = _rand((k, k, C, K), 2) * 0.2
    y = O.conv2d_same(x, w, None, s)
    dy = _rand(tuple(y.shape), 3)
    y.backward(dy)
    accum = _rand((N, Hh, W, C), 4)
    mask = _rand((N, Hh, W, C), 5)
    ref = (x.grad + accum) * torch.where(mask > 0, 1.0, 0.2)
    got = H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), stride=s,
                         accum=accum.float().to(dev), mask=mask.float().to(dev), act=2, alpha=0.2)
    _close(got, ref)


def test_conv_wgrad_split_and_beta(dev):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = 4, 16, 8, 64, 128, 3, 1
    x = _rand((N, Hh, W, C), 1)
    w = (_rand((k, k, C, K), 2) * 0.2).requires_grad_(True)
    y = O.conv2d_same(x, w, None, s)
    dy = _rand(tuple(y.shape), 3)
    y.backward(dy)
    base = _rand((k, k, C, K), 6)
    for split in (1, 5):
        out = base.float().to(dev).clone()
        H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (k, k, C, K), stride=s, out=out, beta=1.0,
                       split_k=split)
        _close(out, w.grad + base)


@pytest.mark.parametrize("shape", [(4, 16, 8, 64, 128, 3, 1), (2, 16, 8, 3, 64, 5, 2), (2, 16, 8, 256, 3, 3, 1),
                                   (2, 9, 7, 36, 40, 3, 2)])
def test_wgrad_fused_bias_gradient(dev, shape):
    """db = sum over pixels of dy comes out of the wgrad launch (split and unsplit, beta 0 / 1)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = (_rand((k, k, C, K), 2) * 0.2).requires_grad_(True)
    b = _rand((K,), 3).requires_grad_(True)
    y = O.conv2d_same(x, w, b, s)
    dy = _rand(tuple(y.shape), 4)
    y.backward(dy)
    base = _rand((K,), 5)
    for split in (0, 1, 3):
        db = base.float().to(dev).clone()
        dw = H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (k, k, C, K), stride=s, split_k=split,
                            out=torch.empty(k, k, C, K, device=dev), beta=0.0, db=db, db_beta=1.0)
        _close(dw, w.grad)
        _close(db, b.grad + base)


def test_conv_large_decoder_shape(dev):
    """dec4-sized layer (Market B=2): checks the many-tile path against the oracle."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 128, 64, 256, 256
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2) * 0.05
    ref = O.conv2d_same(x.float(), w.float(), None, 1)
    got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), None)
    _close(got, ref, tol=1e-4)


def test_bad_descriptor_raises(dev):
    import dpig_amd.hip_ops as H
    x = torch.zeros(1, 4, 4, 8, device=dev)
    w = torch.zeros(3, 3, 9, 8, device=dev)
    with pytest.raises(RuntimeError):
        H.conv2d_fwd(x, w)
    w7 = torch.zeros(7, 7, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        H.conv2d_fwd(x, w7)


def test_tiled_embedding_conv_collapse(dev):
    """G.stem (models.py:520-528): conv3x3(concat([tile(emb), pose])) == class-GEMM + thin pose conv,
    forward and every gradient, against the dense oracle on the materialised tensor."""
    import dpig_amd.autograd as A
    from oracle import ops as O
    B, Hh, W, E, P, K = 3, 10, 6, 44, 18, 32
    emb = _rand((B, E), 1).requires_grad_(True)
    pose = _rand((B, Hh, W, P), 2)
    w = (_rand((3, 3, E + P, K), 3) * 0.2).requires_grad_(True)
    b = _rand((K,), 4).requires_grad_(True)
    x = torch.cat([emb.reshape(B, 1, 1, E).expand(B, Hh, W, E), pose], dim=3)
    y = O.relu(O.conv2d_same(x, w, b, 1))
    dy = _rand(tuple(y.shape), 5)
    y.backward(dy)
    ge = emb.detach().float().to(dev).requires_grad_(True)
    gw = w.detach().float().to(dev).requires_grad_(True)
    gb = b.detach().float().to(dev).requires_grad_(True)
    gy = A.tiled_emb_conv(ge, pose.float().to(dev), gw, gb)
    _close(gy, y)
    gy.backward(dy.float().to(dev))
    _close(ge.grad, emb.grad, 5e-5)
    _close(gw.grad, w.grad, 5e-5)
    _close(gb.grad, b.grad, 5e-5)


def test_border_class_sum(dev):
    import dpig_amd.hip_ops as H
    a = _rand((2, 7, 5, 12), 1)
    ref = torch.zeros(2, 9, 12, dtype=torch.float64)
    for y in range(7):
        for x in range(5):
            cy = 0 if y == 0 else (2 if y == 6 else 1)
            cx = 0 if x == 0 else (2 if x == 4 else 1)
            ref[:, cy * 3 + cx] += a[:, y, x]
    _close(H.border_class_sum(a.float().to(dev)), ref)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_thin_cout3_epilogue_and_accumulate(dev, act):
    """The 3-output-channel image conv (vector-ALU kernels): fused bias + activation, wgrad with beta = 1 and the
    bias gradient, bitwise repeatable (no atomics)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 3, 7, 33, 128, 3
    x = _rand((N, Hh, W, C), 1)
    w = (_rand((3, 3, C, K), 2) * 0.2).requires_grad_(True)
    b = _rand((K,), 3).requires_grad_(True)
    z = O.conv2d_same(x, w, b, 1)
    ref = z if act == 0 else (torch.relu(z) if act == 1 else torch.where(z > 0, z, 0.2 * z))
    got = H.conv2d_fwd(x.float().to(dev), w.detach().float().to(dev), b.detach().float().to(dev), act=act, alpha=0.2)
    _close(got, ref)
    dy = _rand(tuple(z.shape), 4)
    z.backward(dy)
    base_w, base_b = _rand((3, 3, C, K), 5), _rand((K,), 6)
    outs = []
    for _ in range(2):
        dw = base_w.float().to(dev).clone(); db = base_b.float().to(dev).clone()
        H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (3, 3, C, K), out=dw, beta=1.0, db=db, db_beta=1.0)
        outs.append((dw, db))
    _close(outs[0][0], w.grad + base_w)
    _close(outs[0][1], b.grad + base_b)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


BF16_SHAPES = [(2, 16, 8, 32, 128, 3, 1), (2, 24, 24, 128, 128, 3, 1), (3, 3, 3, 64, 64, 3, 1), (2, 16, 8, 64, 128, 3, 2), (1, 12, 12, 128, 256, 3, 1), (2, 9, 7, 36, 40, 3, 1),
               (2, 8, 4, 64, 128, 5, 2), (3, 6, 6, 96, 132, 1, 1), (2, 9, 7, 36, 40, 3, 2), (1, 16, 8, 200, 64, 3, 1)]


def _bf(t):
    """Round to bfloat16 (nearest even) and return fp64: what the bf16 matrix pipe multiplies."""
    return t.float().bfloat16().double()


@pytest.mark.parametrize("shape", BF16_SHAPES)
@pytest.mark.parametrize("split_k", [0, 3])
def test_conv_bf16_compute(dev, shape, split_k):
    """DpigConvDesc.compute = BF16: operands rounded to bf16, exact products, fp32 accumulation -> equal to the fp64
    oracle on the ROUNDED operands to fp32-accumulation accuracy (fwd with bias + LeakyReLU, dgrad, wgrad).  A GEMM
    with at most 32 output columns is served by the fp32 pipe (then the unrounded oracle is the reference)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2) * 0.2
    b = _rand((K,), 3)
    dy_shape = tuple(O.conv2d_same(x, w, None, s).shape)
    dy = _rand(dy_shape, 4)
    H.set_compute("bf16c")
    try:
        # forward: N-dim = K
        xe, we = (_bf(x), _bf(w)) if K > 32 else (x, w)
        got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), stride=s, act=2, alpha=0.2,
                           split_k=split_k)
        _close(got, O.leaky_relu(O.conv2d_same(xe, we, b, s), 0.2))
        # dgrad multiplies dy with w: N-dim = C
        de, we = (_bf(dy), _bf(w)) if C > 32 else (dy, w)
        xr = x.clone().requires_grad_(True)
        (O.conv2d_same(xr, we, None, s) * de).sum().backward()
        dx = H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), stride=s, split_k=split_k)
        _close(dx, xr.grad)
        # wgrad multiplies x with dy (layers with > 32 output channels and >= 32 input channels); the bias gradient stays an exact
        # fp32 column sum of dy
        elig = K > 32 and not (C < 32 and k > 1)
        xe, de = (_bf(x), _bf(dy)) if elig else (x, dy)
        wr = w.clone().requires_grad_(True)
        (O.conv2d_same(xe, wr, None, s) * de).sum().backward()
        db = torch.zeros(K, device=dev)
        dw = H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (k, k, C, K), stride=s, split_k=split_k,
                            out=torch.empty(k, k, C, K, device=dev), beta=0.0, db=db, db_beta=0.0)
        _close(dw, wr.grad)
        _close(db, dy.reshape(-1, K).sum(0))
    finally:
        H.set_compute("f32")


def test_conv_upsample_fused_bf16(dev):
    """The upsample-commuted 1x1 conv (fwd with 2x2 replication, 4-tap dgrad, source-shift wgrad) on the bf16 pipe."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 8, 4, 96, 64
    x = _rand((N, Hh, W, C), 1)
    w = _rand((1, 1, C, K), 2) * 0.3
    b = _rand((K,), 3)
    xb, wb = _bf(x), _bf(w)
    ref = O.relu(O.conv2d_same(O.upsample2x(xb), wb, b, 1))
    dy = _rand(tuple(ref.shape), 5)
    dyb = _bf(dy)
    H.set_compute("bf16c")
    try:
        _close(H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), act=1, upsample2x=True), ref)
        xr = x.clone().requires_grad_(True)
        (O.conv2d_same(O.upsample2x(xr), wb, None, 1) * dyb).sum().backward()
        _close(H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), upsample2x=True), xr.grad)
        wr = w.clone().requires_grad_(True)
        (O.conv2d_same(O.upsample2x(xb), wr, None, 1) * dyb).sum().backward()
        _close(H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (1, 1, C, K), upsample2x=True), wr.grad)
    finally:
        H.set_compute("f32")


@pytest.mark.parametrize("shape", BF16_SHAPES)
@pytest.mark.parametrize("split_k", [0, 3])
def test_conv_split_bf16_compute(dev, shape, split_k):
    """DpigConvDesc.compute = BF16X3 (set_compute('bf16x3')): fp32 tensors, every operand split into two bf16 terms, three
    bf16 MFMAs per product block.  Held to the SAME bar as the exact fp32 kernels (TOL x max|ref| against the fp64 oracle
    on the UNROUNDED operands): forward with bias + LeakyReLU, dgrad, wgrad with the fused bias gradient."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2) * 0.2
    b = _rand((K,), 3)
    ref = O.conv2d_same(x, w, b, s)
    dy = _rand(tuple(ref.shape), 4)
    H.set_compute("bf16x3")
    try:
        assert H.get_compute() == "bf16x3"
        got = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), stride=s, act=2, alpha=0.2, split_k=split_k)
        _close(got, O.leaky_relu(ref, 0.2))
        xr = x.clone().requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        (O.conv2d_same(xr, wr, None, s) * dy).sum().backward()
        dx = H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), stride=s, split_k=split_k)
        _close(dx, xr.grad)
        db = torch.zeros(K, device=dev)
        dw = H.conv2d_wgrad(x.float().to(dev), dy.float().to(dev), (k, k, C, K), stride=s, split_k=split_k,
                            out=torch.empty(k, k, C, K, device=dev), beta=0.0, db=db, db_beta=0.0)
        _close(dw, wr.grad)
        _close(db, dy.reshape(-1, K).sum(0))
    finally:
        H.set_compute("f32")


def test_conv_split_bf16_epilogues_and_upsample(dev):
    """Split-bf16 pipe with the fused epilogues the model uses: residual add + ReLU with the pre-activation kept, masked dgrad
    accumulation, and the upsample-commuted 1x1 conv (2x2 replication fwd, 4-tap dgrad, source-shift wgrad)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 12, 10, 64, 64
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2) * 0.2
    b = _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    pre = O.conv2d_same(x, w, b, 1) + res
    dy = _rand(tuple(pre.shape), 5)
    acc = _rand((N, Hh, W, C), 6)
    H.set_compute("bf16x3")
    try:
        y = H.conv2d_fwd(x.float().to(dev), w.float().to(dev), b.float().to(dev), act=1, residual=res.float().to(dev))
        _close(y, O.relu(pre))
        xr = x.clone().requires_grad_(True)
        (O.conv2d_same(xr, w, None, 1) * dy).sum().backward()
        dx = H.conv2d_dgrad(dy.float().to(dev), w.float().to(dev), (N, Hh, W, C), accum=acc.float().to(dev))
        _close(dx, xr.grad + acc)
        Cu, Ku = 96, 64
        xu = _rand((N, 8, 4, Cu), 7)
        wu = _rand((1, 1, Cu, Ku), 8) * 0.3
        refu = O.relu(O.conv2d_same(O.upsample2x(xu), wu, b, 1))
        dyu = _rand(tuple(refu.shape), 9)
        _close(H.conv2d_fwd(xu.float().to(dev), wu.float().to(dev), b.float().to(dev), act=1, upsample2x=True), refu)
        xr = xu.clone().requires_grad_(True)
        wr = wu.clone().requires_grad_(True)
        (O.conv2d_same(O.upsample2x(xr), wr, None, 1) * dyu).sum().backward()
        _close(H.conv2d_dgrad(dyu.float().to(dev), wu.float().to(dev), (N, 8, 4, Cu), upsample2x=True), xr.grad)
        _close(H.conv2d_wgrad(xu.float().to(dev), dyu.float().to(dev), (1, 1, Cu, Ku), upsample2x=True), wr.grad)
    finally:
        H.set_compute("f32")


@pytest.mark.parametrize("shape", [(2, 24, 24, 128, 128, 3, 1), (2, 16, 8, 64, 128, 3, 2), (2, 8, 4, 64, 128, 5, 2), (3, 6, 6, 96, 136, 1, 1),
                                   (1, 16, 8, 200, 64, 3, 1), (2, 9, 7, 36, 40, 3, 1)])
def test_conv_split_bf16_filter_shadows(dev, shape):
    """dpig_conv2d_fwd_x3 / _dgrad_x3: the filter's hi / lo terms come from precomputed shadows by LDS-DMA instead of being
    split in the k-loop.  Same two roundings, same products, same order => bit-identical to the in-loop split, for the plain
    and split-K plans, the stride-2 dgrad classes and (36 channels: not a multiple of 8) the fallback to the fp32 filter."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1).float().to(dev)
    w = (_rand((k, k, C, K), 2) * 0.2).float().to(dev)
    b = _rand((K,), 3).float().to(dev)
    H.set_compute("bf16x3")
    planes_default, emit_default = H.X3_PLANES[0], H.X3_EMIT[0]
    H.X3_EMIT[0] = 2                                            # let every eligible epilogue leave its output's image
    try:
        for sk in (0, 3):
            y0 = H.conv2d_fwd(x, w, b, stride=s, act=2, alpha=0.2, split_k=sk)
            dy = torch.randn(y0.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
            dx0 = H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, split_k=sk)
            sh = H.FilterShadows([w], split=True)
            if C % 8 or K % 8:                                  # no 16-byte k-granules: no shadows, the fp32 filter is split in the loop
                assert not hasattr(w, "_dpig_shadow_x3")
            else:
                hi, lo = w._dpig_shadow_x3[0].float(), w._dpig_shadow_x3[2].float()
                assert torch.equal(hi, w.bfloat16().float()) and torch.equal(lo, (w - hi).bfloat16().float())
                assert torch.equal(w._dpig_shadow_x3[1].float(), hi.permute(0, 1, 3, 2)) and torch.equal(w._dpig_shadow_x3[3].float(), lo.permute(0, 1, 3, 2))
            for planes in (False, True):                        # filter shadows alone; + the activation's split32 image (both by DMA)
                H.X3_PLANES[0] = planes
                for t in (x, dy):
                    H.tag_s32(t, None)
                y1 = H.conv2d_fwd(x, w, b, stride=s, act=2, alpha=0.2, split_k=sk)
                dx1 = H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, split_k=sk)
                if planes and not (C % 8 or K % 8) and k > 1:
                    assert H.cached_s32(x) is not None          # the image was made (and is kept for the tensor's next consumer)
                    s32 = H.cached_s32(x).float()
                    xp = torch.nn.functional.pad(x, (0, s32.shape[3] * 32 - C)).reshape(N, Hh, W, -1, 32)
                    assert torch.equal(s32[..., :32], xp.bfloat16().float()) and torch.equal(s32[..., 32:], (xp - s32[..., :32]).bfloat16().float())
                assert torch.equal(y1, y0) and torch.equal(dx1, dx0), (sk, planes, float((y1 - y0).abs().max()), float((dx1 - dx0).abs().max()))
                dbw = torch.zeros(K, device=dev)
                dw1 = H.conv2d_wgrad(x, dy, (k, k, C, K), stride=s, split_k=sk, out=torch.empty(k, k, C, K, device=dev), beta=0.0,
                                     db=dbw, db_beta=0.0)
                if not planes:
                    dw0, db0 = dw1, dbw                        # the register path (fp32 tensors split in the loop)
                else:                                          # both operands from their images: dw bit for bit, db to dy's 16 bits
                    assert torch.equal(dw1, dw0), (sk, float((dw1 - dw0).abs().max()))
                    assert float((dbw - db0).abs().max()) <= 2e-5 * float(db0.abs().max())
                for t in (y1, dx1):                            # an image the epilogue left equals the one dpig_split32 makes
                    img = H.cached_s32(t)
                    if img is not None:
                        assert planes
                        H.tag_s32(t, None)
                        assert torch.equal(H.split32(t, 99), img)
                # an in-place write to the tensor invalidates its image (version counter): the next consumer re-splits
                if planes and H.cached_s32(x) is not None:
                    stale = H.cached_s32(x)
                    x.mul_(1.0)
                    assert H.cached_s32(x) is None and not hasattr(x, "_dpig_s32")
                    assert torch.equal(H.split32(x, 99), stale)
            H.X3_PLANES[0] = planes_default
            sh.detach()
    finally:
        H.X3_PLANES[0], H.X3_EMIT[0] = planes_default, emit_default
        H.set_compute("f32")


def test_conv_bf16_falls_back_to_fp32_when_ineligible(dev):
    import dpig_amd.hip_ops as H
    x = _rand((2, 16, 8, 256), 1).float().to(dev)
    w = (_rand((3, 3, 256, 3), 2) * 0.2).float().to(dev)
    ref = H.conv2d_fwd(x, w)
    H.set_compute("bf16c")
    try:
        got = H.conv2d_fwd(x, w)          # 3 output columns: the fp32 path serves it
    finally:
        H.set_compute("f32")
    assert torch.equal(got, ref)


def test_edge_cases_empty_huge_and_misaligned(dev):
    """Empty batches and single images beyond a launch's 2 GiB range are refused with a status (nothing is launched);
    operands that are not 16-byte aligned take the dword-load variants and still give the right answer.  (Batches beyond
    2 GiB are served in runs of whole images: test_batches_beyond_two_gib.)"""
    import ctypes
    import dpig_amd.hip_ops as H
    from dpig_amd._lib import lib
    from oracle import ops as O
    x = torch.zeros(1, 4, 4, 8, device=dev)
    w = torch.zeros(3, 3, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        H.conv2d_fwd(torch.zeros(0, 4, 4, 8, device=dev), w)
    # a descriptor whose every IMAGE is 4096 x 4096 x 64 floats (4 GiB), on a tiny allocation: refused before any launch
    d = H._desc(16, 4096, 4096, 64, 64, 3, 3, 1, 64, 64)
    y = torch.zeros(16, device=dev)
    rc = lib().dpig_conv2d_fwd(ctypes.byref(d), x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), None, None, 0, None)
    assert rc != 0 and b"exceeds" in lib().dpig_last_error()
    # misaligned views (offset by one float) of x, w and the output
    N, Hh, W, C, K = 2, 9, 7, 36, 40
    xs = torch.zeros(N * Hh * W * C + 1, device=dev); ws = torch.zeros(9 * C * K + 1, device=dev)
    xr, wr = _rand((N, Hh, W, C), 1), _rand((3, 3, C, K), 2) * 0.2
    xs[1:].copy_(xr.float().reshape(-1)); ws[1:].copy_(wr.float().reshape(-1))
    xv, wv = xs[1:].view(N, Hh, W, C), ws[1:].view(3, 3, C, K)
    assert xv.data_ptr() % 16 != 0 and wv.data_ptr() % 16 != 0
    ref = O.conv2d_same(xr, wr, None, 1)
    _close(H.conv2d_fwd(xv, wv), ref)
    dy = _rand(tuple(ref.shape), 3)
    xg = xr.clone().requires_grad_(True); wg = wr.clone().requires_grad_(True)
    O.conv2d_same(xg, wg, None, 1).backward(dy)
    _close(H.conv2d_dgrad(dy.float().to(dev), wv, (N, Hh, W, C)), xg.grad)
    _close(H.conv2d_wgrad(xv, dy.float().to(dev), (3, 3, C, K)), wg.grad)


@pytest.mark.parametrize("storage", ["f32", "bf16"])
def test_batches_beyond_two_gib(dev, storage):
    """A launch addresses each tensor through one buffer descriptor (< 2 GiB).  Bigger batches are cut into runs of whole
    images by the entry points themselves (forward / dgrad: independent runs; wgrad: runs accumulated in image order).
    x is 2.4 GiB here; every image of the big call must equal the same kernel run on that image alone, the filter gradient
    the sum over three sub-batches that each fit one launch, and the call is repeatable bit for bit."""
    import dpig_amd.hip_ops as H
    bf = storage == "bf16"
    N, Hh, W, C, K = (72, 128, 128, 1024, 64) if bf else (72, 128, 128, 512, 64)
    g = torch.Generator(device=dev).manual_seed(8)
    dt = torch.bfloat16 if bf else torch.float32
    x = torch.empty(N, Hh, W, C, device=dev, dtype=dt)
    for n in range(0, N, 8):                                   # (randn of 2.4 GiB in one piece would need a 2x temporary)
        x[n:n + 8] = torch.randn(min(8, N - n), Hh, W, C, device=dev, generator=g).to(dt)
    assert x.numel() * x.element_size() > 2 ** 31
    w = torch.randn(3, 3, C, K, device=dev, generator=g) * 0.05
    b = torch.randn(K, device=dev, generator=g)
    y = H.conv2d_fwd(x, w, b, act=1)
    assert torch.equal(H.conv2d_fwd(x, w, b, act=1), y)
    tol = (2.0 ** -7 if bf else 2e-5)
    for n in (0, 17, 62, 63, 71):                               # 63 images per launch here: both sides of the run boundary
        yn = H.conv2d_fwd(x[n:n + 1], w, b, act=1)
        assert float((y[n:n + 1].float() - yn.float()).abs().max()) <= tol * float(yn.float().abs().max()), n
    dy = torch.randn(N, Hh, W, K, device=dev, generator=g).to(dt)
    dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C))
    for n in (0, 62, 63, 71):
        dn = H.conv2d_dgrad(dy[n:n + 1], w, (1, Hh, W, C))
        assert float((dx[n:n + 1].float() - dn.float()).abs().max()) <= tol * float(dn.float().abs().max()), n
    del dx
    db = torch.zeros(K, device=dev)
    dw = H.conv2d_wgrad(x, dy, (3, 3, C, K), out=torch.empty(3, 3, C, K, device=dev), beta=0.0, db=db, db_beta=0.0)
    ref = torch.zeros_like(dw); refb = torch.zeros_like(db)
    for n0 in range(0, N, 24):
        dbp = torch.zeros(K, device=dev)
        ref += H.conv2d_wgrad(x[n0:n0 + 24], dy[n0:n0 + 24], (3, 3, C, K), out=torch.empty(3, 3, C, K, device=dev), beta=0.0, db=dbp, db_beta=0.0)
        refb += dbp
    assert float((dw - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float((db - refb).abs().max()) <= 2e-5 * float(refb.abs().max())
    dw2 = H.conv2d_wgrad(x, dy, (3, 3, C, K), out=torch.empty(3, 3, C, K, device=dev), beta=0.0)
    assert torch.equal(dw2, dw)


def test_round2_entry_points_refuse_bad_arguments(dev):
    """Status codes (never a launch) of the entry points added in round 2: statistics-carrying forward on a plan that cannot
    carry them, a statistics merge whose tile count does not cover the rows, split shadows with a lo plane inside the hi
    plane; and the shadow-fed x3 forward without shadows simply splits the fp32 filter in the loop."""
    import ctypes
    import dpig_amd.hip_ops as H
    from dpig_amd._lib import lib, ptr, stream_ptr
    x = torch.randn(1, 8, 8, 64, device=dev)
    w = torch.randn(3, 3, 64, 64, device=dev) * 0.1
    y = torch.empty(1, 8, 8, 64, device=dev)
    st = torch.empty(1, 2, 64, device=dev)
    d = H._desc(1, 8, 8, 64, 64, 3, 3, 1, 64, 64)
    assert lib().dpig_conv2d_bn_stats_tiles(ctypes.byref(d)) == 0            # 1 row tile: the heuristic plan is split-K
    assert lib().dpig_conv2d_fwd_stats(ctypes.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(st), stream_ptr()) != 0
    assert b"statistics" in lib().dpig_last_error()
    assert lib().dpig_conv2d_bn_stats_tiles_ws(ctypes.byref(d)) == 1         # ... with a workspace the split plan carries them
    assert lib().dpig_conv2d_fwd_stats_ws(ctypes.byref(d), ptr(x), ptr(w), None, ptr(y), ptr(st), None, 0, stream_ptr()) != 0   # no workspace
    d1 = H._desc(1, 8, 8, 64, 64, 3, 3, 1, 64, 64, split_k=1)
    assert lib().dpig_conv2d_bn_stats_tiles(ctypes.byref(d1)) == 1
    assert lib().dpig_conv2d_fwd_stats(ctypes.byref(d1), ptr(x), ptr(w), None, ptr(y), None, stream_ptr()) != 0
    m = torch.empty(64, device=dev)
    assert lib().dpig_bn_stats_finalize(ptr(st), 1, 64, 128, 64, 1e-5, ptr(m), ptr(m), stream_ptr()) == 0
    assert lib().dpig_bn_stats_finalize(ptr(st), 1, 200, 128, 64, 1e-5, ptr(m), ptr(m), stream_ptr()) != 0    # 2 tiles needed
    assert lib().dpig_bn_stats_finalize(ptr(st), 2, 64, 128, 64, 1e-5, ptr(m), ptr(m), stream_ptr()) != 0     # 1 tile too many
    sh = torch.empty(4 * w.numel(), dtype=torch.bfloat16, device=dev)
    assert lib().dpig_filter_shadow_split(ptr(w), ptr(sh), None, w.numel() - 8, 9, 64, 64, stream_ptr()) != 0  # planes overlap
    assert lib().dpig_filter_shadow_split(ptr(w), ptr(sh), None, w.numel(), 9, 64, 64, stream_ptr()) == 0
    H.set_compute("bf16x3")
    try:
        d2 = H._desc(1, 8, 8, 64, 64, 3, 3, 1, 64, 64)
        ref = H.conv2d_fwd(x, w)
        ws, wn = H._ws(d2, 0, dev)
        assert lib().dpig_conv2d_fwd_x3(ctypes.byref(d2), ptr(x), None, ptr(w), None, None, None, None, ptr(y), None, None, None,
                                        ptr(ws), wn, stream_ptr()) == 0
        assert torch.equal(y, ref)
    finally:
        H.set_compute("f32")
