"""Size-independent properties of the conv kernels at BASELINE configs[1] sizes (Market 128x64, B=16), where the
fp64 oracle is too slow to be the reference:

  * adjointness   <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)>   (the three kernels are one bilinear form),
  * linearity     conv(a x1 + x2) = a conv(x1) + conv(x2),
  * a spatially constant input gives the 9 border-class values of sum-over-valid-taps(w) (analytic),
  * bit-for-bit repeatability (no atomics anywhere on the path).

Dot products are accumulated in fp64 on the device; tolerances are fp32 round-off over the 1e7..1e8-term sums."""
import pytest
import torch

pytestmark = pytest.mark.gpu

# (name, N, H, W, C, K, k, stride, upsample)
LAYERS = [
    ("dec4 3x3", 16, 128, 64, 256, 256, 3, 1, False),
    ("enc down 3x3 s2", 16, 128, 64, 128, 256, 3, 2, False),
    ("roi tower 48x48", 112, 48, 48, 128, 128, 3, 1, False),
    ("dec up 1x1 on 2x-upsampled", 16, 64, 32, 512, 128, 1, 1, True),
    ("critic 5x5 s2", 16, 64, 32, 64, 128, 5, 2, False),
    ("image conv 256->3", 16, 128, 64, 256, 3, 3, 1, False),
    ("stem 3->128", 16, 128, 64, 3, 128, 3, 1, False),
]


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("layer", LAYERS, ids=[l[0] for l in LAYERS])
def test_adjoint_identities_and_linearity(dev, layer):
    import dpig_amd.hip_ops as H
    _, N, Hh, W, C, K, k, s, up = layer
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, Hh, W, C, device=dev, generator=g)
    x2 = torch.randn(N, Hh, W, C, device=dev, generator=g)
    w = torch.randn(k, k, C, K, device=dev, generator=g) * 0.05
    y = H.conv2d_fwd(x, w, None, stride=s, upsample2x=up)
    dy = torch.randn(y.shape, device=dev, generator=g)
    dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, upsample2x=up)
    dw = H.conv2d_wgrad(x, dy, (k, k, C, K), stride=s, upsample2x=up)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = (float((y.double() ** 2).sum()) * float((dy.double() ** 2).sum())) ** 0.5
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    # linearity in the input
    y12 = H.conv2d_fwd(0.5 * x + x2, w, None, stride=s, upsample2x=up)
    y2 = H.conv2d_fwd(x2, w, None, stride=s, upsample2x=up)
    err = float((y12 - (0.5 * y + y2)).abs().max())
    assert err <= 2e-5 * float(y12.abs().max()), err
    # repeatability
    assert torch.equal(H.conv2d_fwd(x, w, None, stride=s, upsample2x=up), y)
    assert torch.equal(H.conv2d_dgrad(dy, w, (N, Hh, W, C), stride=s, upsample2x=up), dx)
    assert torch.equal(H.conv2d_wgrad(x, dy, (k, k, C, K), stride=s, upsample2x=up), dw)


# the decoder 3x3 layers of SURVEY 8(d)'s target list at the FULL Market batch (B = 16): (name, H, W, C)
DECODER_LAYERS = [("dec4", 128, 64, 256), ("dec3", 64, 32, 512), ("dec2", 32, 16, 768), ("dec1", 16, 8, 1024), ("dec0", 8, 4, 768)]


@pytest.mark.parametrize("layer", DECODER_LAYERS, ids=[l[0] for l in DECODER_LAYERS])
def test_full_batch_decoder_layers_against_the_fp64_oracle(dev, layer):
    """B = 16 decoder convs (implicit GEMMs 131072x256x2304 ... 512x768x6912) against oracle.ops.conv2d_same evaluated in fp64 on the host,
    forward (+ bias + ReLU in the epilogue), dgrad and wgrad + bias gradient (the oracle's own autograd).  Operands are fp32-representable,
    so the only difference is the kernels' fp32 accumulation: bars 1e-4 of max|ref| (k <= 9216 terms), 2e-4 for the filter gradient
    (k = up to 131072 pixels)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    _, Hh, W, C = layer
    N = 16
    g = torch.Generator().manual_seed(Hh * 7 + C)
    x = (torch.rand((N, Hh, W, C), generator=g) * 2 - 1)
    w = (torch.rand((3, 3, C, C), generator=g) * 2 - 1) * (1.5 / (9 * C) ** 0.5)
    b = torch.rand((C,), generator=g) - 0.5
    dy = (torch.rand((N, Hh, W, C), generator=g) * 2 - 1)
    xd, wd, bd = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = O.conv2d_same(xd, wd, bd, 1)
    rdx, rdw, rdb = torch.autograd.grad(ref, [xd, wd, bd], dy.double())
    xg, wg, bg, dyg = x.to(dev), w.to(dev), b.to(dev), dy.to(dev)

    def close(got, want, tol):
        err = (got.double().cpu() - want).abs().max().item()
        assert err <= tol * want.abs().max().item(), (err, want.abs().max().item())
    close(H.conv2d_fwd(xg, wg, bg), ref.detach(), 1e-4)
    close(H.conv2d_fwd(xg, wg, bg, act=1), torch.relu(ref.detach()), 1e-4)
    close(H.conv2d_dgrad(dyg, wg, (N, Hh, W, C)), rdx, 1e-4)
    dw = torch.empty((3, 3, C, C), device=dev)
    db = torch.empty((C,), device=dev)
    H.conv2d_wgrad(xg, dyg, (3, 3, C, C), out=dw, db=db)
    close(dw, rdw, 2e-4)
    close(db, rdb, 2e-4)


def test_constant_input_gives_border_classes(dev):
    """SAME 3x3 conv of an all-ones image: every output pixel equals the sum of the filter taps that fall inside the
    image, i.e. one of 9 values per output channel (what the tiled-embedding collapse relies on)."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K = 16, 128, 64, 128, 128
    g = torch.Generator(device=dev).manual_seed(2)
    w = torch.randn(3, 3, C, K, device=dev, generator=g) * 0.05
    y = H.conv2d_fwd(torch.ones(N, Hh, W, C, device=dev), w, None)
    ws = w.double().sum(2)                                           # [3,3,K]
    rows = {0: ws[1:].sum(0), 1: ws.sum(0), 2: ws[:2].sum(0)}       # top / interior / bottom: valid filter rows
    for cy, yy in ((0, 0), (1, Hh // 2), (2, Hh - 1)):
        for cx, xx in ((0, 0), (1, W // 2), (2, W - 1)):
            r = rows[cy]
            want = (r[1:].sum(0), r.sum(0), r[:2].sum(0))[cx]
            got = y[:, yy, xx, :].double()
            assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    inner = y[:, 1:-1, 1:-1, :]
    assert float((inner - inner[:1, :1, :1, :]).abs().max()) <= 1e-5 * float(inner.abs().max())


def test_bf16_mode_adjointness_full_size(dev):
    """bf16 matrix-pipe mode at full size: fwd / dgrad / wgrad still describe one bilinear form on the ROUNDED operands:
    <conv(bf(x); bf(w)), bf(dy)> computed three ways (feeding already-rounded tensors makes the kernels' own rounding
    the identity)."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K = 16, 128, 64, 256, 256
    g = torch.Generator(device=dev).manual_seed(3)
    bf = lambda t: t.bfloat16().float()
    x, w = bf(torch.randn(N, Hh, W, C, device=dev, generator=g)), bf(torch.randn(3, 3, C, K, device=dev, generator=g) * 0.05)
    dy = bf(torch.randn(N, Hh, W, K, device=dev, generator=g))
    H.set_compute("bf16c")
    try:
        y = H.conv2d_fwd(x, w, None)
        dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C))
        dw = H.conv2d_wgrad(x, dy, (3, 3, C, K))
    finally:
        H.set_compute("f32")
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = (float((y.double() ** 2).sum()) * float((dy.double() ** 2).sum())) ** 0.5
    assert abs(a - b) <= 2e-6 * scale and abs(a - c) <= 2e-6 * scale, (a, b, c, scale)
    yf = H.conv2d_fwd(x, w, None)                    # fp32 pipe on the same (bf16-representable) operands: same products
    assert float((y - yf).abs().max()) <= 2e-5 * float(yf.abs().max())


# BASELINE configs[3] / [4] (DeepFashion 256x256, B=8) and configs[2] (Market stage-II, B=64: 448 ROI crops) layer shapes
BF16_LAYERS = [
    ("df dec4 3x3 256ch @256x256", 8, 256, 256, 256, 256, 3, 1, False),
    ("df E.res 3x3 128ch @256x256", 8, 256, 256, 128, 128, 3, 1, False),
    ("df dec3 3x3 512ch @128x128", 8, 128, 128, 512, 512, 3, 1, False),
    ("df roi tower N=56 64x64x128", 56, 64, 64, 128, 128, 3, 1, False),
    ("df roi down 3x3 s2", 56, 64, 64, 128, 256, 3, 2, False),
    ("df dec up 1x1 on 2x-upsampled", 8, 128, 128, 512, 256, 1, 1, True),
    ("df critic 5x5 s2 on the [x;G] pair", 16, 128, 128, 64, 128, 5, 2, False),
    ("stage-II roi tower N=448 48x48x128", 448, 48, 48, 128, 128, 3, 1, False),
    ("stage-II roi tower N=448 12x12x384", 448, 12, 12, 384, 384, 3, 1, False),
]


@pytest.mark.parametrize("layer", BF16_LAYERS, ids=[l[0] for l in BF16_LAYERS])
def test_bf16_storage_full_size_layers(dev, layer):
    """The bf16-STORAGE kernels (dpig_conv2d_*_bf16) at the sizes of BASELINE configs 2-4, where the fp64 oracle is too
    slow: products of bf16 numbers are exact in fp32 and both families accumulate in fp32, so on the same bf16-valued
    operands the bf16 kernels must reproduce the (oracle-verified) fp32 kernels -- the filter gradient to fp32 round-off,
    activations to the final rounding to bf16 -- and fwd / dgrad / wgrad must still be one bilinear form.  Bit-for-bit
    repeatable."""
    import dpig_amd.hip_ops as H
    _, N, Hh, W, C, K, k, s, up = layer
    BF = torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(4)
    xb = torch.randn(N, Hh, W, C, device=dev, generator=g).to(BF)
    w = (torch.randn(k, k, C, K, device=dev, generator=g) * 0.05).to(BF).float()      # bf16-representable master
    yb = H.conv2d_fwd(xb, w, None, stride=s, upsample2x=up)
    assert yb.dtype == BF
    dyb = torch.randn(yb.shape, device=dev, generator=g).to(BF)
    dxb = H.conv2d_dgrad(dyb, w, (N, Hh, W, C), stride=s, upsample2x=up)
    dw = H.conv2d_wgrad(xb, dyb, (k, k, C, K), stride=s, upsample2x=up)
    assert dxb.dtype == BF and dw.dtype == torch.float32
    x32, dy32 = xb.float(), dyb.float()
    y32 = H.conv2d_fwd(x32, w, None, stride=s, upsample2x=up)
    dx32 = H.conv2d_dgrad(dy32, w, (N, Hh, W, C), stride=s, upsample2x=up)
    dw32 = H.conv2d_wgrad(x32, dy32, (k, k, C, K), stride=s, upsample2x=up)

    def close_bf16(got, ref):
        err = (got.float() - ref).abs()
        bound = ref.abs() * 2.0 ** -8 + 2e-5 * float(ref.abs().max())
        assert not bool((err > bound).any()), float((err - bound).max())
    close_bf16(yb, y32)
    close_bf16(dxb, dx32)
    assert float((dw - dw32).abs().max()) <= 2e-5 * float(dw32.abs().max())
    a, b, c = _dot(yb, dyb), _dot(xb, dxb), _dot(w, dw)
    scale = (float((yb.double() ** 2).sum()) * float((dyb.double() ** 2).sum())) ** 0.5
    assert abs(a - c) <= 2e-5 * scale and abs(b - c) <= 2e-5 * scale, (a, b, c, scale)
    assert torch.equal(H.conv2d_fwd(xb, w, None, stride=s, upsample2x=up), yb)
    assert torch.equal(H.conv2d_dgrad(dyb, w, (N, Hh, W, C), stride=s, upsample2x=up), dxb)
    assert torch.equal(H.conv2d_wgrad(xb, dyb, (k, k, C, K), stride=s, upsample2x=up), dw)
