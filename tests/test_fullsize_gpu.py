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


# ---- sampled fp64 oracle at FULL size (VERDICT r4 "missing" 3) ----------------------------------------------------------------
# Every kernel family that serves a DeepFashion / stage-II shape is compared with oracle.ops.conv2d_same -- not with another HIP
# kernel: the oracle is evaluated in fp64 on >= 4096 sampled output positions (forward), input positions (dgrad) and on sampled
# (tap, ci, co) filter elements with their FULL pixel sums (wgrad); oracle.ops.conv2d_same*_sampled are pinned to the dense
# conv2d_same and its autograd gradients by tests/test_oracle.py::test_sampled_conv_equals_dense.  Operands are bf16-representable,
# so ONE reference serves the fp32 kernels (bar: fp32 accumulation, 1e-4 / 2e-4 of max|ref| as for the Market decoder layers above)
# and the bf16-storage kernels (bar: one rounding of the result, 2^-8 |ref| + 5e-5 max|ref|; filter gradients are fp32).
SAMPLED_LAYERS = BF16_LAYERS + [
    ("df dec2 3x3 768ch @64x64 (384 tiles of 256x256)", 8, 64, 64, 768, 768, 3, 1, False),
    ("df dec1 3x3 1024ch @32x32", 8, 32, 32, 1024, 1024, 3, 1, False),
    ("df dec0 3x3 768ch @16x16", 8, 16, 16, 768, 768, 3, 1, False),
    ("df roi tower N=56 8x8x512", 56, 8, 8, 512, 512, 3, 1, False),
    ("df image conv 256->3 @256x256", 8, 256, 256, 256, 3, 3, 1, False),
    ("df stem 3->128 @256x256", 8, 256, 256, 3, 128, 3, 1, False),
    ("df critic stem 5x5 s2 3->64 on the [x;G] pair", 16, 256, 256, 3, 64, 5, 2, False),
    ("stage-II roi tower N=448 6x6x512", 448, 6, 6, 512, 512, 3, 1, False),
]
# (id, arithmetic, forward/dgrad tile family (mode, variant), wgrad tile family, eight-wave bits)
SAMPLED_VARIANTS = [
    ("f32", "f32", None, None, None),                        # gather_gemm_kernel / wgrad_kernel / thin3_* / fewc_* on fp32 tensors
    ("bf16-auto", "bf16", (1, 0), (1, 0), 3),                 # what the trainers run: bhq / bhq32 / bq / bh / bg8, bwq / bw8, thin bf16
    ("bf16-128tile", "bf16", (0, 0), (0, 0), 3),              # bh_kernel / bg8_kernel, bw8_kernel
    ("bf16-128tile-4wave", "bf16", (0, 0), (0, 0), 0),        # bh_kernel / bg_kernel, bw_kernel
    ("bf16-256x256", "bf16", (2, 1), (2, 1), 3),              # bhq_kernel where eligible else bq_kernel<2,4>; bwq_kernel<2,4>
    ("bf16-512x128", "bf16", (2, 2), (2, 2), 3),              # bhq32_kernel where eligible else bq_kernel<4,2>; bwq_kernel<4,2>
]


def _sample_positions(N, Hh, W, count, gen):
    """`count` positions: uniformly random + every corner / border class of the first and the last image."""
    n = torch.randint(0, N, (count,), generator=gen)
    y = torch.randint(0, Hh, (count,), generator=gen)
    x = torch.randint(0, W, (count,), generator=gen)
    edge_y = torch.tensor([0, 0, 0, Hh // 2, Hh - 1, Hh - 1, Hh - 1, Hh // 2, min(1, Hh - 1), max(Hh - 2, 0)])
    edge_x = torch.tensor([0, W // 2, W - 1, W - 1, W - 1, W // 2, 0, 0, min(1, W - 1), max(W - 2, 0)])
    k = edge_y.numel()
    for j, img in enumerate((0, N - 1)):
        n[j * k:(j + 1) * k], y[j * k:(j + 1) * k], x[j * k:(j + 1) * k] = img, edge_y, edge_x
    return n, y, x


_SAMPLED_CACHE = {}


def _sampled_reference(layer, dev):
    """Operands (bf16-representable fp32, on the device and on the host), sample indices and the oracle's fp64 values -- once per
    layer, shared by the variants (the parametrisation iterates variants fastest)."""
    from oracle import ops as O
    name, N, Hh, W, C, K, k, s, up = layer
    if name in _SAMPLED_CACHE:
        return _SAMPLED_CACHE[name]
    _SAMPLED_CACHE.clear()                                   # one layer's tensors at a time (the largest is 537 MB per operand)
    gd = torch.Generator(device=dev).manual_seed(11)
    g = torch.Generator().manual_seed(11)
    bf = lambda t: t.bfloat16().float()
    Ho, Wo = (2 * Hh, 2 * W) if up else (O.same_pad(Hh, k, s)[0], O.same_pad(W, k, s)[0])
    xd = bf(torch.rand((N, Hh, W, C), device=dev, generator=gd) * 2 - 1)
    wd = bf((torch.rand((k, k, C, K), device=dev, generator=gd) * 2 - 1) * (1.5 / (k * k * C) ** 0.5))
    bd = bf(torch.rand((K,), device=dev, generator=gd) - 0.5)
    dyd = bf(torch.rand((N, Ho, Wo, K), device=dev, generator=gd) * 2 - 1)
    x, w, b, dy = xd.cpu(), wd.cpu(), bd.cpu(), dyd.cpu()
    P = 4096
    on, oy, ox = _sample_positions(N, Ho, Wo, P, g)
    inn, iy, ix = _sample_positions(N, Hh, W, P, g)
    taps = [(r, c) for r in range(k) for c in range(k)]
    ci = torch.randperm(C, generator=g)[:min(C, 48)]
    co = torch.randperm(K, generator=g)[:min(K, 48)]
    ref = dict(
        y=O.conv2d_same_sampled(x, w, b, s, on, oy, ox, upsample2x=up),
        dx=O.conv2d_same_dgrad_sampled(dy, w, (N, Hh, W, C), s, inn, iy, ix, upsample2x=up),
        dw=O.conv2d_same_wgrad_sampled(x, dy, (k, k, C, K), s, taps, ci, co, upsample2x=up),
        db=dy.double().sum(dim=(0, 1, 2)))
    del x, dy
    out = (xd, wd, bd, dyd, (on, oy, ox), (inn, iy, ix), (taps, ci, co), ref)
    _SAMPLED_CACHE[name] = out
    return out


FOUR_WAVE_LAYERS = ("df dec3 3x3 512ch @128x128", "df roi down 3x3 s2", "df critic 5x5 s2 on the [x;G] pair", "stage-II roi tower N=448 12x12x384")


@pytest.mark.parametrize("variant", SAMPLED_VARIANTS, ids=[v[0] for v in SAMPLED_VARIANTS])
@pytest.mark.parametrize("layer", SAMPLED_LAYERS, ids=[l[0] for l in SAMPLED_LAYERS])
def test_sampled_oracle_full_size(dev, layer, variant):
    import dpig_amd.hip_ops as H
    name, N, Hh, W, C, K, k, s, up = layer
    vid, arith, ftile, wtile, wave8 = variant
    thin = C == 3 or K == 3
    if thin and vid not in ("f32", "bf16-auto"):
        pytest.skip("the vector-ALU / 16x16x32 thin kernels have no tile families")
    if vid == "bf16-128tile-4wave" and name not in FOUR_WAVE_LAYERS:
        pytest.skip("the four-wave 128-tile kernels are sampled on four layers (same tiles, plan and k order as the eight-wave forms)")
    x, w, b, dy, (on, oy, ox), (inn, iy, ix), (taps, ci, co), ref = _sampled_reference(layer, dev)
    BF = torch.bfloat16
    store = lambda t: t.to(BF) if (arith == "bf16" and t.shape[-1] % 8 == 0) else t
    xg, dyg, wg, bg = store(x), store(dy), w, b
    if arith == "bf16":
        H.set_compute("bf16")
        H.set_large_tile(*ftile)
        H.set_large_tile_wgrad(*wtile)
        H.set_wave8(wave8)
    try:
        y = H.conv2d_fwd(xg, wg, bg, stride=s, upsample2x=up)
        dx = H.conv2d_dgrad(dyg, wg, (N, Hh, W, C), stride=s, upsample2x=up)
        dw = torch.empty((k, k, C, K), device=dev)
        db = torch.empty((K,), device=dev)
        H.conv2d_wgrad(xg, dyg, (k, k, C, K), stride=s, upsample2x=up, out=dw, db=db)
    finally:
        if arith == "bf16":
            H.set_compute("f32")
            H.set_large_tile(1, 0)
            H.set_large_tile_wgrad(1, 0)
            H.set_wave8(3)

    def check(got, want, tol32, what):
        want = want.double()
        scale = max(want.abs().max().item(), 1e-30)
        err = (got.double().cpu() - want).abs()
        bound = torch.full_like(want, tol32 * scale) if got.dtype == torch.float32 else want.abs() * 2.0 ** -8 + 5e-5 * scale
        bad = err > bound
        assert not bool(bad.any()), "%s [%s, %s]: %d of %d sampled elements off, worst %.3e of max|ref| %.3e" % (
            what, name, vid, int(bad.sum()), bad.numel(), float(err.max()), scale)
    check(y[on.to(dev), oy.to(dev), ox.to(dev)], ref["y"], 1e-4, "forward")
    check(dx[inn.to(dev), iy.to(dev), ix.to(dev)], ref["dx"], 1e-4, "dgrad")
    tr = torch.tensor([t[0] for t in taps], device=dev)
    tc = torch.tensor([t[1] for t in taps], device=dev)
    check(dw[tr, tc][:, ci.to(dev)][:, :, co.to(dev)], ref["dw"], 2e-4, "wgrad")
    check(db, ref["db"], 2e-4, "bias gradient")


def teardown_module(module):
    _SAMPLED_CACHE.clear()
