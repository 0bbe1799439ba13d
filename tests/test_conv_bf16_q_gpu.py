"""GPU parity of the large-tile bf16-storage gather-GEMM (csrc/dpig_conv_bf16_q.hip: 8-wave workgroups, 256 x 256 and
512 x 128 block tiles, counted-vmcnt LDS-DMA pipeline across raw barriers) through dpig_conv2d_fwd_bf16 / _dgrad_bf16.

The kernels are forced (dpig_conv_bf16_set_large_tile(2, variant)) onto small layers so that every structural edge is hit
against the fp64 oracle on the bf16-rounded operands: partial row tiles, partial column tiles, odd and even k-tile counts,
two k-tiles only, halo taps, stride 2, 5x5, 1x1, every fused epilogue (lean bodies and the generic one).  At BASELINE
sizes the result must equal the 128 x 128 kernels' bit for bit (same products, same k order, fp32 accumulation) and must
be repeatable launch after launch (the DMA pipeline's hazards are timing dependent, a race would show up as a difference).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * scale


def _r(t):
    return t.float().to(BF).double()


def _close_bf16(got, ref):
    assert got.dtype == BF
    ref = ref.double()
    err = (got.double().cpu() - ref).abs()
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * max(ref.abs().max().item(), 1e-6)
    bad = err > bound
    assert not bad.any(), "%d elements off; worst err %.3e (ref scale %.3e)" % (int(bad.sum()), err.max().item(), ref.abs().max().item())


@pytest.fixture(params=[1, 2], ids=["256x256", "512x128"])
def large_tile(request):
    import dpig_amd.hip_ops as H
    H.set_large_tile(2, request.param)
    yield request.param
    H.set_large_tile(1, 0)


# (N, H, W, C, K, k, stride)
FWD_SHAPES = [
    (3, 24, 20, 128, 192, 3, 1),    # 1440 rows (partial row tile both variants), 18 k-tiles
    (2, 16, 12, 64, 264, 3, 1),     # 264 columns: a partial column tile of 8; 9 k-tiles (odd)
    (2, 33, 17, 64, 128, 5, 2),     # 5x5 stride 2 (TF pads (1,2) / (2,2)), 25 k-tiles
    (2, 16, 16, 192, 64, 1, 1),     # 1x1: 3 k-tiles
    (1, 16, 16, 128, 64, 1, 1),     # 1x1: exactly 2 k-tiles (prologue + one loop trip + drain)
    (2, 9, 7, 64, 72, 3, 2),        # odd input, stride 2, tiny M (126 rows)
    (5, 3, 3, 640, 64, 3, 1),       # 3x3 image: every tap is mostly halo, 90 k-tiles
]


@pytest.mark.parametrize("shape", FWD_SHAPES)
def test_forward_against_oracle(dev, large_tile, shape):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2, 0.2)
    b = _rand((K,), 3)
    ref = O.leaky_relu(O.conv2d_same(_r(x), _r(w), b.float().double(), s), 0.2)
    got = H.conv2d_fwd(x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), stride=s, act=2, alpha=0.2)
    _close_bf16(got, ref)


def test_fused_epilogues(dev, large_tile):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 3, 24, 20, 128, 192
    x = _rand((N, Hh, W, C), 1)
    w = _rand((3, 3, C, K), 2, 0.2)
    b = _rand((K,), 3)
    res = _rand((N, Hh, W, K), 4)
    xd, wd, bd, rd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), res.float().to(dev).to(BF)
    xr = _r(x).requires_grad_(True)
    conv0 = O.conv2d_same(xr, _r(w), None, 1)
    conv = conv0.detach() + b.float().double()
    # no bias, no activation
    _close_bf16(H.conv2d_fwd(xd, wd, None), conv0.detach())
    # residual before the activation
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd), O.relu(conv + _r(res)))
    # the res-block tail: act -> out_act, (stored act) + skip -> out
    out, out_act = torch.empty((N, Hh, W, K), dtype=BF, device=dev), torch.empty((N, Hh, W, K), dtype=BF, device=dev)
    H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
    _close_bf16(out_act, O.relu(conv))
    assert torch.equal(out.cpu(), (out_act.float() + rd.float()).to(BF).cpu())
    # post-activation residual WITHOUT the second output (generic epilogue body)
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True), O.relu(conv) + _r(res))
    # class-indexed residual (tiled-embedding collapse, fp32 [N, 9, K]; generic body)
    e9 = _rand((N, 9, K), 5).float()
    yy, xx = torch.meshgrid(torch.arange(Hh), torch.arange(W), indexing="ij")
    cls = torch.where(yy == 0, 0, torch.where(yy == Hh - 1, 2, 1)) * 3 + torch.where(xx == 0, 0, torch.where(xx == W - 1, 2, 1))
    ref = O.relu(conv + e9.double()[:, cls.reshape(-1), :].reshape(N, Hh, W, K))
    _close_bf16(H.conv2d_fwd(xd, wd, bd, act=1, residual=e9.to(dev), res_class=True), ref)
    # channel slices of wider buffers on both sides
    xbig = torch.zeros((N, Hh, W, C + 64), dtype=BF, device=dev)
    xbig[..., 64:] = xd
    ybig = torch.full((N, Hh, W, K + 64), 7.0, dtype=BF, device=dev)
    H.conv2d_fwd(xbig[..., 64:], wd, bd, act=1, out=ybig[..., :K])
    _close_bf16(ybig[..., :K], O.relu(conv))
    assert (ybig[..., K:] == 7.0).all()
    # stride-1 dgrad: plain, * mask, (+ accum) * mask   (Cs = K = 192, columns = C = 128)
    dy = _rand((N, Hh, W, K), 6)
    conv0.backward(_r(dy))
    dyd = dy.float().to(dev).to(BF)
    acc, m = _rand((N, Hh, W, C), 7), _rand((N, Hh, W, C), 8)
    ad, md = acc.float().to(dev).to(BF), m.float().to(dev).to(BF)
    _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C)), xr.grad)
    _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), mask=md, act=1), xr.grad * (_r(m) > 0))
    _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad, mask=md, act=2, alpha=0.2),
                (xr.grad + _r(acc)) * torch.where(_r(m) > 0, 1.0, 0.2))
    _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad), xr.grad + _r(acc))


def test_upsampled_1x1(dev, large_tile):
    """nearest-2x upsample + 1x1 conv computed at low resolution with the 2 x 2 replicating epilogue (models.py:569-570)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = 2, 12, 10, 128, 64
    x = _rand((N, Hh, W, C), 3)
    w1 = _rand((1, 1, C, K), 4, 0.3)
    b = _rand((K,), 5)
    ref = O.relu(O.conv2d_same(O.upsample2x(_r(x)), _r(w1), b.float().double(), 1))
    got = H.conv2d_fwd(x.float().to(dev).to(BF), w1.float().to(dev), b.float().to(dev), act=1, upsample2x=True)
    _close_bf16(got, ref)


# ---- the halo-staged kernels: bhq_kernel (256 x 256: 16 x 16-pixel patches) and bhq32_kernel (512 x 128: 32 x 16-pixel patches, 32-channel
# k-tiles).  Tiles run over the STACK of all images' pixel rows (wave rows of 8 pixel rows stay inside one image: H % 8 == 0), so a
# tile may straddle two images; (variant, (N, H, W, C, K)): W a multiple of 16, C of 64, N * H a multiple of 16 / 32
HALO_SHAPES = [
    (2, (2, 32, 16, 64, 128)),      # one patch per image, two 32-channel chunks (the next chunk's halo is staged exactly once)
    (2, (1, 64, 48, 128, 128)),     # 2 x 3 patches: interior halos on every side
    (2, (3, 32, 32, 192, 104)),     # 104 columns: a partial column tile (dead filter rows), 6 chunks
    (2, (1, 32, 16, 64, 264)),      # three column tiles, the last one with 8 live columns
    (2, (4, 24, 16, 64, 128)),      # 24-row images: every tile straddles two images (rows 0-23 | 0-7, 8-23 | 0-15, ...)
    (2, (4, 48, 48, 128, 128)),     # the Market ROI-tower level (48 x 48 crops): tiles 1 and 2 of every three straddle
    (1, (4, 24, 16, 64, 256)),      # bhq_kernel, 16-row tiles over 24-row images
    (1, (2, 40, 32, 128, 192)),     # bhq_kernel, 40 = 2.5 tiles per image, a partial column tile
]


@pytest.mark.parametrize("variant,shape", HALO_SHAPES)
def test_halo_staged_kernels_against_oracle(dev, variant, shape):
    """Forward (every fused epilogue of the family) and the stride-1 dgrad of 3x3 layers on bhq_kernel / bhq32_kernel against the fp64 oracle on
    the bf16-rounded operands, and launch-after-launch repeatability (the counted-vmcnt pipeline's hazards are timing dependent)."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K = shape
    x = _rand((N, Hh, W, C), 21)
    w = _rand((3, 3, C, K), 22, 0.2)
    b = _rand((K,), 23)
    res = _rand((N, Hh, W, K), 24)
    dy = _rand((N, Hh, W, K), 26)
    acc, m = _rand((N, Hh, W, C), 27), _rand((N, Hh, W, C), 28)
    xd, wd, bd, rd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev), res.float().to(dev).to(BF)
    dyd, ad, md = dy.float().to(dev).to(BF), acc.float().to(dev).to(BF), m.float().to(dev).to(BF)
    xr = _r(x).requires_grad_(True)
    conv0 = O.conv2d_same(xr, _r(w), None, 1)
    conv = conv0.detach() + b.float().double()
    conv0.backward(_r(dy))
    try:
        H.set_large_tile(2, variant)
        y1 = H.conv2d_fwd(xd, wd, bd, act=1)
        _close_bf16(y1, O.relu(conv))
        _close_bf16(H.conv2d_fwd(xd, wd, None), conv0.detach())
        _close_bf16(H.conv2d_fwd(xd, wd, bd, act=2, alpha=0.2, residual=rd), O.leaky_relu(conv + _r(res), 0.2))
        out, out_act = torch.empty((N, Hh, W, K), dtype=BF, device=dev), torch.empty((N, Hh, W, K), dtype=BF, device=dev)
        H.conv2d_fwd(xd, wd, bd, act=1, residual=rd, res_after_act=True, out=out, out_act=out_act)
        _close_bf16(out_act, O.relu(conv))
        assert torch.equal(out.cpu(), (out_act.float() + rd.float()).to(BF).cpu())
        dx1 = H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), mask=md, act=1)
        _close_bf16(dx1, xr.grad * (_r(m) > 0))
        _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C)), xr.grad)
        _close_bf16(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), accum=ad, mask=md, act=2, alpha=0.2),
                    (xr.grad + _r(acc)) * torch.where(_r(m) > 0, 1.0, 0.2))
        # channel slices of wider buffers on both sides
        xbig = torch.zeros((N, Hh, W, C + 64), dtype=BF, device=dev)
        xbig[..., 64:] = xd
        ybig = torch.full((N, Hh, W, K + 64), 7.0, dtype=BF, device=dev)
        H.conv2d_fwd(xbig[..., 64:], wd, bd, act=1, out=ybig[..., :K])
        assert torch.equal(ybig[..., :K], y1) and (ybig[..., K:] == 7.0).all()
        for _ in range(4):
            assert torch.equal(H.conv2d_fwd(xd, wd, bd, act=1), y1)
            assert torch.equal(H.conv2d_dgrad(dyd, wd, (N, Hh, W, C), mask=md, act=1), dx1)
    finally:
        H.set_large_tile(1, 0)


def _full_size_operands(dev, N, Hh, W, C, K):
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.rand((N, Hh, W, C), generator=g) * 2 - 1).to(dev).to(BF)
    w = ((torch.rand((3, 3, C, K), generator=g) * 2 - 1) * 0.05).to(dev)
    b = (torch.rand((K,), generator=g) * 2 - 1).to(dev)
    dy = (torch.rand((N, Hh, W, K), generator=g) * 2 - 1).to(dev).to(BF)
    m = (torch.rand((N, Hh, W, C), generator=g) * 2 - 1).to(dev).to(BF)
    return x, w, b, dy, m


def _one_ulp(a, b):
    """bf16 tensors equal to within one unit in the last place + fp32 round-off of a 2304..6912-term sum (different fp32
    summation orders of the same products)"""
    d = (a.float() - b.float()).abs()
    return bool((d <= b.float().abs() * 2.0 ** -7 + 1e-4 * b.float().abs().max()).all())


@pytest.mark.parametrize("layer", [(8, 128, 128, 256, 256), (8, 256, 256, 128, 128), (16, 128, 64, 256, 256),
                                   (8, 64, 64, 768, 768), (56, 64, 64, 128, 128)])
def test_full_size_layers_repeat_and_agree(dev, layer):
    """BASELINE configs[1]/[3] layer sizes (3x3 stride 1), forward (+ bias + ReLU) and dgrad (* mask): every launch of the
    large-tile kernels reproduces itself bit for bit (five launches: the DMA pipeline's hazards are timing dependent, a race
    would show as a difference).  The 256 x 256 variant runs these layers on the halo-staged kernel (bhq_kernel: k order (chunk,
    tap), as the 128 x 128 halo-patch kernel), the 512 x 128 variant tap-major: the variants agree with each other and with the
    128 x 128 kernel to one bf16 ulp (same products, fp32 accumulation in a different order); with the halo staging switched off
    (DPIG_BF16_QH=0 builds / non-16-multiple images) the two variants share the k order and are bit-identical
    (test_equals_the_tap_major_128_tile_kernel_bit_for_bit covers that path)."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K = layer
    x, w, b, dy, m = _full_size_operands(dev, N, Hh, W, C, K)
    w._dpig_shadow = H.filter_shadows(w)
    try:
        H.set_large_tile(0, 0)
        y0 = H.conv2d_fwd(x, w, b, act=1)
        dx0 = H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1)
        refs = {}
        for variant in (1, 2):
            H.set_large_tile(2, variant)
            ref = None
            for rep in range(5):
                y = H.conv2d_fwd(x, w, b, act=1)
                dx = H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1)
                if ref is None:
                    ref = (y, dx)
                assert torch.equal(y, ref[0]), "forward differs (variant %d, launch %d): %d elements" % (
                    variant, rep, int((y != ref[0]).sum()))
                assert torch.equal(dx, ref[1]), "dgrad differs (variant %d, launch %d): %d elements" % (
                    variant, rep, int((dx != ref[1]).sum()))
            refs[variant] = ref
            assert _one_ulp(ref[0], y0) and _one_ulp(ref[1], dx0)
        assert _one_ulp(refs[1][0], refs[2][0]) and _one_ulp(refs[1][1], refs[2][1])
        # the halo-staged 256 x 256 kernel and the 128 x 128 halo-patch kernel share the k order (chunk, tap) and the per-k-tile MFMA
        # order: the same fp32 sums, bit for bit
        assert torch.equal(refs[1][0], y0) and torch.equal(refs[1][1], dx0)
    finally:
        H.set_large_tile(1, 0)


def test_equals_the_tap_major_128_tile_kernel_bit_for_bit(dev):
    """A 9 x 17 image does not tile into 8 x 16 / 16 x 8 patches, so the 128 x 128 family runs its tap-by-tap kernel
    (bg_kernel), whose k order (tap, channel chunk) is the large-tile kernels': outputs must be identical."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K = 192, 9, 17, 256, 256
    x, w, b, dy, m = _full_size_operands(dev, N, Hh, W, C, K)
    w._dpig_shadow = H.filter_shadows(w)
    try:
        H.set_large_tile(0, 0)
        y0 = H.conv2d_fwd(x, w, b, act=1)
        dx0 = H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1)
        for variant in (1, 2):
            H.set_large_tile(2, variant)
            assert torch.equal(H.conv2d_fwd(x, w, b, act=1), y0)
            assert torch.equal(H.conv2d_dgrad(dy, w, (N, Hh, W, C), mask=m, act=1), dx0)
    finally:
        H.set_large_tile(1, 0)


# ---- filter gradient on the large-tile kernel (csrc/dpig_conv_bf16_wq.hip) --------------------------------------------------
@pytest.fixture(params=[1, 2], ids=["2x256", "4x128"])
def large_tile_wgrad(request):
    import dpig_amd.hip_ops as H
    H.set_large_tile_wgrad(2, request.param)
    yield request.param
    H.set_large_tile_wgrad(1, 0)


def _close_f32(got, ref, tol=2e-5):
    ref = ref.double()
    err = (got.double().cpu() - ref).abs().max().item()
    scale = max(ref.abs().max().item(), 1e-6)
    assert err <= tol * scale, "max err %.3e vs scale %.3e" % (err, scale)


WGRAD_SHAPES = [
    (3, 24, 20, 128, 192, 3),    # 1440 pixels (22.5 k-tiles), 9 items: a half-empty item tile / a partial column tile
    (2, 16, 12, 64, 264, 3),     # 64 input channels (half an item), 264 output channels
    (2, 16, 16, 192, 64, 1),     # 1x1: one tap, two channel blocks
    (5, 3, 3, 640, 64, 3),       # 45 pixels: one partial k-tile, images smaller than a k-tile, every tap mostly halo
    (1, 40, 40, 256, 256, 3),    # 25 k-tiles
]


@pytest.mark.parametrize("shape", WGRAD_SHAPES)
@pytest.mark.parametrize("split_k", [0, 3])
def test_wgrad_against_oracle(dev, large_tile_wgrad, shape, split_k):
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2, 0.2)
    xr = _r(x)
    wr = _r(w).requires_grad_(True)
    y = O.conv2d_same(xr, wr, None, 1)
    dy = _rand(tuple(y.shape), 3)
    y.backward(_r(dy))
    xd, dyd = x.float().to(dev).to(BF), dy.float().to(dev).to(BF)
    dw = torch.full((k, k, C, K), 2.0, device=dev)
    db = torch.full((K,), 3.0, device=dev)
    H.conv2d_wgrad(xd, dyd, (k, k, C, K), out=dw, beta=1.0, split_k=split_k, db=db, db_beta=1.0)
    _close_f32(dw, wr.grad + 2.0)
    _close_f32(db, _r(dy).sum((0, 1, 2)) + 3.0)
    dw0 = torch.empty((k, k, C, K), device=dev)
    H.conv2d_wgrad(xd, dyd, (k, k, C, K), out=dw0, beta=0.0, split_k=split_k)
    _close_f32(dw0, wr.grad)


@pytest.mark.parametrize("layer", [(8, 128, 128, 256, 256), (8, 64, 64, 768, 768), (16, 64, 32, 512, 512), (8, 64, 64, 384, 384)])
def test_wgrad_full_size_repeats_and_agrees(dev, layer):
    """BASELINE layer sizes: the large-tile filter gradient reproduces itself bit for bit launch after launch and agrees with
    the 128 x 128 kernel (another split of the pixel range: another fp32 summation order) to fp32 round-off."""
    import dpig_amd.hip_ops as H
    N, Hh, W, C, K = layer
    x, w, b, dy, m = _full_size_operands(dev, N, Hh, W, C, K)
    try:
        H.set_large_tile_wgrad(0, 0)
        dw0 = torch.empty((3, 3, C, K), device=dev)
        db0 = torch.empty((K,), device=dev)
        H.conv2d_wgrad(x, dy, (3, 3, C, K), out=dw0, db=db0)
        scale = dw0.abs().max().item()
        for variant in (1, 2):
            H.set_large_tile_wgrad(2, variant)
            ref = None
            for rep in range(3):
                dw = torch.empty((3, 3, C, K), device=dev)
                db = torch.empty((K,), device=dev)
                H.conv2d_wgrad(x, dy, (3, 3, C, K), out=dw, db=db)
                if ref is None:
                    ref = (dw, db)
                assert torch.equal(dw, ref[0]) and torch.equal(db, ref[1]), "launch %d differs (variant %d)" % (rep, variant)
            assert (ref[0] - dw0).abs().max().item() <= 2e-5 * scale
            assert (ref[1] - db0).abs().max().item() <= 2e-5 * db0.abs().max().item()
    finally:
        H.set_large_tile_wgrad(1, 0)


# ---- the eight-wave form of the 128 x 128 kernel (csrc/dpig_conv_bf16.hip bg8_kernel / bg8_multi_kernel) --------------------------
WAVE8_SHAPES = FWD_SHAPES + [
    (16, 8, 4, 640, 640, 3, 1),     # Market G level 5: 512 rows, 90 k-tiles -> split-K plan
    (16, 16, 8, 512, 384, 3, 2),    # stride 2 (the dgrad runs its four parity classes in one launch)
    (2, 12, 10, 72, 200, 3, 1),     # channel tail (72 = 64 + 8) and a partial column tile
    (16, 64, 64, 64, 128, 3, 1),    # 512 patches of 8 x 16: the halo-patch kernel (bh8_kernel<4>)
    (36, 48, 40, 72, 128, 3, 1),    # 16 x 8 patches (bh8_kernel<3>), channel tail in the halo
]


@pytest.mark.parametrize("shape", WAVE8_SHAPES)
def test_eight_wave_128_tile_kernel(dev, shape):
    """Same tile, plan and k order as the four-wave kernels, so the results must be the same bits -- forward with a fused epilogue,
    dgrad with the activation mask (stride 2: bg8_multi_kernel; halo patches: bh8_kernel), filter + bias gradient (bw8_kernel), with
    and without split-K -- and
    forward and filter gradient must meet the oracle."""
    import dpig_amd.hip_ops as H
    from oracle import ops as O
    N, Hh, W, C, K, k, s = shape
    x = _rand((N, Hh, W, C), 1)
    w = _rand((k, k, C, K), 2, 0.2)
    b = _rand((K,), 3)
    xd, wd, bd = x.float().to(dev).to(BF), w.float().to(dev), b.float().to(dev)
    Ho, Wo = -(-Hh // s), -(-W // s)
    dy = _rand((N, Ho, Wo, K), 5).float().to(dev).to(BF)
    m = _rand((N, Hh, W, C), 6).float().to(dev).to(BF)
    try:
        H.set_large_tile(0, 0)
        H.set_large_tile_wgrad(0, 0)
        out = {}
        for mode in (0, 7):
            H.set_wave8(mode)
            for rep in range(3):
                y = H.conv2d_fwd(xd, wd, bd, stride=s, act=2, alpha=0.2)
                dx = H.conv2d_dgrad(dy, wd, (N, Hh, W, C), stride=s, mask=m, act=1)
                db = torch.empty(K, device=dev)
                dw = H.conv2d_wgrad(xd, dy, (k, k, C, K), stride=s, db=db)
                if rep == 0:
                    out[mode] = (y, dx, dw, db)
                assert all(torch.equal(t0, t1) for t0, t1 in zip((y, dx, dw, db), out[mode])), "launch %d differs (mode %d)" % (rep, mode)
        for name, t0, t1 in zip(("forward", "dgrad", "wgrad", "bias gradient"), out[0], out[7]):
            assert torch.equal(t0, t1), "%s: %d elements differ" % (name, int((t0 != t1).sum()))
        wref = torch.autograd.functional.vjp(lambda ww: O.conv2d_same(_r(x), ww, None, s), w.double() * 0, dy.double().cpu())[1]
        err = (out[7][2].double().cpu() - wref).abs().max().item()
        assert err <= 2e-5 * max(wref.abs().max().item(), 1e-6) + 1e-6, "wgrad off the oracle by %.3e (scale %.3e)" % (err, wref.abs().max().item())
        ref = O.leaky_relu(O.conv2d_same(_r(x), _r(w), b.float().double(), s), 0.2)
        _close_bf16(out[7][0], ref)
    finally:
        H.set_wave8(3)
        H.set_large_tile(1, 0)
        H.set_large_tile_wgrad(1, 0)
