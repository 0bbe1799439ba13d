"""Differentiable torch-CPU restatement of the TensorFlow-1.4 ops on the DPIG hot path.

TEST INFRASTRUCTURE (see oracle/__init__.py) -- parity unpinned by the reference; pinned against
oracle/naive.py and analytic KATs.  Activations are NHWC, filters HWIO, exactly as the reference
generator (main.py:18 forces NHWC) and tflib store them.  Gradients come from torch autograd over
these forward definitions (TF autodiff, trainer.py:137-140).
"""
import math

import torch
import torch.nn.functional as F


def same_pad(inp, k, stride):
    """tf 'SAME' (tflib/ops/conv2d.py:110; slim.conv2d default): out=ceil(in/s),
    pad_total=max((out-1)*s+k-in,0), pad_before=floor(pad_total/2), remainder after."""
    out = -(-inp // stride)
    total = max((out - 1) * stride + k - inp, 0)
    return out, total // 2, total - total // 2


def conv2d_same(x, w, b=None, stride=1):
    """tf.nn.conv2d(x, w, [1,s,s,1], 'SAME') + bias_add.  x NHWC, w HWIO (cross-correlation).
    Follows tflib/ops/conv2d.py:106-120 and slim.conv2d as used in models.py:396-573."""
    kh, kw = w.shape[0], w.shape[1]
    _, pt, pb = same_pad(x.shape[1], kh, stride)
    _, pl, pr = same_pad(x.shape[2], kw, stride)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))                 # asymmetric pad: torch's padding= is wrong for s=2
    wn = w.permute(3, 2, 0, 1).contiguous()          # HWIO -> OIHW (contiguous: torch-CPU's backward wants it for 1-pixel shapes)
    y = F.conv2d(xn, wn, None, stride=stride)
    y = y.permute(0, 2, 3, 1)
    if b is not None:
        y = y + b
    return y


# ---- the same conv evaluated at SAMPLED positions (full-size layers: a dense fp64 conv of a 256x256 batch takes minutes on the
# host, a few thousand output positions take seconds).  Same definitions as conv2d_same / its autograd gradients -- pinned against
# them on small dense problems by tests/test_oracle.py::test_sampled_conv_equals_dense -- restricted to the requested elements.
# Elements are gathered in the tensors' own dtype (a 537-MB fp32 batch is not copied to fp64) and all arithmetic is float64.
def _sampled_patches(x, kh, kw, stride, n, oy, ox, upsample2x):
    """[P, kh, kw, C] input windows behind output positions (n, oy, ox) of conv2d_same (zero where the window is padding)."""
    N, H, W, C = x.shape
    if upsample2x:                       # 1x1 conv on the nearest-2x upsampled map (models.py:569-570): source pixel (oy//2, ox//2)
        assert kh == 1 and kw == 1 and stride == 1
        return x[n, oy // 2, ox // 2].double().reshape(-1, 1, 1, C)
    _, pt, _ = same_pad(H, kh, stride)
    _, pl, _ = same_pad(W, kw, stride)
    P = n.shape[0]
    out = torch.zeros((P, kh, kw, C), dtype=torch.float64)
    for r in range(kh):
        iy = oy * stride + r - pt
        for s in range(kw):
            ix = ox * stride + s - pl
            ok = (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
            out[ok, r, s] = x[n[ok], iy[ok], ix[ok]].double()
    return out


def conv2d_same_sampled(x, w, b, stride, n, oy, ox, upsample2x=False):
    """conv2d_same(x, w, b, stride)[n, oy, ox, :] for index vectors (n, oy, ox): [P, Cout].  With upsample2x the conv is the 1x1
    conv of the nearest-2x upsampled x (utils.py:61-72 + models.py:570) and (oy, ox) index the upsampled grid."""
    kh, kw, C, K = w.shape
    pat = _sampled_patches(x, kh, kw, stride, n, oy, ox, upsample2x)
    y = pat.reshape(pat.shape[0], -1) @ w.double().reshape(kh * kw * C, K)
    return y if b is None else y + b.double()


def conv2d_same_dgrad_sampled(dy, w, in_shape, stride, n, iy, ix, upsample2x=False):
    """(d conv2d_same / d x)^T dy at input positions (n, iy, ix): [P, Cin].  dy is the full output gradient [N, Ho, Wo, K]."""
    kh, kw, C, K = w.shape
    N, H, W, _ = in_shape
    P = n.shape[0]
    out = torch.zeros((P, C), dtype=torch.float64)
    w = w.double()
    if upsample2x:                       # every low-resolution pixel feeds its 2 x 2 replicas
        for a in range(2):
            for b2 in range(2):
                out += dy[n, 2 * iy + a, 2 * ix + b2].double() @ w[0, 0].t()
        return out
    Ho, pt, _ = same_pad(H, kh, stride)
    Wo, pl, _ = same_pad(W, kw, stride)
    for r in range(kh):
        ty = iy + pt - r                 # oy * stride = iy + pt - r
        for s in range(kw):
            tx = ix + pl - s
            ok = (ty >= 0) & (tx >= 0) & (ty % stride == 0) & (tx % stride == 0) & (ty // stride < Ho) & (tx // stride < Wo)
            if bool(ok.any()):
                out[ok] += dy[n[ok], ty[ok] // stride, tx[ok] // stride].double() @ w[r, s].t()
    return out


def conv2d_same_wgrad_sampled(x, dy, wshape, stride, taps, ci, co, upsample2x=False):
    """(d conv2d_same / d w)^T dy for the filter taps `taps` (list of (r, s)) and the channel index vectors ci, co:
    [len(taps), len(ci), len(co)], each element the FULL sum over the batch's pixels."""
    kh, kw, C, K = wshape
    N, H, W, _ = x.shape
    dys = dy[..., co].double()
    out = []
    if upsample2x:
        assert kh == 1 and kw == 1 and taps == [(0, 0)]
        pooled = dys[:, 0::2, 0::2] + dys[:, 0::2, 1::2] + dys[:, 1::2, 0::2] + dys[:, 1::2, 1::2]
        return (x[..., ci].double().reshape(-1, len(ci)).t() @ pooled.reshape(-1, len(co)))[None]
    Ho, pt, _ = same_pad(H, kh, stride)
    Wo, pl, _ = same_pad(W, kw, stride)
    xs = x[..., ci].double()
    for (r, s) in taps:
        # output rows oy whose source row oy * stride + r - pt lies inside the image (same for columns)
        oy = [o for o in range(Ho) if 0 <= o * stride + r - pt < H]
        ox = [o for o in range(Wo) if 0 <= o * stride + s - pl < W]
        if not oy or not ox:
            out.append(torch.zeros((len(ci), len(co)), dtype=torch.float64))
            continue
        ys = slice(oy[0] * stride + r - pt, oy[-1] * stride + r - pt + 1, stride)
        xsl = slice(ox[0] * stride + s - pl, ox[-1] * stride + s - pl + 1, stride)
        a = xs[:, ys, xsl].reshape(-1, len(ci))
        g = dys[:, oy[0]:oy[-1] + 1, ox[0]:ox[-1] + 1].reshape(-1, len(co))
        out.append(a.t() @ g)
    return torch.stack(out)


def conv2d_transpose_same(x, w, b=None, stride=2):
    """tf.nn.conv2d_transpose(x, w, out=[N,2H,2W,Cout], strides 2, 'SAME') + bias
    (tflib/ops/deconv2d.py:89-112).  w is (k,k,Cout,Cin).  Defined as the gradient of
    conv2d_same w.r.t. its input, which is TF's own definition of the op."""
    n, h, wd, _ = x.shape
    cout = w.shape[2]
    ref = torch.zeros(n, h * stride, wd * stride, cout, dtype=x.dtype, requires_grad=True)
    with torch.enable_grad():
        y = conv2d_same(ref, w.detach() if not w.requires_grad else w, None, stride)
    assert y.shape == x.shape, (y.shape, x.shape)
    (out,) = torch.autograd.grad(y, ref, x, create_graph=x.requires_grad or w.requires_grad)
    if b is not None:
        out = out + b
    return out


def relu(x):
    return torch.relu(x)


def leaky_relu(x, alpha=0.2):
    """wgan_gp.py:23-24: tf.maximum(alpha*x, x)."""
    return torch.maximum(alpha * x, x)


def batchnorm_train(x, scale, offset, eps=1e-5):
    """tf.nn.fused_batch_norm training mode (tflib/ops/batchnorm.py:30): stats over N,H,W,
    biased variance, eps inside the sqrt.  x is NHWC here (the reference passes NCHW)."""
    mean = x.mean(dim=(0, 1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(0, 1, 2), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * scale + offset


def batchnorm_inference_blend(x, scale, offset, moving_mean, moving_variance, eps=1e-5):
    """`_fused_batch_norm_inference` (tflib/ops/batchnorm.py:31-37): each item's own moments over (H, W), blended 1/B : (B-1)/B with
    the moving statistics, then tf.nn.batch_normalization with eps inside the sqrt.  x is NHWC here (the reference passes NCHW)."""
    bs = float(x.shape[0])
    mean = x.mean(dim=(1, 2), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2), keepdim=True)
    mean = (1. / bs) * mean + ((bs - 1.) / bs) * moving_mean
    var = (1. / bs) * var + ((bs - 1.) / bs) * moving_variance
    return (x - mean) / torch.sqrt(var + eps) * scale + offset


def batchnorm_moving_update(x, moving_mean, moving_variance, stats_iter):
    """`_force_updates` (tflib/ops/batchnorm.py:57-68): running averages with weight 1/(stats_iter+1) of the batch moments that
    tf.nn.fused_batch_norm RETURNS in training mode -- the mean, and the variance with Bessel's correction n/(n-1), n = N*H*W
    (the normalisation itself uses the biased variance).  x NHWC; returns the new (moving_mean, moving_variance)."""
    C = x.shape[-1]
    xf = x.reshape(-1, C)
    n = xf.shape[0]
    bm = xf.mean(0)
    bv = ((xf - bm) ** 2).sum(0) / (n - 1)
    it = float(stats_iter)
    return (it / (it + 1)) * moving_mean + (1 / (it + 1)) * bm, (it / (it + 1)) * moving_variance + (1 / (it + 1)) * bv


def batchnorm_unfused(x, axes, scale, offset, eps=1e-5):
    """The non-fused branch (tflib/ops/batchnorm.py:74-87): tf.nn.moments over `axes` with keep_dims, parameters shaped like the kept
    dims (batch dim forced to 1 when 0 is not in axes), tf.nn.batch_normalization.  x in the caller's own layout."""
    mean = x.mean(dim=tuple(axes), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=tuple(axes), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * scale + offset


def layernorm(x, scale, offset, eps=1e-5):
    """tflib/ops/layernorm.py:6-20 with norm_axes [1,2,3]: per-sample moments over (C,H,W),
    per-channel scale/offset.  x NHWC."""
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * scale + offset


def linear(x, w, b=None):
    """tf.matmul + bias_add (tflib/ops/linear.py:132-146; slim.fully_connected)."""
    y = x @ w
    if b is not None:
        y = y + b
    return y


def upsample2x(x):
    """tf.image.resize_nearest_neighbor, align_corners=False, exact 2x (utils.py:61-72)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def crop_and_resize(img, boxes, box_ind, crop_h, crop_w):
    """tf.image.crop_and_resize, bilinear, extrapolation 0 (models.py:415).  img NHWC, boxes
    [n,4] normalised (y1,x1,y2,x2).  Differentiable w.r.t. img only (TF gives no box grad here)."""
    n, H, W, C = img.shape
    nb = boxes.shape[0]
    boxes = boxes.to(img.dtype)
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    ii = torch.arange(crop_h, dtype=img.dtype)
    jj = torch.arange(crop_w, dtype=img.dtype)
    if crop_h > 1:
        in_y = y1[:, None] * (H - 1) + ii[None, :] * ((y2 - y1) * (H - 1) / (crop_h - 1))[:, None]
    else:
        in_y = (0.5 * (y1 + y2) * (H - 1))[:, None].expand(nb, 1)
    if crop_w > 1:
        in_x = x1[:, None] * (W - 1) + jj[None, :] * ((x2 - x1) * (W - 1) / (crop_w - 1))[:, None]
    else:
        in_x = (0.5 * (x1 + x2) * (W - 1))[:, None].expand(nb, 1)
    oky = ~((in_y < 0) | (in_y > H - 1))
    okx = ~((in_x < 0) | (in_x > W - 1))
    ty = torch.floor(in_y).clamp(0, H - 1)
    by = torch.ceil(in_y).clamp(0, H - 1)
    lx = torch.floor(in_x).clamp(0, W - 1)
    rx = torch.ceil(in_x).clamp(0, W - 1)
    ly = (in_y - torch.floor(in_y))[:, :, None, None]
    lw = (in_x - torch.floor(in_x))[:, None, :, None]
    bi = box_ind.long()[:, None, None]
    tyi, byi, lxi, rxi = ty.long(), by.long(), lx.long(), rx.long()
    tl = img[bi, tyi[:, :, None], lxi[:, None, :]]
    tr = img[bi, tyi[:, :, None], rxi[:, None, :]]
    bl = img[bi, byi[:, :, None], lxi[:, None, :]]
    br = img[bi, byi[:, :, None], rxi[:, None, :]]
    top = tl + (tr - tl) * lw
    bot = bl + (br - bl) * lw
    out = top + (bot - top) * ly
    ok = (oky[:, :, None] & okx[:, None, :])[..., None].to(img.dtype)
    return out * ok


def sigmoid_cross_entropy_with_logits(logits, labels):
    """max(x,0) - x*z + log1p(exp(-|x|)) (trainer.py:239-243)."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))


def tf_adam_step(p, g, m, v, lr, beta1, beta2, eps, t):
    """tf.train.AdamOptimizer (trainer.py:137-140): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    epsilon OUTSIDE the bias-corrected sqrt.  Returns new (p, m, v)."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    m = beta1 * m + (1.0 - beta1) * g
    v = beta2 * v + (1.0 - beta2) * g * g
    p = p - lr_t * m / (torch.sqrt(v) + eps)
    return p, m, v


# ---- input-pipeline pose maps (utils.py:237-318) ----------------------------------------------------
POSE_STENCIL = ([(a, 0) for a in (-4, 4)] + [(a, b) for a in (-3, 3) for b in range(-2, 3)] +
                [(a, b) for a in (-2, 2) for b in range(-3, 4)] + [(a, b) for a in (-1, 1) for b in range(-3, 4)] +
                [(0, b) for b in range(-4, 5)])          # the 49 (row, col) shifts of tf_poseInflate (utils.py:300-314)


def coord2channel_simple_rcv(rcv, keypoint_num=18, is_normalized=True, img_H=128, img_W=64):
    """utils.py:237-285: RCV [B, K*3] or [B,K,3] = (row, col, visibility) per keypoint -> [B,H,W,K] map that is
    2*v-1 at the (clipped, truncated) keypoint pixel and -1 elsewhere.  scatter_nd ADDS duplicates; there are
    none here (one update per (b,k))."""
    B = rcv.shape[0]
    rcv = rcv.reshape(B, keypoint_num, 3)
    R, C, V = rcv[..., 0], rcv[..., 1], rcv[..., 2]
    if is_normalized:
        R = torch.clamp((R + 1) / 2.0 * img_H, min=0.0, max=float(img_H - 1))
        C = torch.clamp((C + 1) / 2.0 * img_W, min=0.0, max=float(img_W - 1))
    Ri, Ci = R.to(torch.int64), C.to(torch.int64)            # tf.to_int32 truncates
    land = torch.zeros(B, img_H, img_W, keypoint_num, dtype=rcv.dtype)
    bi = torch.arange(B)[:, None].expand(B, keypoint_num)
    ki = torch.arange(keypoint_num)[None, :].expand(B, keypoint_num)
    land.index_put_((bi, Ri, Ci, ki), torch.full((B, keypoint_num), 2.0, dtype=rcv.dtype), accumulate=True)
    land = land * V[:, None, None, :]
    return land - 1


def tf_poseInflate(pose, keypoint_num=18, radius=4, img_H=128, img_W=64):
    """utils.py:287-318: g=(pose+1)/2; g + sum of the 49 zero-padded shifts of g; min(.,1); back to [-1,1].
    shift (a,b): result[i,j] = g[i+a, j+b] (pad_to_bounding_box by `radius`, crop at (a+radius, b+radius))."""
    g = (pose + 1) / 2
    pad = torch.nn.functional.pad(g, (0, 0, radius, radius, radius, radius))          # NHWC: pad W then H
    out = g.clone()
    for a, b in POSE_STENCIL:
        out = out + pad[:, a + radius:a + radius + img_H, b + radius:b + radius + img_W, :]
    return torch.clamp(out, max=1.0) * 2 - 1


# ---- evaluation metric of trainer.generate() / score.py: skimage compare_ssim on gray uint8 images ----------------
def rgb2gray_u8(img_u8):
    """skimage.color.rgb2gray on a uint8 RGB image: float64 in [0,1], weights (0.2125, 0.7154, 0.0721)."""
    import numpy as np
    x = np.asarray(img_u8, dtype=np.float64) / 255.0
    return x[..., 0] * 0.2125 + x[..., 1] * 0.7154 + x[..., 2] * 0.0721


def ssim_skimage(X, Y, data_range, win_size=7, K1=0.01, K2=0.03):
    """skimage.measure.compare_ssim(X, Y, data_range=..., multichannel=False) with its defaults (trainer.py:516-521):
    7x7 uniform window, sample covariance (NP/(NP-1)), mean of the SSIM map cropped by (win-1)/2 on every side."""
    import numpy as np
    from scipy.ndimage import uniform_filter
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    NP = win_size ** 2
    cov_norm = NP / (NP - 1.0)
    ux, uy = uniform_filter(X, size=win_size), uniform_filter(Y, size=win_size)
    uxx, uyy, uxy = uniform_filter(X * X, size=win_size), uniform_filter(Y * Y, size=win_size), uniform_filter(X * Y, size=win_size)
    vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
    pad = (win_size - 1) // 2
    return float(S[pad:-pad, pad:-pad].mean())


def ssim_G_x(G_255, x_pm1):
    """The per-image SSIM list of trainer.generate() (trainer.py:516-521): G in 0..255 floats, x in [-1,1]."""
    import numpy as np
    out = []
    for i in range(G_255.shape[0]):
        g = rgb2gray_u8(np.clip(np.asarray(G_255[i]), 0, 255).astype(np.uint8))
        x = rgb2gray_u8(np.clip((np.asarray(x_pm1[i]) + 1) * 127.5, 0, 255).astype(np.uint8))
        out.append(ssim_skimage(g, x, data_range=x.max() - x.min()))
    return np.array(out)
