"""Independent numpy restatement (explicit loops / einsum, float64) of the TF-1.4 kernels that the
hot path reaches.  TEST INFRASTRUCTURE: its only job is to pin oracle/ops.py (two independent
restatements of the published TensorFlow algorithms must agree) -- parity is otherwise unpinned
because the reference ships no vectors (see oracle/__init__.py).

Each function cites the reference call site it restates; the kernel arithmetic itself follows
TensorFlow 1.4.1 (tensorflow/core/kernels/{conv_ops,crop_and_resize_op,resize_nearest_neighbor_op,
fused_batch_norm_op}.cc and python/training/adam.py as published).
"""
import math

import numpy as np


def same_pad(inp, k, stride):
    out = int(math.ceil(inp / float(stride)))
    total = max((out - 1) * stride + k - inp, 0)
    return out, total // 2


def conv2d_same(x, w, b=None, stride=1):
    """tflib/ops/conv2d.py:106-120 / slim.conv2d models.py:396: direct 7-loop definition."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    n, H, W, C = x.shape
    R, S, _, K = w.shape
    Ho, pt = same_pad(H, R, stride)
    Wo, pl = same_pad(W, S, stride)
    y = np.zeros((n, Ho, Wo, K))
    for oy in range(Ho):
        for ox in range(Wo):
            acc = np.zeros((n, K))
            for ky in range(R):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(S):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    acc += x[:, iy, ix, :] @ w[ky, kx]
            y[:, oy, ox, :] = acc
    if b is not None:
        y = y + np.asarray(b, np.float64)
    return y


def conv2d_same_dgrad(dy, w, in_shape, stride=1):
    """Gradient of conv2d_same w.r.t. x (scatter form of the definition above)."""
    dy = np.asarray(dy, np.float64)
    w = np.asarray(w, np.float64)
    n, H, W, C = in_shape
    R, S, _, K = w.shape
    Ho, pt = same_pad(H, R, stride)
    Wo, pl = same_pad(W, S, stride)
    dx = np.zeros((n, H, W, C))
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(R):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(S):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    dx[:, iy, ix, :] += dy[:, oy, ox, :] @ w[ky, kx].T
    return dx


def conv2d_same_wgrad(x, dy, wshape, stride=1):
    x = np.asarray(x, np.float64)
    dy = np.asarray(dy, np.float64)
    n, H, W, C = x.shape
    R, S, _, K = wshape
    Ho, pt = same_pad(H, R, stride)
    Wo, pl = same_pad(W, S, stride)
    dw = np.zeros((R, S, C, K))
    for oy in range(Ho):
        for ox in range(Wo):
            for ky in range(R):
                iy = oy * stride + ky - pt
                if iy < 0 or iy >= H:
                    continue
                for kx in range(S):
                    ix = ox * stride + kx - pl
                    if ix < 0 or ix >= W:
                        continue
                    dw[ky, kx] += x[:, iy, ix, :].T @ dy[:, oy, ox, :]
    return dw


def batchnorm_train(x, scale, offset, eps=1e-5):
    """tflib/ops/batchnorm.py:30 (fused_batch_norm, training): biased variance, eps in sqrt."""
    x = np.asarray(x, np.float64)
    C = x.shape[-1]
    flat = x.reshape(-1, C)
    mean = flat.sum(0) / flat.shape[0]
    var = ((flat - mean) ** 2).sum(0) / flat.shape[0]
    return (x - mean) / np.sqrt(var + eps) * scale + offset


def layernorm(x, scale, offset, eps=1e-5):
    """tflib/ops/layernorm.py:7-19 with axes (C,H,W) per sample (x NHWC here)."""
    x = np.asarray(x, np.float64)
    out = np.empty_like(x)
    for i in range(x.shape[0]):
        mu = x[i].mean()
        var = ((x[i] - mu) ** 2).mean()
        out[i] = (x[i] - mu) / math.sqrt(var + eps) * scale + offset
    return out


def upsample2x(x):
    """utils.py:61-72: out[y,x] = in[min(floor(y*in/out), in-1)] with out = 2*in."""
    x = np.asarray(x)
    n, H, W, C = x.shape
    y = np.empty((n, 2 * H, 2 * W, C), x.dtype)
    for oy in range(2 * H):
        for ox in range(2 * W):
            y[:, oy, ox, :] = x[:, min(int(math.floor(oy * H / (2.0 * H))), H - 1),
                                min(int(math.floor(ox * W / (2.0 * W))), W - 1), :]
    return y


def crop_and_resize(img, boxes, box_ind, ch, cw):
    """models.py:415 -- tensorflow/core/kernels/crop_and_resize_op.cc (bilinear, extrapolation 0)."""
    img = np.asarray(img, np.float64)
    n, H, W, C = img.shape
    nb = len(boxes)
    out = np.zeros((nb, ch, cw, C))
    for b in range(nb):
        y1, x1, y2, x2 = [float(v) for v in boxes[b]]
        bi = int(box_ind[b])
        hs = (y2 - y1) * (H - 1) / (ch - 1) if ch > 1 else 0.0
        ws = (x2 - x1) * (W - 1) / (cw - 1) if cw > 1 else 0.0
        for i in range(ch):
            in_y = y1 * (H - 1) + i * hs if ch > 1 else 0.5 * (y1 + y2) * (H - 1)
            if in_y < 0 or in_y > H - 1:
                continue
            ty, by = int(math.floor(in_y)), int(math.ceil(in_y))
            ly = in_y - ty
            for j in range(cw):
                in_x = x1 * (W - 1) + j * ws if cw > 1 else 0.5 * (x1 + x2) * (W - 1)
                if in_x < 0 or in_x > W - 1:
                    continue
                lx_, rx = int(math.floor(in_x)), int(math.ceil(in_x))
                lw = in_x - lx_
                top = img[bi, ty, lx_] + (img[bi, ty, rx] - img[bi, ty, lx_]) * lw
                bot = img[bi, by, lx_] + (img[bi, by, rx] - img[bi, by, lx_]) * lw
                out[b, i, j] = top + (bot - top) * ly
    return out


def crop_and_resize_grad_image(dout, boxes, box_ind, img_shape):
    """CropAndResizeGradImage: scatter the 4 bilinear weights."""
    dout = np.asarray(dout, np.float64)
    n, H, W, C = img_shape
    nb, ch, cw, _ = dout.shape
    dimg = np.zeros(img_shape)
    for b in range(nb):
        y1, x1, y2, x2 = [float(v) for v in boxes[b]]
        bi = int(box_ind[b])
        hs = (y2 - y1) * (H - 1) / (ch - 1) if ch > 1 else 0.0
        ws = (x2 - x1) * (W - 1) / (cw - 1) if cw > 1 else 0.0
        for i in range(ch):
            in_y = y1 * (H - 1) + i * hs if ch > 1 else 0.5 * (y1 + y2) * (H - 1)
            if in_y < 0 or in_y > H - 1:
                continue
            ty, by = int(math.floor(in_y)), int(math.ceil(in_y))
            ly = in_y - ty
            for j in range(cw):
                in_x = x1 * (W - 1) + j * ws if cw > 1 else 0.5 * (x1 + x2) * (W - 1)
                if in_x < 0 or in_x > W - 1:
                    continue
                lx_, rx = int(math.floor(in_x)), int(math.ceil(in_x))
                lw = in_x - lx_
                g = dout[b, i, j]
                dimg[bi, ty, lx_] += (1 - ly) * (1 - lw) * g
                dimg[bi, ty, rx] += (1 - ly) * lw * g
                dimg[bi, by, lx_] += ly * (1 - lw) * g
                dimg[bi, by, rx] += ly * lw * g
    return dimg


def sigmoid_cross_entropy_with_logits(x, z):
    x = np.asarray(x, np.float64)
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def tf_adam(p, grads, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """python/training/adam.py: sequence of updates for a list of gradients; returns final p."""
    p = np.asarray(p, np.float64).copy()
    m = np.zeros_like(p)
    v = np.zeros_like(p)
    for t, g in enumerate(grads, start=1):
        g = np.asarray(g, np.float64)
        lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
        m = beta1 * m + (1 - beta1) * g
        v = beta2 * v + (1 - beta2) * g * g
        p = p - lr_t * m / (np.sqrt(v) + eps)
    return p


def pose_disc_map(rcv, keypoint_num, img_H, img_W, radius=4):
    """Independent restatement of the pose target the input pipeline builds (utils.py:237-318 chained): the
    reference's own numpy variant fills a Euclidean disc of `radius` around each visible keypoint
    (py_poseInflate, utils.py:320-340); the 49-shift stencil of tf_poseInflate is exactly that disc for radius 4."""
    B = rcv.shape[0]
    rcv = np.asarray(rcv, dtype=np.float64).reshape(B, keypoint_num, 3)
    out = -np.ones((B, img_H, img_W, keypoint_num))
    for b in range(B):
        for k in range(keypoint_num):
            r = min(max((rcv[b, k, 0] + 1) / 2.0 * img_H, 0.0), img_H - 1.0)
            c = min(max((rcv[b, k, 1] + 1) / 2.0 * img_W, 0.0), img_W - 1.0)
            r, c, v = int(r), int(c), rcv[b, k, 2]
            for i in range(-radius, radius + 1):
                for j in range(-radius, radius + 1):
                    if i * i + j * j <= radius * radius and 0 <= r + i < img_H and 0 <= c + j < img_W:
                        hits = 2 if (i == 0 and j == 0) else 1          # the stencil visits the centre twice
                        out[b, r + i, c + j, k] = min(v * hits, 1.0) * 2 - 1
    return out


def ssim_window_loops(X, Y, data_range, win=7, K1=0.01, K2=0.03):
    """Independent loop restatement of the windowed SSIM (Wang et al. 2004) as skimage evaluates it: for every window
    fully inside the image, unbiased variances / covariance of the 49 samples."""
    X = np.asarray(X, dtype=np.float64); Y = np.asarray(Y, dtype=np.float64)
    H, W = X.shape
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    tot, cnt = 0.0, 0
    for i in range(H - win + 1):
        for j in range(W - win + 1):
            a = X[i:i + win, j:j + win].ravel(); b = Y[i:i + win, j:j + win].ravel()
            ma, mb = a.mean(), b.mean()
            va, vb = a.var(ddof=1), b.var(ddof=1)
            cab = ((a - ma) * (b - mb)).sum() / (a.size - 1)
            tot += ((2 * ma * mb + C1) * (2 * cab + C2)) / ((ma * ma + mb * mb + C1) * (va + vb + C2)); cnt += 1
    return tot / cnt
