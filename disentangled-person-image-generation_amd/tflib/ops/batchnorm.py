"""tflib.ops.batchnorm drop-in (reference tflib/ops/batchnorm.py:6-87)."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU
from ._layout import nchw_to_nhwc_view, nhwc_to_nchw_view


def set_sync(enabled, group=None):
    """Extension for data-parallel runs: take the batch statistics over all ranks of `group`.  The
    reference is single-process; with this switch a world-size-N run at B/N per rank reproduces its batch-B
    statistics exactly."""
    A.set_sync_batchnorm(enabled, group)


def Batchnorm(name, axes, inputs, is_training=None, stats_iter=None, update_moving_stats=True, fused=True,
              fused_act=None, alpha=0.2):
    """Training-mode batch norm with the reference's contract: with axes [0,2,3] (NCHW conv data) or
    [0,2] and `is_training is None` the batch statistics are ALWAYS used and the moving statistics
    are created (`<name>.moving_mean/.moving_variance`, non-trainable) but never updated
    (batchnorm.py:23-30,51-52).  eps = 1e-5 inside the sqrt, biased variance (fused_batch_norm).
    `fused_act` (extension) folds the following LeakyReLU/ReLU into the normalise kernel."""
    act = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}[fused_act]
    if ((axes == [0, 2, 3]) or (axes == [0, 2])) and fused:
        if axes == [0, 2]:
            inputs = inputs.unsqueeze(3)
        C = inputs.shape[1]
        offset = lib.param(name + '.offset', np.zeros(C, dtype='float32'))
        scale = lib.param(name + '.scale', np.ones(C, dtype='float32'))
        moving_mean = lib.param(name + '.moving_mean', np.zeros(C, dtype='float32'), trainable=False)
        moving_variance = lib.param(name + '.moving_variance', np.ones(C, dtype='float32'), trainable=False)

        x = nchw_to_nhwc_view(inputs)
        if is_training is None or bool(is_training):
            # statistics already left by the producing Conv2D(..., bn_stats=True), if its launch plan could carry them
            y = A.batchnorm(x, scale, offset, 1e-5, act, alpha, stats=getattr(inputs, '_dpig_bnstats', None))
            if is_training is not None and update_moving_stats:
                # batchnorm.py:57-68: running average with weight 1/(stats_iter+1); TF's returned
                # batch_var is Bessel-corrected
                with torch.no_grad():
                    it = float(stats_iter)
                    xf = x.reshape(-1, C)
                    bm = xf.mean(0)
                    bv = xf.var(0, unbiased=True)
                    moving_mean.mul_(it / (it + 1)).add_(bm / (it + 1))
                    moving_variance.mul_(it / (it + 1)).add_(bv / (it + 1))
        else:
            # inference version which blends in the current item's statistics (batchnorm.py:31-37);
            # not on the hot path (every reference call site passes is_training=None)
            bs = float(inputs.shape[0])
            mean = x.mean(dim=(1, 2), keepdim=True)
            var = x.var(dim=(1, 2), unbiased=False, keepdim=True)
            mean = mean / bs + (bs - 1.) / bs * moving_mean
            var = var / bs + (bs - 1.) / bs * moving_variance
            y = (x - mean) / torch.sqrt(var + 1e-5) * scale + offset
            if act != ACT_NONE:
                y = A.activation(y.contiguous(), act, alpha)
        out = nhwc_to_nchw_view(y)
        if axes == [0, 2]:
            return out[:, :, :, 0]
        return out
    else:
        # unfused fallback of the reference (batchnorm.py:74-87): moments over `axes`, params shaped
        # like the kept dims.  Small/unused on the hot path; expressed with tensor plumbing.
        mean = inputs.mean(dim=axes, keepdim=True)
        var = inputs.var(dim=axes, unbiased=False, keepdim=True)
        shape = list(mean.shape)
        if 0 not in axes:
            print("WARNING ({}): didn't find 0 in axes, but not using separate BN params for each item in batch".format(name))
            shape[0] = 1
        offset = lib.param(name + '.offset', np.zeros(shape, dtype='float32'))
        scale = lib.param(name + '.scale', np.ones(shape, dtype='float32'))
        return (inputs - mean) / torch.sqrt(var + 1e-5) * scale + offset
