"""tflib.ops.conv2d drop-in: `Conv2D(name, input_dim, output_dim, filter_size, inputs, ...)` on logical-NCHW
tensors with the reference's signature and variable names (`<name>.Filters` HWIO, `<name>.Biases`, `<name>.g`;
reference tflib/ops/conv2d.py:20-123): SAME padding, stride 1 or 2, optional PixelCNN mask and weight norm,
computed by `dpig_conv2d_*`."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU
from . import _init
from ._layout import nchw_to_nhwc_view, nhwc_to_nchw_view

_SW = _init.Switches()
_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}


def enable_default_weightnorm():
    _SW.weightnorm = True


def set_weights_stdev(weights_stdev):
    _SW.stdev = weights_stdev


def unset_weights_stdev():
    _SW.stdev = None


def _pixelcnn_mask(kind, groups, k, cin, cout):
    """Causal filter mask of PixelCNN type 'a' / 'b' (conv2d.py:40-58): nothing below the centre row or right of the
    centre tap; at the centre, channel group i of the input may feed group j of the output only if i < j ('a') or
    i <= j ('b')."""
    m = np.ones((k, k, cin, cout), dtype='float32')
    c = k // 2
    m[c + 1:] = 0.
    m[c, c + 1:] = 0.
    for i in range(groups):
        for j in range(groups):
            if i > j or (kind == 'a' and i == j):
                m[c, c, i::groups, j::groups] = 0.
    return m


def Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, mask_type=None, stride=1,
           weightnorm=None, biases=True, gain=1., fused_act=None, alpha=0.2, bn_stats=False):
    """inputs / result: (batch, channels, height, width).  mask_type: None or ('a'|'b', n_channel_groups).
    `fused_act` in {None,'relu','lrelu'} (extension) folds the caller's next activation into the conv epilogue.
    `bn_stats=True` (extension): the caller's next op is `Batchnorm` over this output -- the conv epilogue leaves the batch
    statistics with the result (`_dpig_bnstats`) so that Batchnorm skips its own statistics passes."""
    k = filter_size
    fan_in, fan_out = input_dim * k ** 2, output_dim * k ** 2 / (stride ** 2)
    if mask_type is not None:                      # roughly half of the taps are masked away
        fan_in, fan_out = fan_in / 2., fan_out / 2.
    fresh = None
    if name + '.Filters' not in lib._params:
        fresh = _init.conv_filter_values(_SW, (k, k, input_dim, output_dim), fan_in, fan_out, he_init, gain)
    filters = lib.param(name + '.Filters', fresh)
    if _SW.weightnorm if weightnorm is None else weightnorm:
        filters = _init.weight_normalised(name, filters, fresh, reduce_axes=(0, 1, 2))
    if mask_type is not None:
        filters = filters * torch.as_tensor(_pixelcnn_mask(mask_type[0], mask_type[1], k, input_dim, output_dim),
                                            device=filters.device)
    bias = _init.zero_bias(name + '.Biases', output_dim) if biases else None
    y = A.conv2d(nchw_to_nhwc_view(inputs), filters, bias, stride=stride, act=_ACT[fused_act], alpha=alpha,
                 bn_stats=bn_stats and fused_act is None)
    out = nhwc_to_nchw_view(y)
    if getattr(y, '_dpig_bnstats', None) is not None:
        out._dpig_bnstats = y._dpig_bnstats
    return out
