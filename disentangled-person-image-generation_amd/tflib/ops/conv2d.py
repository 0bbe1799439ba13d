"""tflib.ops.conv2d drop-in (reference tflib/ops/conv2d.py:6-123)."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU
from ._layout import nchw_to_nhwc_view, nhwc_to_nchw_view

_default_weightnorm = False


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


_weights_stdev = None


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def Conv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, mask_type=None, stride=1,
           weightnorm=None, biases=True, gain=1., fused_act=None, alpha=0.2):
    """
    inputs: tensor of shape (batch size, num channels, height, width)
    mask_type: one of None, 'a', 'b'

    returns: tensor of shape (batch size, num channels, height, width)

    Same signature and semantics as the reference (conv2d.py:20): SAME padding, HWIO filter
    `<name>.Filters` initialised uniform(+-stdev*sqrt(3)), optional weight-norm `<name>.g`,
    PixelCNN mask, bias `<name>.Biases`.  Extension (keyword-only in spirit): `fused_act` in
    {None,'relu','lrelu'} fuses the activation the caller would apply next into the conv epilogue.
    """
    mask = None
    if mask_type is not None:
        mask_type, mask_n_channels = mask_type
        mask = np.ones((filter_size, filter_size, input_dim, output_dim), dtype='float32')
        center = filter_size // 2
        # Mask out future locations; filter shape is (height, width, input channels, output channels)
        mask[center + 1:, :, :, :] = 0.
        mask[center, center + 1:, :, :] = 0.
        # Mask out future channels
        for i in range(mask_n_channels):
            for j in range(mask_n_channels):
                if (mask_type == 'a' and i >= j) or (mask_type == 'b' and i > j):
                    mask[center, center, i::mask_n_channels, j::mask_n_channels] = 0.

    def uniform(stdev, size):
        return np.random.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')

    fan_in = input_dim * filter_size ** 2
    fan_out = output_dim * filter_size ** 2 / (stride ** 2)
    if mask_type is not None:  # only approximately correct
        fan_in /= 2.
        fan_out /= 2.
    if he_init:
        filters_stdev = np.sqrt(4. / (fan_in + fan_out))
    else:  # Normalized init (Glorot & Bengio)
        filters_stdev = np.sqrt(2. / (fan_in + fan_out))

    if name + '.Filters' in lib._params:
        filter_values = None
    elif _weights_stdev is not None:
        filter_values = uniform(_weights_stdev, (filter_size, filter_size, input_dim, output_dim))
    else:
        filter_values = uniform(filters_stdev, (filter_size, filter_size, input_dim, output_dim))
    if filter_values is not None:
        filter_values *= gain
    filters = lib.param(name + '.Filters', filter_values)

    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        if name + '.g' in lib._params:
            target_norms = lib.param(name + '.g')
        else:
            init = filter_values if filter_values is not None else filters.detach().cpu().numpy()
            target_norms = lib.param(name + '.g', np.sqrt(np.sum(np.square(init), axis=(0, 1, 2))))
        norms = torch.sqrt(torch.sum(filters * filters, dim=(0, 1, 2)))
        filters = filters * (target_norms / norms)
    if mask is not None:
        filters = filters * torch.as_tensor(mask, device=filters.device)

    _biases = lib.param(name + '.Biases', np.zeros(output_dim, dtype='float32')) if biases else None

    act = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}[fused_act]
    x = nchw_to_nhwc_view(inputs)
    y = A.conv2d(x, filters, _biases, stride=stride, act=act, alpha=alpha)
    return nhwc_to_nchw_view(y)
