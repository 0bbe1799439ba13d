"""tflib.ops.deconv2d drop-in (reference tflib/ops/deconv2d.py:6-115): stride-2 SAME transposed conv."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
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


def Deconv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, weightnorm=None, biases=True,
             gain=1., mask_type=None):
    """
    inputs: tensor of shape (batch size, input_dim, height, width)   [logical NCHW]
    returns: tensor of shape (batch size, output_dim, 2*height, 2*width)
    """
    if mask_type is not None:
        raise Exception('Unsupported configuration')

    def uniform(stdev, size):
        return np.random.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')

    stride = 2
    fan_in = input_dim * filter_size ** 2 / (stride ** 2)
    fan_out = output_dim * filter_size ** 2
    if he_init:
        filters_stdev = np.sqrt(4. / (fan_in + fan_out))
    else:  # Normalized init (Glorot & Bengio)
        filters_stdev = np.sqrt(2. / (fan_in + fan_out))

    if name + '.Filters' in lib._params:
        filter_values = None
    elif _weights_stdev is not None:
        filter_values = uniform(_weights_stdev, (filter_size, filter_size, output_dim, input_dim))
    else:
        filter_values = uniform(filters_stdev, (filter_size, filter_size, output_dim, input_dim))
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
            target_norms = lib.param(name + '.g', np.sqrt(np.sum(np.square(init), axis=(0, 1, 3))))
        norms = torch.sqrt(torch.sum(filters * filters, dim=(0, 1, 3)))
        filters = filters * (target_norms / norms).unsqueeze(1)

    _biases = lib.param(name + '.Biases', np.zeros(output_dim, dtype='float32')) if biases else None
    x = nchw_to_nhwc_view(inputs)
    y = A.conv2d_transpose(x, filters, _biases)
    return nhwc_to_nchw_view(y)
