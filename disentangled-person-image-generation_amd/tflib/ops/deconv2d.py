"""tflib.ops.deconv2d drop-in: `Deconv2D(name, input_dim, output_dim, filter_size, inputs, ...)`, the stride-2 SAME
transposed convolution of the reference (tflib/ops/deconv2d.py:21-115; filter `<name>.Filters` stored
[k, k, output_dim, input_dim] as `tf.nn.conv2d_transpose` wants it), computed by `dpig_conv2d_dgrad`."""
from ... import autograd as A
from ... import tflib as lib
from . import _init
from ._layout import nchw_to_nhwc_view, nhwc_to_nchw_view

_SW = _init.Switches()


def enable_default_weightnorm():
    _SW.weightnorm = True


def set_weights_stdev(weights_stdev):
    _SW.stdev = weights_stdev


def unset_weights_stdev():
    _SW.stdev = None


def Deconv2D(name, input_dim, output_dim, filter_size, inputs, he_init=True, weightnorm=None, biases=True,
             gain=1., mask_type=None):
    """inputs: (batch, input_dim, H, W) -> (batch, output_dim, 2H, 2W)."""
    if mask_type is not None:
        raise Exception('Unsupported configuration')
    k, stride = filter_size, 2
    fan_in, fan_out = input_dim * k ** 2 / (stride ** 2), output_dim * k ** 2
    fresh = None
    if name + '.Filters' not in lib._params:
        fresh = _init.conv_filter_values(_SW, (k, k, output_dim, input_dim), fan_in, fan_out, he_init, gain)
    filters = lib.param(name + '.Filters', fresh)
    if _SW.weightnorm if weightnorm is None else weightnorm:
        filters = _init.weight_normalised(name, filters, fresh, reduce_axes=(0, 1, 3), broadcast=lambda r: r.unsqueeze(1))
    bias = _init.zero_bias(name + '.Biases', output_dim) if biases else None
    return nhwc_to_nchw_view(A.conv2d_transpose(nchw_to_nhwc_view(inputs), filters, bias))
