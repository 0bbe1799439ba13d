"""Weight creation shared by the conv / deconv / linear ops.

The reference keeps, per op module, two process-wide switches (a forced init scale and default weight
normalisation) and spells the initialiser table out in each op (`tflib/ops/conv2d.py:6-18,77-104`,
`linear.py:6-26,48-104`, `deconv2d.py:6-19,50-87`).  Here the switches are one small object per op module and the
table is data: a fan rule gives the variance, every scheme draws ONE `np.random.uniform` (or the SVD of one normal
draw for 'orthogonal'), so seeded runs create the same numbers the reference's numpy calls would."""
import numpy as np
import torch

from ... import tflib as lib

SQRT3 = float(np.sqrt(3.0))


class Switches(object):
    """Per-module state behind set_weights_stdev()/unset_weights_stdev()/enable_default_weightnorm()."""
    __slots__ = ("stdev", "weightnorm")

    def __init__(self):
        self.stdev = None
        self.weightnorm = False


def draw_uniform(stdev, shape):
    """U(-stdev*sqrt(3), +stdev*sqrt(3)): variance stdev^2, float32."""
    lim = stdev * SQRT3
    return np.random.uniform(low=-lim, high=lim, size=shape).astype('float32')


def draw_orthogonal(shape):
    """Rows/columns of the SVD of a standard-normal matrix (the lasagne recipe the reference cites)."""
    if len(shape) < 2:
        raise RuntimeError("Only shapes of length 2 or more are supported.")
    flat = (shape[0], int(np.prod(shape[1:])))
    u, _, v = np.linalg.svd(np.random.normal(0.0, 1.0, flat), full_matrices=False)
    return (u if u.shape == flat else v).reshape(shape).astype('float32')


# variance of the named fan-based schemes as a function of (fan_in, fan_out)
FAN_VARIANCE = {
    'lecun': lambda fi, fo: 1.0 / fi,
    'glorot': lambda fi, fo: 2.0 / (fi + fo),
    'he': lambda fi, fo: 2.0 / fi,
    'glorot_he': lambda fi, fo: 4.0 / (fi + fo),
}


def conv_filter_values(sw, shape, fan_in, fan_out, he_init, gain):
    """Filter tensor of a (de)convolution: forced scale if set, else sqrt(4/(fi+fo)) ('he_init') or sqrt(2/(fi+fo))."""
    stdev = sw.stdev if sw.stdev is not None else float(np.sqrt((4.0 if he_init else 2.0) / (fan_in + fan_out)))
    return draw_uniform(stdev, shape) * gain


def linear_weight_values(sw, initialization, n_in, n_out, gain):
    """[n_in, n_out] matrix of `Linear`; None means Glorot (the forced scale overrides the fan-based schemes only)."""
    shape = (n_in, n_out)
    scheme = 'glorot' if initialization is None else initialization
    if isinstance(scheme, str) and scheme in FAN_VARIANCE:
        stdev = sw.stdev if sw.stdev is not None else float(np.sqrt(FAN_VARIANCE[scheme](n_in, n_out)))
        values = draw_uniform(stdev, shape)
    elif scheme == 'orthogonal':
        values = draw_orthogonal(shape)
    elif isinstance(scheme, (tuple, list)) and scheme[0] == 'uniform':
        values = np.random.uniform(low=-scheme[1], high=scheme[1], size=shape).astype('float32')
    else:
        raise Exception('Invalid initialization!')
    return values * gain


def weight_normalised(name, weight, fresh_values, reduce_axes, broadcast=None):
    """w * g / ||w|| with the norm over `reduce_axes`; `<name>.g` starts at the initial norms (so the op is the
    identity at creation) and is trained alongside w."""
    if name + '.g' in lib._params:
        g = lib.param(name + '.g')
    else:
        src = fresh_values if fresh_values is not None else weight.detach().cpu().numpy()
        g = lib.param(name + '.g', np.sqrt(np.sum(np.square(src), axis=reduce_axes)))
    ratio = g / torch.sqrt(torch.sum(weight * weight, dim=reduce_axes))
    return weight * (ratio if broadcast is None else broadcast(ratio))


def zero_bias(name, n):
    return lib.param(name, np.zeros(n, dtype='float32'))
