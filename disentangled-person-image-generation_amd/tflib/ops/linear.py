"""tflib.ops.linear drop-in (reference tflib/ops/linear.py:6-147)."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU

_default_weightnorm = False


def enable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = True


def disable_default_weightnorm():
    global _default_weightnorm
    _default_weightnorm = False


_weights_stdev = None


def set_weights_stdev(weights_stdev):
    global _weights_stdev
    _weights_stdev = weights_stdev


def unset_weights_stdev():
    global _weights_stdev
    _weights_stdev = None


def Linear(name, input_dim, output_dim, inputs, biases=True, initialization=None, weightnorm=None, gain=1.,
           fused_act=None, alpha=0.2):
    """
    initialization: None, `lecun`, 'glorot', `he`, 'glorot_he', `orthogonal`, `("uniform", range)`
    Same init table as the reference (linear.py:48-104), `<name>.W` [in,out], `<name>.b`.
    """
    def uniform(stdev, size):
        if _weights_stdev is not None:
            stdev = _weights_stdev
        return np.random.uniform(low=-stdev * np.sqrt(3), high=stdev * np.sqrt(3), size=size).astype('float32')

    weight_values = None
    if name + '.W' not in lib._params:
        if initialization == 'lecun':
            weight_values = uniform(np.sqrt(1. / input_dim), (input_dim, output_dim))
        elif initialization == 'glorot' or (initialization is None):
            weight_values = uniform(np.sqrt(2. / (input_dim + output_dim)), (input_dim, output_dim))
        elif initialization == 'he':
            weight_values = uniform(np.sqrt(2. / input_dim), (input_dim, output_dim))
        elif initialization == 'glorot_he':
            weight_values = uniform(np.sqrt(4. / (input_dim + output_dim)), (input_dim, output_dim))
        elif initialization == 'orthogonal' or (initialization is None and input_dim == output_dim):
            # From lasagne
            def sample(shape):
                if len(shape) < 2:
                    raise RuntimeError("Only shapes of length 2 or more are supported.")
                flat_shape = (shape[0], int(np.prod(shape[1:])))
                a = np.random.normal(0.0, 1.0, flat_shape)
                u, _, v = np.linalg.svd(a, full_matrices=False)
                q = u if u.shape == flat_shape else v
                return q.reshape(shape).astype('float32')
            weight_values = sample((input_dim, output_dim))
        elif initialization[0] == 'uniform':
            weight_values = np.random.uniform(low=-initialization[1], high=initialization[1],
                                              size=(input_dim, output_dim)).astype('float32')
        else:
            raise Exception('Invalid initialization!')
        weight_values *= gain
    weight = lib.param(name + '.W', weight_values)

    if weightnorm is None:
        weightnorm = _default_weightnorm
    if weightnorm:
        if name + '.g' in lib._params:
            target_norms = lib.param(name + '.g')
        else:
            init = weight_values if weight_values is not None else weight.detach().cpu().numpy()
            target_norms = lib.param(name + '.g', np.sqrt(np.sum(np.square(init), axis=0)))
        norms = torch.sqrt(torch.sum(weight * weight, dim=0))
        weight = weight * (target_norms / norms)

    b = lib.param(name + '.b', np.zeros((output_dim,), dtype='float32')) if biases else None
    act = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}[fused_act]
    if inputs.dim() == 2:
        return A.linear(inputs, weight, b, act, alpha)
    lead = inputs.shape[:-1]
    y = A.linear(inputs.reshape(-1, input_dim), weight, b, act, alpha)
    return y.reshape(tuple(lead) + (output_dim,))
