"""tflib.ops.linear drop-in: `Linear(name, input_dim, output_dim, inputs, ...)` with the reference's signature,
variable names (`<name>.W` [in,out], `<name>.b`, `<name>.g`) and initialiser schemes
(reference tflib/ops/linear.py:28-147), computing `x @ W + b` with `dpig_linear_*`."""
from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU
from . import _init

_SW = _init.Switches()
_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}


def enable_default_weightnorm():
    _SW.weightnorm = True


def disable_default_weightnorm():
    _SW.weightnorm = False


def set_weights_stdev(weights_stdev):
    _SW.stdev = weights_stdev


def unset_weights_stdev():
    _SW.stdev = None


def Linear(name, input_dim, output_dim, inputs, biases=True, initialization=None, weightnorm=None, gain=1.,
           fused_act=None, alpha=0.2):
    """initialization: None (Glorot), 'lecun', 'glorot', 'he', 'glorot_he', 'orthogonal' or ('uniform', range).
    `fused_act` in {None,'relu','lrelu'} (extension) folds the caller's next activation into the GEMM epilogue.
    Inputs of any rank: the last axis is the feature axis."""
    fresh = None
    if name + '.W' not in lib._params:
        fresh = _init.linear_weight_values(_SW, initialization, input_dim, output_dim, gain)
    W = lib.param(name + '.W', fresh)
    if _SW.weightnorm if weightnorm is None else weightnorm:
        W = _init.weight_normalised(name, W, fresh, reduce_axes=0)
    b = _init.zero_bias(name + '.b', (output_dim,)) if biases else None
    lead = tuple(inputs.shape[:-1])
    y = A.linear(inputs.reshape(-1, input_dim) if inputs.dim() != 2 else inputs, W, b, _ACT[fused_act], alpha)
    return y if inputs.dim() == 2 else y.reshape(lead + (output_dim,))
