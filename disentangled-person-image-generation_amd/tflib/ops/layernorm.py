"""tflib.ops.layernorm drop-in (reference tflib/ops/layernorm.py:6-20)."""
import numpy as np
import torch

from ... import autograd as A
from ... import tflib as lib
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU
from ._layout import nchw_to_nhwc_view, nhwc_to_nchw_view


def Layernorm(name, norm_axes, inputs, fused_act=None, alpha=0.2):
    """Per-sample moments over `norm_axes`; per-'neuron' (first of norm_axes) offset/scale; eps 1e-5.
    The conv case norm_axes == [1,2,3] on logical-NCHW data runs on the HIP kernel; other axes use
    tensor plumbing (never reached by the reference, wgan_gp.py:36-37 raises for them)."""
    act = {None: ACT_NONE, 'relu': ACT_RELU, 'lrelu': ACT_LRELU}[fused_act]
    n_neurons = inputs.shape[norm_axes[0]]
    offset = lib.param(name + '.offset', np.zeros(n_neurons, dtype='float32'))
    scale = lib.param(name + '.scale', np.ones(n_neurons, dtype='float32'))
    if list(norm_axes) == [1, 2, 3] and inputs.dim() == 4:
        x = nchw_to_nhwc_view(inputs)
        y = A.layernorm(x, scale, offset, 1e-5, act, alpha)
        return nhwc_to_nchw_view(y)
    mean = inputs.mean(dim=list(norm_axes), keepdim=True)
    var = inputs.var(dim=list(norm_axes), unbiased=False, keepdim=True)
    bshape = [-1] + [1 for _ in range(len(norm_axes) - 1)]
    y = (inputs - mean) / torch.sqrt(var + 1e-5) * scale.reshape(bshape) + offset.reshape(bshape)
    if act != ACT_NONE:
        y = A.activation(y.contiguous(), act, alpha)
    return y
