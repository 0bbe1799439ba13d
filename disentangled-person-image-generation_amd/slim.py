"""The second face of the drop-in boundary: the `slim.conv2d` /
`slim.fully_connected` call shape that models.py uses for the encoders and the U-Net decoder
(reference models.py:396-573), implemented on the same HIP conv/linear kernels as tflib.ops.

TF-slim conventions reproduced:
  * SAME padding, bias always on, activation applied after the bias;
  * weights xavier-uniform  U(+-sqrt(6/(fan_in+fan_out))), fan_in=k*k*Cin, fan_out=k*k*Cout,
    biases zero; filters HWIO; FC weights [Cin, Cout];
  * variables named `<scopes>/Conv[_k]/{weights,biases}` / `<scopes>/fully_connected[_k]/...`
    in creation order, re-entering a scope with reuse=True resolves to the same variables.
Variables live in the tflib registry (`lib.param`) under those names.
"""
import contextlib

import numpy as np
import torch

from . import autograd as A
from . import tflib as lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU

# ---- minimal tf.variable_scope ----------------------------------------------------------------
_scope_stack = []          # list of scope names
_counters = {}             # "<path>/<default_name>" -> number of times opened


def _path():
    return "/".join(_scope_stack)


@contextlib.contextmanager
def variable_scope(name, reuse=False):
    """tf.variable_scope(name, reuse=...): yields the full scope path.  On exit the sub-scope
    counters are cleared (TF's close_variable_subscopes), so re-entering the same scope
    regenerates the same `Conv`, `Conv_1`, ... names -- with reuse=True that is weight sharing."""
    _scope_stack.append(name)
    path = _path()
    try:
        yield path
    finally:
        _scope_stack.pop()
        for k in [k for k in _counters if k.startswith(path + "/")]:
            del _counters[k]


def _unique(default_name):
    key = _path() + "/" + default_name
    n = _counters.get(key, 0)
    _counters[key] = n + 1
    return key if n == 0 else "%s_%d" % (key, n)


def get_variables(scope_path):
    """tf.contrib.framework.get_variables(vs): all variables whose name starts with the scope."""
    return [p for n, p in lib._params.items() if n.startswith(scope_path + "/")]


def reset_scopes():
    _scope_stack.clear()
    _counters.clear()


# ---- activation handling ----------------------------------------------------------------------
def relu(x):
    """tf.nn.relu stand-in; recognised by conv2d/fully_connected and fused into the kernel epilogue."""
    return A.activation(x.contiguous(), ACT_RELU)


def leaky_relu(x, alpha=0.2):
    return A.activation(x.contiguous(), ACT_LRELU, alpha)


def _act_code(activation_fn):
    if activation_fn is None:
        return ACT_NONE, None
    if activation_fn is relu or activation_fn is torch.relu or activation_fn is torch.nn.functional.relu:
        return ACT_RELU, None
    if activation_fn is leaky_relu:
        return ACT_LRELU, None
    return ACT_NONE, activation_fn      # foreign activation: applied unfused on the result


def _xavier_uniform(shape, fan_in, fan_out):
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return np.random.uniform(-limit, limit, size=shape).astype("float32")


def _conv_vars(scope, k, cin, cout):
    wname, bname = scope + "/weights", scope + "/biases"
    w_init = None if wname in lib._params else _xavier_uniform((k, k, cin, cout), k * k * cin, k * k * cout)
    w = lib.param(wname, w_init)
    b = lib.param(bname, np.zeros(cout, dtype="float32"))
    return w, b


def conv2d(inputs, num_outputs, kernel_size, stride=1, activation_fn=relu, data_format="NHWC", scope=None,
           upsample2x=False, out=None):
    """slim.conv2d(x, num_outputs, k, s, activation_fn=..., data_format=...) on NHWC tensors.
    `upsample2x=True` (extension) means "the input is utils.upscale(x, 2)": the nearest-neighbour
    upsample of models.py:569 is folded into the 1x1 conv (exact: the two ops commute)."""
    if data_format != "NHWC":
        raise Exception("only NHWC is supported (main.py:18 forces NHWC for the generator path)")
    name = scope if scope is not None else _unique("Conv")
    cin = inputs.shape[-1]
    w, b = _conv_vars(name, kernel_size, cin, num_outputs)
    act, foreign = _act_code(activation_fn)
    y = A.conv2d(inputs, w, b, stride=stride, act=act, upsample2x=upsample2x, out=out if foreign is None else None)
    return foreign(y) if foreign is not None else y


def conv2d_tiled_embedding(emb, pose, num_outputs, scope=None):
    """slim.conv2d(concat([tile(emb), pose], -1), num_outputs, 3, 1, activation_fn=relu) -- same `Conv`
    variables ([3,3,E+P,num_outputs] weights), evaluated without materialising the tiled embedding
    (autograd._TiledEmbConvFn)."""
    name = scope if scope is not None else _unique("Conv")
    cin = emb.shape[-1] + pose.shape[-1]
    w, b = _conv_vars(name, 3, cin, num_outputs)
    return A.tiled_emb_conv(emb, pose, w, b)


def res_block(inputs, channel_num, kernel_size=3, activation_fn=relu, data_format="NHWC", out=None):
    """The three reference lines
            x = slim.conv2d(x, channel_num, 3, 1, activation_fn=...)
            x = slim.conv2d(x, channel_num, 3, 1, activation_fn=...)
            x = x + res
    (models.py:398-400, 425-427, 458-460, 534-536, 564-566) as one fused op: same two `Conv`
    variable scopes, same arithmetic; the skip add rides in the second conv's epilogue and the
    backward pass fuses the ReLU masks / skip-gradient add into the dgrad epilogues."""
    if data_format != "NHWC":
        raise Exception("only NHWC is supported")
    cin = inputs.shape[-1]
    n1 = _unique("Conv")
    n2 = _unique("Conv")
    w1, b1 = _conv_vars(n1, kernel_size, cin, channel_num)
    w2, b2 = _conv_vars(n2, kernel_size, channel_num, channel_num)
    act, foreign = _act_code(activation_fn)
    if act == ACT_RELU and cin == channel_num and foreign is None:
        return A.resblock(inputs, w1, b1, w2, b2, out=out)      # `out` (extension): a channel slice to write the result into
    x = A.conv2d(inputs, w1, b1, act=act)
    x = foreign(x) if foreign is not None else x
    x = A.conv2d(x, w2, b2, act=act)
    x = foreign(x) if foreign is not None else x
    return x + inputs


def fully_connected(inputs, num_outputs, activation_fn=relu, scope=None):
    """slim.fully_connected(x, n, activation_fn=...) (models.py:431,464,545,554; FC nets 474-515)."""
    name = scope if scope is not None else _unique("fully_connected")
    cin = inputs.shape[-1]
    wname = name + "/weights"
    w_init = None if wname in lib._params else _xavier_uniform((cin, num_outputs), cin, num_outputs)
    w = lib.param(wname, w_init)
    b = lib.param(name + "/biases", np.zeros(num_outputs, dtype="float32"))
    act, foreign = _act_code(activation_fn)
    y = A.linear(inputs.reshape(-1, cin), w, b, act)
    y = y.reshape(tuple(inputs.shape[:-1]) + (num_outputs,))
    return foreign(y) if foreign is not None else y
