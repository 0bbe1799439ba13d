"""Drop-in mirror of the reference's `tflib` parameter registry (tflib/__init__.py:8-48).

`param(name, init_value)` creates a named trainable tensor on first use and returns the SAME
tensor on every later call -- this is what lets `DCGANDiscriminator` be called twice (real, fake)
with shared weights (trainer.py:601-602).  `params_with_name(substr)` selects trainables by
substring (trainer.py:603).  Module-global, unsynchronised state, exactly like the reference.

Parameters are torch tensors in HBM (fp32, requires_grad for trainables).  Where the reference
holds `tf.Variable`s, we hold `torch.nn.Parameter`s; non-trainable ones (batchnorm.py:26-27
`moving_mean/variance`) are plain tensors and, as in the reference (SURVEY C-7), still show up
in `params_with_name`.
"""
import numpy as np
import torch

_params = {}
_param_aliases = {}
_device = None


def set_device(device):
    """Where new params are created (the reference relies on TF's default device placement)."""
    global _device
    _device = torch.device(device) if device is not None else None


def get_device():
    if _device is not None:
        return _device
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def param(name, *args, **kwargs):
    """tflib/__init__.py:10-34.  `args[0]` is the initial value (numpy array / tensor / scalar);
    `trainable=False` makes a buffer instead of a trainable parameter."""
    if name not in _params:
        if not args:
            raise Exception("param(%r): initial value required for a new parameter" % name)
        trainable = kwargs.pop("trainable", True)
        kwargs.pop("name", None)
        init = args[0]
        if isinstance(init, torch.Tensor):
            t = init.detach().to(device=get_device(), dtype=torch.float32).clone()
        else:
            t = torch.as_tensor(np.asarray(init, dtype=np.float32)).to(get_device())
        if trainable:
            t = torch.nn.Parameter(t, requires_grad=True)
        t.param = True
        t.dpig_name = name
        _params[name] = t
    result = _params[name]
    while id(result) in _param_aliases:
        result = _param_aliases[id(result)]
    return result


# The reference's Conv2D / Deconv2D / Linear create their tf.Variable(name=...) INSIDE `with tf.name_scope(name)`
# (tflib/ops/conv2d.py:27,88; linear.py:37,108; deconv2d.py:36,71), so the graph -- and therefore every checkpoint a
# `tf.train.Saver()` writes -- knows them as `<op name>/<op name>.<suffix>` (e.g. `Discriminator.1/Discriminator.1.Filters`,
# `Fg_FCDis_Discriminator.Input.Linear/Fg_FCDis_Discriminator.Input.Linear.W`).  Batchnorm / Layernorm parameters
# (batchnorm.py:23-27, layernorm.py:12-13) are created outside any name scope and keep their registry name.
_TF_SCOPED_SUFFIXES = ('.Filters', '.Biases', '.g', '.W', '.b')


def tf_variable_name(name):
    """Registry name -> the variable name TensorFlow stores it under (checkpoint key; slots append `/Adam` etc.)."""
    if '/' not in name:
        for sfx in _TF_SCOPED_SUFFIXES:
            if name.endswith(sfx) and len(name) > len(sfx):
                return name[:-len(sfx)] + '/' + name
    return name


def params_with_name(name):
    return [p for n, p in _params.items() if name in n]


def named_params_with_name(name):
    """(name, tensor) pairs -- convenience for checkpoint export."""
    return [(n, p) for n, p in _params.items() if name in n]


def delete_all_params():
    _params.clear()


def alias_params(replace_dict):
    for old, new in replace_dict.items():
        _param_aliases[id(old)] = new


def delete_param_aliases():
    _param_aliases.clear()


def print_model_settings(locals_):
    print("Uppercase local vars:")
    all_vars = [(k, v) for (k, v) in locals_.items()
                if (k.isupper() and k != "T" and k != "SETTINGS" and k != "ALL_SETTINGS")]
    for var_name, var_value in sorted(all_vars, key=lambda x: x[0]):
        print("\t{}: {}".format(var_name, var_value))
