"""Operator library with the reference's `tflib.ops` signatures, backed by the
gfx950 kernels of libdpig_hip.so.  Tensors are logical NCHW at this boundary, like the reference
(tflib/ops/conv2d.py:21-26); physically they are NHWC (torch channels_last), so the
`tf.transpose(x,[0,3,1,2])` of trainer.py:601-602 is a free view here."""
from . import batchnorm, conv2d, deconv2d, layernorm, linear  # noqa: F401
