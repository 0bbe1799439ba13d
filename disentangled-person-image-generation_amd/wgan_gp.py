"""GAN-mode object + discriminators of the hot path, mirroring the reference `wgan_gp.py`
(class WGAN_GP :95-117, Batchnorm switch :34-40, LeakyReLU :23-24, FCDiscriminator :399-405,
DCGANDiscriminator :407-440) on top of the `tflib.ops` drop-in operators.

Inputs/outputs are logical NCHW like the reference; a tensor obtained from an NHWC image by
`x.permute(0,3,1,2)` (trainer.py:601 `tf.transpose(x,[0,3,1,2])`) is consumed without a copy.
The activation that follows each conv / norm in the reference is folded into that op's kernel
(`fused_act=`); the arithmetic and parameter names are unchanged.
"""
from . import tflib as lib
from .tflib import ops  # noqa: F401  (lib.ops.* like the reference's `import tflib.ops.conv2d`)
from . import autograd as A
from ._lib import ACT_LRELU


def LeakyReLU(x, alpha=0.2):
    """wgan_gp.py:23-24: tf.maximum(alpha*x, x)."""
    return A.activation(x.contiguous(), ACT_LRELU, alpha)


def ReLULayer(name, n_in, n_out, inputs):
    return lib.ops.linear.Linear(name + '.Linear', n_in, n_out, inputs, initialization='he', fused_act='relu')


def LeakyReLULayer(name, n_in, n_out, inputs):
    return lib.ops.linear.Linear(name + '.Linear', n_in, n_out, inputs, initialization='he', fused_act='lrelu')


def Batchnorm(name, axes, inputs, MODE, fused_act=None):
    """wgan_gp.py:34-40: LayerNorm inside the discriminator for wgan-gp, BatchNorm otherwise."""
    if ('Discriminator' in name) and (MODE == 'wgan-gp'):
        if axes != [0, 2, 3]:
            raise Exception('Layernorm over non-standard axes is unsupported')
        return lib.ops.layernorm.Layernorm(name, [1, 2, 3], inputs, fused_act=fused_act)
    else:
        return lib.ops.batchnorm.Batchnorm(name, axes, inputs, fused=True, fused_act=fused_act)


class WGAN_GP(object):
    def __init__(self, DATA_DIR='', MODE='wgan-gp', DIM=64, BATCH_SIZE=64, ITERS=200000, LAMBDA=10,
                 G_OUTPUT_DIM=128 * 64 * 3, IMG_H=128, IMG_W=64, verbose=False):
        self.DATA_DIR = DATA_DIR
        self.MODE = MODE  # dcgan, wgan, wgan-gp, lsgan
        self.DIM = DIM
        self.BATCH_SIZE = BATCH_SIZE
        self.ITERS = ITERS
        self.LAMBDA = LAMBDA
        self.G_OUTPUT_DIM = G_OUTPUT_DIM
        self.IMG_H = IMG_H
        self.IMG_W = IMG_W
        self.CRITIC_ITERS = 5  # How many iterations to train the critic for
        self.N_GPUS = 1
        self.DEVICES = ['/gpu:{}'.format(i) for i in range(self.N_GPUS)]
        if verbose:
            lib.print_model_settings(locals().copy())

    def FCDiscriminator(self, inputs, input_dim, FC_DIM=512, n_layers=3, reuse=False, name=''):
        """wgan_gp.py:399-405 (stage-II embedding critic)."""
        output = LeakyReLULayer(name + 'Discriminator.Input', input_dim, FC_DIM, inputs)
        for i in range(n_layers):
            output = LeakyReLULayer(name + 'Discriminator.{}'.format(i), FC_DIM, FC_DIM, output)
        output = lib.ops.linear.Linear(name + 'Discriminator.Out', FC_DIM, 1, output)
        return output.reshape(-1)

    def DCGANDiscriminator(self, inputs, input_dim=3, dim=64, bn=True, nonlinearity=LeakyReLU, name=''):
        """wgan_gp.py:407-440.  Conv5x5s2 -> LReLU -> [Conv5x5s2 -> BN|LN -> LReLU] x3 -> reshape
        [-1, 8*4*8*dim] -> Linear -> [-1].  The hard-coded reshape is kept: at 256x256 it yields 8
        logit rows per image."""
        output = inputs
        fuse = 'lrelu' if nonlinearity is LeakyReLU else None

        def post(o):
            return o if fuse is not None else nonlinearity(o)

        # a BatchNorm (not the wgan-gp LayerNorm) follows the conv: let the conv epilogue carry the batch statistics
        bn_stats = bool(bn) and self.MODE != 'wgan-gp'

        lib.ops.conv2d.set_weights_stdev(0.02)
        lib.ops.deconv2d.set_weights_stdev(0.02)
        lib.ops.linear.set_weights_stdev(0.02)

        output = lib.ops.conv2d.Conv2D(name + 'Discriminator.1', input_dim, dim, 5, output, stride=2, fused_act=fuse)
        output = post(output)

        output = lib.ops.conv2d.Conv2D(name + 'Discriminator.2', dim, 2 * dim, 5, output, stride=2,
                                       fused_act=None if bn else fuse, bn_stats=bn_stats)
        if bn:
            output = Batchnorm(name + 'Discriminator.BN2', [0, 2, 3], output, self.MODE, fused_act=fuse)
        output = post(output)

        output = lib.ops.conv2d.Conv2D(name + 'Discriminator.3', 2 * dim, 4 * dim, 5, output, stride=2,
                                       fused_act=None if bn else fuse, bn_stats=bn_stats)
        if bn:
            output = Batchnorm(name + 'Discriminator.BN3', [0, 2, 3], output, self.MODE, fused_act=fuse)
        output = post(output)

        output = lib.ops.conv2d.Conv2D(name + 'Discriminator.4', 4 * dim, 8 * dim, 5, output, stride=2,
                                       fused_act=None if bn else fuse, bn_stats=bn_stats)
        if bn:
            output = Batchnorm(name + 'Discriminator.BN4', [0, 2, 3], output, self.MODE, fused_act=fuse)
        output = post(output)

        # tf.reshape on the logical NCHW tensor: flatten order (c, h, w)
        output = A.nchw_flatten(output, 8 * 4 * 8 * dim)       # (one transpose kernel when the data is physically NHWC)
        output = lib.ops.linear.Linear(name + 'Discriminator.Output', 8 * 4 * 8 * dim, 1, output)

        lib.ops.conv2d.unset_weights_stdev()
        lib.ops.deconv2d.unset_weights_stdev()
        lib.ops.linear.unset_weights_stdev()

        return output.reshape(-1)
