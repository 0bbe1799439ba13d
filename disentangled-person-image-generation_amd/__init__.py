"""dpig_amd -- MI355X-native conv hot path of Disentangled-Person-Image-Generation.

The directory is named `disentangled-person-image-generation_amd` (not an importable identifier);
import it as `dpig_amd` through the repo-root shim `dpig_amd.py`.

Layout:
  csrc/            hand-written gfx950 HIP kernels + the C ABI (include/dpig_hip.h)
  _lib.py          ctypes binding (fails loudly when libdpig_hip.so is missing: no CPU fallback)
  hip_ops.py       functional tensor-in/tensor-out layer over the C ABI
  autograd.py      torch.autograd.Function wrappers (plumbing only)
  tflib/           drop-in mirror of the reference's tflib param registry + ops signatures
  slim.py          the second face of the boundary: slim.conv2d / slim.fully_connected call shape
  models.py        Fg/Bg/appearance encoders, U-Net decoder, FC nets (reference models.py)
  wgan_gp.py       WGAN_GP mode object + DCGAN / FC discriminators (reference wgan_gp.py)
  trainer.py       stage-I G+D step: losses, TF-Adam, step order (reference trainer.py)
"""
__version__ = "0.1.0"
