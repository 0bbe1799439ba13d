"""DeepFashion 256x256 trainers.  Stage I (model 101), mirroring the reference `trainer_256.py:10-134`:
appearance encoder `GeneratorCNN_ID_Encoder_BodyROIVis(repeat_num+1, roi_size=64)` (:40-41), generator with
`repeat_num-1` levels (:53-55), and the discriminator applied ONCE to the concatenated pair [x; G] (:61-66) --
joint BatchNorm statistics, so g_loss depends on the real half too.  With the hard-coded reshape of
`DCGANDiscriminator` (wgan_gp.py:433) a 256x256 image yields 8 logit rows; `tf.split(D_z, 2)` then
gives the first 8B rows to the real images.

Stage II (run_DF_train.sh:39-77), at the end of this file: model 102 `DPIG_Encoder_subSampleAppNet_GAN_BodyROI_256`
(trainer_256.py:266-400, the appearance-embedding GAN), model 103 `DPIG_PoseRCV_AE_BodyROI_256` (:404-509, the pose
auto-encoder) and model 104 `DPIG_subnetSamplePoseRCV_GAN_BodyROI_256` (:511-700, the pose-embedding GAN)."""
import torch

from . import autograd as A
from . import hip_ops as H
from . import models
from . import slim
from . import tflib as lib
from .trainer import DPIG_Encoder_GAN_BodyROI_FgBg, FlatParams, GradAllReduce, clip_disc_weights, gan_loss, get_optimizers
from .trainer_stage2 import DPIG_PoseRCV_AE_BodyROI, DPIG_subnetSamplePoseRCV_GAN_BodyROI
from .wgan_gp import WGAN_GP


class DPIG_Encoder_GAN_BodyROI_256(DPIG_Encoder_GAN_BodyROI_FgBg):
    def encode(self, batch):
        with slim.variable_scope("Encoder"):
            embs, _, enc_var = models.GeneratorCNN_ID_Encoder_BodyROIVis(
                batch["x"], batch["part_bbox"], batch["part_vis"], self.part_num, 32, self.repeat_num + 1,
                self.conv_hidden_num, self.data_format, activation_fn=slim.relu, keep_part_prob=1.0, roi_size=64,
                reuse=self.built)
        return embs, enc_var

    def generate(self, embs, pose):
        B = embs.shape[0]
        embs_rep = embs.reshape(B, 1, 1, -1).expand(B, self.img_H, self.img_W, embs.shape[1])
        embs_rep._dpig_src = embs     # the builder's collapsed first conv reads the [B,E] source (no expand -> select round trip in autograd)
        with slim.variable_scope("ID_AE"):
            G, _, g_var = self.Generator_fn(embs_rep, pose, self.channel, self.z_num, self.repeat_num - 1,
                                            self.conv_hidden_num, self.data_format, activation_fn=slim.relu,
                                            reuse=self.built)
        return G, g_var

    def disc_pair(self, x, G, need_real=True):
        pair = torch.cat([x, G], dim=0)                       # trainer_256.py:61
        D_z = self.discriminate(pair)
        D_z_pos, D_z_neg = torch.split(D_z, D_z.shape[0] // 2)
        return D_z_pos, D_z_neg


class DPIG_Encoder_subSampleAppNet_GAN_BodyROI_256(object):
    """Model 102 (trainer_256.py:266-400): the stage-I appearance encoder, rebuilt as `GeneratorCNN_ID_Encoder_BodyROI`
    (no visibility flags, repeat_num + 1 levels, 48 x 48 crops: :305-311) and frozen (restored from `pretrained_path`,
    :268-271, 292-294), yields the real embeddings [B, 224]; ONE `GaussianFCRes` mapper under `Gaussian_FC` (width 512,
    LeakyReLU 0.2, :322-324) the fake ones; the critic `FCDis_Discriminator.*` sees the pair [real; fake] in one call
    (:326-331); MODE='wgan' (:300-302): g = -mean D(fake), d = mean D(fake) - mean D(real), RMSProp, critic clipped to
    +-0.01 after each of its 5 updates per step (:362-373).  Only the mapper and the critic are trained (:352-354)."""

    def __init__(self, config, device):
        self.config = config
        self.device = torch.device(device)
        self.batch_size = config.batch_size
        self.img_H, self.img_W = config.img_H, config.img_W
        self.repeat_num, self.conv_hidden_num = config.repeat_num, config.conv_hidden_num
        self.data_format = config.data_format
        self.part_num = 7
        lib.set_device(self.device)
        self.g_lr = torch.full((1,), config.g_lr, dtype=torch.float32, device=self.device)
        self.d_lr = torch.full((1,), config.d_lr, dtype=torch.float32, device=self.device)
        self.wgan_gp_encoder = WGAN_GP(DATA_DIR='', MODE='wgan', DIM=64, BATCH_SIZE=self.batch_size, ITERS=200000, LAMBDA=10,
                                       G_OUTPUT_DIM=7 * 32)
        self.step = 0
        self.built = False

    def encode(self, batch):
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        with torch.no_grad(), slim.variable_scope("Encoder"):
            embs, _, enc_var = models.GeneratorCNN_ID_Encoder_BodyROI(
                batch["x"], batch["part_bbox"], self.part_num, 32, self.repeat_num + 1, self.conv_hidden_num,
                self.data_format, activation_fn=slim.relu, keep_part_prob=1.0, reuse=self.built)
        return H.to_f32(embs), enc_var

    def mapper(self, dim, z=None):
        with slim.variable_scope("Gaussian_FC"):
            return models.GaussianFCRes([self.batch_size, dim], dim, repeat_num=4, hidden_num=512, data_format=self.data_format,
                                        activation_fn=slim.leaky_relu, z=z, device=self.device, reuse=self.built)

    def critic_pair(self, real, fake):
        pair = torch.cat([real, fake], dim=0)                  # trainer_256.py:326
        D_z = self.wgan_gp_encoder.FCDiscriminator(pair, input_dim=pair.shape[-1], FC_DIM=512, n_layers=3, name='FCDis_')
        return torch.split(D_z, D_z.shape[0] // 2)

    def init_net(self, batch):
        real, self.Encoder_var = self.encode(batch)
        self.dim = real.shape[1]
        with torch.no_grad():
            fake, g_var = self.mapper(self.dim)
            self.critic_pair(real, fake)
        self.built = True
        from . import tfckpt
        tfckpt.restore_from_config(self.config)
        self.G_var_app_embs = g_var
        self.D_var_embs = lib.params_with_name('FCDis_Discriminator.')
        self.G_flat, self.D_flat = FlatParams(self.G_var_app_embs), FlatParams(self.D_var_embs)
        self.g_opt, self.d_opt = get_optimizers(self.wgan_gp_encoder, self.G_flat, self.D_flat, self.g_lr, self.d_lr)
        self.allreduce = GradAllReduce()
        self.allreduce.broadcast(self.G_flat.flat)
        self.allreduce.broadcast(self.D_flat.flat)
        if getattr(self.config, "compute_dtype", "f32") in ("bf16", "bf16x3"):
            self.encoder_shadows = H.FilterShadows(self.Encoder_var, split=self.config.compute_dtype == "bf16x3")
        if getattr(self.config, "compute_dtype", "f32") == "f32w":
            self.encoder_wino = H.WinoFilters(self.Encoder_var)      # frozen: Winograd images made once

    def g_optim_embs(self, batch, z=None):
        self.G_flat.zero_grad()
        self.D_flat.set_requires_grad(False)
        real, _ = self.encode(batch)
        fake, _ = self.mapper(self.dim, z)
        _, D_neg = self.critic_pair(real, fake)
        g_loss, _ = gan_loss(self.wgan_gp_encoder, None, D_neg)
        with A.wgrad_overlap():
            g_loss.backward()
        self.D_flat.set_requires_grad(True)
        self.G_flat.finalize()
        self.g_opt.step(self.allreduce(self.G_flat.grad))
        return g_loss.detach()

    def d_optim_embs(self, batch, z=None):
        self.D_flat.zero_grad()
        real, _ = self.encode(batch)
        with torch.no_grad():
            fake, _ = self.mapper(self.dim, z)
        D_pos, D_neg = self.critic_pair(real, fake)
        _, d_loss = gan_loss(self.wgan_gp_encoder, D_pos, D_neg)
        with A.wgrad_overlap():
            d_loss.backward()
        self.D_flat.finalize()
        self.d_opt.step(self.allreduce(self.D_flat.grad))
        if self.wgan_gp_encoder.MODE == 'wgan':
            clip_disc_weights(self.D_flat)
        return d_loss.detach()

    def train_step(self, batch):
        """trainer_256.py:362-373.  `batch`: one batch, or a sequence -- every `sess.run(g_optim_embs / d_optim_embs)` of the reference
        dequeues a fresh batch, so the mapper update and the five critic updates of a step see six batches (a single one is reused)."""
        many = isinstance(batch, (list, tuple))
        nxt = iter(range(10 ** 9))
        pick = (lambda: batch[next(nxt) % len(batch)]) if many else (lambda: batch)
        out = {}
        if self.step > 0:
            out["g_loss_embs"] = self.g_optim_embs(pick())
        iters = 1 if self.wgan_gp_encoder.MODE in ('dcgan', 'lsgan') else self.wgan_gp_encoder.CRITIC_ITERS
        for _ in range(iters):
            out["d_loss_embs"] = self.d_optim_embs(pick())
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out


class DPIG_PoseRCV_AE_BodyROI_256(DPIG_PoseRCV_AE_BodyROI):
    """Model 103 (trainer_256.py:404-509): the pose auto-encoder at 256 x 256 -- the graph of model 2 (trainer.py:626-713) with
    the keypoint rows / columns normalised by img_H = img_W = 256 (:448-452)."""


class DPIG_subnetSamplePoseRCV_GAN_BodyROI_256(DPIG_subnetSamplePoseRCV_GAN_BodyROI):
    """Model 104 (trainer_256.py:511-700): the pose-embedding GAN at 256 x 256 -- the graph of model 4 (trainer.py:868-1040)
    with the 256 x 256 normalisation; the frozen appearance encoder / generator the reference also builds (:598-615) take no
    part in either loss (they serve `generate()`, the inference harness)."""
