"""Stage-II appearance-embedding GAN (model 3), mirroring the reference
`DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI` (`trainer.py:715-868`):

  * the stage-I encoder E is frozen and only runs forward (`:727-741`) -> real embeddings fg [B,224], bg [B,128];
  * two `GaussianFCRes` mappers (`models.py:474-486`; z ~ N(0, 0.2), 4 residual blocks, width 512 / 256,
    LeakyReLU(0.2)) under scopes `Gaussian_FC_Fg` / `Gaussian_FC_Bg` (`:752-758`);
  * two `FCDiscriminator` critics named `Fg_FCDis_` / `Bg_FCDis_` (`:764-777`), MODE='wgan' (`:720-725`):
    g = -mean(D(fake)), d = mean(D(fake)) - mean(D(real)) (`:218-220`), RMSProp, critic weights clipped to
    +-0.01 after every critic update (`:119-128`);
  * loop (`:821-845`): per step, for Fg then Bg: one mapper update (skipped at step 0) + CRITIC_ITERS=5 critic
    updates, each followed by the clip.

Every critic update re-runs the (conv-heavy) encoder forward, which is where the time goes.
"""
import torch

from . import models
from . import slim
from . import tflib as lib
from . import hip_ops as H
from .trainer import Config, FlatParams, GradAllReduce, clip_disc_weights, gan_loss, get_optimizers
from .wgan_gp import WGAN_GP, LeakyReLU  # noqa: F401  (trainer.py:23 star-import: alpha 0.2)


class DPIG_Encoder_subSampleAppNetFgBg_GAN_BodyROI(object):
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
        self.wgan_gp_fg = WGAN_GP(DATA_DIR='', MODE='wgan', DIM=64, BATCH_SIZE=self.batch_size, ITERS=200000, LAMBDA=10)
        self.wgan_gp_bg = WGAN_GP(DATA_DIR='', MODE='wgan', DIM=64, BATCH_SIZE=self.batch_size, ITERS=200000, LAMBDA=10)
        self.sides = {"fg": dict(scope="Gaussian_FC_Fg", name="Fg_FCDis_", hidden=512, wg=self.wgan_gp_fg),
                      "bg": dict(scope="Gaussian_FC_Bg", name="Bg_FCDis_", hidden=256, wg=self.wgan_gp_bg)}
        self.step = 0
        self.built = False

    # ---- graph pieces ------------------------------------------------------------------------------
    def encode(self, batch):
        """Frozen stage-I encoder (restored from --pretrained_path in the reference, trainer.py:180-183)."""
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        with torch.no_grad(), slim.variable_scope("Encoder"):
            embs, _, _, enc_var = models.GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch(
                batch["x"], batch["mask_r6"], batch["part_bbox"], batch["part_vis"], self.part_num, 32,
                self.repeat_num, self.conv_hidden_num, self.data_format, activation_fn=slim.relu,
                keep_part_prob=1.0, reuse=self.built)
        n_fg = self.part_num * 32
        return embs[:, :n_fg], embs[:, n_fg:], enc_var

    def mapper(self, side, dim, z=None):
        cfg = self.sides[side]
        with slim.variable_scope(cfg["scope"]):
            return models.GaussianFCRes([self.batch_size, dim], dim, repeat_num=4, hidden_num=cfg["hidden"],
                                        data_format=self.data_format, activation_fn=slim.leaky_relu, z=z,
                                        device=self.device, reuse=self.built)

    def critic(self, side, x):
        cfg = self.sides[side]
        return cfg["wg"].FCDiscriminator(x, input_dim=x.shape[-1], FC_DIM=512, n_layers=3, name=cfg["name"])

    def init_net(self, batch):
        fg, bg, self.Encoder_var = self.encode(batch)
        self.dims = {"fg": fg.shape[1], "bg": bg.shape[1]}
        self.flats, self.opts = {}, {}
        for side, real in (("fg", fg), ("bg", bg)):
            with torch.no_grad():
                app, g_var = self.mapper(side, self.dims[side])
                self.critic(side, app)
            d_var = lib.params_with_name(self.sides[side]["name"] + 'Discriminator.')
            gf, df = FlatParams(g_var), FlatParams(d_var)
            self.flats[side] = (gf, df)
            self.opts[side] = get_optimizers(self.sides[side]["wg"], gf, df, self.g_lr, self.d_lr)
        self.built = True
        from . import tfckpt
        tfckpt.restore_from_config(self.config)      # the frozen stage-I encoder comes from `pretrained_path` (trainer.py:180-183)
        self.allreduce = GradAllReduce()
        for gf, df in self.flats.values():
            self.allreduce.broadcast(gf.flat)
            self.allreduce.broadcast(df.flat)
        if getattr(self.config, "compute_dtype", "f32") == "bf16":
            # the frozen encoder's filters: bf16 shadows made once (the FC mappers / critics run on fp32 weights)
            self.encoder_shadows = H.FilterShadows(self.Encoder_var)

    # ---- optimizer ops -------------------------------------------------------------------------------
    def g_optim_embs(self, side, z=None):
        gf, df = self.flats[side]
        gf.zero_grad()
        df.set_requires_grad(False)
        app, _ = self.mapper(side, self.dims[side], z=z)
        g_loss, _ = gan_loss(self.sides[side]["wg"], None, self.critic(side, app))
        g_loss.backward()
        df.set_requires_grad(True)
        gf.finalize()
        self.opts[side][0].step(self.allreduce(gf.grad))
        return g_loss.detach()

    def d_optim_embs(self, side, batch, z=None):
        gf, df = self.flats[side]
        df.zero_grad()
        fg, bg, _ = self.encode(batch)
        real = fg if side == "fg" else bg
        with torch.no_grad():
            app, _ = self.mapper(side, self.dims[side], z=z)
        real = real.contiguous()
        _, d_loss = gan_loss(self.sides[side]["wg"], self.critic(side, real), self.critic(side, app),
                             Discriminator=lambda t: self.critic(side, t), real_data=real, fake_data=app)
        d_loss.backward()
        df.finalize()
        self.opts[side][1].step(self.allreduce(df.grad))
        if self.sides[side]["wg"].MODE == 'wgan':
            clip_disc_weights(df)
        return d_loss.detach()

    def train_step(self, batch):
        """trainer.py:821-845."""
        out = {}
        for side in ("fg", "bg"):
            wg = self.sides[side]["wg"]
            if self.step > 0:
                out["g_loss_embs_" + side] = self.g_optim_embs(side)
            iters = 1 if wg.MODE in ('dcgan', 'lsgan') else wg.CRITIC_ITERS
            for _ in range(iters):
                out["d_loss_embs_" + side] = self.d_optim_embs(side, batch)
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out
