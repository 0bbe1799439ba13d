"""Stage-II trainers, mirroring the reference `trainer.py`: the appearance-embedding GAN (model 3, below), the pose
auto-encoder (model 2, `DPIG_PoseRCV_AE_BodyROI`, :626-713) and the pose-embedding GAN (model 4,
`DPIG_subnetSamplePoseRCV_GAN_BodyROI`, :868-1040) at the end of this file.

Stage-II appearance-embedding GAN (model 3), mirroring the reference
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

from . import autograd as A
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
        if getattr(self.config, "compute_dtype", "f32") in ("bf16", "bf16x3"):
            # the frozen encoder's filters: bf16 shadows made once (the FC mappers / critics run on fp32 weights)
            self.encoder_shadows = H.FilterShadows(self.Encoder_var, split=self.config.compute_dtype == "bf16x3")
        if getattr(self.config, "compute_dtype", "f32") == "f32w":
            self.encoder_wino = H.WinoFilters(self.Encoder_var)      # frozen: Winograd images made once

    # ---- optimizer ops -------------------------------------------------------------------------------
    def _update(self, side, which):
        """Gradient exchange + optimizer step (+ the wgan weight clip) of one side's mapper (which = 0) or critic (1).  Data parallel:
        kept OUT of the captured graphs (the collective library's calls are not capturable on every backend; `enable_graphs`)."""
        f = self.flats[side][which]
        self.opts[side][which].step(self.allreduce(f.grad))
        if which == 1 and self.sides[side]["wg"].MODE == 'wgan':
            clip_disc_weights(f)

    def g_optim_embs(self, side, z=None, update=True):
        gf, df = self.flats[side]
        gf.zero_grad()
        df.set_requires_grad(False)
        app, _ = self.mapper(side, self.dims[side], z=z)
        g_loss, _ = gan_loss(self.sides[side]["wg"], None, self.critic(side, app))
        with A.wgrad_overlap():
            g_loss.backward()
        df.set_requires_grad(True)
        gf.finalize()
        if update:
            self._update(side, 0)
        return g_loss.detach()

    def d_optim_embs(self, side, batch, z=None, update=True):
        gf, df = self.flats[side]
        df.zero_grad()
        fg, bg, _ = self.encode(batch)
        real = fg if side == "fg" else bg
        with torch.no_grad():
            app, _ = self.mapper(side, self.dims[side], z=z)
        real = real.contiguous()
        _, d_loss = gan_loss(self.sides[side]["wg"], self.critic(side, real), self.critic(side, app),
                             Discriminator=lambda t: self.critic(side, t), real_data=real, fake_data=app)
        with A.wgrad_overlap():
            d_loss.backward()
        df.finalize()
        if update:
            self._update(side, 1)
        return d_loss.detach()

    def enable_graphs(self, batch, warmup=2):
        """hipGraph replay of the step's optimizer ops (the launch-bound part of this trainer: ten frozen-encoder forwards of ~300 launches
        each plus the FC mappers / critics): one captured graph per (side, op) -- `d_optim_embs` = encoder forward + mapper forward + critic
        forward / backward + RMSProp + clip, `g_optim_embs` = mapper + critic forward / backward + RMSProp -- replayed in the loop order of
        trainer.py:821-845.  The batch lives in static buffers (`_feed` copies a new one in); the samplers' noise comes from the device
        generator, which torch advances per replay.  Weights, optimizer slots and the generator state are put back after the warm-up, so
        enabling graphs does not move the training trajectory.  Data parallel (world > 1): a graph ends with the backward pass; the
        gradient all-reduce, the optimizer step and the clip follow eagerly on the same stream, as in the stage-I trainer (found by
        `scripts/run_scale.sh --dry`: with the exchange inside the capture the two-rank launch died in `capture_end`)."""
        from ._lib import workspace
        dev = self.device
        self._static = {k: v.clone() for k, v in batch.items()}
        snap = [(f.flat.clone(), f.m.clone(), f.v.clone()) for pair in self.flats.values() for f in pair]
        osnap = [(o, o.t, o.state.clone() if hasattr(o, "state") else None) for pair in self.opts.values() for o in pair]
        rng = torch.cuda.get_rng_state(dev)
        lrs = (self.g_lr.clone(), self.d_lr.clone())          # (the eager step halves them in place at the lr_update_step boundary)
        side_stream = torch.cuda.Stream(device=dev)
        side_stream.wait_stream(torch.cuda.current_stream(dev))
        step0, self._graphs = self.step, None
        with torch.cuda.stream(side_stream):
            self.step = 1
            for _ in range(warmup):
                self._train_step_eager(self._static)
        torch.cuda.current_stream(dev).wait_stream(side_stream)
        torch.cuda.synchronize(dev)
        self.step = step0
        with torch.no_grad():
            for f, (w, m, v) in zip([f for pair in self.flats.values() for f in pair], snap):
                f.flat.copy_(w); f.m.copy_(m); f.v.copy_(v)
            for o, t, st in osnap:
                o.t = t
                if st is not None:
                    o.state.copy_(st)
            self.g_lr.copy_(lrs[0]); self.d_lr.copy_(lrs[1])
        torch.cuda.set_rng_state(rng, dev)
        torch.cuda.synchronize(dev)
        graphs, pool = {}, None
        self._graph_update = fold = not self.allreduce.enabled       # exchange + optimizer step inside the graph?
        for side in ("fg", "bg"):
            for op in ("g", "d"):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                    out = self.g_optim_embs(side, update=fold) if op == "g" else self.d_optim_embs(side, self._static, update=fold)
                pool = g.pool()
                graphs[(side, op)] = (g, out)
                if fold:
                    o = self.opts[side][0 if op == "g" else 1]
                    o.t -= 1                  # capturing recorded one step() without executing it
        self._graphs = graphs
        workspace.pin()

    def static_batch(self):
        """The captured graphs' input buffers: a producer that fills them and passes them to `train_step` feeds the replay without a copy."""
        return self._static

    def _feed(self, batch):
        for k, v in batch.items():
            st = self._static.get(k)
            if st is None or tuple(st.shape) != tuple(v.shape) or st.dtype != v.dtype:
                raise RuntimeError("train_step: batch field %r (%s) does not match the captured graphs' input (%s)" % (
                    k, tuple(v.shape), None if st is None else tuple(st.shape)))
            if st.data_ptr() != v.data_ptr():
                st.copy_(v, non_blocking=True)

    def train_step(self, batch):
        """trainer.py:821-845.  `batch`: one batch, or a sequence -- every `sess.run(d_optim_embs)` of the reference dequeues a fresh
        batch (trainer.py:553-555, 832-841), so the ten critic updates of a step see ten batches (a single one is reused for all)."""
        if getattr(self, "_graphs", None) is None:
            return self._train_step_eager(batch)
        many = isinstance(batch, (list, tuple))
        out, k = {}, 0
        for side in ("fg", "bg"):
            wg = self.sides[side]["wg"]
            if self.step > 0:
                g, o = self._graphs[(side, "g")]
                g.replay()
                if self._graph_update:
                    self.opts[side][0].t += 1
                else:
                    self._update(side, 0)
                out["g_loss_embs_" + side] = o
            iters = 1 if wg.MODE in ('dcgan', 'lsgan') else wg.CRITIC_ITERS
            for _ in range(iters):
                self._feed(batch[k % len(batch)] if many else batch)
                k += 1
                g, o = self._graphs[(side, "d")]
                g.replay()
                if self._graph_update:
                    self.opts[side][1].t += 1
                else:
                    self._update(side, 1)
                out["d_loss_embs_" + side] = o
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out

    def _train_step_eager(self, batch):
        many = isinstance(batch, (list, tuple))
        out, k = {}, 0
        for side in ("fg", "bg"):
            wg = self.sides[side]["wg"]
            if self.step > 0:
                out["g_loss_embs_" + side] = self.g_optim_embs(side)
            iters = 1 if wg.MODE in ('dcgan', 'lsgan') else wg.CRITIC_ITERS
            for _ in range(iters):
                out["d_loss_embs_" + side] = self.d_optim_embs(side, batch[k % len(batch)] if many else batch)
                k += 1
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out


# ======================================================================================================================
# pose branch of stage II
# ======================================================================================================================
def normalise_pose_rcv(pose_rcv, keypoint_num, img_H, img_W):
    """trainer.py:639-644: (row, col, visibility) pixel triplets -> rows / cols in [-1, 1], flattened [B, 3 K]."""
    B = pose_rcv.shape[0]
    p = pose_rcv.reshape(B, keypoint_num, 3).to(torch.float32)
    R = p[..., 0:1] / float(img_H) * 2.0 - 1
    C = p[..., 1:2] / float(img_W) * 2.0 - 1
    return torch.cat([R, C, p[..., 2:3]], dim=-1).reshape(B, -1)


class DPIG_PoseRCV_AE_BodyROI(object):
    """Model 2 (trainer.py:626-713): the pose auto-encoder.  PoseEncoderFCRes (54 -> 32) and PoseDecoderFCRes (32 ->
    36 coordinates + 18 visibilities through sigmoid / binaryRound, straight-through gradient) under scope `PoseAE`;
    reconstruct_loss = mean((pose_rcv_norm - G_pose_rcv)^2); Adam(g_lr, beta1 = 0.5) on 20 x that loss over both nets
    (`_define_loss_optim` :665-667); the loop runs g_optim from step 1 on (:678-680)."""

    def __init__(self, config, device):
        from .trainer import TFAdam
        self.config = config
        self.device = torch.device(device)
        self.batch_size = config.batch_size
        self.img_H, self.img_W = config.img_H, config.img_W
        self.data_format = config.data_format
        self.keypoint_num = 18
        self.sample_pose = bool(getattr(config, "sample_pose", False))
        lib.set_device(self.device)
        self.g_lr = torch.full((1,), config.g_lr, dtype=torch.float32, device=self.device)
        self.d_lr = torch.full((1,), config.d_lr, dtype=torch.float32, device=self.device)
        self._Adam = TFAdam
        self.step = 0
        self.built = False

    def autoencode(self, pose_rcv):
        """-> (pose_rcv_norm [B,54], pose_embs [B,32], G_pose_rcv [B,18,3], variables)."""
        from . import autograd as A  # noqa: F401
        B = pose_rcv.shape[0]
        with slim.variable_scope("PoseAE"):
            pose_rcv_norm = normalise_pose_rcv(pose_rcv, self.keypoint_num, self.img_H, self.img_W)
            pose_embs, enc_var = models.PoseEncoderFCRes(pose_rcv_norm, z_num=32, repeat_num=4, hidden_num=512,
                                                         data_format=self.data_format, activation_fn=slim.leaky_relu,
                                                         reuse=self.built)
            if self.sample_pose:                   # sampling new poses at test time (trainer.py:649-650)
                pose_embs = torch.randn(tuple(pose_embs.shape), device=pose_embs.device) * 0.2
            coord, visible, dec_var = models.PoseDecoderFCRes(pose_embs, self.keypoint_num, repeat_num=4, hidden_num=512,
                                                              data_format=self.data_format, activation_fn=slim.leaky_relu,
                                                              reuse=self.built)
        G_pose_rcv = torch.cat([coord.reshape(B, self.keypoint_num, 2), visible.unsqueeze(-1)], dim=-1)
        return pose_rcv_norm, pose_embs, G_pose_rcv, enc_var + dec_var

    def reconstruct_loss(self, pose_rcv_norm, G_pose_rcv):
        from . import autograd as A
        diff = (pose_rcv_norm.reshape(-1) - G_pose_rcv.reshape(-1)).contiguous()
        return A.logit_sq_mean(diff, 0.0)          # mean of squares: one reduction kernel

    def init_net(self, batch):
        with torch.no_grad():
            _, _, _, var = self.autoencode(batch["pose_rcv"])
        self.built = True
        from . import tfckpt
        tfckpt.restore_from_config(self.config)
        self.G_var_pose = var
        self.G_flat = FlatParams(var)
        self.g_opt = self._Adam(self.G_flat, self.g_lr, beta1=0.5)      # tf.train.AdamOptimizer(lr, beta1=0.5): beta2 0.999
        self.allreduce = GradAllReduce()
        self.allreduce.broadcast(self.G_flat.flat)

    def g_optim(self, batch):
        self.G_flat.zero_grad()
        norm, _, G_pose_rcv, _ = self.autoencode(batch["pose_rcv"])
        loss = self.reconstruct_loss(norm, G_pose_rcv)
        with A.wgrad_overlap():
            (loss * 20).backward()
        self.G_flat.finalize()
        self.g_opt.step(self.allreduce(self.G_flat.grad))
        return {"reconstruct_loss": loss.detach(), "G_pose_rcv": G_pose_rcv.detach()}

    def train_step(self, batch):
        out = {}
        if self.step > 0:
            out.update(self.g_optim(batch))
        else:
            with torch.no_grad():
                norm, _, G_pose_rcv, _ = self.autoencode(batch["pose_rcv"])
                out["reconstruct_loss"] = self.reconstruct_loss(norm, G_pose_rcv)
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out

    def G_pose(self, G_pose_rcv):
        """`coord2channel_simple_rcv` of the decoded keypoints (trainer.py:657): the [B,H,W,18] point maps."""
        from . import utils
        return utils.coord2channel_simple_rcv(G_pose_rcv.reshape(G_pose_rcv.shape[0], -1), self.keypoint_num, True,
                                              self.img_H, self.img_W)


class DPIG_subnetSamplePoseRCV_GAN_BodyROI(DPIG_PoseRCV_AE_BodyROI):
    """Model 4 (trainer.py:868-1040): the pose-embedding GAN.  The pose encoder of the auto-encoder (scope `PoseAE`,
    frozen: restored from `pretrained_poseAE_path`) turns real poses into embeddings [B,32]; a GaussianFCRes mapper under
    `PoseGaussian` turns z ~ N(0, 0.2) into fake ones; the critic `Pose_emb_Discriminator.*` (FCDiscriminator) sees the
    PAIR [real; fake] in one call (:907-909); MODE='wgan' (:873): g = -mean D(fake), d = mean D(fake) - mean D(real),
    RMSProp, critic clipped to +-0.01 after each of its 5 updates per step (:962-973).  The pose decoder maps the sampled
    embedding to keypoints (`sample`), which the stage-I generator turns into a person (tester.py)."""

    def __init__(self, config, device):
        super(DPIG_subnetSamplePoseRCV_GAN_BodyROI, self).__init__(config, device)
        self.wgan_gp_encoder = WGAN_GP(DATA_DIR='', MODE='wgan', DIM=64, BATCH_SIZE=self.batch_size, ITERS=200000, LAMBDA=10,
                                       G_OUTPUT_DIM=self.keypoint_num * 3)

    def encode_pose(self, pose_rcv):
        with torch.no_grad(), slim.variable_scope("PoseAE"):
            norm = normalise_pose_rcv(pose_rcv, self.keypoint_num, self.img_H, self.img_W)
            embs, enc_var = models.PoseEncoderFCRes(norm, z_num=32, repeat_num=4, hidden_num=512, data_format=self.data_format,
                                                    activation_fn=slim.leaky_relu, reuse=self.built)
        return embs, enc_var

    def mapper(self, z=None):
        with slim.variable_scope("PoseGaussian"):
            return models.GaussianFCRes([self.batch_size, 32], 32, repeat_num=4, hidden_num=512, data_format=self.data_format,
                                        activation_fn=slim.leaky_relu, z=z, device=self.device, reuse=self.built)

    def decode_pose(self, embs):
        with torch.no_grad(), slim.variable_scope("PoseAE"):
            coord, visible, dec_var = models.PoseDecoderFCRes(embs, self.keypoint_num, repeat_num=4, hidden_num=512,
                                                              data_format=self.data_format, activation_fn=slim.leaky_relu,
                                                              reuse=self.built)
        B = embs.shape[0]
        return torch.cat([coord.reshape(B, self.keypoint_num, 2), visible.unsqueeze(-1)], dim=-1), dec_var

    def critic_pair(self, real, fake):
        pair = torch.cat([real, fake], dim=0)                  # trainer.py:907-910
        D_z = self.wgan_gp_encoder.FCDiscriminator(pair, input_dim=pair.shape[-1], FC_DIM=512, n_layers=3, name='Pose_emb_')
        return torch.split(D_z, D_z.shape[0] // 2)

    def init_net(self, batch):
        real, self.G_var_encoder = self.encode_pose(batch["pose_rcv"])
        with torch.no_grad():
            fake, g_var = self.mapper()
            _, self.G_var_decoder = self.decode_pose(fake)
            self.critic_pair(real, fake)
        self.built = True
        from . import tfckpt
        tfckpt.restore_from_config(self.config)
        self.G_var_embs = g_var
        self.D_var_embs = lib.params_with_name('Pose_emb_Discriminator.')
        self.G_flat, self.D_flat = FlatParams(self.G_var_embs), FlatParams(self.D_var_embs)
        self.g_opt, self.d_opt = get_optimizers(self.wgan_gp_encoder, self.G_flat, self.D_flat, self.g_lr, self.d_lr)
        self.allreduce = GradAllReduce()
        self.allreduce.broadcast(self.G_flat.flat)
        self.allreduce.broadcast(self.D_flat.flat)

    def g_optim_embs(self, batch, z=None):
        self.G_flat.zero_grad()
        self.D_flat.set_requires_grad(False)
        real, _ = self.encode_pose(batch["pose_rcv"])
        fake, _ = self.mapper(z)
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
        real, _ = self.encode_pose(batch["pose_rcv"])
        with torch.no_grad():
            fake, _ = self.mapper(z)
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
        """trainer.py:962-973."""
        out = {}
        if self.step > 0:
            out["g_loss_embs"] = self.g_optim_embs(batch)
        iters = 1 if self.wgan_gp_encoder.MODE in ('dcgan', 'lsgan') else self.wgan_gp_encoder.CRITIC_ITERS
        for _ in range(iters):
            out["d_loss_embs"] = self.d_optim_embs(batch)
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out

    def sample(self, z=None):
        """G_pose_rcv of a sampled embedding (trainer.py:895-902): [B,18,3] normalised (row, col, visibility)."""
        with torch.no_grad():
            fake, _ = self.mapper(z)
            rcv, _ = self.decode_pose(fake)
        return rcv
