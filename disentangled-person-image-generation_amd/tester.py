"""Inference / sampling harnesses, mirroring the reference `tester.py`: the stage-I encoder + generator
(+ critic) chained with the stage-II / III samplers, forward only.  `DPIG_FourNetsFgBg_testOnly` (`tester.py:256-417`) is
described here; the other five pipelines of the file are `_Pipeline` subclasses at the end of this module
(`DPIG_FourNets_testOnly` :4-253, `..._testOnlySampleFactor` :419-613, `..._testOnlyCondition` :616-772,
`DPIG_ThreeNetsApp_testOnlyCondition_256` :775-914, `DPIG_ThreeNetsApp_testOnlySampleFactor_256` :917-1138).

    pose      : the (row, col, visibility) keypoints are normalised to [-1,1] (`tester.py:329-333`), encoded by
                `PoseEncoderFCRes` (z=32) and decoded back by `PoseDecoderFCRes`; with `sample_pose` the decoded
                (or a Gaussian-mapped, `PoseGaussian`) embedding drives the generator instead of the input pose.
    appearance: `fg_embs | bg_embs` of the Fg/Bg encoder, or -- with `sample_app` -- the outputs of the two
                `GaussianFCRes` mappers `Gaussian_FC_Fg` (512) / `Gaussian_FC_Bg` (256); `one_app_per_batch` repeats
                the first foreground embedding over the batch (`tester.py:376-397`).
    output    : `G` denormalised to 0..255 (`denorm_img`), the critic's mean score per image, the pose reconstruction
                loss, and the device SSIM against the input image.

The pose target map is rasterised on the device straight from the (normalised) coordinates
(`dpig_pose_rasterize`) -- the reference leaves the graph for `py_poseInflate` here (`tester.py:399-400`).
Variable scopes / names are the reference's, so weights trained by `trainer.py` / `trainer_stage2.py` in the same
process (or loaded into `tflib` under those names) are picked up."""
import torch

from . import hip_ops as H
from . import models
from . import slim
from . import tflib as lib
from . import utils
from .wgan_gp import WGAN_GP

LeakyReLU = slim.leaky_relu     # wgan_gp.LeakyReLU (alpha 0.2), as a fused FC activation


def denorm_img(norm):
    """utils.py:88-89 (NHWC in, NHWC out)."""
    return torch.clamp((norm + 1) * 127.5, 0, 255)


class DPIG_FourNetsFgBg_testOnly(object):
    def __init__(self, config, device, sample_app=False, sample_pose=False, one_app_per_batch=False,
                 sample_pose_embedding=False):
        self.config = config
        self.device = torch.device(device)
        self.batch_size = config.batch_size
        self.img_H, self.img_W, self.channel = config.img_H, config.img_W, 3
        self.repeat_num, self.conv_hidden_num, self.z_num = config.repeat_num, config.conv_hidden_num, config.z_num
        self.data_format = config.data_format
        self.keypoint_num, self.part_num, self.roi_emb_dim = 18, 7, 32
        self.sample_app, self.sample_pose, self.one_app_per_batch = sample_app, sample_pose, one_app_per_batch
        self.sample_pose_embedding = sample_pose_embedding
        lib.set_device(self.device)
        self.wgan_gp = WGAN_GP(DATA_DIR='', MODE=getattr(config, "gan_mode", "dcgan"), DIM=64, BATCH_SIZE=self.batch_size,
                               ITERS=200000, LAMBDA=10, G_OUTPUT_DIM=self.img_H * self.img_W * 3)
        self.built = False

    @torch.no_grad()
    def run(self, batch, pose_rcv, z_fg=None, z_bg=None, z_pose=None):
        """batch: dict with x, mask_r6, part_bbox, part_vis (as the trainers take); pose_rcv: [B, 18*3] pixel
        coordinates + visibility.  z_*: optional fixed noise for the Gaussian mappers (else drawn on the device)."""
        if not self.built:
            from . import tfckpt
            if tfckpt.wants_restore(self.config):        # tester.py:17-64: build the graph, then restore into it
                self._forward(batch, pose_rcv, z_fg, z_bg, z_pose)
                tfckpt.restore_from_config(self.config)
        return self._forward(batch, pose_rcv, z_fg, z_bg, z_pose)

    def _forward(self, batch, pose_rcv, z_fg=None, z_bg=None, z_pose=None):
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        B, K = self.batch_size, self.keypoint_num
        reuse = self.built
        out = {}
        # ---- pose (tester.py:327-352) ---------------------------------------------------------------------
        rcv = pose_rcv.reshape(B, K, 3).float()
        rcv_norm = torch.stack([rcv[..., 0] / float(self.img_H) * 2.0 - 1, rcv[..., 1] / float(self.img_W) * 2.0 - 1,
                                rcv[..., 2]], dim=-1)
        with slim.variable_scope("PoseAE"):
            pose_embs, _ = models.PoseEncoderFCRes(rcv_norm.reshape(B, -1), z_num=32, repeat_num=4, hidden_num=512,
                                                   data_format=self.data_format, activation_fn=LeakyReLU, reuse=reuse)
        with slim.variable_scope("PoseGaussian"):
            G_pose_embs, _ = models.GaussianFCRes([B, pose_embs.shape[-1]], pose_embs.shape[-1], repeat_num=4,
                                                  hidden_num=512, data_format=self.data_format, activation_fn=LeakyReLU,
                                                  z=z_pose, device=self.device, reuse=reuse)
        with slim.variable_scope("PoseAE"):
            dec_in = G_pose_embs if self.sample_pose_embedding else pose_embs
            G_pose_coord, G_pose_visible, _ = models.PoseDecoderFCRes(dec_in, K, repeat_num=4, hidden_num=512,
                                                                      data_format=self.data_format,
                                                                      activation_fn=LeakyReLU, reuse=reuse)
        if self.sample_pose:
            G_pose_rcv = torch.cat([G_pose_coord.reshape(B, K, 2), G_pose_visible.unsqueeze(-1)], dim=-1)
        else:
            G_pose_rcv = rcv_norm
        out["reconstruct_loss"] = torch.mean((rcv_norm - G_pose_rcv) ** 2)
        out["G_pose_rcv"] = G_pose_rcv
        pose_map = utils.pose_target_from_rcv(G_pose_rcv.reshape(B, -1).contiguous(), K, True, self.img_H, self.img_W)
        # ---- appearance (tester.py:356-397) ---------------------------------------------------------------
        with slim.variable_scope("Encoder"):
            embs, _, _, _ = models.GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch(
                batch["x"], batch["mask_r6"], batch["part_bbox"], batch["part_vis"], self.part_num, self.roi_emb_dim,
                self.repeat_num, self.conv_hidden_num, self.data_format, activation_fn=slim.relu, keep_part_prob=1.0,
                reuse=reuse)
        n_fg = self.part_num * self.roi_emb_dim
        fg_embs, bg_embs = embs[:, :n_fg], embs[:, n_fg:]
        with slim.variable_scope("Gaussian_FC_Fg"):
            app_fg, _ = models.GaussianFCRes([B, n_fg], n_fg, repeat_num=4, hidden_num=512, data_format=self.data_format,
                                             activation_fn=LeakyReLU, z=z_fg, device=self.device, reuse=reuse)
        with slim.variable_scope("Gaussian_FC_Bg"):
            n_bg = bg_embs.shape[-1]
            app_bg, _ = models.GaussianFCRes([B, n_bg], n_bg, repeat_num=4, hidden_num=256, data_format=self.data_format,
                                             activation_fn=LeakyReLU, z=z_bg, device=self.device, reuse=reuse)
        if self.sample_app:
            fg = app_fg[:1].expand(B, -1) if self.one_app_per_batch else app_fg
            embs = torch.cat([fg, app_bg], dim=-1)
        elif self.one_app_per_batch:
            embs = torch.cat([fg_embs[:1].expand(B, -1), bg_embs], dim=-1)
        embs = embs.contiguous()
        out["embs"] = embs
        # ---- generator + critic (tester.py:399-417) ---------------------------------------------------------
        embs_rep = embs.reshape(B, 1, 1, -1).expand(B, self.img_H, self.img_W, embs.shape[1])
        with slim.variable_scope("ID_AE"):
            G, _, _ = models.GeneratorCNN_ID_UAEAfterResidual(embs_rep, pose_map, self.channel, self.z_num, self.repeat_num,
                                                             self.conv_hidden_num, self.data_format,
                                                             activation_fn=slim.relu, reuse=reuse)
        out["G"] = denorm_img(G)
        score = self.wgan_gp.DCGANDiscriminator(G.permute(0, 3, 1, 2), input_dim=3)
        out["G_dis_score"] = score.reshape(B, -1).mean(dim=1)
        out["pose_map"] = pose_map
        out["ssim_G_x"] = utils.ssim_G_x(out["G"], batch["x"])
        self.built = True
        return out


# ---- the other pipelines of tester.py: one configurable forward chain -----------------------------------------------------
def _first_over_batch(t):
    """tf.tile(tf.slice(t, [0, ...], [1, ...]), [batch_size, ...]): the first sample's value for every sample."""
    return t[:1].expand(t.shape[0], *t.shape[1:]).contiguous()


class _Pipeline(object):
    """Forward chain shared by the remaining `tester.py` classes.  What differs between them (class attributes):

    ENCODER     'fgbg' (models.py:390-471) | 'body_roi' (:275-325, 48 x 48 crops, no visibility) | 'roi_vis' (:328-388, 64 x 64)
    ENC_EXTRA   levels added to the encoder's repeat_num (the 256 x 256 graphs: +1, `tester.py:825, 1031`)
    GEN_LESS    levels removed from the generator's repeat_num (256 x 256: 1, `tester.py:832, 1051`)
    POSE_Z      PoseEncoderFCRes width (100 in `DPIG_FourNets_testOnly`, else 32); None: no pose branch (condition pipelines:
                the target pose MAP is an input)
    HOLD_FIRST  a factor that is not sampled is the FIRST sample's value tiled over the batch (SampleFactor pipelines,
                `tester.py:508-509, 538-541, 548-551, 1010-1011, 1040-1043`) instead of every sample's own
    CRITIC      score the generated images with the DCGAN critic (the 128 x 64 pipelines)
    Flags of an instance: sample_pose, and sample_app | (sample_fg, sample_bg) | one_app_per_batch as the reference's."""
    ENCODER, ENC_EXTRA, GEN_LESS, POSE_Z, HOLD_FIRST, CRITIC = 'fgbg', 0, 0, 32, False, True

    def __init__(self, config, device, sample_app=False, sample_fg=False, sample_bg=False, sample_pose=False,
                 one_app_per_batch=False):
        self.config = config
        self.device = torch.device(device)
        self.batch_size = config.batch_size
        self.img_H, self.img_W, self.channel = config.img_H, config.img_W, 3
        self.repeat_num, self.conv_hidden_num, self.z_num = config.repeat_num, config.conv_hidden_num, config.z_num
        self.data_format = config.data_format
        self.keypoint_num, self.part_num, self.roi_emb_dim = 18, 7, 32
        self.sample_app, self.sample_fg, self.sample_bg = sample_app, sample_fg, sample_bg
        self.sample_pose, self.one_app_per_batch = sample_pose, one_app_per_batch
        lib.set_device(self.device)
        self.wgan_gp = WGAN_GP(DATA_DIR='', MODE=getattr(config, "gan_mode", "dcgan"), DIM=64, BATCH_SIZE=self.batch_size,
                               ITERS=200000, LAMBDA=10, G_OUTPUT_DIM=self.img_H * self.img_W * 3)
        self.built = False

    @torch.no_grad()
    def run(self, batch, pose_rcv=None, pose_target=None, x_target=None, z_app=None, z_fg=None, z_bg=None, z_pose=None):
        """batch: x, part_bbox, part_vis (+ mask_r6 for the Fg/Bg encoder); pose_rcv [B, 54] pixel keypoints + visibility (pose
        pipelines) or pose_target [B, H, W, 18] (condition pipelines); x_target: the image SSIM is taken against (condition
        pipelines, `tester.py:693-697`; default: batch['x']); z_*: fixed noise of the Gaussian mappers."""
        kw = dict(pose_rcv=pose_rcv, pose_target=pose_target, x_target=x_target, z_app=z_app, z_fg=z_fg, z_bg=z_bg, z_pose=z_pose)
        if not self.built:
            from . import tfckpt
            if tfckpt.wants_restore(self.config):
                self._forward(batch, **kw)
                tfckpt.restore_from_config(self.config)
        return self._forward(batch, **kw)

    def _pose(self, pose_rcv, z_pose, out, reuse):
        B, K = self.batch_size, self.keypoint_num
        rcv = pose_rcv.reshape(B, K, 3).float()
        rcv_norm = torch.stack([rcv[..., 0] / float(self.img_H) * 2.0 - 1, rcv[..., 1] / float(self.img_W) * 2.0 - 1,
                                rcv[..., 2]], dim=-1)
        with slim.variable_scope("PoseAE"):
            pose_embs, _ = models.PoseEncoderFCRes(rcv_norm.reshape(B, -1), z_num=self.POSE_Z, repeat_num=4, hidden_num=512,
                                                   data_format=self.data_format, activation_fn=LeakyReLU, reuse=reuse)
        with slim.variable_scope("PoseGaussian"):     # built (and restorable) like the reference's graph; its output is unused
            models.GaussianFCRes([B, pose_embs.shape[-1]], pose_embs.shape[-1], repeat_num=4, hidden_num=512,
                                 data_format=self.data_format, activation_fn=LeakyReLU, z=z_pose, device=self.device, reuse=reuse)
        with slim.variable_scope("PoseAE"):
            coord, visible, _ = models.PoseDecoderFCRes(pose_embs, K, repeat_num=4, hidden_num=512, data_format=self.data_format,
                                                        activation_fn=LeakyReLU, reuse=reuse)
        if self.sample_pose:
            G_pose_rcv = torch.cat([coord.reshape(B, K, 2), visible.unsqueeze(-1)], dim=-1)
        else:
            G_pose_rcv = _first_over_batch(rcv_norm) if self.HOLD_FIRST else rcv_norm
        out["reconstruct_loss"] = torch.mean((rcv_norm - G_pose_rcv) ** 2)
        out["G_pose_rcv"] = G_pose_rcv
        return utils.pose_target_from_rcv(G_pose_rcv.reshape(B, -1).contiguous(), K, True, self.img_H, self.img_W)

    def _appearance(self, batch, z_app, z_fg, z_bg, reuse):
        B = self.batch_size
        rep = self.repeat_num + self.ENC_EXTRA
        with slim.variable_scope("Encoder"):
            if self.ENCODER == 'fgbg':
                embs, _, _, _ = models.GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch(
                    batch["x"], batch["mask_r6"], batch["part_bbox"], batch["part_vis"], self.part_num, self.roi_emb_dim, rep,
                    self.conv_hidden_num, self.data_format, activation_fn=slim.relu, keep_part_prob=1.0, reuse=reuse)
            elif self.ENCODER == 'roi_vis':
                embs, _, _ = models.GeneratorCNN_ID_Encoder_BodyROIVis(
                    batch["x"], batch["part_bbox"], batch["part_vis"], self.part_num, self.roi_emb_dim, rep, self.conv_hidden_num,
                    self.data_format, activation_fn=slim.relu, keep_part_prob=1.0, roi_size=64, reuse=reuse)
            else:
                embs, _, _ = models.GeneratorCNN_ID_Encoder_BodyROI(
                    batch["x"], batch["part_bbox"], self.part_num, self.roi_emb_dim, rep, self.conv_hidden_num, self.data_format,
                    activation_fn=slim.relu, reuse=reuse)
        return embs

    def _sample_appearance(self, embs, z_app, z_fg, z_bg, reuse):
        """The Gaussian mappers of this pipeline and which factor they replace; overridden per class."""
        return embs

    def _forward(self, batch, pose_rcv=None, pose_target=None, x_target=None, z_app=None, z_fg=None, z_bg=None, z_pose=None):
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        B = self.batch_size
        reuse = self.built
        out = {}
        if self.POSE_Z is not None:
            pose_map = self._pose(pose_rcv, z_pose, out, reuse)
        else:
            pose_map = pose_target
        embs = self._appearance(batch, z_app, z_fg, z_bg, reuse)
        embs = self._sample_appearance(embs, z_app, z_fg, z_bg, reuse).contiguous()
        out["embs"] = embs
        embs_rep = embs.reshape(B, 1, 1, -1).expand(B, self.img_H, self.img_W, embs.shape[1])
        embs_rep._dpig_src = embs
        with slim.variable_scope("ID_AE"):
            G, _, _ = models.GeneratorCNN_ID_UAEAfterResidual(embs_rep, pose_map, self.channel, self.z_num,
                                                             self.repeat_num - self.GEN_LESS, self.conv_hidden_num,
                                                             self.data_format, activation_fn=slim.relu, reuse=reuse)
        out["G"] = denorm_img(G)
        if self.CRITIC:
            score = self.wgan_gp.DCGANDiscriminator(G.permute(0, 3, 1, 2), input_dim=3)
            out["G_dis_score"] = score.reshape(B, -1).mean(dim=1)
        out["pose_map"] = pose_map
        out["ssim_G_x"] = utils.ssim_G_x(out["G"], x_target if x_target is not None else batch["x"])
        self.built = True
        return out

    def _gaussian(self, scope, n, hidden, z, reuse):
        with slim.variable_scope(scope):
            e, _ = models.GaussianFCRes([self.batch_size, n], n, repeat_num=4, hidden_num=hidden, data_format=self.data_format,
                                        activation_fn=LeakyReLU, z=z, device=self.device, reuse=reuse)
        return e


class DPIG_FourNets_testOnly(_Pipeline):
    """`tester.py:4-253`: BodyROI encoder (no Fg/Bg split, no visibility), pose auto-encoder of width 100, ONE appearance mapper
    `Gaussian_FC`; `sample_app` replaces the embedding by the mapper's output (`one_app_per_batch`: its first row for everybody,
    :119-123), `sample_pose` drives the generator with the decoded pose."""
    ENCODER, POSE_Z = 'body_roi', 100

    def _sample_appearance(self, embs, z_app, z_fg, z_bg, reuse):
        rnd = self._gaussian("Gaussian_FC", embs.shape[-1], 512, z_app, reuse)
        if self.one_app_per_batch:
            rnd = _first_over_batch(rnd)
        return rnd if self.sample_app else embs


class DPIG_FourNetsFgBg_testOnlySampleFactor(_Pipeline):
    """`tester.py:419-613`: sample the foreground, background and pose factors independently; a factor that is not sampled is
    held at the first sample's value over the whole batch."""
    HOLD_FIRST = True

    def _sample_appearance(self, embs, z_app, z_fg, z_bg, reuse):
        n_fg = self.part_num * self.roi_emb_dim
        fg_embs, bg_embs = embs[:, :n_fg], embs[:, n_fg:]
        app_fg = self._gaussian("Gaussian_FC_Fg", n_fg, 512, z_fg, reuse)
        app_bg = self._gaussian("Gaussian_FC_Bg", bg_embs.shape[-1], 256, z_bg, reuse)
        fg = app_fg if self.sample_fg else _first_over_batch(fg_embs)
        bg = app_bg if self.sample_bg else _first_over_batch(bg_embs)
        return torch.cat([fg, bg], dim=-1)


class DPIG_FourNetsFgBg_testOnlyCondition(_Pipeline):
    """`tester.py:616-772`: pose transfer -- the appearance of x rendered in a given target pose map; scored by the critic and by
    SSIM against the target image."""
    POSE_Z = None


class DPIG_ThreeNetsApp_testOnlyCondition_256(_Pipeline):
    """`tester.py:775-914`: the DeepFashion 256 x 256 pose-transfer pipeline (BodyROIVis encoder with repeat_num + 1 levels on
    64 x 64 crops, generator with repeat_num - 1 levels, no critic)."""
    ENCODER, ENC_EXTRA, GEN_LESS, POSE_Z, CRITIC = 'roi_vis', 1, 1, None, False


class DPIG_ThreeNetsApp_testOnlySampleFactor_256(_Pipeline):
    """`tester.py:917-1138`: DeepFashion sampling -- appearance from `Gaussian_FC` (`sample_app`) or the first sample's embedding,
    pose from the pose auto-encoder (`sample_pose`) or the first sample's keypoints."""
    ENCODER, ENC_EXTRA, GEN_LESS, HOLD_FIRST, CRITIC = 'roi_vis', 1, 1, True, False

    def _sample_appearance(self, embs, z_app, z_fg, z_bg, reuse):
        rnd = self._gaussian("Gaussian_FC", embs.shape[-1], 512, z_app, reuse)
        return rnd if self.sample_app else _first_over_batch(embs)
