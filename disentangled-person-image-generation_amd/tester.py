"""Inference / sampling harness, mirroring the reference `tester.py:256-417`
(`DPIG_FourNetsFgBg_testOnly`): the stage-I encoder + generator + critic chained with the stage-II / III samplers,
forward only.

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
