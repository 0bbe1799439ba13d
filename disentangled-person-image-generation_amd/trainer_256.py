"""DeepFashion 256x256 stage-I trainer (model 101), mirroring the reference `trainer_256.py:10-134`:
appearance encoder `GeneratorCNN_ID_Encoder_BodyROIVis(repeat_num+1, roi_size=64)` (:40-41), generator with
`repeat_num-1` levels (:53-55), and the discriminator applied ONCE to the concatenated pair [x; G] (:61-66) --
joint BatchNorm statistics, so g_loss depends on the real half too.  With the hard-coded reshape of
`DCGANDiscriminator` (wgan_gp.py:433) a 256x256 image yields 8 logit rows; `tf.split(D_z, 2)` then
gives the first 8B rows to the real images."""
import torch

from . import models
from . import slim
from .trainer import DPIG_Encoder_GAN_BodyROI_FgBg


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
