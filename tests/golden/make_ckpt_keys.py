"""Generate tests/golden/ckpt_keys_model1.json -- the variable names and shapes a TensorFlow checkpoint of the reference's
stage-I Market model (main.py --model=1: 128x64, conv_hidden_num=128, z_num=64, 7 parts x 32, D_arch=DCGAN, MODE=dcgan, Adam)
holds, written down from the reference's graph-building code as a flat list of layer records -- NOT through this repo's model
or oracle code, so that the names `tfckpt.save` emits are checked against an independent reading of the reference:

  * slim layers are auto-named per variable scope in creation order: Conv, Conv_1, ... / fully_connected, fully_connected_1 with
    variables `weights` [k,k,Cin,Cout] | [In,Out] and `biases` [Cout]  (TF-slim convention)
  * scopes: Encoder/G_encoder (trainer.py:570 + models.py:391), ID_AE/G (trainer.py:592 + models.py:519)
  * tflib conv2d / linear build their variables inside `with tf.name_scope(name)` (conv2d.py:27, linear.py:37), so the
    variable is stored as `Discriminator.1/Discriminator.1.Filters`, `Discriminator.Output/Discriminator.Output.W`;
    batchnorm.py:23-27 has no name scope: `Discriminator.BN2.offset|scale|moving_mean|moving_variance` stay bare
  * optimizer slots: `<var>/Adam`, `<var>/Adam_1` for every trained variable + `beta1_power`, `beta2_power` (G) and
    `beta1_power_1`, `beta2_power_1` (D) (two tf.train.AdamOptimizer instances, trainer.py:136-140)
  * `step`, `g_lr`, `d_lr` (trainer.py:47,54-55)

TensorFlow is not available here: the list is a convention-following reading, not a dump of a TF-written bundle (the bundle
FORMAT stays unpinned; see tests/test_tfckpt.py).

    python tests/golden/make_ckpt_keys.py
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))
H, W, HID, ZNUM, PARTS, PART_Z, ROI = 128, 64, 128, 64, 7, 32, 48
REPEAT = 5            # int(np.log2(128)) - 2  (trainer.py:66)


class Scope(object):
    def __init__(self, prefix, out):
        self.prefix, self.out, self.nconv, self.nfc = prefix, out, 0, 0

    def conv(self, cin, cout, k=3):
        name = "Conv" if self.nconv == 0 else "Conv_%d" % self.nconv
        self.nconv += 1
        self.out.append(("%s/%s/weights" % (self.prefix, name), [k, k, cin, cout]))
        self.out.append(("%s/%s/biases" % (self.prefix, name), [cout]))

    def fc(self, nin, nout):
        name = "fully_connected" if self.nfc == 0 else "fully_connected_%d" % self.nfc
        self.nfc += 1
        self.out.append(("%s/%s/weights" % (self.prefix, name), [nin, nout]))
        self.out.append(("%s/%s/biases" % (self.prefix, name), [nout]))


def tower(sc, hidden, repeat):
    """models.py:420-431 / 454-464 / 529-540: [c1, c2 (+res)] then a stride-2 conv to the next width, `repeat` blocks."""
    for idx in range(repeat):
        ch = hidden * (idx + 1)
        sc.conv(ch, ch)
        sc.conv(ch, ch)
        if idx < repeat - 1:
            sc.conv(ch, hidden * (idx + 2))


def model1_variables():
    g, d = [], []
    enc = Scope("Encoder/G_encoder", g)                  # models.py:390-471
    enc.conv(3, HID); enc.conv(HID, HID); enc.conv(HID, HID)
    tower(enc, HID, REPEAT)                              # shared ROI tower on the 7*B crops (48 -> 3 after 4 stride-2 convs)
    enc.fc((ROI // 16) * (ROI // 16) * HID * REPEAT, PART_Z)
    tower(enc, HID, REPEAT)                              # background tower at full resolution
    enc.fc((H // 16) * (W // 16) * HID * REPEAT, PART_Z * 4)
    emb = PARTS * PART_Z + PART_Z * 4                    # 352
    ae = Scope("ID_AE/G", g)                             # models.py:518-576 on [tiled embedding | 18 pose channels]
    ae.conv(emb + 18, HID)
    tower(ae, HID, REPEAT)
    ae.fc((H // 16) * (W // 16) * HID * REPEAT, ZNUM)
    ae.fc(ZNUM, (H // 16) * (W // 16) * HID)
    for idx in range(REPEAT):                            # decoder: concat skip, [c1, c2 (+res)], upscale + 1x1
        ch = HID + HID * (REPEAT - idx) if idx == 0 else HID * (REPEAT - idx) + HID * (REPEAT - idx)
        ae.conv(ch, ch)
        ae.conv(ch, ch)
        if idx < REPEAT - 1:
            ae.conv(ch, HID * (REPEAT - idx - 1), k=1)
    ae.conv(2 * HID, 3)
    # wgan_gp.py:407-440, DIM=64, MODE='dcgan': BatchNorm (fused) after convs 2-4
    dim = 64
    for i, (cin, cout) in enumerate([(3, dim), (dim, 2 * dim), (2 * dim, 4 * dim), (4 * dim, 8 * dim)], start=1):
        d.append(("Discriminator.%d/Discriminator.%d.Filters" % (i, i), [5, 5, cin, cout]))
        d.append(("Discriminator.%d/Discriminator.%d.Biases" % (i, i), [cout]))
        if i > 1:
            for leaf in ("offset", "scale"):
                d.append(("Discriminator.BN%d.%s" % (i, leaf), [cout]))
    d.append(("Discriminator.Output/Discriminator.Output.W", [8 * 4 * 8 * dim, 1]))
    d.append(("Discriminator.Output/Discriminator.Output.b", [1]))
    moving = [("Discriminator.BN%d.%s" % (i, leaf), [c]) for i, c in ((2, 2 * dim), (3, 4 * dim), (4, 8 * dim))
              for leaf in ("moving_mean", "moving_variance")]
    return g, d, moving


def main():
    g, d, moving = model1_variables()
    keys = {n: s for n, s in g + d + moving}
    for n, s in g + d:
        keys[n + "/Adam"] = s
        keys[n + "/Adam_1"] = s
    for n in ("beta1_power", "beta2_power", "beta1_power_1", "beta2_power_1", "g_lr", "d_lr", "step"):
        keys[n] = []
    out = {"model": "main.py --model=1 (Market-1501 128x64, conv_hidden_num=128, z_num=64, DCGAN critic, MODE=dcgan)",
           "g_trainable": [n for n, _ in g], "d_trainable": [n for n, _ in d], "keys": keys}
    path = os.path.join(HERE, "ckpt_keys_model1.json")
    json.dump(out, open(path, "w"), indent=0, sort_keys=True)
    print("wrote", path, len(keys), "keys;", sum(1 for _ in g), "G vars,", sum(1 for _ in d), "D vars;",
          sum(int(__import__("numpy").prod(s)) for _, s in g + d) / 1e6, "M trained parameters")


if __name__ == "__main__":
    main()
