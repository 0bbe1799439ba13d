"""CPU restatement (torch-CPU, differentiable) of the reference model graphs on the hot path.
TEST INFRASTRUCTURE -- see oracle/__init__.py; parity unpinned by the reference.

Layer for layer after the reference (each conv / add / concat is its own op, nothing fused):
  encoder_fgbg      models.py:390-471   GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch
  encoder_roi       models.py:328-388   GeneratorCNN_ID_Encoder_BodyROIVis
  generator_uae     models.py:518-576   GeneratorCNN_ID_UAEAfterResidual
  dcgan_discriminator wgan_gp.py:407-440 (+ Batchnorm switch :34-40, LeakyReLU :23-24)
  fc_discriminator  wgan_gp.py:399-405
  gan_loss          trainer.py:217-252
  stage1_*          trainer.py:568-625 build_model + :336-347 step order, Adam :136-140

Variables are kept in a ParamStore under the TF-slim / tflib names so the
same values can be loaded into the HIP implementation.
"""
import math

import numpy as np
import torch

from . import ops as O


class ParamStore(object):
    """name -> torch tensor (requires_grad).  Creation order and init follow slim / tflib:
    'xavier' = U(+-sqrt(6/(fan_in+fan_out))) (slim default), 'stdev' = U(+-s*sqrt(3)) (tflib
    `uniform`, conv2d.py:55-60), 'he_lin' = tflib Linear 'he' (linear.py:71-75)."""

    def __init__(self, seed=0, dtype=torch.float64):
        self.rng = np.random.default_rng(seed)
        self.dtype = dtype
        self.p = {}
        self.trainable = {}

    def get(self, name, shape=None, kind="zeros", fan=None, stdev=None, trainable=True):
        if name not in self.p:
            if kind == "zeros":
                v = np.zeros(shape)
            elif kind == "ones":
                v = np.ones(shape)
            elif kind == "xavier":
                lim = math.sqrt(6.0 / (fan[0] + fan[1]))
                v = self.rng.uniform(-lim, lim, size=shape)
            elif kind == "stdev":
                v = self.rng.uniform(-stdev * math.sqrt(3), stdev * math.sqrt(3), size=shape)
            else:
                raise ValueError(kind)
            v = v.astype(np.float32)            # initial values are fp32 like the reference's variables
            t = torch.tensor(v, dtype=self.dtype, requires_grad=trainable)
            self.p[name] = t
            self.trainable[name] = trainable
        return self.p[name]

    def names_with(self, substr, trainable_only=True):
        return [n for n in self.p if substr in n and (self.trainable[n] or not trainable_only)]

    def state_numpy(self):
        return {n: t.detach().to(torch.float32).numpy().copy() for n, t in self.p.items()}


class _Scope(object):
    """TF-slim variable naming: <path>/Conv[_k], <path>/fully_connected[_k] in creation order."""

    def __init__(self, path):
        self.path = path
        self.count = {}

    def uniq(self, default):
        n = self.count.get(default, 0)
        self.count[default] = n + 1
        return "%s/%s" % (self.path, default if n == 0 else "%s_%d" % (default, n))


# Emulation of the library's bf16 STORAGE mode (DESIGN: "a tensor is stored bf16 iff its channel count is a multiple of 8"; conv
# filters are read from bf16 shadows; accumulation, biases, FC layers and norms are fp32).  STORE = None: the reference's arithmetic.
# STORE = f (e.g. lambda t: t.bfloat16().to(t.dtype)): every activation the library keeps in bf16 is passed through f where the
# library rounds it -- conv outputs after their fused activation, residual sums, ROI crops, the masked fg / bg images, the reshaped
# FC output that enters the decoder -- and every filter of a bf16 matrix-pipe conv (Cin, Cout >= 32, multiples of 8).  Used by
# tests/test_variants_gpu.py::test_stage1_bf16_storage_mode to hold the bf16 model to a bound that follows from the storage format
# instead of an empirical one.
STORE = None


def _st(x):
    return STORE(x) if (STORE is not None and x.dim() == 4 and x.shape[-1] % 8 == 0) else x


def _conv(P, sc, x, cout, k, stride, act, round_w=True):
    name = sc.uniq("Conv")
    cin = x.shape[-1]
    w = P.get(name + "/weights", (k, k, cin, cout), "xavier", fan=(k * k * cin, k * k * cout))
    b = P.get(name + "/biases", (cout,), "zeros")
    if STORE is not None and round_w and cin >= 32 and cout >= 32 and cin % 8 == 0 and cout % 8 == 0:
        w = STORE(w)
    y = O.conv2d_same(x, w, b, stride)
    return _st(act(y) if act is not None else y)


def _fc(P, sc, x, cout, act):
    name = sc.uniq("fully_connected")
    cin = x.shape[-1]
    w = P.get(name + "/weights", (cin, cout), "xavier", fan=(cin, cout))
    b = P.get(name + "/biases", (cout,), "zeros")
    y = O.linear(x, w, b)
    return act(y) if act is not None else y


def _tower(P, sc, x, z_out, repeat_num, hidden_num, act):
    for idx in range(repeat_num):
        channel_num = hidden_num * (idx + 1)
        res = x
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _st(x + res)
        if idx < repeat_num - 1:
            x = _conv(P, sc, x, hidden_num * (idx + 2), 3, 2, act)
    x = x.reshape(x.shape[0], -1)            # NHWC flatten order (h, w, c)
    return _fc(P, sc, x, z_out, None)


def _crops(x, ROI_bboxs, bbox_num, roi_size):
    B, H, W, _ = x.shape
    rois = []
    for i in range(bbox_num):
        bbox = ROI_bboxs[:, i, :].to(x.dtype)
        y1 = bbox[:, 0:1] / float(H)
        x1 = bbox[:, 1:2] / float(W)
        y2 = bbox[:, 2:3] / float(H)
        x2 = bbox[:, 3:4] / float(W)
        nb = torch.cat([y1, x1, y2, x2], dim=-1)
        rois.append(_st(O.crop_and_resize(x, nb, torch.arange(B), roi_size, roi_size)))
    return torch.cat(rois, dim=0)


def encoder_fgbg(P, x, fg_mask, ROI_bboxs, ROI_vis, bbox_num=7, z_num=32, repeat_num=5, hidden_num=128,
                 roi_size=48, scope="Encoder/G_encoder", taps=None):
    """models.py:390-471."""
    sc = _Scope(scope)
    act = O.relu
    B = x.shape[0]
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    res = x
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _st(x + res)
    if taps is not None:
        taps["E.stem"] = x
    m = fg_mask.to(x.dtype)
    x_fg = _st(x * m)
    x_bg = _st(x * (1.0 - m))
    body = _crops(x_fg, ROI_bboxs, bbox_num, roi_size)
    if taps is not None:
        taps["E.rois"] = body
    body = _tower(P, sc, body, z_num, repeat_num, hidden_num, act)
    fea_list = list(torch.split(body, B, dim=0))
    for i in range(bbox_num):
        fea_list[i] = fea_list[i] * ROI_vis[:, i:i + 1].to(x.dtype)
    x_bg = _tower(P, sc, x_bg, z_num * 4, repeat_num, hidden_num, act)
    fea_list.append(x_bg)
    return torch.cat(fea_list, dim=-1)


def encoder_roi(P, x, ROI_bboxs, ROI_vis, bbox_num=7, z_num=32, repeat_num=7, hidden_num=128, roi_size=64,
                scope="Encoder/G_encoder"):
    """models.py:328-388."""
    sc = _Scope(scope)
    act = O.relu
    B = x.shape[0]
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    res = x
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _st(x + res)
    body = _crops(x, ROI_bboxs, bbox_num, roi_size)
    body = _tower(P, sc, body, z_num, repeat_num, hidden_num, act)
    fea_list = list(torch.split(body, B, dim=0))
    for i in range(bbox_num):
        fea_list[i] = fea_list[i] * ROI_vis[:, i:i + 1].to(x.dtype)
    return torch.cat(fea_list, dim=-1)


def encoder_body_roi(P, x, ROI_bboxs, bbox_num=7, z_num=32, repeat_num=7, hidden_num=128, roi_size=48,
                     scope="Encoder/G_encoder"):
    """models.py:275-325 (`GeneratorCNN_ID_Encoder_BodyROI`: the ROI encoder WITHOUT the visibility multiply; the DeepFashion
    stage-II trainers build it with repeat_num + 1 levels and the default 48 x 48 crops, trainer_256.py:310-311, 604-605)."""
    sc = _Scope(scope)
    act = O.relu
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    res = x
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _conv(P, sc, x, hidden_num, 3, 1, act)
    x = _st(x + res)
    body = _crops(x, ROI_bboxs, bbox_num, roi_size)
    body = _tower(P, sc, body, z_num, repeat_num, hidden_num, act)
    return torch.cat(list(torch.split(body, x.shape[0], dim=0)), dim=-1)


def generator_uae(P, x, pose, input_channel=3, z_num=64, repeat_num=5, hidden_num=128, scope="ID_AE/G",
                  taps=None):
    """models.py:518-576.  x = tiled embedding [B,H,W,E]."""
    sc = _Scope(scope)
    act = O.relu
    if pose is not None:
        x = torch.cat([x, pose], dim=3)
    enc = []
    x = _conv(P, sc, x, hidden_num, 3, 1, act, round_w=False)     # (tiled-embedding collapse: fp32 filter sums)
    if taps is not None:
        taps["G.stem"] = x
    for idx in range(repeat_num):
        channel_num = hidden_num * (idx + 1)
        res = x
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _st(x + res)
        enc.append(x)
        if idx < repeat_num - 1:
            x = _conv(P, sc, x, hidden_num * (idx + 2), 3, 2, act)
    x_shape = list(x.shape)
    x = x.reshape(x_shape[0], -1)
    z = x = _fc(P, sc, x, z_num, None)
    if taps is not None:
        taps["G.z"] = z
    x = _fc(P, sc, z, x_shape[1] * x_shape[2] * hidden_num, None)
    x = _st(x.reshape(-1, x_shape[1], x_shape[2], hidden_num))
    for idx in range(repeat_num):
        x = torch.cat([x, enc[repeat_num - 1 - idx]], dim=-1)
        res = x
        channel_num = x.shape[-1]
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _conv(P, sc, x, channel_num, 3, 1, act)
        x = _st(x + res)
        if taps is not None:
            taps["G.dec%d" % idx] = x
        if idx < repeat_num - 1:
            x = O.upsample2x(x)
            x = _conv(P, sc, x, hidden_num * (repeat_num - idx - 1), 1, 1, act)
    out = _conv(P, sc, x, input_channel, 3, 1, None)
    return out, z


def dcgan_discriminator(P, img_nhwc, mode="dcgan", dim=64, name="", taps=None):
    """wgan_gp.py:407-440 on NHWC data; the final flatten reproduces tf.reshape of the reference's
    NCHW tensor (order c,h,w) including the hard-coded 8*4*8*dim width."""
    def conv(nm, x, cin, cout):
        w = P.get(nm + ".Filters", (5, 5, cin, cout), "stdev", stdev=0.02)
        b = P.get(nm + ".Biases", (cout,), "zeros")
        if STORE is not None and cin >= 32:              # (bf16-storage emulation: layers 2-4 read bf16 filter shadows)
            w = STORE(w)
        return O.conv2d_same(x, w, b, 2)

    def norm(nm, x):
        c = x.shape[-1]
        offset = P.get(nm + ".offset", (c,), "zeros")
        scale = P.get(nm + ".scale", (c,), "ones")
        if mode == "wgan-gp":
            return O.layernorm(x, scale, offset)
        P.get(nm + ".moving_mean", (c,), "zeros", trainable=False)
        P.get(nm + ".moving_variance", (c,), "ones", trainable=False)
        return O.batchnorm_train(x, scale, offset)

    pre = name + "Discriminator."
    z1 = conv(pre + "1", img_nhwc, img_nhwc.shape[-1], dim)
    o = _st(O.leaky_relu(z1))
    z2 = norm(pre + "BN2", _st(conv(pre + "2", o, dim, 2 * dim)))
    o = _st(O.leaky_relu(z2))
    z3 = norm(pre + "BN3", _st(conv(pre + "3", o, 2 * dim, 4 * dim)))
    o = _st(O.leaky_relu(z3))
    z4 = norm(pre + "BN4", _st(conv(pre + "4", o, 4 * dim, 8 * dim)))
    o = _st(O.leaky_relu(z4))
    if taps is not None:
        taps.update({"D.1pre": z1, "D.2pre": z2, "D.3pre": z3, "D.4pre": z4, "D.4": o})
    o = o.permute(0, 3, 1, 2).reshape(-1, 8 * 4 * 8 * dim)
    w = P.get(pre + "Output.W", (8 * 4 * 8 * dim, 1), "stdev", stdev=0.02)
    b = P.get(pre + "Output.b", (1,), "zeros")
    return O.linear(o, w, b).reshape(-1)


def fc_discriminator(P, x, input_dim, fc_dim=512, n_layers=3, name=""):
    """wgan_gp.py:399-405 with LeakyReLULayer :30-32 ('he' init: stdev sqrt(2/n_in))."""
    def lin(nm, x, nin, nout, he=True):
        sd = math.sqrt(2.0 / nin) if he else math.sqrt(2.0 / (nin + nout))
        w = P.get(nm + ".W", (nin, nout), "stdev", stdev=sd)
        b = P.get(nm + ".b", (nout,), "zeros")
        return O.linear(x, w, b)
    o = O.leaky_relu(lin(name + "Discriminator.Input.Linear", x, input_dim, fc_dim))
    for i in range(n_layers):
        o = O.leaky_relu(lin(name + "Discriminator.%d.Linear" % i, o, fc_dim, fc_dim))
    return lin(name + "Discriminator.Out", o, fc_dim, 1, he=False).reshape(-1)


def gan_loss(mode, disc_real, disc_fake):
    """trainer.py:217-252 (dcgan / wgan / lsgan)."""
    if mode == "dcgan":
        gen = O.sigmoid_cross_entropy_with_logits(disc_fake, torch.ones_like(disc_fake)).mean()
        dis = O.sigmoid_cross_entropy_with_logits(disc_fake, torch.zeros_like(disc_fake)).mean()
        if disc_real is not None:
            dis = dis + O.sigmoid_cross_entropy_with_logits(disc_real, torch.ones_like(disc_real)).mean()
        dis = dis / 2.0
    elif mode == "wgan":
        gen = -disc_fake.mean()
        dis = disc_fake.mean() - (disc_real.mean() if disc_real is not None else 0.0)
    elif mode == "lsgan":
        gen = ((disc_fake - 1) ** 2).mean()
        dis = (((disc_real - 1) ** 2).mean() + (disc_fake ** 2).mean()) / 2.0
    else:
        raise ValueError(mode)
    return gen, dis


def stage1_forward(P, batch, hidden_num=128, z_num=64, repeat_num=5, taps=None):
    """trainer.py:568-607: E -> tile -> G.  batch: dict of torch CPU tensors."""
    x = batch["x"]
    B, H, W, _ = x.shape
    embs = encoder_fgbg(P, x, batch["mask_r6"], batch["part_bbox"], batch["part_vis"], 7, 32, repeat_num,
                        hidden_num, taps=taps)
    if taps is not None:
        taps["embs"] = embs
    embs_rep = embs.reshape(B, 1, 1, -1).expand(B, H, W, embs.shape[1])
    G, z = generator_uae(P, embs_rep, batch["pose"], 3, z_num, repeat_num, hidden_num, taps=taps)
    if taps is not None:
        taps["G"] = G
    return embs, G


def stage1_g_loss(P, batch, **kw):
    """g_loss = sce(D(G), 1) + 20 * mean|G - x|   (trainer.py:605-607, 623)."""
    _, G = stage1_forward(P, batch, **kw)
    d_fake = dcgan_discriminator(P, G, "dcgan")
    g_only, _ = gan_loss("dcgan", None, d_fake)
    l1 = (G - batch["x"]).abs().mean()
    return g_only + 20.0 * l1, {"g_loss_only": g_only, "L1Loss": l1, "G": G, "D_z_neg": d_fake}


def stage1_d_loss(P, batch, **kw):
    """d_loss = (sce(D(G),0) + sce(D(x),1)) / 2 with D called separately on real and fake
    (trainer.py:601-602: two independent BN statistic sets)."""
    with torch.no_grad():
        _, G = stage1_forward(P, batch, **kw)
    d_real = dcgan_discriminator(P, batch["x"], "dcgan")
    d_fake = dcgan_discriminator(P, G, "dcgan")
    _, d = gan_loss("dcgan", d_real, d_fake)
    return d, {"D_z_pos": d_real, "D_z_neg": d_fake}


def g_var_names(P):
    return [n for n in P.p if (n.startswith("Encoder/") or n.startswith("ID_AE/"))]


def d_var_names(P):
    return [n for n in P.p if "Discriminator." in n and P.trainable[n]]


class OracleAdam(object):
    """TF Adam over a name list of a ParamStore."""

    def __init__(self, P, names, lr, beta1=0.5, beta2=0.999, eps=1e-8):
        self.P, self.names, self.lr, self.b1, self.b2, self.eps, self.t = P, names, lr, beta1, beta2, eps, 0
        self.m = {n: torch.zeros_like(P.p[n]) for n in names}
        self.v = {n: torch.zeros_like(P.p[n]) for n in names}

    def step(self, grads):
        self.t += 1
        with torch.no_grad():
            for n in self.names:
                g = grads.get(n)
                if g is None:
                    g = torch.zeros_like(self.P.p[n])
                p, m, v = O.tf_adam_step(self.P.p[n], g, self.m[n], self.v[n], self.lr, self.b1, self.b2, self.eps,
                                         self.t)
                self.P.p[n].copy_(p)
                self.m[n], self.v[n] = m, v


def gradient_penalty(D, real, fake, alpha, lam=10.0):
    """trainer.py:222-236 / wgan_gp.py:605-619 in the canonical form (per-sample alpha, L2 norm over all non-batch axes;
    SURVEY C-3):  lam * mean_b (|| d D(xhat_b) / d xhat_b ||_2 - 1)^2,  xhat = real + alpha (fake - real).  Differentiable
    w.r.t. the critic's parameters (create_graph)."""
    B = real.shape[0]
    xh = (real + alpha.reshape([B] + [1] * (real.dim() - 1)) * (fake - real)).detach().requires_grad_(True)
    (g,) = torch.autograd.grad(D(xh).sum(), xh, create_graph=True)
    return lam * ((g.reshape(B, -1).pow(2).sum(1).sqrt() - 1.0) ** 2).mean()


def gan_losses(mode, D, real, fake, alpha=None, lam=10.0):
    """`_gan_loss` for all four modes (trainer.py:217-252): (gen_cost, disc_cost)."""
    d_real, d_fake = D(real), D(fake)
    if mode == "wgan-gp":
        gen, dis = gan_loss("wgan", d_real, d_fake)
        return gen, dis + gradient_penalty(D, real, fake, alpha, lam)
    return gan_loss(mode, d_real, d_fake)


def batch_to_torch(batch, dtype=torch.float64):
    out = {}
    for k, v in batch.items():
        t = torch.as_tensor(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


# ---- stage-II pieces (trainer.py:715-868) and the DeepFashion 256 variant (trainer_256.py:31-88) -------------
def gaussian_fc_res(P, z, out_channel, repeat_num=4, hidden_num=512, scope="Gaussian_FC_Fg/G_FC", alpha=0.2):
    """models.py:474-486 with the LeakyReLU(0.2) the stage-II trainer passes (trainer.py:753, B-11)."""
    sc = _Scope(scope)
    act = lambda t: O.leaky_relu(t, alpha)  # noqa: E731
    z = _fc(P, sc, z, hidden_num, act)
    for _ in range(repeat_num):
        res = z
        z = _fc(P, sc, z, hidden_num, act)
        z = _fc(P, sc, z, hidden_num, act)
        z = res + z
    return _fc(P, sc, z, out_channel, None)


def stage2_losses(P, real, z, side="Fg", hidden=512):
    """wgan mode (trainer.py:218-220): g = -mean(D(fake)); d = mean(D(fake)) - mean(D(real))."""
    fake = gaussian_fc_res(P, z, real.shape[1], 4, hidden, scope="Gaussian_FC_%s/G_FC" % side)
    d_fake = fc_discriminator(P, fake, fake.shape[1], name="%s_FCDis_" % side)
    d_real = fc_discriminator(P, real, real.shape[1], name="%s_FCDis_" % side)
    return -d_fake.mean(), d_fake.mean() - d_real.mean(), fake


def stage2_256_losses(P, real, z, hidden=512):
    """Model 102 (trainer_256.py:320-330): ONE Gaussian mapper under `Gaussian_FC`, the critic `FCDis_Discriminator.*` applied
    to the pair [real; fake] in one call, wgan mode."""
    fake = gaussian_fc_res(P, z, real.shape[1], 4, hidden, scope="Gaussian_FC/G_FC")
    pair = torch.cat([real, fake], dim=0)
    d = fc_discriminator(P, pair, pair.shape[1], name="FCDis_")
    d_real, d_fake = torch.split(d, d.shape[0] // 2)
    return -d_fake.mean(), d_fake.mean() - d_real.mean(), fake


def stage1_256_forward(P, batch, hidden_num=128, z_num=64, repeat_num=6, taps=None):
    """trainer_256.py:31-66: E = BodyROIVis(repeat_num+1, roi 64); G with repeat_num-1 levels; D on [x; G].
    `taps` (a dict) additionally receives the generator's G.stem / G.z / G.dec* activations and the critic's pre-norm maps."""
    x = batch["x"]
    B, H, W, _ = x.shape
    embs = encoder_roi(P, x, batch["part_bbox"], batch["part_vis"], 7, 32, repeat_num + 1, hidden_num, roi_size=64)
    embs_rep = embs.reshape(B, 1, 1, -1).expand(B, H, W, embs.shape[1])
    G, _ = generator_uae(P, embs_rep, batch["pose"], 3, z_num, repeat_num - 1, hidden_num, taps=taps)
    D_z = dcgan_discriminator(P, torch.cat([x, G], dim=0), "dcgan", taps=taps)
    D_pos, D_neg = torch.split(D_z, D_z.shape[0] // 2)
    g_only, d_loss = gan_loss("dcgan", D_pos, D_neg)
    l1 = (G - x).abs().mean()
    return {"embs": embs, "G": G, "D_z": D_z, "g_loss": g_only + 20.0 * l1, "d_loss": d_loss, "L1Loss": l1}


def tf_rmsprop_step(p, g, ms, mom, lr, decay=0.9, momentum=0.0, eps=1e-10):
    """tf.train.RMSPropOptimizer (rms slot initialised to ones): returns new (p, ms, mom)."""
    ms = decay * ms + (1 - decay) * g * g
    mom = momentum * mom + lr * g / torch.sqrt(ms + eps)
    return p - mom, ms, mom


def pose_encoder_fc_res(P, pose_rcv, z_num=32, repeat_num=4, hidden_num=512, scope="PoseAE/G_Pose_Encoder", alpha=0.2):
    """models.py:488-499 with the LeakyReLU(0.2) of trainer.py:647-648."""
    sc = _Scope(scope)
    act = lambda t: O.leaky_relu(t, alpha)  # noqa: E731
    x = _fc(P, sc, pose_rcv, hidden_num, act)
    for _ in range(repeat_num):
        res = x
        x = _fc(P, sc, x, hidden_num, act)
        x = _fc(P, sc, x, hidden_num, act)
        x = res + x
    return _fc(P, sc, x, z_num, None)


def pose_decoder_fc_res(P, z, keypoint_num=18, repeat_num=4, hidden_num=512, scope="PoseAE/G_Pose_Decoder", alpha=0.2):
    """models.py:501-515: coords head linear, visibility head sigmoid + binaryRound (forward value = round)."""
    sc = _Scope(scope)
    act = lambda t: O.leaky_relu(t, alpha)  # noqa: E731
    x = _fc(P, sc, z, hidden_num, None)
    for _ in range(repeat_num):
        res = x
        x = _fc(P, sc, x, hidden_num, act)
        x = _fc(P, sc, x, hidden_num, act)
        x = res + x
    coord = _fc(P, sc, x, keypoint_num * 2, None)
    vis_prob = torch.sigmoid(_fc(P, sc, x, keypoint_num, None))
    return coord, torch.round(vis_prob), vis_prob


def normalise_pose_rcv(pose_rcv, keypoint_num=18, img_H=128, img_W=64):
    """trainer.py:639-644."""
    B = pose_rcv.shape[0]
    p = pose_rcv.reshape(B, keypoint_num, 3)
    return torch.cat([p[..., 0:1] / float(img_H) * 2.0 - 1, p[..., 1:2] / float(img_W) * 2.0 - 1, p[..., 2:3]], dim=-1).reshape(B, -1)


def pose_ae_loss(P, pose_rcv, img_H=128, img_W=64):
    """Model 2 (trainer.py:636-661): reconstruct_loss = mean((pose_rcv_norm - G_pose_rcv)^2) with the visibility head's
    binaryRound as a straight-through op (models.py:97-108: forward round, gradient of the identity)."""
    B = pose_rcv.shape[0]
    norm = normalise_pose_rcv(pose_rcv, 18, img_H, img_W)
    z = pose_encoder_fc_res(P, norm)
    coord, vis_round, vis_prob = pose_decoder_fc_res(P, z)
    vis_st = vis_prob + (vis_round - vis_prob).detach()        # straight-through
    G_pose_rcv = torch.cat([coord.reshape(B, 18, 2), vis_st.unsqueeze(-1)], dim=-1)
    return ((norm.reshape(B, 18, 3) - G_pose_rcv) ** 2).mean(), z, G_pose_rcv


def pose_gan_losses(P, pose_rcv, z, img_H=128, img_W=64):
    """Model 4 (trainer.py:878-911): wgan losses of the pose-embedding GAN, critic on the pair [real; fake]."""
    norm = normalise_pose_rcv(pose_rcv, 18, img_H, img_W)
    real = pose_encoder_fc_res(P, norm).detach()
    fake = gaussian_fc_res(P, z, 32, 4, 512, scope="PoseGaussian/G_FC")
    coord, vis_round, _ = pose_decoder_fc_res(P, fake.detach())
    D_z = fc_discriminator(P, torch.cat([real, fake], 0), 32, name="Pose_emb_")
    d_real, d_fake = torch.split(D_z, D_z.shape[0] // 2)
    return -d_fake.mean(), d_fake.mean() - d_real.mean(), fake, real


# ---- tester.py pipelines (forward only), layer for layer ----------------------------------------------------------------------
def tester_pipeline(P, kind, batch, rcv=None, pose_target=None, z_app=None, z_fg=None, z_bg=None, z_pose=None, sample_app=False,
                    sample_fg=False, sample_bg=False, sample_pose=False, one_app_per_batch=False, hidden_num=128, z_num=64,
                    repeat_num=5, img_H=128, img_W=64):
    """The five pipelines of tester.py next to `DPIG_FourNetsFgBg_testOnly`:
      'four_nets'          :75-135    BodyROI encoder, PoseAE z=100, `Gaussian_FC`
      'sample_factor'      :479-571   Fg/Bg encoder, per-factor sampling, unsampled factors held at the first sample
      'condition'          :657-686   Fg/Bg encoder + generator on a given pose map + critic
      'condition_256'      :815-834   BodyROIVis(repeat+1, 64 x 64 crops) + generator(repeat-1), no critic
      'sample_factor_256'  :985-1053  the same with the pose auto-encoder and `Gaussian_FC`
    Returns dict(embs, pose_map, G, [G_pose_rcv], [score])."""
    B = batch["x"].shape[0]
    first = lambda t: t[:1].expand(B, *t.shape[1:])  # noqa: E731
    out = {}
    pose = kind in ("four_nets", "sample_factor", "sample_factor_256")
    if pose:
        norm = normalise_pose_rcv(rcv, 18, img_H, img_W)
        pz = 100 if kind == "four_nets" else 32
        pose_embs = pose_encoder_fc_res(P, norm, z_num=pz)
        gaussian_fc_res(P, z_pose, pz, 4, 512, scope="PoseGaussian/G_FC")                      # in the graph, unused
        coord, vis, _ = pose_decoder_fc_res(P, pose_embs)
        if sample_pose:
            G_pose_rcv = torch.cat([coord.reshape(B, 18, 2), vis.unsqueeze(-1)], -1)
        else:
            G_pose_rcv = norm.reshape(B, 18, 3) if kind == "four_nets" else first(norm.reshape(B, 18, 3))
        out["G_pose_rcv"] = G_pose_rcv
        out["reconstruct_loss"] = ((norm.reshape(B, 18, 3) - G_pose_rcv) ** 2).mean()
        pose_map = O.tf_poseInflate(O.coord2channel_simple_rcv(G_pose_rcv.reshape(B, -1), 18, True, img_H, img_W), 18, 4, img_H, img_W)
    else:
        pose_map = pose_target
    if kind in ("sample_factor", "condition"):
        embs = encoder_fgbg(P, batch["x"], batch["mask_r6"], batch["part_bbox"], batch["part_vis"], 7, 32, repeat_num, hidden_num)
    elif kind == "four_nets":
        embs = encoder_body_roi(P, batch["x"], batch["part_bbox"], 7, 32, repeat_num, hidden_num, 48)
    else:
        embs = encoder_roi(P, batch["x"], batch["part_bbox"], batch["part_vis"], 7, 32, repeat_num + 1, hidden_num, 64)
    if kind == "four_nets":
        rnd = gaussian_fc_res(P, z_app, embs.shape[-1], 4, 512, scope="Gaussian_FC/G_FC")
        if one_app_per_batch:
            rnd = first(rnd)
        embs = rnd if sample_app else embs
    elif kind == "sample_factor":
        fg_e, bg_e = embs[:, :224], embs[:, 224:]
        app_fg = gaussian_fc_res(P, z_fg, 224, 4, 512, scope="Gaussian_FC_Fg/G_FC")
        app_bg = gaussian_fc_res(P, z_bg, bg_e.shape[-1], 4, 256, scope="Gaussian_FC_Bg/G_FC")
        embs = torch.cat([app_fg if sample_fg else first(fg_e), app_bg if sample_bg else first(bg_e)], -1)
    elif kind == "sample_factor_256":
        rnd = gaussian_fc_res(P, z_app, embs.shape[-1], 4, 512, scope="Gaussian_FC/G_FC")
        embs = rnd if sample_app else first(embs)
    out["embs"] = embs
    embs_rep = embs.reshape(B, 1, 1, -1).expand(B, img_H, img_W, embs.shape[1])
    gen_rep = repeat_num - 1 if kind.endswith("256") else repeat_num
    G, _ = generator_uae(P, embs_rep, pose_map, 3, z_num, gen_rep, hidden_num)
    out["G"], out["pose_map"] = G, pose_map
    if not kind.endswith("256"):
        out["score"] = dcgan_discriminator(P, G, "dcgan").reshape(B, -1).mean(1)
    return out
