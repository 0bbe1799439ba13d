"""Stage-I training step of DPIG on the HIP kernels, mirroring the reference `trainer.py`:
`_gan_loss` (:217-252), `_getOptimizer` (:116-149), `_getDiscriminator` (:151-158),
`DPIG_Encoder_GAN_BodyROI_FgBg.build_model` (:568-625) and the step order of `train()` (:336-347):
one "G+D step" = g_optim on one batch (skipped at step 0) then d_optim x {1 (dcgan/lsgan) | 5}.

Host code is graph wiring only: every tensor op on the hot path is a kernel of libdpig_hip.so.
Parameters live in two flat HBM buffers (G-side = Encoder + ID_AE, D-side = Discriminator.*), the
wgrad kernels write gradients straight into the matching flat gradient buffers, one Adam launch
updates a whole side, and under data parallelism the flat gradient buffer is what RCCL
all-reduces (bucketed slices of one allocation).
"""
import math

import torch

from . import autograd as A
from . import hip_ops as H
from . import models
from . import slim
from . import tflib as lib
from .wgan_gp import WGAN_GP


class FlatParams(object):
    """Re-homes a list of parameters into one flat fp32 buffer + flat grad / Adam-moment buffers.

    Each tensor starts on a 16-byte boundary so the kernels' float4 paths apply to every slice."""

    def __init__(self, params):
        seen, plist = set(), []
        for p in params:
            if isinstance(p, torch.nn.Parameter) and p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append(p)
        self.params = plist
        dev = plist[0].device
        offs, total = [], 0
        for p in plist:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.numel = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets = offs
        for p, o in zip(plist, offs):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            p._dpig_grad = self.grad[o:o + n].view(p.shape)
            p._dpig_touched = [False]

    def enable_bf16_shadows(self, split=False):
        """'bf16' mode: persistent bf16 shadows of the conv filters (hip_ops.FilterShadows), refreshed after every
        optimizer step from the fp32 masters this object owns.  split=True ('bf16x3' mode): the two-term shadows."""
        self.shadows = H.FilterShadows(self.params, flat=self.flat, split=split)

    def enable_wino(self):
        """'f32w' mode: persistent Winograd images of the 3x3 filters (hip_ops.WinoFilters), refreshed after every optimizer step."""
        self.wino = H.WinoFilters(self.params)

    def refresh_shadows(self):
        sh = getattr(self, "shadows", None)
        if sh is not None:
            sh.refresh()
        wn = getattr(self, "wino", None)
        if wn is not None:
            wn.refresh()

    def zero_grad(self):
        """No memset: the first kernel that touches a slice overwrites it (beta = 0)."""
        for p in self.params:
            p._dpig_touched[0] = False
            p.grad = None

    def finalize(self, lo=0, hi=None):
        """Parameters (of the index range [lo, hi)) that received no gradient this step must contribute zeros."""
        for p in self.params[lo:hi]:
            if not p._dpig_touched[0]:
                if p.grad is not None:           # gradient came through plain autograd
                    p._dpig_grad.copy_(p.grad)
                    p.grad = None
                else:
                    p._dpig_grad.zero_()

    def set_requires_grad(self, flag):
        for p in self.params:
            p.requires_grad_(flag)


class TFAdam(object):
    """tf.train.AdamOptimizer on a FlatParams (trainer.py:131-146): one fused kernel launch.
    `lr` is a 1-element device tensor (the reference keeps it in a tf.Variable, :56-59)."""

    def __init__(self, flat, lr_dev, beta1=0.9, beta2=0.999, eps=1e-8):
        self.flat, self.lr, self.b1, self.b2, self.eps, self.t = flat, lr_dev, beta1, beta2, eps, 0
        # {int32 t, float corr} on the device: the launch sequence is hipGraph-replayable
        self.state = torch.zeros(2, dtype=torch.int32, device=flat.flat.device)

    def step(self, grad_scale=1.0):
        self.t += 1          # host mirror, bookkeeping only
        f = self.flat
        H.adam_step_dev(f.flat, f.grad, f.m, f.v, self.lr, self.state, self.b1, self.b2, self.eps, grad_scale)
        f.refresh_shadows()


class TFRMSProp(object):
    """tf.train.RMSPropOptimizer(lr) with TF's defaults decay=0.9, momentum=0, epsilon=1e-10 and TF's slot
    initialisation (rms = ones, momentum = zeros) -- wgan / lsgan modes, trainer.py:119-122,142-146."""

    def __init__(self, flat, lr_dev, decay=0.9, momentum=0.0, eps=1e-10):
        self.flat, self.lr, self.decay, self.mu, self.eps, self.t = flat, lr_dev, decay, momentum, eps, 0
        flat.m.fill_(1.0)      # `rms` slot
        flat.v.zero_()         # `momentum` slot

    def step(self, grad_scale=1.0):
        self.t += 1
        f = self.flat
        H.rmsprop_step(f.flat, f.grad, f.m, f.v, self.lr, self.decay, self.mu, self.eps, grad_scale)
        f.refresh_shadows()


def get_optimizers(wgan_gp, G_flat, D_flat, g_lr, d_lr):
    """trainer.py:116-149 `_getOptimizer`: the optimizer pair each GAN mode trains with.  The wgan
    weight clipping op of :124-128 is `clip_disc_weights` below."""
    if wgan_gp.MODE in ('wgan', 'lsgan'):
        return TFRMSProp(G_flat, g_lr), TFRMSProp(D_flat, d_lr)
    elif wgan_gp.MODE == 'wgan-gp':
        return (TFAdam(G_flat, g_lr, beta1=0.5, beta2=0.9, eps=1e-8), TFAdam(D_flat, d_lr, beta1=0.5, beta2=0.9, eps=1e-8))
    elif wgan_gp.MODE == 'dcgan':
        return (TFAdam(G_flat, g_lr, beta1=0.5, beta2=0.999, eps=1e-8), TFAdam(D_flat, d_lr, beta1=0.5, beta2=0.999, eps=1e-8))
    raise Exception()


def optimizer_slots(flat, opt, ordinal=0):
    """The optimizer state of one FlatParams under the names tf.train.Saver gives it: per variable `<var>/Adam` (m) and
    `<var>/Adam_1` (v), or `<var>/RMSProp` (rms) and `<var>/RMSProp_1` (momentum); for Adam also the bias-correction
    powers `beta1_power`, `beta2_power` (suffix `_<ordinal>` for every optimizer after the first one built: the
    generator's is built first, trainer.py:116-149).  TF stores beta^(t+1) after t steps (the power is multiplied
    AFTER each update)."""
    import numpy as np
    kind = "Adam" if isinstance(opt, TFAdam) else "RMSProp"
    out = {}
    for p, o in zip(flat.params, flat.offsets):
        n = p.numel()
        tfn = lib.tf_variable_name(p.dpig_name)
        out["%s/%s" % (tfn, kind)] = flat.m[o:o + n].reshape(p.shape).detach().cpu().numpy().copy()
        out["%s/%s_1" % (tfn, kind)] = flat.v[o:o + n].reshape(p.shape).detach().cpu().numpy().copy()
    if kind == "Adam":
        t = int(opt.state[0])                    # the device counter is the truth (graph replays advance it)
        sfx = "" if ordinal == 0 else "_%d" % ordinal
        out["beta1_power" + sfx] = np.array(opt.b1 ** (t + 1), dtype=np.float32)
        out["beta2_power" + sfx] = np.array(opt.b2 ** (t + 1), dtype=np.float32)
        # the float32 powers underflow (0.5^150 = 0): the exact step count rides along under a name of our own, which
        # TensorFlow ignores on restore
        out[ADAM_STEP_KEY + sfx] = np.array(t, dtype=np.int64)
    return out


ADAM_STEP_KEY = "dpig_amd/adam_step"


def adam_step_from_powers(values, opt, sfx=""):
    """Number of Adam updates behind a checkpoint.  Exact when our own step key is present; otherwise from TensorFlow's
    beta powers (beta^(t+1) after t updates): beta2's first (it decays slowest), then beta1's; when BOTH have underflowed
    to 0 in float32 the bias correction has long converged to 1, which any large t reproduces."""
    import math
    if (ADAM_STEP_KEY + sfx) in values:
        return int(values[ADAM_STEP_KEY + sfx])
    for key, beta in (("beta2_power" + sfx, opt.b2), ("beta1_power" + sfx, opt.b1)):
        if key in values:
            power = float(values[key])
            if 0.0 < power < 1.0:
                return max(0, int(round(math.log(power) / math.log(beta))) - 1)
            if power >= 1.0:
                return 0
    return 10 ** 7


def load_optimizer_slots(flat, opt, values, ordinal=0):
    """Inverse of `optimizer_slots` from a {name: array} dict (e.g. tfckpt.load_checkpoint).  All or nothing: returns
    False, touching nothing, when any slot of this optimizer is missing or mis-shaped."""
    import math
    import numpy as np
    kind = "Adam" if isinstance(opt, TFAdam) else "RMSProp"
    sfx = "" if ordinal == 0 else "_%d" % ordinal
    def slot_names(p):
        # TF-scoped keys (`<op>/<op>.Filters/Adam`); checkpoints written before the TF naming carry the bare registry name
        # (`<op>.Filters/Adam`) -- the same fallback tfckpt.restore gives the weights
        for base in (lib.tf_variable_name(p.dpig_name), p.dpig_name):
            a, b = "%s/%s" % (base, kind), "%s/%s_1" % (base, kind)
            if a in values and b in values:
                return a, b
        return "%s/%s" % (lib.tf_variable_name(p.dpig_name), kind), "%s/%s_1" % (lib.tf_variable_name(p.dpig_name), kind)
    need = [slot_names(p) for p in flat.params]
    for p, (a, b) in zip(flat.params, need):
        if a not in values or b not in values or tuple(values[a].shape) != tuple(p.shape) or tuple(values[b].shape) != tuple(p.shape):
            return False
    if kind == "Adam" and ("beta1_power" + sfx) not in values:
        return False
    with torch.no_grad():
        for p, o, (a, b) in zip(flat.params, flat.offsets, need):
            n = p.numel()
            flat.m[o:o + n].copy_(torch.from_numpy(np.array(values[a], dtype=np.float32).reshape(-1)).to(flat.m.device))
            flat.v[o:o + n].copy_(torch.from_numpy(np.array(values[b], dtype=np.float32).reshape(-1)).to(flat.v.device))
        if kind == "Adam":
            t = adam_step_from_powers(values, opt, sfx)
            opt.t = t
            opt.state.zero_()
            opt.state[0] = t                     # the tick kernel recomputes the correction from t + 1
    return True


def clip_disc_weights(D_flat, lo=-.01, hi=.01):
    """trainer.py:124-128: clip every `Discriminator` parameter to [-0.01, 0.01] -- one launch on the flat buffer
    (the 16-byte alignment padding between tensors is zero and stays zero)."""
    H.clip_(D_flat.flat, lo, hi)
    D_flat.refresh_shadows()


class GradAllReduce(object):
    """Data-parallel gradient exchange: sum-all-reduce of the flat gradient buffer over RCCL in ~32 MB slices (xGMI is
    point-to-point: a few large collectives, not hundreds of small ones); the 1/world factor is folded into the Adam
    kernel's grad_scale.

    `compress='bf16'` ('bf16' storage mode): the slice is rounded to bf16 into a persistent staging buffer
    (dpig_cvt_f32_to_bf16), all-reduced there -- half the bytes on every xGMI link -- and widened back into the fp32
    gradient buffer Adam reads (dpig_cvt_bf16_to_f32).  `codec` = (encode(src_f32, dst_bf16), decode(src_bf16, dst_f32))
    replaces the two kernels (the gloo test on CPU tensors)."""

    def __init__(self, bucket_bytes=32 << 20, compress=None, codec=None):
        import torch.distributed as dist
        self.dist = dist
        self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        self.world = dist.get_world_size() if self.enabled else 1
        self.compress = compress if compress in (None, 'bf16') else None
        self.bucket = bucket_bytes // (2 if self.compress else 4)
        self.codec = codec or (lambda s, d: H.to_bf16(s, out=d), lambda s, d: H.to_f32(s, out=d))
        self._stage = {}

    def _staging(self, t):
        """bf16 twin of the flat buffer `t` is a slice of (same offsets), allocated once per flat buffer."""
        base = t._base if t._base is not None else t
        key = (base.data_ptr(), base.numel())
        if key not in self._stage:
            self._stage[key] = torch.empty(base.numel(), dtype=torch.bfloat16, device=base.device)
        off = (t.data_ptr() - base.data_ptr()) // 4
        return self._stage[key][off:off + t.numel()]

    def __call__(self, flat_grad):
        if not self.enabled:
            return 1.0
        return self.finish(self.start(flat_grad))

    def start(self, flat_grad):
        """Launch the bucketed all-reduce of a slice asynchronously (RCCL runs it on its own stream, ordered after the
        work already queued on the current stream) and return the handles; `finish` waits for them."""
        if not self.enabled:
            return []
        n = flat_grad.numel()
        if n == 0:
            return []
        buf = flat_grad
        pending = None
        if self.compress:
            buf = self._staging(flat_grad)
            self.codec[0](flat_grad, buf)
            pending = (buf, flat_grad)
        hs = [(self.dist.all_reduce(buf[o:min(o + self.bucket, n)], async_op=True), None) for o in range(0, n, self.bucket)]
        if pending is not None:
            hs.append((None, pending))
        return hs

    def finish(self, handles):
        for h, _ in handles:
            if h is not None:
                h.wait()
        for _, pending in handles:
            if pending is not None:
                self.codec[1](pending[0], pending[1])
        return 1.0 / self.world if self.enabled else 1.0

    def broadcast(self, flat_params):
        if self.enabled:
            self.dist.broadcast(flat_params, src=0)


def pose_input(batch, img_H, img_W, keypoint_num=18):
    """The generator's pose operand of a batch: the dense target map `batch['pose']` when the batch carries it, else the
    keypoints `batch['pose_rcv']` ([B, 18*3] pixel (row, col, visibility) as the records hold them) wrapped so that the first
    conv consumes them directly -- the map the reference builds in the graph (trainer.py:556-560) is never materialised."""
    if "pose" in batch:
        return batch["pose"]
    return A.PoseKeypoints(batch["pose_rcv"], img_H, img_W, keypoint_num, is_normalized=False)


def critic_variables(name='', registry=None):
    """The DCGAN critic's variables under `name` in creation order (wgan_gp.py:407-440), keyed WITHOUT the prefix: the operand
    list of the fused penalty call.  None unless every one exists as a dense fp32 tensor (e.g. BatchNorm critics have no
    LayerNorm variables)."""
    reg = lib._params if registry is None else registry
    out = {}
    for grp in H.CRITIC_KEYS[:4]:
        for k in grp:
            out[k] = reg.get(name + k)
    out[H.CRITIC_KEYS[4]] = reg.get(name + H.CRITIC_KEYS[4])
    if any(v is None or v.dtype != torch.float32 for v in out.values()):
        return None
    return out


def gradient_penalty(Discriminator, real_data, fake_data, LAMBDA=10., alpha=None, fused=None):
    """trainer.py:222-236 / wgan_gp.py:605-619: LAMBDA * mean_b (||grad_xhat D(xhat)||_2 - 1)^2 with
    xhat = real + alpha*(fake - real), alpha ~ U[0,1) per sample.  As written the reference only type-checks
    for flat [B,D] inputs (SURVEY C-3); this is the canonical form (per-sample alpha, L2 norm over all
    non-batch axes), identical on flat data.  The penalty's gradient w.r.t. the critic's parameters needs
    the backward pass differentiated once more: every piece is a kernel (autograd second-level functions); the
    interpolation is one launch, the norm / penalty / seed-of-the-second-sweep another (`dpig_gp_penalty`)."""
    B = real_data.shape[0]
    if alpha is None:
        alpha = torch.rand([B] + [1] * (real_data.dim() - 1), device=real_data.device)
    if fused is not None:
        # Discriminator == DCGANDiscriminator over NHWC images: one library call (`dpig_gp_double_backward`) instead of the tape
        return A.gp_fused(real_data.detach(), fake_data.detach(), alpha.reshape(B), LAMBDA, fused["dim"], fused["params"])
    interpolates = H.gp_interpolate(real_data.detach(), fake_data.detach(), alpha.reshape(B)).requires_grad_(True)
    D_int = Discriminator(interpolates)
    with A.no_param_grads():      # d(sum D(xhat))/dtheta is not part of the loss: input gradient only
        gradients = torch.autograd.grad(D_int.sum(), interpolates, create_graph=True)[0]
    # penalty value + its derivative w.r.t. `gradients` (the seed of the second sweep) in one fused pass
    return A.gp_penalty(gradients, LAMBDA)


def gan_loss(wgan_gp, disc_real, disc_fake, Discriminator=None, real_data=None, fake_data=None, alpha=None, fused_gp=None):
    """trainer.py:217-252 (`_gan_loss`), modes dcgan / wgan / wgan-gp / lsgan.  Returns (gen_cost, disc_cost);
    either input may be None when that side is not needed (TF prunes the unused branch)."""
    mode = wgan_gp.MODE
    gen_cost = disc_cost = None
    if mode == 'dcgan':
        if disc_fake is not None:
            gen_cost = A.sce_mean(disc_fake, 1.0)
        if disc_fake is not None and disc_real is not None:
            disc_cost = (A.sce_mean(disc_fake, 0.0) + A.sce_mean(disc_real, 1.0)) / 2.
    elif mode == 'wgan':
        if disc_fake is not None:
            gen_cost = -A.logit_mean(disc_fake)
        if disc_fake is not None and disc_real is not None:
            disc_cost = A.logit_mean(disc_fake) - A.logit_mean(disc_real)
    elif mode == 'lsgan':
        if disc_fake is not None:
            gen_cost = A.logit_sq_mean(disc_fake, 1.0)
        if disc_fake is not None and disc_real is not None:
            disc_cost = (A.logit_sq_mean(disc_real, 1.0) + A.logit_sq_mean(disc_fake, 0.0)) / 2.
    elif mode == 'wgan-gp':
        if disc_fake is not None:
            gen_cost = -A.logit_mean(disc_fake)
        if disc_fake is not None and disc_real is not None:
            disc_cost = A.logit_mean(disc_fake) - A.logit_mean(disc_real)
            disc_cost = disc_cost + gradient_penalty(Discriminator, real_data, fake_data, wgan_gp.LAMBDA, alpha, fused=fused_gp)
    else:
        raise Exception()
    return gen_cost, disc_cost


class Config(object):
    """The config.py flags the hot path reads (defaults = config.py / run_market_train.sh)."""

    def __init__(self, **kw):
        self.batch_size = 16
        self.img_H, self.img_W = 128, 64
        self.conv_hidden_num = 128       # config.py:23
        self.z_num = 64
        self.g_lr = 2e-5                 # run_market_train.sh:12
        self.d_lr = 2e-5
        self.lr_update_step = 50000
        self.D_arch = 'DCGAN'
        self.gan_mode = 'dcgan'          # trainer.py:257 hard-codes MODE='dcgan' for stage I; 'wgan-gp' exercises
                                         # the gradient-penalty branch the reference keeps dormant
        self.data_format = 'NHWC'        # main.py:18
        self.sync_bn = False             # data parallel: D's BatchNorm statistics over all ranks
        self.grad_exchange = None        # None: 'bf16' in 'bf16' mode, fp32 otherwise.  'bf16' = the gradient all-reduce moves
                                         # bf16 (half the xGMI bytes; fp32 gradients and optimizer unchanged); 'f32' forces fp32
        self.split_backward = None       # None: in data-parallel runs only.  The generator-side backward runs in stages
                                         # (critic + generator, background tower, ROI tower, encoder stem) so that the
                                         # all-reduce of each finished stage's gradient slice overlaps the next stage's
                                         # backward pass; gradients are bit-identical to one backward pass
        self.compute_dtype = 'f32'       # 'f32' is the reference's arithmetic.  'bf16' (BASELINE configs 3-5):
                                         # activations / their gradients / filter shadows stored as bf16, bf16 matrix
                                         # pipe, fp32 accumulation, master weights, gradients and optimizer.  'bf16c':
                                         # fp32 tensors, conv operands rounded to bf16 (hip_ops module docstring)
        self.pretrained_path = None      # V2 checkpoint prefixes (trainer.py:180-212): 'Encoder' + 'ID_AE' variables,
        self.pretrained_poseAE_path = None   # the 'PoseAE' variables,
        self.ckpt_path = None            # every variable of the model
        self.restore_optimizer = False   # with ckpt_path: also the Adam / RMSProp slots, if the checkpoint holds them
        self.model_dir = None            # where save_checkpoint() writes model.ckpt-<step>
        self.__dict__.update(kw)
        self.repeat_num = int(math.log2(self.img_H)) - 2    # trainer.py:75


class DPIG_Encoder_GAN_BodyROI_FgBg(object):
    """Reference trainer.py:567-625 (model 1) + the base-class loop :326-347."""

    def __init__(self, config, device):
        self.config = config
        self.device = torch.device(device)
        self.batch_size = config.batch_size
        self.img_H, self.img_W, self.channel = config.img_H, config.img_W, 3
        self.repeat_num, self.conv_hidden_num, self.z_num = config.repeat_num, config.conv_hidden_num, config.z_num
        self.data_format = config.data_format
        self.keypoint_num, self.part_num = 18, 7
        lib.set_device(self.device)
        self.g_lr = torch.full((1,), config.g_lr, dtype=torch.float32, device=self.device)
        self.d_lr = torch.full((1,), config.d_lr, dtype=torch.float32, device=self.device)
        # _define_input (trainer.py:254-259)
        self.Generator_fn = models.GeneratorCNN_ID_UAEAfterResidual
        self.wgan_gp = WGAN_GP(DATA_DIR='', MODE=config.gan_mode, DIM=64, BATCH_SIZE=self.batch_size, ITERS=200000,
                               LAMBDA=10, G_OUTPUT_DIM=self.img_H * self.img_W * 3)
        self.Discriminator_fn = self._getDiscriminator(self.wgan_gp, arch=config.D_arch)
        self.step = 0
        self.built = False
        self.allreduce = None
        self._graphs = None

    def _getDiscriminator(self, wgan_gp, arch='DCGAN'):
        if 'DCGAN' == arch:
            return wgan_gp.DCGANDiscriminator
        elif 'FCDis' == arch:
            return wgan_gp.FCDiscriminator
        raise Exception('You must choose an architecture!')

    # ---- graph pieces (build_model, trainer.py:568-607) ------------------------------------------
    def encode(self, batch):
        with slim.variable_scope("Encoder"):
            embs, _, _, enc_var = models.GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch(
                batch["x"], batch["mask_r6"], batch["part_bbox"], batch["part_vis"], self.part_num, 32,
                self.repeat_num, self.conv_hidden_num, self.data_format, activation_fn=slim.relu,
                keep_part_prob=1.0, reuse=self.built)
        return embs, enc_var

    def generate(self, embs, pose):
        # embs_rep: tile the [B,352] embedding over H x W (trainer.py:588-590)
        B = embs.shape[0]
        embs_rep = embs.reshape(B, 1, 1, -1).expand(B, self.img_H, self.img_W, embs.shape[1])
        embs_rep._dpig_src = embs     # the builder's collapsed first conv reads the [B,E] source (no expand -> select round trip in autograd)
        with slim.variable_scope("ID_AE"):
            G, _, g_var = self.Generator_fn(embs_rep, pose, self.channel, self.z_num, self.repeat_num,
                                            self.conv_hidden_num, self.data_format, activation_fn=slim.relu,
                                            reuse=self.built)
        return G, g_var

    def discriminate(self, img_nhwc):
        # tf.transpose(x, [0,3,1,2]) (trainer.py:601-602): a free view here
        return self.Discriminator_fn(img_nhwc.permute(0, 3, 1, 2), input_dim=3)

    def _fused_gp(self):
        """Operands of the one-call penalty (`dpig_gp_double_backward`) when this trainer's critic is the image DCGAN critic in
        MODE 'wgan-gp' (config.fused_gp, default on; in 'bf16' storage mode the call keeps its activations in bf16:
        DPIG_COMPUTE_BF16_STORE); None -> the taped double backward."""
        if self.wgan_gp.MODE != 'wgan-gp' or not getattr(self.config, "fused_gp", True):
            return None
        if getattr(self.config, "D_arch", "DCGAN") != 'DCGAN':
            return None
        params = critic_variables('')
        return None if params is None else {"dim": 64, "params": params}

    def disc_pair(self, x, G, need_real=True):
        """(D_z_pos, D_z_neg).  Model 1 calls D separately on real and fake (trainer.py:601-602): two
        independent BatchNorm statistic sets; g_loss never needs the real pass (TF prunes it)."""
        D_z_pos = self.discriminate(x) if need_real else None
        return D_z_pos, self.discriminate(G)

    def init_net(self, batch):
        """Create every variable (one forward, like TF graph construction) and set up optimizers."""
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        with torch.no_grad():
            embs, enc_var = self.encode(batch)
            G, g_var = self.generate(embs, pose_input(batch, self.img_H, self.img_W))
            self.discriminate(batch["x"])
        self.built = True
        self.restore_from_config()
        self.G_var = g_var + enc_var                       # trainer.py:596
        self.D_var = lib.params_with_name('Discriminator.')  # trainer.py:603
        self.G_flat = FlatParams(self.G_var)
        self.D_flat = FlatParams(self.D_var)
        dec_ids = set(id(p) for p in g_var)
        self._n_dec = sum(1 for p in self.G_flat.params if id(p) in dec_ids)        # decoder params come first
        assert all(id(p) in dec_ids for p in self.G_flat.params[:self._n_dec])
        self._enc_off = self.G_flat.offsets[self._n_dec] if self._n_dec < len(self.G_flat.params) else self.G_flat.numel
        self._stages = self._backward_stages()
        self.g_opt, self.d_opt = get_optimizers(self.wgan_gp, self.G_flat, self.D_flat, self.g_lr, self.d_lr)
        gx = getattr(self.config, "grad_exchange", None)
        if gx is None:
            gx = 'bf16' if getattr(self.config, "compute_dtype", "f32") == "bf16" else 'f32'
        self.allreduce = GradAllReduce(compress='bf16' if gx == 'bf16' else None)
        self.allreduce.broadcast(self.G_flat.flat)
        self.allreduce.broadcast(self.D_flat.flat)
        if getattr(self.config, "compute_dtype", "f32") in ("bf16", "bf16x3"):
            split = self.config.compute_dtype == "bf16x3"
            self.G_flat.enable_bf16_shadows(split)
            self.D_flat.enable_bf16_shadows(split)
        if getattr(self.config, "compute_dtype", "f32") == "f32w":
            self.G_flat.enable_wino()                    # (the critic has no 3x3 convs)
        lib.ops.batchnorm.set_sync(bool(getattr(self.config, "sync_bn", False)) and self.allreduce.enabled)
        if getattr(self.config, "ckpt_path", None):
            self.restore_counters(self.config.ckpt_path)
            if getattr(self.config, "restore_optimizer", False):
                ok = self.restore_optimizer(self.config.ckpt_path)
                if not all(ok):
                    raise Exception("Config.restore_optimizer: checkpoint %s holds no complete set of %s slots" % (
                        self.config.ckpt_path, " / ".join(n for n, o in zip(("generator", "critic"), ok) if not o)))

    def refresh_shadows(self):
        """After writing parameter values from outside the optimizer ('bf16' mode): re-derive the bf16 filter shadows."""
        self.G_flat.refresh_shadows()
        self.D_flat.refresh_shadows()

    # ---- checkpoints (trainer.py:180-212 restores, :366 saves; TF V2 bundle format, tfckpt.py) ----------------
    def restore_from_config(self):
        """The three restores of the reference's init_net, in its order; in place, so flat buffers / graphs survive."""
        from . import tfckpt
        return tfckpt.restore_from_config(self.config)

    def restore_counters(self, prefix):
        """`step`, `g_lr`, `d_lr` of a full checkpoint (trainer.py:47,54-59: tf.Variables the reference's Saver stores
        and restores with everything else): a resumed run continues the step count and the halved learning rates."""
        from . import tfckpt
        have = set(n for n, _, _ in tfckpt.list_variables(prefix))
        want = [n for n in ("step", "g_lr", "d_lr") if n in have]
        if not want:
            return []
        v = tfckpt.load_checkpoint(prefix, want)
        if "step" in v:
            self.step = int(v["step"])
        if "g_lr" in v:
            self.g_lr.fill_(float(v["g_lr"]))
        if "d_lr" in v:
            self.d_lr.fill_(float(v["d_lr"]))
        return want

    def restore_optimizer(self, prefix):
        """The optimizer slots of a full checkpoint (a `tf.train.Saver()` holds them beside the variables).  Returns
        (generator restored, critic restored)."""
        from . import tfckpt
        values = tfckpt.load_checkpoint(prefix, names=lambda n: n.rsplit("/", 1)[-1].startswith(("Adam", "RMSProp"))
                                        or n.startswith(("beta1_power", "beta2_power", ADAM_STEP_KEY)))
        return (load_optimizer_slots(self.G_flat, self.g_opt, values, 0),
                load_optimizer_slots(self.D_flat, self.d_opt, values, 1))

    def save_checkpoint(self, model_dir=None, include_optimizer=False):
        """`saver.save(sess, model_dir/model.ckpt, global_step=step)`: every variable + the `step` counter, and with
        `include_optimizer` the Adam / RMSProp slots under TF's names (what the reference's full Saver writes)."""
        import os
        import numpy as np
        from . import tfckpt
        model_dir = model_dir or getattr(self.config, "model_dir", None)
        if not model_dir:
            raise Exception("save_checkpoint: no model_dir")
        prefix = os.path.join(model_dir, "model.ckpt-%d" % self.step)
        extra = {"step": np.array(self.step, dtype=np.int32),                   # trainer.py:47
                 "g_lr": np.array(float(self.g_lr), dtype=np.float32),          # trainer.py:54-55
                 "d_lr": np.array(float(self.d_lr), dtype=np.float32)}
        if include_optimizer:
            extra.update(optimizer_slots(self.G_flat, self.g_opt, 0))
            extra.update(optimizer_slots(self.D_flat, self.d_opt, 1))
        tfckpt.save(prefix, extra=extra)
        return prefix

    # ---- hipGraph capture of the two optimizer ops ----------------------------------------------
    def _sync_bn_active(self):
        return bool(A._SYNC_BN_GROUP[0]) and self.allreduce.enabled

    def _capture(self, fn, pool=None):
        """(replayable, fn's result): one hipGraph -- or, with cross-rank batch-norm statistics, a chain of graphs with the statistics'
        all-reduces between them (autograd.SegmentedCapture: the critic's BatchNorm layers are `eager_island`s)."""
        if getattr(self, "_cap_stream", None) is None:
            # ONE capture stream for every graph of this trainer: a later graph's backward pass replays nodes on the stream they
            # were recorded on (an earlier capture's), which must be the stream being captured
            self._cap_stream = torch.cuda.Stream(device=self.device)
        if self._sync_bn_active():
            seg = A.SegmentedCapture(self.device, pool=pool, stream=self._cap_stream)
            out = seg.capture(fn)
            return seg, out
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool, stream=self._cap_stream, capture_error_mode="thread_local"):
            out = fn()
        return g, out

    def enable_graphs(self, batch_g, batch_d, warmup=2):
        """Capture g_optim and d_optim into hipGraphs.  Shapes are static, parameters/gradients/Adam state
        live at fixed addresses and the Adam step counter is on the device, so a replay is the whole host
        cost of ~900 launches.  Single GPU: one graph per optimizer op (fwd + bwd + Adam).  Data parallel:
        the graph stops after the backward pass; the RCCL all-reduce of the flat gradient buffer and the
        Adam launch follow eagerly on the same stream (collectives are kept out of the capture).
        Inputs are copied into static buffers; outputs are static tensors overwritten by each replay."""
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        self._static_g = {k: v.clone() for k, v in batch_g.items()}
        self._static_d = {k: v.clone() for k, v in batch_d.items()}
        # The warm-up (sizes the workspace, pages every kernel in) runs real optimizer steps: weights, optimizer slots
        # and step counters are put back afterwards, so enabling graphs does not move the training trajectory (nor
        # advance a restored checkpoint's Adam state).
        snap = [(f.flat.clone(), f.m.clone(), f.v.clone()) for f in (self.G_flat, self.D_flat)]
        osnap = [(getattr(o, "state", None), o.t) for o in (self.g_opt, self.d_opt)]
        osnap = [(s.clone() if s is not None else None, t) for s, t in osnap]
        # ... and so are the device RNG stream (the wgan-gp alpha, the stage-II samplers draw from it) and every non-trainable
        # registry tensor (BatchNorm moving statistics when update_moving_stats is on)
        rng = torch.cuda.get_rng_state(self.device)
        trainable = set(id(p) for f in (self.G_flat, self.D_flat) for p in f.params)
        bsnap = [(t, t.detach().clone()) for t in lib._params.values() if id(t) not in trainable and not t.requires_grad]
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._g_optim_eager(self._static_g)
                self._d_optim_eager(self._static_d)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        with torch.no_grad():
            for f, (w, m, v) in zip((self.G_flat, self.D_flat), snap):
                f.flat.copy_(w); f.m.copy_(m); f.v.copy_(v)
                f.refresh_shadows()
            for o, (st, t) in zip((self.g_opt, self.d_opt), osnap):
                if st is not None:
                    o.state.copy_(st)
                o.t = t
            for t, v in bsnap:
                t.copy_(v)
        torch.cuda.set_rng_state(rng, self.device)
        del snap, bsnap
        torch.cuda.synchronize(self.device)
        self._graph_update = not (self.allreduce.enabled or self._split())   # fold all-reduce + Adam into the graph?
        # (capture_error_mode thread_local: a process-group watchdog thread polling events must not invalidate the capture)
        self._gg2 = None
        if self._split():
            # one graph per backward stage of g_optim: [forward + critic / generator backward], then the encoder's stages (background
            # tower, ROI tower, stem); the all-reduce of a finished stage's gradient slice is launched between two replays and runs
            # under the next stage's kernels (collectives stay out of the captures)
            A.CUTS = {}
            try:
                def stage0():
                    g_loss, embs, out_g = self._g_forward(self._static_g)
                    return g_loss, embs, out_g, self._g_backward_decoder(g_loss, embs)
                gg, (g_loss, embs, out_g, d_embs) = self._capture(stage0)
                cuts = A.CUTS
            finally:
                A.CUTS = None
            ctx = {"embs": embs, "d_embs": d_embs, "cuts": cuts}
            self._gg2 = []
            for st in self._stages[1:]:
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs, pool=gg.pool if isinstance(gg, A.SegmentedCapture) else gg.pool(), stream=self._cap_stream,
                                      capture_error_mode="thread_local"):
                    self._g_backward_stage(st[0], ctx)
                self._gg2.append((gs, st))
            self._keep = (g_loss, ctx)
            del g_loss, embs, d_embs
        else:
            gg, out_g = self._capture(lambda: self._g_optim_eager(self._static_g, update=self._graph_update))
        gd, out_d = self._capture(lambda: self._d_optim_eager(self._static_d, update=self._graph_update),
                                  pool=gg.pool if isinstance(gg, A.SegmentedCapture) else gg.pool())
        self._graphs = (gg, out_g, gd, out_d)
        from ._lib import workspace
        workspace.pin()            # the graphs hold the workspace address: a later, larger request must not free it
        if self._graph_update:
            self.g_opt.t -= 1          # capturing recorded one step() each without executing it
            self.d_opt.t -= 1

    def static_batches(self):
        """(g_optim's, d_optim's) input buffers of the captured graphs: a producer that writes the next batch INTO them (a prefetcher's
        device slot, bench.py's resident synthetic batches) and passes them back to `train_step` feeds the replay without a copy."""
        return self._static_g, self._static_d

    def _feed(self, static, batch):
        for k, v in batch.items():
            if static[k].data_ptr() != v.data_ptr():
                static[k].copy_(v, non_blocking=True)

    def g_optim(self, batch):
        if self._graphs is None:
            return self._g_optim_eager(batch)
        self._feed(self._static_g, batch)
        self._graphs[0].replay()
        if self._gg2 is not None:
            h = self.allreduce.start(self._stage_slice(self._stages[0]))
            for gs, st in self._gg2:
                gs.replay()
                h += self.allreduce.start(self._stage_slice(st))
            self.g_opt.step(self.allreduce.finish(h))
        elif self._graph_update:
            self.g_opt.t += 1
        else:
            self.g_opt.step(self.allreduce(self.G_flat.grad))
        return self._graphs[1]

    def d_optim(self, batch):
        if self._graphs is None:
            return self._d_optim_eager(batch)
        self._feed(self._static_d, batch)
        self._graphs[2].replay()
        if self._graph_update:
            self.d_opt.t += 1
        else:
            self.d_opt.step(self.allreduce(self.D_flat.grad))
        return self._graphs[3]

    # ---- the two optimizer ops -----------------------------------------------------------------
    def _split(self):
        sb = getattr(self.config, "split_backward", None)
        return self.allreduce.enabled if sb is None else bool(sb)

    def _g_forward(self, batch):
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        self.G_flat.zero_grad()
        self.D_flat.set_requires_grad(False)
        embs, _ = self.encode(batch)
        G, _ = self.generate(embs, pose_input(batch, self.img_H, self.img_W))
        _, D_z_neg = self.disc_pair(batch["x"], G, need_real=False)
        g_loss_only, _ = gan_loss(self.wgan_gp, None, D_z_neg)
        L1Loss = A.l1_mean(G, batch["x"])
        g_loss = g_loss_only + L1Loss * 20           # trainer.py:623
        out = {"g_loss": g_loss.detach(), "L1Loss": L1Loss.detach(), "g_loss_only": g_loss_only.detach(), "G": G.detach()}
        return g_loss, embs, out

    def _sunk_or_copy(self, params, grads):
        """Gradients our kernels wrote into the flat buffer come back as None; anything plain autograd produced is
        copied into its slice."""
        for p, g in zip(params, grads):
            if g is not None:
                if p._dpig_touched[0]:
                    p._dpig_grad.add_(g)
                else:
                    p._dpig_grad.copy_(g)
                p._dpig_touched[0] = True

    def _g_backward_decoder(self, g_loss, embs):
        """Stage 1: critic (dgrad only) + generator, down to the embedding; the decoder slice of the flat gradient
        is complete afterwards."""
        dec = self.G_flat.params[:self._n_dec]
        with A.wgrad_overlap():
            res = torch.autograd.grad(g_loss, [embs] + dec, allow_unused=True)
        self._sunk_or_copy(dec, res[1:])
        self.D_flat.set_requires_grad(True)
        self.G_flat.finalize(0, self._n_dec)
        return res[0]

    def _backward_stages(self):
        """The generator-side backward as stages in completion order, each a contiguous slice of the flat gradient buffer:
        [generator] -> [background tower] -> [ROI tower] -> [encoder stem] (the towers share only the stem, models.py:390-471; a
        single-tower encoder has no background stage).  [(name, first param index, one past the last)]; the encoder is cut at the
        tensors the builders record (autograd.CUTS) into the variables their TF-slim names assign to each part."""
        n0, params = self._n_dec, self.G_flat.params
        # (the scope counters are a process-global that every encoder build overwrites: take THIS trainer's, right after its own build)
        self._marks = dict(A.MARKS)
        kinds = [models.encoder_stage_of(p.dpig_name, self._marks) for p in params[n0:]]
        stages = [("generator", 0, n0)]
        if not kinds or any(k is None for k in kinds):
            return stages + ([("encoder", n0, len(params))] if kinds else [])
        runs = []
        for i, k in enumerate(kinds):                      # creation order: stem, ROI tower, background tower -- contiguous runs
            if runs and runs[-1][0] == k:
                runs[-1][2] = n0 + i + 1
            else:
                runs.append([k, n0 + i, n0 + i + 1])
        if sorted(r[0] for r in runs) != sorted(set(kinds)):
            return stages + [("encoder", n0, len(params))]     # (not contiguous: one encoder stage)
        by = {r[0]: (r[1], r[2]) for r in runs}
        for k in ("bg", "roi", "stem"):
            if k in by:
                stages.append((k, by[k][0], by[k][1]))
        return stages

    def _stage_slice(self, st):
        """The flat-gradient slice of a stage (the 16-byte alignment gaps between tensors ride along: they are zero)."""
        _, lo, hi = st
        off = self.G_flat.offsets
        return self.G_flat.grad[off[lo]:(off[hi] if hi < len(off) else self.G_flat.numel)]

    def _g_backward_stage(self, name, ctx):
        """One encoder stage of the backward pass; `ctx` carries the cut tensors and the gradients that have reached them."""
        _, lo, hi = next(s for s in self._stages if s[0] == name)
        ps = self.G_flat.params[lo:hi]
        cuts = ctx["cuts"].get("E.towers_in")
        with A.wgrad_overlap():
            if name == "encoder" or cuts is None:          # no finer cut available: the whole encoder from the embedding gradient
                res = torch.autograd.grad(ctx["embs"], ps, grad_outputs=ctx["d_embs"], allow_unused=True)
                self._sunk_or_copy(ps, res)
            elif name in ("bg", "roi"):
                t = cuts[1] if name == "bg" else cuts[0]
                last_tower = name == "roi" or not any(s[0] == "roi" for s in self._stages)
                res = torch.autograd.grad(ctx["embs"], [t] + ps, grad_outputs=ctx["d_embs"], allow_unused=True, retain_graph=not last_tower)
                ctx["d_" + name] = res[0]
                self._sunk_or_copy(ps, res[1:])
            else:                                          # stem: from the towers' input gradients
                outs = [c for c, g in zip(cuts, (ctx.get("d_roi"), ctx.get("d_bg"))) if g is not None]
                gouts = [g for g in (ctx.get("d_roi"), ctx.get("d_bg")) if g is not None]
                res = torch.autograd.grad(outs, ps, grad_outputs=gouts, allow_unused=True)
                self._sunk_or_copy(ps, res)
        if self.device.type == "cuda":
            A.join_side_streams(self.device)               # (a tower's backward runs on its side stream: autograd.side_branch)
        self.G_flat.finalize(lo, hi)

    def _g_backward_encoder(self, embs, d_embs, cuts=None, between=None):
        """Stages 2..: the encoder, from the embedding gradient; `between(stage)` is called after each finished stage (the
        data-parallel exchange of its gradient slice starts there)."""
        ctx = {"embs": embs, "d_embs": d_embs, "cuts": cuts or {}}
        for st in self._stages[1:]:
            self._g_backward_stage(st[0], ctx)
            if between is not None:
                between(st)

    def _g_optim_eager(self, batch, update=True):
        """sess.run(g_optim): fwd E,G,D(fake); g_loss = sce(D(G),1) + 20*L1; bwd D(dgrad only),G,E; Adam."""
        if self._split():
            A.CUTS = {}
            try:
                g_loss, embs, out = self._g_forward(batch)
                cuts = A.CUTS
            finally:
                A.CUTS = None
            d_embs = self._g_backward_decoder(g_loss, embs)
            h = self.allreduce.start(self._stage_slice(self._stages[0])) if update else []
            self._g_backward_encoder(embs, d_embs, cuts, between=(lambda st: h.extend(self.allreduce.start(self._stage_slice(st)))) if update else None)
            if update:
                self.g_opt.step(self.allreduce.finish(h))
        else:
            g_loss, embs, out = self._g_forward(batch)
            with A.wgrad_overlap():
                g_loss.backward()
            self.D_flat.set_requires_grad(True)
            self.G_flat.finalize()
            if update:
                self.g_opt.step(self.allreduce(self.G_flat.grad))
        return out

    def _d_optim_eager(self, batch, update=True):
        """sess.run(d_optim): fwd E,G (no grad), D(x), D(G); d_loss; bwd D; Adam(D)."""
        H.set_compute(getattr(self.config, "compute_dtype", "f32"))
        self.D_flat.zero_grad()
        if (A.D_OVERLAP[0] and self.wgan_gp.MODE == 'dcgan' and batch["x"].is_cuda and not self._sync_bn_active()
                and type(self).disc_pair is DPIG_Encoder_GAN_BodyROI_FgBg.disc_pair):        # (SyncBN's collectives cannot sit inside a forked stream of a capture)
            return self._d_optim_overlapped(batch, update)
        with torch.no_grad():
            embs, _ = self.encode(batch)
            G, _ = self.generate(embs, pose_input(batch, self.img_H, self.img_W))
        D_z_pos, D_z_neg = self.disc_pair(batch["x"], G, need_real=True)
        _, d_loss = gan_loss(self.wgan_gp, D_z_pos, D_z_neg, Discriminator=self.discriminate,
                             real_data=batch["x"], fake_data=G, alpha=getattr(self, "gp_alpha", None),   # (tests pin alpha)
                             fused_gp=self._fused_gp())
        with A.wgrad_overlap():
            d_loss.backward()
        self.D_flat.finalize()
        if update:
            self.d_opt.step(self.allreduce(self.D_flat.grad))
        return {"d_loss": d_loss.detach()}

    def _d_optim_overlapped(self, batch, update):
        """MODE 'dcgan', separate critic passes (trainer.py:601-602): d_loss = (sce(D(G), 0) + sce(D(x), 1)) / 2 is a sum of a term that
        needs the generator and one that does not.  The real-image term -- critic forward AND backward -- runs on a side stream beside
        the encoder / generator forward; the fake term follows on the current stream and accumulates into the same gradient slices
        (two-term sums: the order of the two contributions does not change a bit)."""
        with A.side_branch(batch["x"], key="critic", enabled=True) as sb:
            l_real = A.sce_mean(self.discriminate(batch["x"]), 1.0)
            (l_real / 2.).backward()
            l_real = l_real.detach()
        with torch.no_grad():
            embs, _ = self.encode(batch)
            G, _ = self.generate(embs, pose_input(batch, self.img_H, self.img_W))
        l_real = sb.join(l_real)
        l_fake = A.sce_mean(self.discriminate(G), 0.0)
        (l_fake / 2.).backward()
        d_loss = (l_fake.detach() + l_real) / 2.
        self.D_flat.finalize()
        if update:
            self.d_opt.step(self.allreduce(self.D_flat.grad))
        return {"d_loss": d_loss}

    def train_step(self, batch_g, batch_d):
        """One iteration of the reference loop (trainer.py:336-347, 362-363).  `batch_d` is one batch, or a sequence of
        batches: every `sess.run(d_optim)` of the reference dequeues a fresh batch (trainer.py:340-345, 553-555), so in
        the wgan / wgan-gp modes the CRITIC_ITERS critic updates of a step see CRITIC_ITERS different batches (a single
        batch is reused for all of them)."""
        out = {}
        if self.step > 0:
            out.update(self.g_optim(batch_g))
        disc_iters = 1 if self.wgan_gp.MODE in ('dcgan', 'lsgan') else self.wgan_gp.CRITIC_ITERS
        many = isinstance(batch_d, (list, tuple))
        for i in range(disc_iters):
            out.update(self.d_optim(batch_d[i % len(batch_d)] if many else batch_d))
            if self.wgan_gp.MODE == 'wgan':
                clip_disc_weights(self.D_flat)
        if self.step % self.config.lr_update_step == self.config.lr_update_step - 1:
            self.g_lr.mul_(0.5)
            self.d_lr.mul_(0.5)
        self.step += 1
        return out
