"""Model builders of the DPIG hot path on the HIP kernels: same function names, argument meaning
and return shapes as the reference `models.py` (file:line cited per function).  Pure functions
`tensor(s) -> (tensor(s), var_list)`; variables are created on first call under TF-slim names.

Activations are NHWC fp32 device tensors.  Differences from the reference are execution-only:
the residual blocks are one fused op (slim.res_block), the 7 ROI crops are one launch, and the
nearest-2x upsample is folded into the following 1x1 conv.  Arithmetic is unchanged.
"""
import os

import numpy as np
import torch

from . import autograd as A
from . import hip_ops as H
from . import slim
from .slim import fully_connected, variable_scope


# Optional activation taps for parity tests: set to a dict to have the builders record the tensors
# named like the oracle's taps (E.stem, E.rois, G.stem, G.z, G.dec<i>).  None = off.
TAPS = None


def _tap(name, t):
    if TAPS is not None:
        TAPS[name] = t.detach()


def _cut(name, *tensors):
    """Backward-stage boundary (autograd.CUTS / MARKS): the tensors, and the scope counters reached so far."""
    path = slim._path()
    if name == "E.towers_in":
        A.MARKS.pop("E.bg_begin", None)                   # (a two-tower builder sets it again further down)
    A.MARKS[name] = (path, slim._counters.get(path + "/Conv", 0), slim._counters.get(path + "/fully_connected", 0))
    if A.CUTS is not None and tensors:
        tensors = tuple(A._CutFn.apply(t) for t in tensors)      # (own backward node per cut tensor, see autograd._CutFn)
        A.CUTS[name] = tensors
    return tensors[0] if len(tensors) == 1 else tensors


def encoder_stage_of(param_name, marks=None):
    """'stem' | 'roi' | 'bg' for a variable of the encoders below, from its TF-slim name (`<scope>/Conv_k/weights`, ...) and the scope
    counters recorded at the stage boundaries (`marks`: a snapshot of autograd.MARKS taken right after the build forward; default the
    live dict); None for any other variable -- another scope, a leaf that is neither `Conv[_k]` nor `fully_connected[_k]`, a
    non-numeric suffix -- so that the caller falls back to ONE 'encoder' stage instead of misfiling the variable."""
    marks = A.MARKS if marks is None else marks
    if "E.towers_in" not in marks:
        return None
    path, conv_t, fc_t = marks["E.towers_in"]
    if not param_name.startswith(path + "/"):
        return None
    leaf = param_name[len(path) + 1:].split("/")[0]
    for base in ("Conv", "fully_connected"):
        if leaf == base:
            k = 0
            break
        if leaf.startswith(base + "_") and leaf[len(base) + 1:].isdigit():
            k = int(leaf[len(base) + 1:])
            break
    else:
        return None

    def before(mark):
        _, c, f = marks[mark]
        return k < (c if base == "Conv" else f)
    if before("E.towers_in"):
        return "stem"
    if "E.bg_begin" in marks and not before("E.bg_begin"):
        return "bg"
    return "roi"


def relu(x):
    return slim.relu(x)


def LeakyReLU(x, alpha=0.3):
    """models.py:137 (alpha 0.3; the trainers resolve LeakyReLU to wgan_gp's 0.2, SURVEY B-11)."""
    return slim.leaky_relu(x, alpha)


def reshape(x, h, w, c, data_format):
    """utils.py:54-59."""
    if data_format == 'NCHW':
        return x.reshape(-1, c, h, w)
    return x.reshape(-1, h, w, c)


def _normalised_boxes(ROI_bboxs, bbox_num, img_H, img_W):
    """models.py:405-413: pixel (y1,x1,y2,x2) -> /img_H, /img_W (division by H, not H-1), stacked
    part-major ([part0: all images, part1: all images, ...]) like tf.concat(body_roi_list, axis=0)."""
    return H.roi_boxes(ROI_bboxs, bbox_num, img_H, img_W)          # one launch (csrc/dpig_glue.hip)


def _roi_tower(body_regions, z_num, repeat_num, hidden_num, data_format, activation_fn):
    """Shared-weight tower on the stacked ROI crops (models.py:346-357 / 420-431)."""
    for idx in range(repeat_num):
        channel_num = hidden_num * (idx + 1)
        body_regions = slim.res_block(body_regions, channel_num, 3, activation_fn=activation_fn,
                                      data_format=data_format)
        if idx < repeat_num - 1:
            body_regions = slim.conv2d(body_regions, hidden_num * (idx + 2), 3, 2, activation_fn=activation_fn,
                                       data_format=data_format)
    body_regions = body_regions.reshape(body_regions.shape[0], -1)
    return fully_connected(body_regions, z_num, activation_fn=None)


def _apply_vis(body_regions, ROI_vis, bbox_num, z_num):
    """models.py:359-368 / 433-442: split per part, multiply by the visibility flag, concat on -1."""
    batch = ROI_vis.shape[0]
    fea = body_regions.reshape(bbox_num, batch, z_num)
    vis = ROI_vis[:, :bbox_num].to(torch.float32).t().reshape(bbox_num, batch, 1)
    fea = fea * vis
    fea_list = [fea[i] for i in range(bbox_num)]
    return fea_list


def GeneratorCNN_ID_Encoder_BodyROI(x, ROI_bboxs, bbox_num, z_num, repeat_num, hidden_num, data_format, activation_fn=relu,
                                    keep_part_prob=1.0, roi_size=48, reuse=False):
    """Reference models.py:275-325: the ROI encoder without visibility flags (the DeepFashion stage-II trainers,
    trainer_256.py:310-311, 604-605)."""
    with variable_scope("G_encoder", reuse=reuse) as vs:
        batch_num = x.shape[0]
        img_H, img_W = float(x.shape[1]), float(x.shape[2])
        x = slim.conv2d(x, hidden_num, 3, 1, activation_fn=activation_fn, data_format=data_format)
        x = slim.res_block(x, hidden_num, 3, activation_fn=activation_fn, data_format=data_format)
        boxes, box_ind = _normalised_boxes(ROI_bboxs, bbox_num, img_H, img_W)
        body_regions = A.crop_and_resize(x, boxes, box_ind, roi_size, roi_size)
        body_regions = _roi_tower(body_regions, z_num, repeat_num, hidden_num, data_format, activation_fn)
        fea = body_regions.reshape(bbox_num, batch_num, z_num)           # tf.split(body_regions, bbox_num, axis=0)
        fea_list = [fea[i] for i in range(bbox_num)]
        if keep_part_prob < 1.0:
            for i in range(bbox_num):
                keep = (torch.rand(1, 1, device=x.device) < keep_part_prob).to(torch.float32)     # bernoulliSample([p]) tiled
                fea_list[i] = fea_list[i] * keep
        fea_all = torch.cat(fea_list, dim=-1)
        variables = slim.get_variables(vs)
    return fea_all, fea_list, variables


def GeneratorCNN_ID_Encoder_BodyROIVis(x, ROI_bboxs, ROI_vis, bbox_num, z_num, repeat_num, hidden_num, data_format,
                                       activation_fn=relu, keep_part_prob=1.0, roi_size=48, reuse=False):
    """Reference models.py:328-388 (DeepFashion appearance encoder)."""
    with variable_scope("G_encoder", reuse=reuse) as vs:
        batch_num = x.shape[0]
        img_H, img_W = float(x.shape[1]), float(x.shape[2])
        x = slim.conv2d(x, hidden_num, 3, 1, activation_fn=activation_fn, data_format=data_format)
        x = slim.res_block(x, hidden_num, 3, activation_fn=activation_fn, data_format=data_format)
        x = _cut("E.towers_in", x)                     # backward stage boundary: [ROI tower] | [stem]

        boxes, box_ind = _normalised_boxes(ROI_bboxs, bbox_num, img_H, img_W)
        body_regions = A.crop_and_resize(x, boxes, box_ind, roi_size, roi_size)
        body_regions = _roi_tower(body_regions, z_num, repeat_num, hidden_num, data_format, activation_fn)
        if keep_part_prob < 1.0:
            fea_list = _apply_vis(body_regions, ROI_vis, bbox_num, z_num)
            for i in range(bbox_num):
                keep = (torch.rand(batch_num, 1, device=x.device) < keep_part_prob).to(torch.float32)
                fea_list[i] = fea_list[i] * keep
            fea_all = torch.cat(fea_list, dim=-1)
        else:
            # visibility multiply + concat (models.py:359-368, 387) as one launch; fea_list = views of the result
            fea_all = A.vis_concat(body_regions, ROI_vis, None, bbox_num, z_num)
            fea_list = [fea_all[:, i * z_num:(i + 1) * z_num] for i in range(bbox_num)]
        variables = slim.get_variables(vs)
    return fea_all, fea_list, variables


def GeneratorCNN_ID_Encoder_BodyROIVis_FgBgFeaTwoBranch(x, fg_mask, ROI_bboxs, ROI_vis, bbox_num, z_num, repeat_num,
                                                        hidden_num, data_format, activation_fn=relu,
                                                        keep_part_prob=1.0, roi_size=48, reuse=False):
    """Reference models.py:390-471 (Market-1501 Fg/Bg two-branch encoder)."""
    with variable_scope("G_encoder", reuse=reuse) as vs:
        batch_num = x.shape[0]
        img_H, img_W = float(x.shape[1]), float(x.shape[2])
        # Encoder stem
        x = slim.conv2d(x, hidden_num, 3, 1, activation_fn=activation_fn, data_format=data_format)
        x = slim.res_block(x, hidden_num, 3, activation_fn=activation_fn, data_format=data_format)

        _tap("E.stem", x)
        x_fg, x_bg = A.mask_split(x, fg_mask)          # x * m, x * (1 - m): one launch (csrc/dpig_glue.hip)
        x_fg, x_bg = _cut("E.towers_in", x_fg, x_bg)  # backward stage boundaries: [Bg tower] | [ROI tower] | [stem]

        boxes, box_ind = _normalised_boxes(ROI_bboxs, bbox_num, img_H, img_W)
        with A.side_branch(x_fg) as fg_branch:          # the ROI tower beside the background branch (A.TWO_STREAM)
            body_regions = A.crop_and_resize(x_fg, boxes, box_ind, roi_size, roi_size)
            conv_fea_list = [body_regions, x_bg]
            _tap("E.rois", body_regions)

            # Share weights for different body regions
            body_regions = _roi_tower(body_regions, z_num, repeat_num, hidden_num, data_format, activation_fn)
        fused_cat = keep_part_prob >= 1.0
        if not fused_cat:
            body_regions, _ = fg_branch.join(body_regions, conv_fea_list[0])     # (the crops are handed to the caller too)
            fea_list = _apply_vis(body_regions, ROI_vis, bbox_num, z_num)
            for i in range(bbox_num):
                keep = (torch.rand(batch_num, 1, device=x.device) < keep_part_prob).to(torch.float32)
                fea_list[i] = fea_list[i] * keep

        # Background branch
        _cut("E.bg_begin")
        for idx in range(repeat_num):
            channel_num = hidden_num * (idx + 1)
            x_bg = slim.res_block(x_bg, channel_num, 3, activation_fn=activation_fn, data_format=data_format)
            if idx < repeat_num - 1:
                x_bg = slim.conv2d(x_bg, hidden_num * (idx + 2), 3, 2, activation_fn=activation_fn,
                                   data_format=data_format)
        x_bg = x_bg.reshape(x_bg.shape[0], -1)
        x_bg = fully_connected(x_bg, z_num * 4, activation_fn=None)

        if fused_cat:
            body_regions, _ = fg_branch.join(body_regions, conv_fea_list[0])     # (the crops are handed to the caller too)
            # visibility multiply (models.py:433-442) + tf.concat(fea_list, -1) (:467-468) as one launch; fea_list = views
            fea_all = A.vis_concat(body_regions, ROI_vis, x_bg, bbox_num, z_num)
            fea_list = [fea_all[:, i * z_num:(i + 1) * z_num] for i in range(bbox_num)] + [fea_all[:, bbox_num * z_num:]]
        else:
            fea_list.append(x_bg)
            fea_all = torch.cat(fea_list, dim=-1)
        variables = slim.get_variables(vs)
    return fea_all, fea_list, conv_fea_list, variables


def GaussianFCRes(z_shape, out_channel, repeat_num, hidden_num, data_format, mean=0.0, stddev=0.2,
                  activation_fn=relu, reuse=False, z=None, device=None):
    """Reference models.py:474-486."""
    with variable_scope("G_FC", reuse=reuse) as vs:
        if z is None:
            z = torch.randn(tuple(z_shape), device=device) * stddev + mean
        z = fully_connected(z, hidden_num, activation_fn=activation_fn)
        for i in range(repeat_num):
            res = z
            z = fully_connected(z, hidden_num, activation_fn=activation_fn)
            z = fully_connected(z, hidden_num, activation_fn=activation_fn)
            z = res + z
        out = fully_connected(z, out_channel, activation_fn=None)
        variables = slim.get_variables(vs)
    return out, variables


def PoseEncoderFCRes(pose_rcv, z_num, repeat_num, hidden_num, data_format, activation_fn=relu, reuse=False):
    """Reference models.py:488-499."""
    with variable_scope("G_Pose_Encoder", reuse=reuse) as vs:
        x = fully_connected(pose_rcv, hidden_num, activation_fn=activation_fn)
        for i in range(repeat_num):
            res = x
            x = fully_connected(x, hidden_num, activation_fn=activation_fn)
            x = fully_connected(x, hidden_num, activation_fn=activation_fn)
            x = res + x
        out = fully_connected(x, z_num, activation_fn=None)
        variables = slim.get_variables(vs)
    return out, variables


class _BinaryRoundST(torch.autograd.Function):
    """models.py:97-108: round() with a straight-through (identity) gradient."""

    @staticmethod
    def forward(ctx, x):
        return torch.round(x)

    @staticmethod
    def backward(ctx, g):
        return g


def binaryRound(x):
    return _BinaryRoundST.apply(x)


def PoseDecoderFCRes(z, keypoint_num, repeat_num, hidden_num, data_format, activation_fn=relu, reuse=False):
    """Reference models.py:501-515."""
    with variable_scope("G_Pose_Decoder", reuse=reuse) as vs:
        x = fully_connected(z, hidden_num, activation_fn=None)
        for i in range(repeat_num):
            res = x
            x = fully_connected(x, hidden_num, activation_fn=activation_fn)
            x = fully_connected(x, hidden_num, activation_fn=activation_fn)
            x = res + x
        re_pose_coord = fully_connected(x, keypoint_num * 2, activation_fn=None)
        re_pose_visible = fully_connected(x, keypoint_num, activation_fn=torch.sigmoid)  # norm to (0, 1)
        re_pose_visible = binaryRound(re_pose_visible)
        variables = slim.get_variables(vs)
    return re_pose_coord, re_pose_visible, variables


def GeneratorCNN_ID_UAEAfterResidual(x, pose, input_channel, z_num, repeat_num, hidden_num, data_format,
                                     activation_fn=relu, min_fea_map_H=8, noise_dim=0, reuse=False):
    """Reference models.py:518-576 (U-Net style decoder G)."""
    with variable_scope("G", reuse=reuse) as vs:
        if data_format != 'NHWC':
            raise Exception("only NHWC is supported (main.py:18)")
        # Encoder
        encoder_layer_list = []
        tiled = (pose is not None and x.dim() == 4 and x.stride(1) == 0 and x.stride(2) == 0 and
                 x.shape[1] >= 2 and x.shape[2] >= 2 and activation_fn is slim.relu)
        if tiled:
            # x is a [B,E] embedding broadcast over H x W (trainer.py:588-590 embs_rep): the first conv
            # collapses exactly to a small GEMM + a thin conv over the pose channels
            src = getattr(x, "_dpig_src", None)       # the [B,E] tensor the caller expanded (saves autograd's zero-fill + sum over [B,H,W,E])
            x = slim.conv2d_tiled_embedding(src if src is not None else x[:, 0, 0, :], pose, hidden_num)
        else:
            if isinstance(pose, A.PoseKeypoints):      # only the collapsed first conv consumes keypoints: rasterise for the dense path
                pose = pose.dense()
            if pose is not None:
                x = torch.cat([x, pose], dim=3)
            x = slim.conv2d(x, hidden_num, 3, 1, activation_fn=activation_fn, data_format=data_format)
        _tap("G.stem", x)
        # The decoder's tf.concat([x, skip], 3) (models.py:560) is made free: each encoder block writes its output straight
        # into the upper channel slice of the buffer the decoder stage will read, and the decoder's producer (the upsampled
        # 1x1 conv of the stage before) writes the lower slice -- A.join_channels then hands out the buffer itself.
        placed = activation_fn is slim.relu and not os.environ.get('DPIG_NO_PLACED_CONCAT')
        dec_slices = []
        for idx in range(repeat_num):
            channel_num = hidden_num * (idx + 1)
            out_slice = None
            if placed and x.shape[-1] == channel_num:
                c_dec = channel_num if idx < repeat_num - 1 else hidden_num
                lo, out_slice = A.channel_slices(x.shape[0], x.shape[1], x.shape[2], c_dec, channel_num, x.dtype, x.device)
                dec_slices.append(lo)
            else:
                dec_slices.append(None)
            x = slim.res_block(x, channel_num, 3, activation_fn=activation_fn, data_format=data_format, out=out_slice)
            encoder_layer_list.append(x)
            if idx < repeat_num - 1:
                x = slim.conv2d(x, hidden_num * (idx + 2), 3, 2, activation_fn=activation_fn,
                                data_format=data_format)

        x_shape = list(x.shape)
        x = x.reshape(x_shape[0], int(np.prod(x_shape[1:])))
        z = x = fully_connected(x, z_num, activation_fn=None)
        _tap("G.z", z)
        if noise_dim > 0:
            noise = torch.rand(z.shape[0], noise_dim, device=z.device) * 2.0 - 1.0
            z = torch.cat([z, noise], dim=1)

        # Decoder
        x = fully_connected(z, x_shape[1] * x_shape[2] * hidden_num, activation_fn=None)
        x = reshape(x, x_shape[1], x_shape[2], hidden_num, data_format)
        lo = dec_slices[repeat_num - 1]
        if lo is not None and lo.shape == x.shape and lo.dtype == x.dtype:
            x = A.place(x, lo)                   # (the 8x4 map out of the FC layer: a tiny copy into its slice)

        for idx in range(repeat_num):
            x = A.join_channels(x, encoder_layer_list[repeat_num - 1 - idx])
            channel_num = x.shape[-1]
            x = slim.res_block(x, channel_num, 3, activation_fn=activation_fn, data_format=data_format)
            _tap("G.dec%d" % idx, x)
            if idx < repeat_num - 1:
                # x = upscale(x, 2, data_format); x = slim.conv2d(x, ..., 1, 1, ...)   (models.py:569-570)
                c_up = hidden_num * (repeat_num - idx - 1)
                lo = dec_slices[repeat_num - 2 - idx]
                if lo is not None and not (lo.shape[-1] == c_up and lo.shape[1] == 2 * x.shape[1] and lo.dtype == x.dtype):
                    lo = None
                x = slim.conv2d(x, c_up, 1, 1, activation_fn=activation_fn, data_format=data_format, upsample2x=True, out=lo)

        out = slim.conv2d(x, input_channel, 3, 1, activation_fn=None, data_format=data_format)
        variables = slim.get_variables(vs)
    return out, z, variables
