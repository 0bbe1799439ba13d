"""Seeded synthetic training batches with the shapes, value ranges and structure of the reference's
input pipeline (trainer.py:537-564; SURVEY.md 8d "Synthetic inputs").  numpy only, so the same
batch can feed the HIP path, the CPU oracle and the golden-vector generator.

  x         [B,H,W,3]  fp32 U(-1,1)                  (process_image: /127.5 - 1, trainer.py:557)
  pose      [B,H,W,18] -1 except a +1 disc around each visible keypoint; the disc is the 49-offset
                        stencil of tf_poseInflate (utils.py:289-318); keypoints dropped with p=0.1
  mask_r6   [B,H,W,1]  {0,1}: dilated union of the keypoint discs (~35 % ones)
  part_bbox [B,7,4]    int32 pixel (y1,x1,y2,x2), min side 8, or the invisible sentinel [0,0,1,1]
  part_vis  [B,7]      {0,1}, Bernoulli(0.9) (0 <=> sentinel box), convert_market.py:609-630
  pose_rcv  [B,54]     the keypoints `pose` was built from: pixel (row, col, visibility) x 18 (a dropped keypoint is (0, 0, 0))
"""
import numpy as np


def _stencil():
    offs = []
    for xo in (-4, 4):
        offs += [(xo, 0)]
    for xo in (-3, 3):
        offs += [(xo, yo) for yo in range(-2, 3)]
    for xo in (-2, 2):
        offs += [(xo, yo) for yo in range(-3, 4)]
    for xo in (-1, 1):
        offs += [(xo, yo) for yo in range(-3, 4)]
    offs += [(0, yo) for yo in range(-4, 5)]
    return offs        # 49 (row, col) offsets


_STENCIL = _stencil()


def make_batch(batch_size, img_H=128, img_W=64, seed=1234, keypoint_num=18, part_num=7):
    rng = np.random.default_rng(seed)
    B, H, W = batch_size, img_H, img_W
    x = rng.uniform(-1.0, 1.0, size=(B, H, W, 3)).astype(np.float32)
    pose = -np.ones((B, H, W, keypoint_num), dtype=np.float32)
    pose_rcv = np.zeros((B, keypoint_num, 3), dtype=np.float32)      # (row, col, visibility): what the records hold (trainer.py:556)
    mask = np.zeros((B, H, W, 1), dtype=np.float32)
    for b in range(B):
        # a crude "person": keypoints scattered around a vertical axis
        cx = rng.uniform(0.35, 0.65) * W
        for k in range(keypoint_num):
            if rng.uniform() < 0.1:
                continue
            r = int(np.clip(rng.uniform(0.08, 0.92) * H, 0, H - 1))
            c = int(np.clip(cx + rng.normal(0, 0.12) * W, 0, W - 1))
            pose_rcv[b, k] = (r, c, 1.0)
            for dr, dc in _STENCIL:
                rr, cc = r + dr, c + dc
                if 0 <= rr < H and 0 <= cc < W:
                    pose[b, rr, cc, k] = 1.0
            r0, r1 = max(r - 12, 0), min(r + 13, H)
            c0, c1 = max(c - 7, 0), min(c + 8, W)
            mask[b, r0:r1, c0:c1, 0] = 1.0
    part_bbox = np.zeros((B, part_num, 4), dtype=np.int32)
    part_vis = (rng.uniform(size=(B, part_num)) < 0.9).astype(np.float32)
    for b in range(B):
        for p in range(part_num):
            if part_vis[b, p] == 0:
                part_bbox[b, p] = [0, 0, 1, 1]
                continue
            h = int(rng.integers(8, max(9, H // 2)))
            w = int(rng.integers(8, max(9, W // 2)))
            y1 = int(rng.integers(0, H - 1 - h + 1))
            x1 = int(rng.integers(0, W - 1 - w + 1))
            part_bbox[b, p] = [y1, x1, min(y1 + h, H - 1), min(x1 + w, W - 1)]
    return {"x": x, "pose": pose, "mask_r6": mask, "part_bbox": part_bbox, "part_vis": part_vis,
            "pose_rcv": pose_rcv.reshape(B, keypoint_num * 3)}


def keypoints_only(batch):
    """The batch as the records deliver it: the keypoints `pose_rcv`, not the target map built from them (trainer.py:556-560)."""
    return {k: v for k, v in batch.items() if k != "pose"}


def make_batch_from_keypoints(batch_size, img_H=128, img_W=64, seed=1234, drop=0.15):
    """A batch whose geometric inputs are DERIVED from one set of keypoints per person the way the reference's converter derives
    them (datasets/convert_market.py:229-283, 578-638 restated in `dataprep`, pinned by tests/golden/prep_reference.npz): body
    mask = closed union of discs along the limbs, 7 part boxes grown from the visible keypoints (sentinel [0,0,1,1] + vis 0 for
    a part with none), pose map = the 49-offset disc per visible keypoint.  Same dict as `make_batch`, plus `keypoints`."""
    from . import dataprep
    rng = np.random.default_rng(seed)
    B, H, W = batch_size, img_H, img_W
    x = rng.uniform(-1.0, 1.0, size=(B, H, W, 3)).astype(np.float32)
    pose = -np.ones((B, H, W, 18), dtype=np.float32)
    mask = np.zeros((B, H, W, 1), dtype=np.float32)
    bbox = np.zeros((B, 7, 4), dtype=np.int32)
    vis = np.zeros((B, 7), dtype=np.float32)
    kps = np.zeros((B, 18, 3), dtype=np.float64)
    # a standing figure: (x, y) of the 18 MSCOCO keypoints as fractions of (W, H), jittered per person
    base = np.array([(.50, .10), (.50, .20), (.36, .21), (.30, .36), (.28, .50), (.64, .21), (.70, .36), (.72, .50), (.42, .52),
                     (.40, .72), (.40, .92), (.58, .52), (.60, .72), (.60, .92), (.54, .08), (.46, .08), (.58, .10), (.42, .10)])
    for b in range(B):
        j = base + rng.normal(0, 0.03, size=base.shape)
        kps[b, :, 0] = np.clip(np.floor(j[:, 0] * W), 0, W - 1)
        kps[b, :, 1] = np.clip(np.floor(j[:, 1] * H), 0, H - 1)
        kps[b, :, 2] = rng.uniform(size=18) >= drop
        if b % 4 == 3:
            kps[b, [8, 9, 10, 11, 12, 13], 2] = 0       # a cropped figure without legs: parts 3, 6, 7 get the sentinel box
        g = dataprep.model_inputs_from_keypoints(kps[b], H, W)
        mask[b], bbox[b], vis[b] = g["mask_r6"], g["part_bbox"], g["part_vis"]
        for k in range(18):
            if kps[b, k, 2]:
                r, c = int(kps[b, k, 1]), int(kps[b, k, 0])
                for dr, dc in _STENCIL:
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < H and 0 <= cc < W:
                        pose[b, rr, cc, k] = 1.0
    return {"x": x, "pose": pose, "mask_r6": mask, "part_bbox": bbox, "part_vis": vis, "keypoints": kps}


def to_device(batch, device):
    import torch
    return {k: torch.as_tensor(v).to(device) for k, v in batch.items()}
