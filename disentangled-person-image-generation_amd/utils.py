"""Host-side mirror of the reference's pose-map helpers of the input pipeline (`utils.py:237-318`, called from
`trainer.py:556-560`), running on the device through the C ABI.

    poses = tf_poseInflate(coord2channel_simple_rcv(poses_rcv, 18, is_normalized=False, img_H=128, img_W=64),
                           keypoint_num=18, radius=4, img_H=128, img_W=64)

keeps working with the same names; `pose_target_from_rcv` is the same result in one launch straight from the
coordinates (no [B,H,W,18] intermediate, no 49 shifted adds).  Keypoints outside the image are dropped."""
from . import hip_ops as H


def coord2channel_simple_rcv(RCV, keypoint_num=18, is_normalized=True, img_H=128, img_W=64):
    return H.pose_points(RCV, img_H, img_W, keypoint_num, is_normalized)


def tf_poseInflate(G_pose, keypoint_num, radius=4, img_H=128, img_W=64):
    if radius != 4:
        raise Exception('only the radius-4 stencil of the reference (utils.py:300-314) is implemented')
    if tuple(G_pose.shape[1:]) != (img_H, img_W, keypoint_num):
        raise Exception('pose map must be [B, img_H, img_W, keypoint_num]')
    return H.pose_inflate(G_pose)


def pose_target_from_rcv(RCV, keypoint_num=18, is_normalized=False, img_H=128, img_W=64):
    return H.pose_rasterize(RCV, img_H, img_W, keypoint_num, is_normalized)


def ssim_G_x(G_255, x_pm1):
    """The SSIM list of `trainer.generate()` (trainer.py:516-521) on the device: G holds 0..255 pixel values (the
    denormalised generator output), x the [-1,1] input batch; returns the per-image SSIM tensor [B] (the reference
    logs its mean).  Replaces skimage's CPU rgb2gray + compare_ssim."""
    return H.ssim_gray_u8(G_255, (x_pm1 + 1) * 127.5)
