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
