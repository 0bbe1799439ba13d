"""Logical-NCHW <-> physical-NHWC helpers shared by the tflib ops."""


def nchw_to_nhwc_view(x):
    """[N,C,H,W] logical tensor -> [N,H,W,C] view (no copy when x is channels_last)."""
    return x.permute(0, 2, 3, 1)


def nhwc_to_nchw_view(y):
    """[N,H,W,C] contiguous -> logical [N,C,H,W] (a channels_last strided view, no copy)."""
    return y.permute(0, 3, 1, 2)
