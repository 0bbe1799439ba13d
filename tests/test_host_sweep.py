"""Host-side sweep of the C ABI's planning surface (no device work, runs without a GPU): every function that sizes a workspace, decides
eligibility or plans a launch is called over a grid of ordinary, degenerate and hostile descriptors.  Properties: it returns (no crash,
no hang), sizes are finite and bounded, "not supported" answers are consistent between the query and the sizing function, and null /
non-positive descriptors are refused.  This file is also part of what the sanitizer build of the library's host code is run against
(SURVEY section 5's sanitizer row; the recipe and its record are named in profiles/r06_asan_ubsan_host.txt)."""
import ctypes
import itertools
import random

import pytest


@pytest.fixture(scope="module")
def H():
    import dpig_amd.hip_ops as H
    return H


def _descs(H):
    rnd = random.Random(7)
    dims = [(1, 1, 1), (1, 2, 2), (2, 3, 3), (1, 5, 7), (16, 8, 4), (16, 128, 64), (8, 256, 256), (112, 48, 48), (56, 1, 1), (3, 63, 31), (1, 2, 1024)]
    chans = [(3, 64), (64, 3), (6, 20), (18, 128), (72, 24), (64, 64), (128, 128), (384, 512), (1024, 1024), (8, 8), (32, 40)]
    ks = [(1, 1), (3, 1), (3, 2), (5, 2), (5, 1), (1, 2)]
    out = []
    for (N, Hh, W), (C, K), (R, st) in itertools.product(dims, chans, ks):
        if rnd.random() < 0.35:
            continue
        for up in ((False, True) if (R == 1 and st == 1) else (False,)):
            for slack in (0, 8, 2048):
                if slack == 2048 and rnd.random() < 0.8:
                    continue
                d = H._desc(N, Hh, W, C, K, R, R, st, C + slack, K + slack, upsample2x=up)
                out.append(d)
    return out


def test_planning_surface_over_a_descriptor_grid(H):
    lib = H.lib()
    descs = _descs(H)
    assert len(descs) > 800
    TB = 1 << 42                                    # no workspace of a legal launch is anywhere near 4 TB
    for d in descs:
        for compute in (H.COMPUTE_F32, H.COMPUTE_BF16, H.COMPUTE_BF16X3, H.COMPUTE_BF16_STORE):
            d.compute = compute
            r = ctypes.byref(d)
            for which in (0, 1, 2):
                assert 0 <= lib.dpig_conv2d_workspace_bytes(r, which) < TB
                sup = lib.dpig_conv2d_bf16_supported(r, which)
                assert sup in (0, 1)
                ws = lib.dpig_conv2d_bf16_workspace_bytes(r, which)
                assert 0 <= ws < TB              # (sized whether or not the bf16 kernels take the layer: callers ask _supported first)
            for which in (0, 1):
                el = lib.dpig_conv2d_wino_eligible(r, which)
                ws = lib.dpig_conv2d_wino_workspace_bytes(r, which)
                assert el in (0, 1) and 0 <= ws < TB
                if d.R != 3 or d.stride != 1 or d.C % 64 or d.K % 64 or d.H % 2 or d.W % 2 or d.upsample2x:
                    assert el == 0 and ws == 0
                el4 = lib.dpig_conv2d_wino4_eligible(r, which)                    # the F(4x4, 3x3) form: sides multiples of 4, a block form
                ws4 = lib.dpig_conv2d_wino4_workspace_bytes(r, which)
                assert el4 in (0, 1) and 0 <= ws4 < TB
                tw = d.W // 4
                if d.R != 3 or d.stride != 1 or d.C % 64 or d.K % 64 or d.H % 4 or d.W % 4 or d.upsample2x or (tw % 2 and tw % 3):
                    assert el4 == 0 and ws4 == 0
                assert compute == H.COMPUTE_F32 or el4 == 0
            el = lib.dpig_conv2d_wgrad_wino_eligible(r)
            ws = lib.dpig_conv2d_wgrad_wino_workspace_bytes(r)
            assert el in (0, 1) and 0 <= ws < TB and (compute == H.COMPUTE_F32 or el == 0)
            assert lib.dpig_conv2d_bn_stats_tiles(r) >= 0 and lib.dpig_conv2d_bn_stats_tiles_ws(r) >= 0
            assert lib.dpig_conv2d_bf16_bn_stats_tiles(r) >= 0


def test_hostile_descriptors_are_refused_not_dereferenced(H):
    lib = H.lib()
    good = H._desc(2, 8, 8, 64, 64, 3, 3, 1, 64, 64)
    for field, value in (("N", 0), ("N", -1), ("H", 0), ("W", -5), ("C", 0), ("K", -64), ("R", 0), ("R", 6), ("S", 7), ("stride", 0), ("stride", 3),
                         ("ldx", 1), ("ldy", 0), ("act", 9), ("act", -1), ("N", 2 ** 30), ("H", 2 ** 20)):
        d = H._desc(2, 8, 8, 64, 64, 3, 3, 1, 64, 64)
        setattr(d, field, value)
        r = ctypes.byref(d)
        for which in (0, 1, 2):
            lib.dpig_conv2d_workspace_bytes(r, which)
            lib.dpig_conv2d_bf16_workspace_bytes(r, which)
            lib.dpig_conv2d_bf16_supported(r, which)
        assert lib.dpig_conv2d_wino_eligible(r, 0) in (0, 1) and lib.dpig_conv2d_wgrad_wino_eligible(r) in (0, 1)
        lib.dpig_conv2d_wino_workspace_bytes(r, 0)
        lib.dpig_conv2d_wgrad_wino_workspace_bytes(r)
        assert lib.dpig_conv2d_wino4_eligible(r, 0) in (0, 1) and lib.dpig_conv2d_wino4_eligible(r, 1) in (0, 1)
        lib.dpig_conv2d_wino4_workspace_bytes(r, 0)
        assert lib.dpig_conv2d_fwd_wino4(r, None, None, None, None, None, None, None, 0, None) != 0
        assert lib.dpig_conv2d_dgrad_wino4(r, None, None, None, None, None, None, 0, None) != 0
        # the launching entry points refuse such a descriptor (or their null tensors) BEFORE touching the device: non-zero status, a message
        rc = lib.dpig_conv2d_fwd(r, None, None, None, None, None, None, None, 0, None)
        assert rc != 0 and lib.dpig_last_error()
        assert lib.dpig_conv2d_dgrad(r, None, None, None, None, None, None, 0, None) != 0
        assert lib.dpig_conv2d_wgrad(r, None, None, None, 0.0, None, 0.0, None, 0, None) != 0
    null = ctypes.POINTER(type(good))()
    assert lib.dpig_conv2d_workspace_bytes(null, 0) == 0 and lib.dpig_conv2d_bf16_workspace_bytes(null, 0) == 0
    assert lib.dpig_conv2d_wino_eligible(null, 0) == 0 and lib.dpig_conv2d_wgrad_wino_eligible(null) == 0
    assert lib.dpig_conv2d_wino4_eligible(null, 0) == 0 and lib.dpig_conv2d_wino4_workspace_bytes(null, 0) == 0
    assert lib.dpig_conv2d_fwd(null, None, None, None, None, None, None, None, 0, None) != 0


def test_scalar_sizing_functions_over_a_grid(H):
    lib = H.lib()
    for rows, cols in itertools.product((0, 1, 7, 128, 4097, 1 << 20, 1 << 31), (0, 1, 3, 8, 64, 1000, 20480)):
        for fn in (lib.dpig_colsum_workspace_bytes, lib.dpig_bn_workspace_bytes, lib.dpig_bn_bf16_workspace_bytes):
            assert 0 <= fn(rows, cols) < (1 << 44)
    for a, b, c in itertools.product((0, 1, 16, 2048), (0, 1, 64, 20480), (0, 1, 128, 1024)):
        for fn in (lib.dpig_ln_fwd_workspace_bytes, lib.dpig_ln_workspace_bytes, lib.dpig_ln_bwd2_workspace_bytes, lib.dpig_ssim_workspace_bytes):
            assert 0 <= fn(a, b, c) < (1 << 44)
        for which in (0, 1, 2):
            assert 0 <= lib.dpig_linear_workspace_bytes(a, b, c, which) < (1 << 44)
        assert 0 <= lib.dpig_crop_resize_bwd_workspace_bytes(a, b, c, 48) < (1 << 44)
        assert 0 <= lib.dpig_border_class_sum_workspace_bytes(a % 17, b % 300, c % 300, 64) < (1 << 44)
    for C, K in itertools.product((0, -64, 3, 64, 96, 128, 1024), repeat=2):
        n = lib.dpig_wino_filter_elems(C, K)
        assert n == (16 * C * K if (C > 0 and K > 0 and C % 64 == 0 and K % 64 == 0) else 0)
        assert lib.dpig_wino4_filter_elems(C, K) == (36 * C * K if (C > 0 and K > 0 and C % 64 == 0 and K % 64 == 0) else 0)
    for inp, k, s in itertools.product((1, 2, 3, 63, 64, 128), (1, 3, 5), (1, 2)):
        out, pad = ctypes.c_int(), ctypes.c_int()
        assert lib.dpig_same_pad(inp, k, s, ctypes.byref(out), ctypes.byref(pad)) == 0
        assert out.value == -(-inp // s) and pad.value == max((out.value - 1) * s + k - inp, 0) // 2
