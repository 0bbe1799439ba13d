"""Host-side planning of the Winograd family (no device work: eligibility = shape rules AND cost model, workspace = the split plan):
the choices measured on the GPU (profiles/r05_wino_layers.txt: on every Market layer shape the model picks the faster kernel) are pinned
here so that a change to the model's constants shows up on the CPU suite."""
import ctypes

import pytest

MARKET = [  # (name, N, H, W, C): BASELINE configs[1] layer shapes at bs = 16 (ROI tower: 7 parts x 16 images)
    ("enc0", 16, 128, 64, 128), ("roi b0", 112, 48, 48, 128), ("enc1", 16, 64, 32, 256), ("roi b1", 112, 24, 24, 256),
    ("enc2", 16, 32, 16, 384), ("roi b2", 112, 12, 12, 384), ("enc3", 16, 16, 8, 512), ("roi b3", 112, 6, 6, 512),
    ("enc4", 16, 8, 4, 640), ("dec0", 16, 8, 4, 768), ("dec1", 16, 16, 8, 1024), ("dec2", 16, 32, 16, 768),
    ("dec3", 16, 64, 32, 512), ("dec4", 16, 128, 64, 256)]
WGRAD_DIRECT = {"enc4"}                  # 8 x 4 C640: 128 tiles x 100 blocks cannot feed the tile-reduction kernel, the direct wgrad is faster (88 vs 64 TF measured); C768 the other way (75 vs 88)


@pytest.fixture(scope="module")
def H():
    import __graft_entry__
    __graft_entry__.build()
    import dpig_amd.hip_ops as H
    return H


def _desc(H, N, Hh, W, C, K, R=3, stride=1):
    return H._desc(N, Hh, W, C, K, R, R, stride, C, K)


def test_cost_model_choices_on_the_market_layers(H):
    lib = H.lib()
    lib.dpig_conv_wino_set_mode(1)
    for name, N, Hh, W, C in MARKET:
        d = _desc(H, N, Hh, W, C, C)
        assert lib.dpig_conv2d_wino_eligible(ctypes.byref(d), 0) == 1, name
        assert lib.dpig_conv2d_wino_eligible(ctypes.byref(d), 1) == 1, name
        assert lib.dpig_conv2d_wgrad_wino_eligible(ctypes.byref(d)) == (0 if name in WGRAD_DIRECT else 1), name


F4_CHOICE = {"enc0", "roi b0", "enc1", "roi b1", "enc2", "roi b2", "dec1", "dec2", "dec3", "dec4"}      # measured faster on each (profiles/r06_wino4_layers.txt)
F4_NO_FORM = {"roi b3", "enc4", "dec0"}          # 6 x 6: sides no multiples of 4; 8 x 4: one tile column


def test_f4_cost_model_choices_on_the_market_layers(H):
    """The F(4x4, 3x3) kernel's selection on the Market layer shapes: the form where it measured faster than F(2x2, 3x3), the 3-wide
    block form on the 12 x 12 level, F(2x2) on 16 x 8 C512 (a cost-model call: 1.15x stand-alone, cancelled by its image refresh) and on
    the maps without a form; mode 0 switches it off, mode 2 takes every layer with the form."""
    lib = H.lib()
    lib.dpig_conv_wino_set_mode(1)
    try:
        lib.dpig_conv_wino4_set_mode(1)
        for name, N, Hh, W, C in MARKET:
            d = _desc(H, N, Hh, W, C, C)
            for which in (0, 1):
                assert lib.dpig_conv2d_wino4_eligible(ctypes.byref(d), which) == (1 if name in F4_CHOICE else 0), (name, which)
        lib.dpig_conv_wino4_set_mode(2)
        for name, N, Hh, W, C in MARKET:
            d = _desc(H, N, Hh, W, C, C)
            assert lib.dpig_conv2d_wino4_eligible(ctypes.byref(d), 0) == (0 if name in F4_NO_FORM else 1), name
        # split plans: 32 x 16 C384 has 96 items for 256 CUs -> two input-channel ranges, their partial outputs in the workspace
        d = _desc(H, 16, 32, 16, 384, 384)
        assert lib.dpig_conv2d_wino4_workspace_bytes(ctypes.byref(d), 0) == 2 * 16 * 32 * 16 * 384 * 4
        d = _desc(H, 16, 128, 64, 256, 256)
        assert lib.dpig_conv2d_wino4_workspace_bytes(ctypes.byref(d), 0) == 0
        lib.dpig_conv_wino4_set_mode(0)
        for name, N, Hh, W, C in MARKET:
            d = _desc(H, N, Hh, W, C, C)
            assert lib.dpig_conv2d_wino4_eligible(ctypes.byref(d), 0) == 0
        assert lib.dpig_conv_wino4_set_mode(3) != 0 and lib.dpig_conv_wino4_set_mode(-1) != 0
    finally:
        lib.dpig_conv_wino4_set_mode(1)


def test_shapes_without_a_winograd_form_are_refused_in_every_mode(H):
    lib = H.lib()
    try:
        for mode in (1, 2):
            lib.dpig_conv_wino_set_mode(mode)
            for d in (_desc(H, 16, 64, 32, 256, 384, stride=2), _desc(H, 16, 64, 32, 256, 256, R=5), _desc(H, 16, 64, 32, 256, 256, R=1),
                      _desc(H, 16, 63, 32, 256, 256), _desc(H, 16, 64, 32, 96, 256), _desc(H, 16, 64, 32, 256, 32)):
                assert lib.dpig_conv2d_wino_eligible(ctypes.byref(d), 0) == 0
                assert lib.dpig_conv2d_wino_eligible(ctypes.byref(d), 1) == 0
                assert lib.dpig_conv2d_wgrad_wino_eligible(ctypes.byref(d)) == 0
                assert lib.dpig_conv2d_wino_workspace_bytes(ctypes.byref(d), 0) == 0
        lib.dpig_conv_wino_set_mode(0)                         # family off: nothing is eligible
        d = _desc(H, 16, 64, 32, 512, 512)
        assert lib.dpig_conv2d_wino_eligible(ctypes.byref(d), 0) == 0 and lib.dpig_conv2d_wgrad_wino_eligible(ctypes.byref(d)) == 0
    finally:
        lib.dpig_conv_wino_set_mode(1)
    assert lib.dpig_conv_wino_set_mode(3) != 0 and lib.dpig_conv_wino_set_mode(-1) != 0


def test_wgrad_refuses_shapes_beyond_its_24_bit_offset_arithmetic(H):
    """wino_wgrad_kernel forms byte offsets with 24-bit multiplies: a row pitch W * ld * 4 or a row count N * H of 2^23 or more passes the
    total-bytes bound but would address wrongly -- the eligibility test (and the entry point) refuse it.  Forward / dgrad use full
    32-bit multiplies and keep accepting the same descriptor."""
    lib = H.lib()
    prev = lib.dpig_conv_wino_get_mode()
    try:
        lib.dpig_conv_wino_set_mode(2)
        ok = H._desc(1, 2, 1024, 64, 64, 3, 3, 1, 1024, 64)          # pitch 1024 * 1024 * 4 = 2^22
        assert lib.dpig_conv2d_wgrad_wino_eligible(ctypes.byref(ok)) == 1
        wide_x = H._desc(1, 2, 1024, 64, 64, 3, 3, 1, 2048, 64)      # x pitch 2^23
        wide_y = H._desc(1, 2, 1024, 64, 64, 3, 3, 1, 64, 2048)      # dy pitch 2^23
        for d in (wide_x, wide_y):
            assert lib.dpig_conv2d_wgrad_wino_eligible(ctypes.byref(d)) == 0
            assert lib.dpig_conv2d_wgrad_wino_workspace_bytes(ctypes.byref(d)) == 0
        assert lib.dpig_conv2d_wino_eligible(ctypes.byref(wide_x), 0) == 1
    finally:
        lib.dpig_conv_wino_set_mode(prev)
    assert lib.dpig_conv_wino_get_mode() == prev


def test_workspace_sizes_follow_the_split_plans(H):
    """Forward / dgrad: 0 for layers whose 64-tile x 64-channel grid fills whole rounds of the 256 CUs, else s partial outputs
    (2 <= s <= 16, every range >= 6 chunks of 8 input channels).  Filter gradient: S slabs of [3][3][C][K] (the two position halves of a workgroup meet in LDS) + S rows of [K]."""
    lib = H.lib()
    lib.dpig_conv_wino_set_mode(1)
    for name, N, Hh, W, C in MARKET:
        d = _desc(H, N, Hh, W, C, C)
        out_bytes = N * Hh * W * C * 4
        items = -(-(N * (Hh // 2) * (W // 2)) // 64) * (C // 64)
        for which in (0, 1):
            ws = lib.dpig_conv2d_wino_workspace_bytes(ctypes.byref(d), which)
            assert ws % out_bytes == 0, name
            s = ws // out_bytes
            assert s == 0 or (2 <= s <= 16 and (C // 8) // s >= 6), (name, s)
            if items % 256 == 0:
                assert s == 0, (name, items)                  # whole rounds already
            if items < 128:
                assert s >= 2, (name, items)                  # fewer workgroups than half the chip: the plan splits
        wsg = lib.dpig_conv2d_wgrad_wino_workspace_bytes(ctypes.byref(d))
        per_split = (9 * C * C + C) * 4
        assert wsg > 0 and wsg % per_split == 0, name
        S = wsg // per_split
        chunks = -(-(N * (Hh // 2) * (W // 2)) // 8)
        assert 1 <= S <= 256 and (S == 1 or chunks // S >= 16), (name, S)


def test_filter_job_planning(H):
    """dpig_wino_filter_jobs_plan: first_block = the running sum of (C / 64)(K / 8); a filter without the form makes the whole plan fail."""
    lib = H.lib()
    shapes = [(64, 64), (128, 64), (256, 256), (64, 192)]
    jobs = (H.WinoFilterJob * len(shapes))()
    for j, (C, K) in zip(jobs, shapes):
        j.w, j.u_fwd, j.u_dgrad, j.C, j.K = 4096, 8192, 12288, C, K
    total = lib.dpig_wino_filter_jobs_plan(ctypes.byref(jobs), len(jobs))
    blocks = [(C // 64) * (K // 8) for C, K in shapes]
    assert total == sum(blocks)
    assert [j.first_block for j in jobs] == [sum(blocks[:i]) for i in range(len(blocks))]
    jobs[2].C = 96
    assert lib.dpig_wino_filter_jobs_plan(ctypes.byref(jobs), len(jobs)) == 0
    jobs[2].C = 256
    jobs[1].w = None
    assert lib.dpig_wino_filter_jobs_plan(ctypes.byref(jobs), len(jobs)) == 0
    assert lib.dpig_wino_filter_jobs_plan(None, 0) == 0
    assert lib.dpig_wino_filter_transform_jobs(None, 0, 0, None) != 0          # refused before any launch
