"""The drop-in boundary: libdpig_hip.so builds for gfx950, loads without a GPU, exports exactly the
symbols include/dpig_hip.h declares, the header is plain C, and the host-side argument checking
works (no compute calls here -- those are the -m gpu tests)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dpig_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    from dpig_amd import _lib
    return _lib


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dpig_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 30
    h = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(h, n), "libdpig_hip.so does not export %s" % n
        assert n in lib.SYMBOLS, "_lib.SYMBOLS does not bind %s" % n
    assert sorted(lib.SYMBOLS) == names, "bindings and header disagree"


def test_header_is_plain_c_and_struct_layout_matches(lib, tmp_path):
    prog = tmp_path / "t.c"
    prog.write_text('#include <stdio.h>\n#include "dpig_hip.h"\n'
                    'int main(void){printf("%zu\\n", sizeof(DpigConvDesc));return 0;}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(prog),
                           "-o", str(exe)])
    size = int(subprocess.check_output([str(exe)]).decode())
    assert size == ctypes.sizeof(lib.DpigConvDesc)


def test_same_pad_and_version(lib):
    h = lib.lib()
    assert h.dpig_version() == 283
    o, p = ctypes.c_int(), ctypes.c_int()
    for inp, k, s, eo, ep in [(128, 3, 1, 128, 1), (128, 3, 2, 64, 0), (64, 5, 2, 32, 1), (7, 3, 2, 4, 1)]:
        h.dpig_same_pad(inp, k, s, ctypes.byref(o), ctypes.byref(p))
        assert (o.value, p.value) == (eo, ep)
        assert lib.same_pad(inp, k, s) == (eo, ep)


def test_kernel_family_switches_validate_their_arguments(lib):
    """The process-wide kernel-choice setters are host state only: callable without a device, out-of-range values are DPIG_EINVAL
    with a message, valid ones leave the defaults' results unchanged (same plans, same bits: the GPU tests compare them)."""
    h = lib.lib()
    for fn, bad, good in ((h.dpig_conv_bf16_set_wave8, (-1, 8), (0, 7, 3)),):
        for v in bad:
            assert fn(v) != 0
            assert b"wave8" in h.dpig_last_error()
        for v in good:
            assert fn(v) == 0
    assert h.dpig_conv_bf16_set_large_tile(3, 0) != 0 and h.dpig_conv_bf16_set_large_tile(1, 3) != 0
    assert h.dpig_conv_bf16_set_large_tile(1, 0) == 0 and h.dpig_conv_bf16_set_large_tile_wgrad(1, 0) == 0


def _desc(lib, **kw):
    d = lib.DpigConvDesc()
    base = dict(N=16, H=8, W=4, C=768, K=768, R=3, S=3, stride=1, pad_t=-1, pad_l=-1, ldx=768, ldy=768)
    base.update(kw)
    for k, v in base.items():
        setattr(d, k, v)
    return d


def test_workspace_queries_and_split_k_plan(lib):
    h = lib.lib()
    # dec0 (M=512 rows): 24 tiles only -> the library splits K and asks for a partial-sum workspace
    d = _desc(lib)
    ws = h.dpig_conv2d_workspace_bytes(ctypes.byref(d), 0)
    assert ws > 0 and ws % (512 * 768 * 4) == 0
    # dec4 has 2048 tiles -> no split, no workspace
    d = _desc(lib, H=128, W=64, C=256, K=256, ldx=256, ldy=256)
    assert h.dpig_conv2d_workspace_bytes(ctypes.byref(d), 0) == 0
    assert h.dpig_conv2d_workspace_bytes(ctypes.byref(d), 2) > 0        # wgrad: 36 tiles -> split over pixels
    d.split_k = 1
    assert h.dpig_conv2d_workspace_bytes(ctypes.byref(d), 2) == 0


def test_bad_arguments_fail_loudly_without_touching_the_gpu(lib):
    h = lib.lib()
    rc = h.dpig_conv2d_fwd(None, None, None, None, None, None, None, None, 0, None)
    assert rc == -22 and b"descriptor" in h.dpig_last_error()
    d = _desc(lib, R=7, S=7)
    assert h.dpig_conv2d_fwd(ctypes.byref(d), 8, 8, None, None, 8, None, None, 0, None) == -22
    d = _desc(lib, stride=3)
    assert h.dpig_conv2d_dgrad(ctypes.byref(d), 8, 8, None, None, 8, None, 0, None) == -22
    d = _desc(lib, ldx=4)
    assert h.dpig_conv2d_wgrad(ctypes.byref(d), 8, 8, 8, 0.0, None, 0.0, None, 0, None) == -22
    d = _desc(lib)        # needs a workspace (split-K) but none is given
    assert h.dpig_conv2d_fwd(ctypes.byref(d), 16, 16, None, None, 16, None, None, 0, None) == -12
    with pytest.raises(RuntimeError):
        lib.check(-22, "unit test")
    assert h.dpig_adam_step(16, 16, 16, 16, 4, 16, 0.5, 0.999, 1e-8, 0, 1.0, None) == -22     # step must be >= 1


def test_product_path_has_no_cpu_fallback(lib):
    import torch
    import dpig_amd.hip_ops as H
    with pytest.raises(RuntimeError):
        H.conv2d_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(3, 3, 8, 8))
    # and nothing under the product package imports the oracle
    pkg = os.path.join(ROOT, "disentangled-person-image-generation_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


# ---- property tests of the host-side planning code (pure host functions of the library; no GPU) ------------------
from hypothesis import given, settings, strategies as st   # noqa: E402


@settings(max_examples=300, deadline=None)
@given(st.integers(1, 4096), st.integers(1, 7), st.integers(1, 2))
def test_same_pad_equals_the_oracle_formula(inp, k, s):
    from dpig_amd import _lib
    from oracle import ops as O
    o, p = ctypes.c_int(), ctypes.c_int()
    assert _lib.lib().dpig_same_pad(inp, k, s, ctypes.byref(o), ctypes.byref(p)) == 0
    eo, before, after = O.same_pad(inp, k, s)
    assert (o.value, p.value) == (eo, before) and before <= after <= before + 1


_dims = st.tuples(st.integers(1, 32), st.sampled_from([3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 128]),
                  st.sampled_from([3, 4, 6, 8, 12, 16, 24, 32, 48, 64]), st.sampled_from([3, 18, 32, 64, 128, 256, 640, 1024]),
                  st.sampled_from([3, 32, 64, 128, 256, 640, 1024]), st.sampled_from([1, 3, 5]), st.sampled_from([1, 2]),
                  st.sampled_from([0, 1]))


@settings(max_examples=200, deadline=None)
@given(_dims)
def test_workspace_plan_invariants(dims):
    """For any layer shape: the query is total (no crash, 0 for bad descriptors), forcing split_k=1 on the GEMM
    kernels needs no workspace, and an automatic plan asks for a whole number of partial-sum slabs, at most 64 (256 = one per CU
    for problems of at most 8 tiles with a very deep reduction: the fully connected layers)."""
    from dpig_amd import _lib
    N, H, W, C, K, k, s, compute = dims
    h = _lib.lib()
    d = _lib.DpigConvDesc()
    for name, v in dict(N=N, H=H, W=W, C=C, K=K, R=k, S=k, stride=s, pad_t=-1, pad_l=-1, ldx=C, ldy=K,
                        compute=compute).items():
        setattr(d, name, v)
    Ho, Wo = -(-H // s), -(-W // s)
    slab = {0: N * Ho * Wo * K * 4, 1: N * H * W * C * 4, 2: (k * k * C * K + K) * 4}
    thin = C <= 4 or K <= 4 or (C == 18 and k == 3)              # vector-ALU kernels have their own (small) workspaces
    for which in (0, 1, 2):
        auto = h.dpig_conv2d_workspace_bytes(ctypes.byref(d), which)
        assert 0 <= auto < (1 << 40)
        if not thin and auto:
            if which == 1 and s == 2:                              # parity classes: slabs of the class sizes, same bound
                assert auto <= 256 * slab[1]
            else:
                assert auto % slab[which] == 0 and auto // slab[which] <= 256, (which, auto, slab[which])
        d.split_k = 1
        if not thin:
            assert h.dpig_conv2d_workspace_bytes(ctypes.byref(d), which) == 0
        d.split_k = 0
    d.stride = 3
    assert h.dpig_conv2d_workspace_bytes(ctypes.byref(d), 0) == 0


def test_missing_library_is_a_loud_error(tmp_path):
    """No .so, no product: the binding raises (and names the build command) instead of computing some other way."""
    env = dict(os.environ, DPIG_LIB_PATH=str(tmp_path / "absent.so"), PYTHONPATH=ROOT)
    code = ("from dpig_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept RuntimeError as e:\n    print('RuntimeError', 'g.build()' in str(e))\n")
    out = subprocess.check_output([sys.executable, "-c", code], env=env, cwd=ROOT).decode()
    assert out.strip() == "RuntimeError True"
