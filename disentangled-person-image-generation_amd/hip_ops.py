"""Thin functional layer over the C ABI: torch tensors in, torch tensors out, no autograd.

Every function launches hand-written gfx950 kernels from libdpig_hip.so on torch's current HIP
stream.  Tensors are NHWC; a tensor may be a channel slice of a wider NHWC buffer (stride of
the W axis = "ld"), which is how channel concats (models.py:524,560) cost nothing.

Arithmetic modes (`set_compute`):
  'f32'    the reference's arithmetic: fp32 tensors, exact fp32 products (BASELINE configs 1-2);
  'bf16'   bf16 STORAGE (BASELINE configs 3-5): activations and their gradients are torch.bfloat16 tensors, conv
           filters are read from bf16 shadows of the fp32 masters, products run on the bf16 matrix pipe, every
           accumulation / bias / residual / normalisation is fp32, parameter gradients and the optimizer stay fp32.
           Ops whose kernels exist in fp32 only (thin 3-channel convs, norms, FC layers, crops) convert at their
           boundary (`to_f32` / `to_bf16` kernels); a tensor is stored as bf16 iff its channel count is a multiple of 8;
  'bf16c'  round 1's intermediate mode: fp32 tensors, conv operands rounded to bf16 on their way into LDS;
  'f32w'   'f32' with the 3x3 stride-1 convs evaluated by Winograd minimal filtering on the fp32 matrix pipe
           (csrc/dpig_conv_wino.hip: fp32 tensors, fp32 products, fp32 accumulation, 2.25x fewer multiplies): forward and dgrad by
           F(2x2,3x3), the filter gradient by F(3x3,2x2) (`conv2d_wgrad` -> dpig_conv2d_wgrad_wino), each where the layer has the form
           (3x3, stride 1, even H and W, C and K multiples of 64) AND the library's cost model says it pays; everything else on the
           'f32' kernels.  bench.py's headline mode since round 5;
  'bf16x3' fp32 tensors, fp32 accuracy class on the bf16 pipe: conv operands are split into two bf16 terms on their way
           into LDS and every product block is three bf16 MFMAs (DPIG_COMPUTE_BF16X3, include/dpig_hip.h).  An opt-in
           mode, held to the exact path's kernel bar (2e-5 max|ref|); 'f32' stays the exact-products default.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU, DpigConvDesc, check, lib, ptr, stream_ptr, workspace


# Optional per-launch timing (bench.py's roofline leg): when PROFILE is a list, every conv launch is
# bracketed by HIP events recorded on the stream the kernel is launched on (torch's current stream)
# and (class, executed FLOPs, start, end) is appended.  None = off (no overhead).
PROFILE = None


class _Timed(object):
    def __init__(self, kind, flops, label=None):
        self.kind, self.flops, self.label = kind, flops, label

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, self.e1) if self.label is None
                           else (self.kind, self.flops, self.e0, self.e1, self.label))
        return False


BF16 = torch.bfloat16
F32 = torch.float32


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("dpig HIP ops need device tensors (got %s): there is no CPU fallback" % t.device)
    if t.dtype != torch.float32:
        raise RuntimeError("dpig HIP ops are fp32 (got %s)" % t.dtype)


def _require_dev(t):
    if not t.is_cuda:
        raise RuntimeError("dpig HIP ops need device tensors (got %s): there is no CPU fallback" % t.device)
    if t.dtype not in (F32, BF16):
        raise RuntimeError("dpig HIP ops take fp32 or bf16 tensors (got %s)" % t.dtype)


def nhwc_ld(t):
    """Channel stride of an NHWC tensor that is dense except for a channel slice; None if not."""
    if t.dim() != 4:
        return None
    n, h, w, c = t.shape
    s = t.stride()
    if c > 1 and s[3] != 1:
        return None
    ld = s[2] if w > 1 else (s[1] // max(w, 1) if h > 1 else (s[0] // max(h * w, 1) if n > 1 else c))
    if ld < c:
        return None
    if (w > 1 and s[2] != ld) or (h > 1 and s[1] != w * ld) or (n > 1 and s[0] != h * w * ld):
        return None
    return ld


def as_nhwc(t):
    """Return (tensor, ld) with tensor usable by the kernels (copy only if the layout is exotic)."""
    ld = nhwc_ld(t)
    if ld is None:
        t = t.contiguous()
        ld = t.shape[3]
    return t, ld


COMPUTE_F32, COMPUTE_BF16, COMPUTE_BF16X3, COMPUTE_BF16_STORE = 0, 1, 2, 3
_COMPUTE = [COMPUTE_F32]      # DpigConvDesc.compute of the fp32-tensor entry points
_STORE_BF16 = [False]         # 'bf16' mode: activations stored as bf16
_WINO = [False]               # 'f32w' mode: 3x3 stride-1 convs through the Winograd F(2x2,3x3) kernel where it pays


def set_compute(dtype):
    """Arithmetic of every subsequent launch (module docstring): 'f32' | 'bf16' (storage) | 'bf16c' (fp32 tensors,
    bf16 matrix pipe) | 'bf16x3' (fp32 tensors, split-bf16 products)."""
    mode = {"f32": "f32", "fp32": "f32", "bf16": "bf16", "bf16c": "bf16c", "bf16x3": "bf16x3", "f32w": "f32w"}[dtype]
    _COMPUTE[0] = {"bf16c": COMPUTE_BF16, "bf16x3": COMPUTE_BF16X3}.get(mode, COMPUTE_F32)
    _STORE_BF16[0] = mode == "bf16"
    _WINO[0] = mode == "f32w"


def get_compute():
    if _WINO[0]:
        return "f32w"
    return "bf16" if _STORE_BF16[0] else {COMPUTE_BF16: "bf16c", COMPUTE_BF16X3: "bf16x3"}.get(_COMPUTE[0], "f32")


def storable_bf16(channels):
    """A tensor is kept in bf16 iff 16-byte accesses fall on channel-vector boundaries."""
    return channels % 8 == 0


def _as_rows(t):
    """(tensor, rows, cols, ld) view of an NHWC / 2-D / 1-D tensor for the row-strided elementwise kernels."""
    if t.dim() == 4:
        t, ld = as_nhwc(t)
        return t, t.shape[0] * t.shape[1] * t.shape[2], t.shape[3], ld
    if t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1) and (t.shape[0] <= 1 or t.stride(0) >= t.shape[1]):
        return t, t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])
    t = t.contiguous()
    n = t.numel()
    cols = t.shape[-1] if t.dim() >= 1 and t.shape[-1] > 0 else 1
    return t, n // max(cols, 1), cols, cols


def to_f32(t, out=None):
    """bf16 -> fp32 copy (exact) through dpig_cvt_bf16_to_f32; fp32 tensors pass through (or are copied into `out`)."""
    if t is None or (t.dtype == F32 and out is None):
        return t
    _require_dev(t)
    if t.dtype == F32:
        out.copy_(t)
        return out
    t, rows, cols, ld = _as_rows(t)
    if out is None:
        out = torch.empty(t.shape, dtype=F32, device=t.device)
        ldo = cols
    else:
        _, _, _, ldo = _as_rows(out)
    if t.numel():
        check(lib().dpig_cvt_bf16_to_f32(ptr(t), ld, ptr(out), ldo, rows, cols, stream_ptr()), "cvt_bf16_to_f32")
    return out


def to_bf16(t, out=None):
    """fp32 -> bf16 copy (round-to-nearest-even) through dpig_cvt_f32_to_bf16; bf16 tensors pass through."""
    if t is None or (t.dtype == BF16 and out is None):
        return t
    _require_dev(t)
    if t.dtype == BF16:
        out.copy_(t)
        return out
    t, rows, cols, ld = _as_rows(t)
    if out is None:
        out = torch.empty(t.shape, dtype=BF16, device=t.device)
        ldo = cols
    else:
        _, _, _, ldo = _as_rows(out)
    if t.numel():
        check(lib().dpig_cvt_f32_to_bf16(ptr(t), ld, ptr(out), ldo, rows, cols, stream_ptr()), "cvt_f32_to_bf16")
    return out


def _like_input(y32, ref):
    """Give an fp32 result the storage type of the activation it derives from."""
    if ref.dtype == BF16 and y32 is not None and y32.dtype == F32 and storable_bf16(y32.shape[-1]):
        return to_bf16(y32)
    return y32


def filter_shadows(w, want_plain=True, want_t=True):
    """bf16 shadows of an fp32 HWIO filter: (plain [R,S,C,K], transposed [R,S,K,C]).  Parameters owned by a
    trainer carry persistent shadows refreshed after every optimizer step (`w._dpig_shadow`, trainer.FlatParams);
    any other tensor gets them made on the spot."""
    sh = getattr(w, "_dpig_shadow", None)
    if sh is not None:
        return sh
    w = w.contiguous()
    R, S, C, K = w.shape
    plain = torch.empty((R, S, C, K), dtype=BF16, device=w.device) if want_plain else None
    trans = torch.empty((R, S, K, C), dtype=BF16, device=w.device) if want_t else None
    check(lib().dpig_filter_shadow_bf16(ptr(w), ptr(plain), ptr(trans), R * S, C, K, stream_ptr()), "filter_shadow")
    return plain, trans


# Every live owner of images DERIVED from fp32 master filters (bf16 shadows, split shadows, Winograd images).  Anything that writes
# parameter values outside the optimizers (tfckpt.restore, tester loading into a live trainer) calls refresh_all_derived(): the
# kernels of 'bf16' / 'bf16x3' / 'f32w' read the images, not the masters.
import weakref
_DERIVED = weakref.WeakSet()


def refresh_all_derived():
    """Re-derive every registered image set from its masters.  Returns how many sets were refreshed."""
    n = 0
    for o in list(_DERIVED):
        if getattr(o, "params", None):
            o.refresh()
            n += 1
    return n


class FilterShadows(object):
    """Persistent bf16 shadows (plain HWIO + per-tap transposed) of every conv filter in `params` that the bf16 kernels
    accept, in ONE allocation; attached to the parameters as `_dpig_shadow`.  `refresh()` re-derives them from the fp32
    masters (call it after anything that changes the weights): one launch for the whole set when the masters are slices
    of one flat buffer (`flat`, trainer.FlatParams), else one small launch per filter."""

    def __init__(self, params, flat=None, split=False):
        """split=True: the two-term (hi, lo) shadows of the 'bf16x3' mode, attached as `_dpig_shadow_x3` =
        (plain_hi, trans_hi, plain_lo, trans_lo); buffer layout [plain hi | plain lo | trans hi | trans lo]."""
        self.split = bool(split)
        self.params = [p for p in params if p.dim() == 4 and _bf16_conv_ok(p.shape[2], p.shape[3])]
        total = sum((p.numel() + 7) // 8 * 8 for p in self.params)
        self.numel = total
        self.flat = None
        if not self.params:
            return
        dev = self.params[0].device
        planes = 2 if self.split else 1
        self.buf = torch.empty(2 * planes * total, dtype=BF16, device=dev)
        self.trans_off = planes * total                  # element offset of the transposed layout; lo planes `total` behind hi
        off, rows, tiles = 0, [], 0
        for p in self.params:
            R, S, C, K = p.shape
            n = p.numel()
            t0 = self.trans_off
            if self.split:
                p._dpig_shadow_x3 = (self.buf[off:off + n].view(R, S, C, K), self.buf[t0 + off:t0 + off + n].view(R, S, K, C),
                                     self.buf[total + off:total + off + n].view(R, S, C, K),
                                     self.buf[t0 + total + off:t0 + total + off + n].view(R, S, K, C))
            else:
                p._dpig_shadow = (self.buf[off:off + n].view(R, S, C, K), self.buf[t0 + off:t0 + off + n].view(R, S, K, C))
            if flat is not None:
                src = (p.data_ptr() - flat.data_ptr()) // 4
                if not (0 <= src and src + n <= flat.numel() and p.is_contiguous()):
                    flat = None
                else:
                    rows.append([src, off, R * S, C, K, tiles])
                    tiles += R * S * ((C + 31) // 32) * ((K + 31) // 32)
            off += (n + 7) // 8 * 8
        if flat is not None:
            self.flat, self.ntiles = flat, tiles
            self.table = torch.tensor(rows, dtype=torch.int64).to(dev)
        _DERIVED.add(self)
        self.refresh()

    def refresh(self):
        if not self.params:
            return
        if self.flat is not None:
            if self.split:
                check(lib().dpig_filter_shadow_split_multi(ptr(self.flat), ptr(self.buf), ptr(self.buf) + 2 * self.trans_off,
                                                           self.numel, ptr(self.table), len(self.params), self.ntiles,
                                                           stream_ptr()), "filter_shadow_split_multi")
            else:
                check(lib().dpig_filter_shadow_bf16_multi(ptr(self.flat), ptr(self.buf), ptr(self.buf) + 2 * self.trans_off,
                                                          ptr(self.table), len(self.params), self.ntiles, stream_ptr()),
                      "filter_shadow_multi")
            return
        for p in self.params:
            R, S, C, K = p.shape
            if self.split:
                plain, trans = p._dpig_shadow_x3[:2]
                check(lib().dpig_filter_shadow_split(ptr(p.data), ptr(plain), ptr(trans), self.numel, R * S, C, K, stream_ptr()),
                      "filter_shadow_split")
            else:
                plain, trans = p._dpig_shadow
                check(lib().dpig_filter_shadow_bf16(ptr(p.data), ptr(plain), ptr(trans), R * S, C, K, stream_ptr()),
                      "filter_shadow")

    def detach(self):
        for p in self.params:
            for a in ("_dpig_shadow", "_dpig_shadow_x3"):
                if hasattr(p, a):
                    delattr(p, a)


def wino_images(w, want_fwd=True, want_dgrad=True):
    """(u_fwd, u_dgrad) transformed images of an HWIO [3,3,C,K] filter for dpig_conv2d_fwd_wino / _dgrad_wino
    (dpig_wino_filter_transform), or None when the filter has no Winograd form.  Parameters owned by a trainer carry persistent
    images refreshed after every optimizer step (`w._dpig_wino`, WinoFilters); any other tensor gets them made on the spot."""
    use = getattr(w, "_dpig_wino_use", None)
    if use is not None:
        use.add(2)
    im = getattr(w, "_dpig_wino", None)
    if im is not None:
        return im
    if w.dim() != 4 or w.shape[0] != 3 or w.shape[1] != 3 or w.dtype != F32 or not w.is_cuda:
        return None
    C, K = int(w.shape[2]), int(w.shape[3])
    n = lib().dpig_wino_filter_elems(C, K)
    if n == 0:
        return None
    w = w.contiguous()
    uf = torch.empty(n, dtype=F32, device=w.device) if want_fwd else None
    ud = torch.empty(n, dtype=F32, device=w.device) if want_dgrad else None
    check(lib().dpig_wino_filter_transform(ptr(w), C, K, ptr(uf), ptr(ud), stream_ptr()), "wino_filter_transform")
    return uf, ud


def wino4_images(w, want_fwd=True, want_dgrad=True):
    """(u_fwd, u_dgrad) F(4x4, 3x3) images of an HWIO [3,3,C,K] filter for dpig_conv2d_fwd_wino4 / _dgrad_wino4, or None when the
    filter has no such form.  A parameter owned by a WinoFilters set gets persistent images on its first request (the set refreshes
    them after every optimizer step from then on); any other tensor gets them made on the spot."""
    im = getattr(w, "_dpig_wino4", None)
    if im is not None:
        return im
    if w.dim() != 4 or w.shape[0] != 3 or w.shape[1] != 3 or w.dtype != F32 or not w.is_cuda:
        return None
    C, K = int(w.shape[2]), int(w.shape[3])
    n = lib().dpig_wino4_filter_elems(C, K)
    if n == 0:
        return None
    owner = getattr(w, "_dpig_wino_owner", None)
    owner = owner() if owner is not None else None
    if owner is not None:
        return owner.want4(w)
    w = w.contiguous()
    uf = torch.empty(n, dtype=F32, device=w.device) if want_fwd else None
    ud = torch.empty(n, dtype=F32, device=w.device) if want_dgrad else None
    check(lib().dpig_wino4_filter_transform(ptr(w), C, K, ptr(uf), ptr(ud), stream_ptr()), "wino4_filter_transform")
    return uf, ud


class WinoFilterJob(ctypes.Structure):
    """DpigWinoFilterJob (include/dpig_hip.h)."""
    _fields_ = [("w", ctypes.c_void_p), ("u_fwd", ctypes.c_void_p), ("u_dgrad", ctypes.c_void_p), ("C", ctypes.c_int32), ("K", ctypes.c_int32),
                ("first_block", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class WinoFilters(object):
    """Persistent Winograd images (forward + dgrad) of every 3x3 filter in `params` that has them, attached to the parameters;
    `refresh()` re-derives them from the fp32 masters (after every optimizer step, like FilterShadows).  The F(2x2, 3x3) images
    (`_dpig_wino`) exist from the start, in one allocation; a filter's F(4x4, 3x3) images (`_dpig_wino4`, 2.25x the size) are made
    when a layer first asks for them (`want4`: the library's cost model decides per layer shape, and only the conv call knows the
    shape).  After the set's first optimizer step -- every layer has run forward and backward once -- filters that only ever asked
    for the F(4x4) form leave the F(2x2) refresh (`prune`; a later F(2x2) request for one of them is served on the spot)."""

    def __init__(self, params):
        self.params = [p for p in params if p.dim() == 4 and tuple(p.shape[:2]) == (3, 3) and
                       lib().dpig_wino_filter_elems(int(p.shape[2]), int(p.shape[3])) > 0]
        if not self.params:
            return
        total = sum(2 * lib().dpig_wino_filter_elems(int(p.shape[2]), int(p.shape[3])) for p in self.params)
        self.buf = torch.empty(total, dtype=F32, device=self.params[0].device)
        off = 0
        me = weakref.ref(self)
        for p in self.params:
            n = lib().dpig_wino_filter_elems(int(p.shape[2]), int(p.shape[3]))
            p._dpig_wino = (self.buf[off:off + n], self.buf[off + n:off + 2 * n])
            p._dpig_wino_owner = me
            p._dpig_wino_use = set()
            if hasattr(p, "_dpig_wino4"):
                delattr(p, "_dpig_wino4")
            off += 2 * n
        self.p2, self.p4 = list(self.params), []
        self.refreshes = 0
        self.pruned = False
        self.keep = []                  # superseded job tables (a captured graph may still name them)
        self.plan2 = self._plan(self.p2, "_dpig_wino")
        self.plan4 = None
        _DERIVED.add(self)
        self.refresh()

    def _plan(self, plist, attr):
        """One launch per refresh for a whole list (dpig_wino[4]_filter_transform_jobs): the job table lives in device memory and names
        the masters where they are NOW -- a parameter whose storage moves afterwards (p.data = ...) falls back to per-filter launches."""
        if not plist:
            return None
        jobs = (WinoFilterJob * len(plist))()
        for j, p in zip(jobs, plist):
            im = getattr(p, attr)
            j.w, j.u_fwd, j.u_dgrad = p.data.data_ptr(), im[0].data_ptr(), im[1].data_ptr()
            j.C, j.K = int(p.shape[2]), int(p.shape[3])
        total = lib().dpig_wino_filter_jobs_plan(ctypes.byref(jobs), len(jobs))
        if total <= 0:
            raise RuntimeError("dpig_wino_filter_jobs_plan refused the filter set")
        dev_jobs = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(plist[0].device)
        return {"jobs": dev_jobs, "n": len(plist), "total": total, "masters": [p.data.data_ptr() for p in plist]}

    def want4(self, p):
        """Persistent F(4x4, 3x3) images for parameter `p` of this set (made and filled now, refreshed with the set from here on)."""
        C, K = int(p.shape[2]), int(p.shape[3])
        n = lib().dpig_wino4_filter_elems(C, K)
        buf = torch.empty(2 * n, dtype=F32, device=p.device)
        p._dpig_wino4 = (buf[:n], buf[n:])
        p._dpig_wino_use.add(4)
        check(lib().dpig_wino4_filter_transform(ptr(p.data), C, K, ptr(p._dpig_wino4[0]), ptr(p._dpig_wino4[1]), stream_ptr()),
              "wino4_filter_transform")
        self.p4.append(p)
        if self.plan4 is not None:
            self.keep.append(self.plan4)
        self.plan4 = self._plan(self.p4, "_dpig_wino4")
        return p._dpig_wino4

    def prune(self):
        """Filters whose layers only ever asked for the F(4x4) images leave the F(2x2) refresh."""
        self.pruned = True
        drop = [p for p in self.p2 if 4 in p._dpig_wino_use and 2 not in p._dpig_wino_use]
        if not drop:
            return 0
        for p in drop:
            delattr(p, "_dpig_wino")
        self.p2 = [p for p in self.p2 if hasattr(p, "_dpig_wino")]
        self.keep.append(self.plan2)
        self.plan2 = self._plan(self.p2, "_dpig_wino")
        return len(drop)

    def _run(self, plan, plist, attr, four):
        if plan is None:
            return
        jobs_fn = lib().dpig_wino4_filter_transform_jobs if four else lib().dpig_wino_filter_transform_jobs
        one_fn = lib().dpig_wino4_filter_transform if four else lib().dpig_wino_filter_transform
        if [p.data.data_ptr() for p in plist] == plan["masters"]:
            check(jobs_fn(ptr(plan["jobs"]), plan["n"], plan["total"], stream_ptr()), "wino_filter_transform_jobs")
            return
        for p in plist:
            uf, ud = getattr(p, attr)
            check(one_fn(ptr(p.data), int(p.shape[2]), int(p.shape[3]), ptr(uf), ptr(ud), stream_ptr()), "wino_filter_transform")

    def refresh(self):
        if not getattr(self, "params", None):
            return
        self.refreshes += 1
        # (the first refresh after the set's layers have run: a refresh that comes before any of them -- a checkpoint restore -- decides nothing)
        if self.refreshes >= 2 and not self.pruned and any(p._dpig_wino_use for p in self.params) \
                and not torch.cuda.is_current_stream_capturing():
            self.prune()
        self._run(self.plan2, self.p2, "_dpig_wino", False)
        self._run(self.plan4, self.p4, "_dpig_wino4", True)

    def detach(self):
        for p in self.params:
            for a in ("_dpig_wino", "_dpig_wino4", "_dpig_wino_owner", "_dpig_wino_use"):
                if hasattr(p, a):
                    delattr(p, a)


def get_wino_mode():
    """dpig_conv_wino_get_mode: the mode in force (DPIG_WINO in the environment unless set_wino_mode changed it)."""
    return int(lib().dpig_conv_wino_get_mode())


def set_wino_mode(mode):
    """dpig_conv_wino_set_mode: 0 never, 1 the library's cost model (default), 2 wherever the layer has a Winograd form (tests).
    Returns the PREVIOUS mode, so that callers restore what was in force (e.g. a process started with the DPIG_WINO=0 kill switch)."""
    prev = get_wino_mode()
    check(lib().dpig_conv_wino_set_mode(int(mode)), "conv_wino_set_mode")
    return prev


def get_wino4_mode():
    """dpig_conv_wino4_get_mode: the F(4x4, 3x3) mode in force (DPIG_WINO4 in the environment unless set_wino4_mode changed it)."""
    return int(lib().dpig_conv_wino4_get_mode())


def set_wino4_mode(mode):
    """dpig_conv_wino4_set_mode: 0 never, 1 the library's cost model (default), 2 wherever the layer has the F(4x4, 3x3) form (tests).
    Returns the PREVIOUS mode."""
    prev = get_wino4_mode()
    check(lib().dpig_conv_wino4_set_mode(int(mode)), "conv_wino4_set_mode")
    return prev


def _bf16_conv_ok(C, K, *lds):
    return C % 8 == 0 and K % 8 == 0 and C >= 32 and K >= 32 and all(ld % 8 == 0 for ld in lds)


def _thin_ok(C, K, ld_wide, R, S, stride, *tensors):
    """Host-side mirror of the vector-ALU kernels' shape conditions (dpig_thin.hip `eligible` / `fewc_kind`)."""
    if not _al16(*tensors):
        return False
    if K == 3:
        return R == 3 and S == 3 and stride == 1 and C % 4 == 0 and 16 <= C <= 256 and ld_wide % 4 == 0
    lp = K // 4
    return C == 3 and K % 4 == 0 and 1 <= lp <= 64 and (lp & (lp - 1)) == 0 and \
        ((R, S, stride) == (3, 3, 1) or (R, S, stride) == (5, 5, 2))


def _al16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def _desc(N, H, W, C, K, R, S, stride, ldx, ldy, ldres=0, ldmask=0, act=ACT_NONE, alpha=0.2, upsample2x=False,
          split_k=0, res_after_act=False, ldy2=0, res_class=False):
    d = DpigConvDesc()
    d.compute = _COMPUTE[0]
    d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride = N, H, W, C, K, R, S, stride
    d.pad_t, d.pad_l = -1, -1
    d.ldx, d.ldy, d.ldres, d.ldmask = ldx, ldy, ldres, ldmask
    d.act, d.alpha, d.upsample2x, d.split_k = act, alpha, int(upsample2x), split_k
    d.res_after_act, d.ldy2, d.res_class = int(res_after_act), ldy2, int(res_class)
    return d


def _ws(d, which, device):
    nbytes = lib().dpig_conv2d_workspace_bytes(ctypes.byref(d), which)
    buf, size = workspace.get(nbytes, device)
    return buf, size


def conv_out_hw(H, W, R, S, stride, upsample2x=False):
    if upsample2x:
        return 2 * H, 2 * W
    return _lib.same_pad(H, R, stride)[0], _lib.same_pad(W, S, stride)[0]


X3_EMIT = [int(os.environ.get('DPIG_X3_EMIT', '1'))]    # 0: never let an epilogue leave its output's split32 image; 1: where the
                                                          # caller says the output feeds another conv (`emit32`); 2: always


def conv2d_fwd(x, w, bias=None, stride=1, act=ACT_NONE, alpha=0.2, residual=None, out=None, upsample2x=False,
               split_k=0, res_after_act=False, out_act=None, res_class=False, emit32=False):
    """y = act(conv_SAME(x, w) + bias + residual) (or act(..) + residual with res_after_act);
    x NHWC, w HWIO.  `out` may be a channel slice.  `out_act` optionally receives the activation
    output before a post-activation residual add."""
    if _STORE_BF16[0] or x.dtype == BF16:
        return _conv2d_fwd_bf16(x, w, bias, stride, act, alpha, residual, out, upsample2x, split_k, res_after_act,
                                out_act, res_class)
    _require_gpu(x)
    x, ldx = as_nhwc(x)
    w = w.contiguous()
    N, H, W, C = x.shape
    R, S, Cw, K = w.shape
    if Cw != C:
        raise RuntimeError("conv2d: filter expects %d input channels, tensor has %d" % (Cw, C))
    Ho, Wo = conv_out_hw(H, W, R, S, stride, upsample2x)
    if out is None:
        out = torch.empty((N, Ho, Wo, K), dtype=torch.float32, device=x.device)
    ldy = nhwc_ld(out)
    if ldy is None or tuple(out.shape) != (N, Ho, Wo, K):
        raise RuntimeError("conv2d: bad output tensor")
    ldres = 0
    if residual is not None:
        if res_class:          # [N, 9, K]: one row per (image, 3x3 border class)
            residual = residual.contiguous()
            if tuple(residual.shape) != (N, 9, K):
                raise RuntimeError("conv2d: class residual must be [N, 9, K]")
            ldres = K
        else:
            residual, ldres = as_nhwc(residual)
    if bias is not None:
        bias = bias.contiguous()
    ldy2 = 0
    if out_act is not None:
        ldy2 = nhwc_ld(out_act)
        if ldy2 is None or tuple(out_act.shape) != (N, Ho, Wo, K):
            raise RuntimeError("conv2d: bad out_act tensor")
    d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, ldres=ldres, act=act, alpha=alpha, upsample2x=upsample2x,
              split_k=split_k, res_after_act=res_after_act, ldy2=ldy2, res_class=res_class)
    if _WINO[0] and R == 3 and S == 3 and stride == 1 and split_k == 0 and lib().dpig_conv2d_wino4_eligible(ctypes.byref(d), 0) \
            and _al16(x, out, residual, out_act, bias):
        im = wino4_images(w, want_dgrad=False)
        if im is not None:
            # executed FLOPs: 36 multiplies per (4x4 tile, ci, co) instead of 144
            wsb, wsn = workspace.get(lib().dpig_conv2d_wino4_workspace_bytes(ctypes.byref(d), 0), x.device)
            with _Timed("conv_fwd_wino4", 2.0 * N * (H // 4) * (W // 4) * 36 * K * C, (N, H, W, C, K, R, stride, 0)):
                check(lib().dpig_conv2d_fwd_wino4(ctypes.byref(d), ptr(x), ptr(im[0]), ptr(bias), ptr(residual), ptr(out), ptr(out_act),
                                                  ptr(wsb), wsn, stream_ptr()), "conv2d_fwd_wino4")
            return out
    if _WINO[0] and R == 3 and S == 3 and stride == 1 and split_k == 0 and lib().dpig_conv2d_wino_eligible(ctypes.byref(d), 0) \
            and _al16(x, out, residual, out_act, bias):
        im = wino_images(w, want_dgrad=False)
        if im is not None:
            # executed FLOPs: 16 multiplies per (2x2 tile, ci, co) instead of 36 -- what the roofline of this launch is priced on
            wsb, wsn = workspace.get(lib().dpig_conv2d_wino_workspace_bytes(ctypes.byref(d), 0), x.device)
            with _Timed("conv_fwd_wino", 2.0 * N * (H // 2) * (W // 2) * 16 * K * C, (N, H, W, C, K, R, stride, 0)):
                check(lib().dpig_conv2d_fwd_wino(ctypes.byref(d), ptr(x), ptr(im[0]), ptr(bias), ptr(residual), ptr(out), ptr(out_act),
                                                 ptr(wsb), wsn, stream_ptr()), "conv2d_fwd_wino")
            return out
    wsb, wsn = _ws(d, 0, x.device)
    mfma = (C % 4 == 0 and K % 4 == 0 and ldx % 4 == 0 and C >= 32 and K >= 32)
    with _Timed("conv_fwd_mfma" if mfma else "conv_fwd_thin", 2.0 * N * H * W // (stride * stride) * K * R * S * C,
                (N, H, W, C, K, R, stride, int(upsample2x))):
        sh = getattr(w, "_dpig_shadow_x3", None) if _COMPUTE[0] == COMPUTE_BF16X3 else None
        if sh is not None:           # the filter's two bf16 terms are kept ready (FilterShadows(split=True))
            x32 = split32(x, R * S * ((K + 127) // 128)) if mfma else None
            y32, wrote = _image_room(N, Ho, Wo, K, x.device) if (mfma and not upsample2x and X3_EMIT[0] > (0 if emit32 else 1)) else (None, None)
            check(lib().dpig_conv2d_fwd_x3(ctypes.byref(d), ptr(x), ptr(x32), ptr(w), ptr(sh[1]), ptr(sh[3]), ptr(bias),
                                           ptr(residual), ptr(out), ptr(out_act), ptr(y32),
                                           ctypes.byref(wrote) if wrote is not None else None, ptr(wsb), wsn, stream_ptr()),
                  "conv2d_fwd_x3")
            # the epilogue left the output's split32 image: the next conv of the chain DMAs it (a caller-supplied `out` must not
            # keep an image of its previous contents otherwise)
            tag_s32(out, y32 if (wrote is not None and wrote.value) else None)
        else:
            check(lib().dpig_conv2d_fwd(ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(out),
                                        ptr(out_act), ptr(wsb), wsn, stream_ptr()), "conv2d_fwd")
    return out


# 'bf16x3' mode: give the conv kernels the ACTIVATIONS' two-term splits as ready images too (dpig_split32 layout), so that
# forward, dgrad and wgrad take BOTH operands by LDS-DMA (PIPE 4, bw3_kernel).  An image costs a pass over its tensor (or
# wider epilogue stores when the producing conv leaves it), and has two consumers (x: forward + wgrad, dy: dgrad + wgrad).
# Measured, same-box A/B: kernels +17 % (fwd / dgrad) and +45 % (wgrad), Market step 452 -> 488 img/s.  DPIG_X3_PLANES=0
# turns it off (every activation is then split in the k-loops' registers).
X3_PLANES = [os.environ.get('DPIG_X3_PLANES', '1') != '0']


def split32(t, reuse=1):
    """The two-term bf16 split of an fp32 NHWC activation in the layout the DMA-fed split k-loop reads
    ([N,H,W,chunks,64]: 32 hi | 32 lo per 32-channel chunk; dpig_split32), made once per tensor and kept on it
    (`_dpig_s32`: the forward input is needed again by wgrad-side consumers, dy by dgrad).  Returns None when it would not pay:
    the image costs a read and a write of the tensor, so a consumer that gathers each element fewer than 4 times
    (`reuse` = taps x column tiles) splits it in its k-loop instead -- unless the image already exists."""
    if not X3_PLANES[0] or t.dtype != torch.float32:
        return None
    s32 = cached_s32(t)
    if s32 is not None:
        return s32
    ld = nhwc_ld(t)
    N, Hh, W, C = t.shape
    if reuse < 4 or ld is None or C % 4 or ld % 4 or t.data_ptr() % 16:
        return None
    nchunk = (C + 31) // 32
    if N * Hh * W * nchunk * 128 >= 2 ** 31 - 1:
        return None
    s32 = torch.empty((N, Hh, W, nchunk, 64), dtype=BF16, device=t.device)
    check(lib().dpig_split32(ptr(t), ld, N * Hh * W, C, ptr(s32), stream_ptr()), "split32")
    tag_s32(t, s32)
    return s32


def tag_s32(t, s32):
    """Attach a split32 image to its tensor together with the tensor's identity at that moment (autograd version counter +
    address): an in-place write (gradient accumulation, `copy_` into a static graph input) bumps the version and the image is
    re-made instead of being DMA-ed stale.  `s32 = None` drops any image the tensor carried."""
    try:
        if s32 is None:
            if hasattr(t, "_dpig_s32"):
                del t._dpig_s32
        else:
            t._dpig_s32 = (s32, t._version, t.data_ptr())
    except Exception:
        pass


def cached_s32(t):
    ent = getattr(t, "_dpig_s32", None)
    if ent is None:
        return None
    s32, ver, addr = ent
    if ver != t._version or addr != t.data_ptr():
        tag_s32(t, None)
        return None
    return s32


def _image_room(N, Hh, W, C, device):
    """Room for a conv output's split32 image (written by the epilogue when the launch allows it) + the flag that says so."""
    if not X3_PLANES[0] or C % 32 or N * Hh * W * (C // 32) * 128 >= 2 ** 31 - 1:
        return None, None
    return torch.empty((N, Hh, W, C // 32, 64), dtype=BF16, device=device), ctypes.c_int(0)


def conv2d_fwd_stats(x, w, bias=None, stride=1, split_k=0):
    """y = conv_SAME(x, w) + bias together with the batch-norm partial statistics of y, left by the conv's epilogue
    (dpig_conv2d_fwd_stats): returns (y, stats) with stats = (float tensor [tiles, 2, K], rows per tile), or (y, None) when
    this problem cannot carry them (thin layers, the upsample fusion, a batch served in several launches; bf16 storage: split-K plans too)
    -- `bn_fwd(y, ..., stats=stats)` accepts both."""
    if _STORE_BF16[0] or x.dtype == BF16:
        _require_dev(x)
        N, H, W, C = x.shape
        R, S, Cw, K = w.shape
        if Cw == C and _bf16_conv_ok(C, K):
            xb, ldx = as_nhwc(to_bf16(x))
            if _bf16_conv_ok(C, K, ldx) and _al16(xb):
                d = _desc(N, H, W, C, K, R, S, stride, ldx, K, split_k=split_k)
                tiles = lib().dpig_conv2d_bf16_bn_stats_tiles(ctypes.byref(d))
                if tiles > 0:
                    Ho, Wo = conv_out_hw(H, W, R, S, stride, False)
                    _, w_t = filter_shadows(w, want_plain=False)
                    out = torch.empty((N, Ho, Wo, K), dtype=BF16, device=x.device)
                    stats = torch.empty((tiles, 2, K), dtype=torch.float32, device=x.device)
                    with _Timed("conv_fwd_bf16", 2.0 * N * H * W // (stride * stride) * K * R * S * C, (N, H, W, C, K, R, stride, 0)):
                        check(lib().dpig_conv2d_fwd_bf16_stats(ctypes.byref(d), ptr(xb), ptr(w_t),
                                                               ptr(bias.contiguous() if bias is not None else None), ptr(out),
                                                               ptr(stats), stream_ptr()), "conv2d_fwd_bf16_stats")
                    return out, (stats, 128)
        return conv2d_fwd(x, w, bias, stride=stride, split_k=split_k), None
    _require_gpu(x)
    x, ldx = as_nhwc(x)
    w = w.contiguous()
    N, H, W, C = x.shape
    R, S, Cw, K = w.shape
    if Cw != C:
        raise RuntimeError("conv2d: filter expects %d input channels, tensor has %d" % (Cw, C))
    Ho, Wo = conv_out_hw(H, W, R, S, stride, False)
    d = _desc(N, H, W, C, K, R, S, stride, ldx, K, split_k=split_k)
    tiles = lib().dpig_conv2d_bn_stats_tiles_ws(ctypes.byref(d))          # (split-K plans included: their reduction pass leaves them)
    if bias is not None:
        bias = bias.contiguous()
    # the stats epilogue stores 16-byte vectors: an operand at an odd offset (a bias that is a view, say) takes the plain conv and
    # bn_fwd's own statistics passes instead of an EINVAL from the launch
    if tiles <= 0 or not _al16(x, w, bias):
        return conv2d_fwd(x, w, bias, stride=stride, split_k=split_k), None
    # a split plan's statistics come from its reduction pass, one workgroup per output tile summing every split in turn: below ~128
    # tiles that pass is latency-bound on a handful of CUs and loses to the all-CU plain reduction followed by bn_fwd's own
    # statistics passes (16 x 8 C256->512 5x5/2 at batch 16: 111 us against 66 us, scripts/bench_critic_convs.py) -- policy here,
    # the entry point itself still takes any plan (a forced split_k keeps the fused path)
    if split_k == 0 and tiles * ((K + 127) // 128) < 128 and lib().dpig_conv2d_bn_stats_tiles(ctypes.byref(d)) <= 0:
        return conv2d_fwd(x, w, bias, stride=stride), None
    out = torch.empty((N, Ho, Wo, K), dtype=torch.float32, device=x.device)
    stats = torch.empty((tiles, 2, K), dtype=torch.float32, device=x.device)
    wsb, wsn = _ws(d, 0, x.device)
    with _Timed("conv_fwd_mfma", 2.0 * N * H * W // (stride * stride) * K * R * S * C, (N, H, W, C, K, R, stride, 0)):
        check(lib().dpig_conv2d_fwd_stats_ws(ctypes.byref(d), ptr(x), ptr(w), ptr(bias), ptr(out), ptr(stats), ptr(wsb), wsn,
                                             stream_ptr()), "conv2d_fwd_stats")
    return out, (stats, 128)


def conv2d_dgrad(dy, w, in_shape, stride=1, accum=None, mask=None, act=ACT_NONE, alpha=0.2, out=None,
                 upsample2x=False, split_k=0, emit32=False):
    """dx = (conv_backward_data(dy, w) + accum) * act'(mask);  in_shape = (N,H,W,C) of the fwd input."""
    if _STORE_BF16[0] or dy.dtype == BF16:
        return _conv2d_dgrad_bf16(dy, w, in_shape, stride, accum, mask, act, alpha, out, upsample2x, split_k)
    _require_gpu(dy)
    dy, ldy = as_nhwc(dy)
    w = w.contiguous()
    N, H, W, C = in_shape
    R, S, Cw, K = w.shape
    if Cw != C or dy.shape[3] != K:
        raise RuntimeError("conv2d_dgrad: channel mismatch")
    Ho, Wo = conv_out_hw(H, W, R, S, stride, upsample2x)
    if tuple(dy.shape) != (N, Ho, Wo, K):
        raise RuntimeError("conv2d_dgrad: dy shape %s, expected %s" % (tuple(dy.shape), (N, Ho, Wo, K)))
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.float32, device=dy.device)
    ldx = nhwc_ld(out)
    if ldx is None or tuple(out.shape) != (N, H, W, C):
        raise RuntimeError("conv2d_dgrad: bad output tensor")
    ldres = ldmask = 0
    if accum is not None:
        accum, ldres = as_nhwc(accum)
    if mask is not None:
        mask, ldmask = as_nhwc(mask)
    d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, ldres=ldres, ldmask=ldmask, act=act, alpha=alpha,
              upsample2x=upsample2x, split_k=split_k)
    if _WINO[0] and R == 3 and S == 3 and stride == 1 and split_k == 0 and lib().dpig_conv2d_wino4_eligible(ctypes.byref(d), 1) \
            and _al16(dy, out, accum, mask):
        im = wino4_images(w, want_fwd=False)
        if im is not None:
            wsb, wsn = workspace.get(lib().dpig_conv2d_wino4_workspace_bytes(ctypes.byref(d), 1), dy.device)
            with _Timed("conv_dgrad_wino4", 2.0 * N * (H // 4) * (W // 4) * 36 * K * C, (N, H, W, C, K, R, stride, 0)):
                check(lib().dpig_conv2d_dgrad_wino4(ctypes.byref(d), ptr(dy), ptr(im[1]), ptr(accum), ptr(mask), ptr(out), ptr(wsb), wsn,
                                                    stream_ptr()), "conv2d_dgrad_wino4")
            return out
    if _WINO[0] and R == 3 and S == 3 and stride == 1 and split_k == 0 and lib().dpig_conv2d_wino_eligible(ctypes.byref(d), 1) \
            and _al16(dy, out, accum, mask):
        im = wino_images(w, want_fwd=False)
        if im is not None:
            wsb, wsn = workspace.get(lib().dpig_conv2d_wino_workspace_bytes(ctypes.byref(d), 1), dy.device)
            with _Timed("conv_dgrad_wino", 2.0 * N * (H // 2) * (W // 2) * 16 * K * C, (N, H, W, C, K, R, stride, 0)):
                check(lib().dpig_conv2d_dgrad_wino(ctypes.byref(d), ptr(dy), ptr(im[1]), ptr(accum), ptr(mask), ptr(out), ptr(wsb), wsn,
                                                   stream_ptr()), "conv2d_dgrad_wino")
            return out
    wsb, wsn = _ws(d, 1, dy.device)
    mfma = (C % 4 == 0 and K % 4 == 0 and ldy % 4 == 0 and C >= 32 and K >= 32)
    with _Timed("conv_dgrad_mfma" if mfma else "conv_dgrad_thin",
                2.0 * N * H * W // (stride * stride) * K * R * S * C, (N, H, W, C, K, R, stride, int(upsample2x))):
        sh = getattr(w, "_dpig_shadow_x3", None) if _COMPUTE[0] == COMPUTE_BF16X3 else None
        if sh is not None:
            taps_hit = 4 if upsample2x else (R * S if stride == 1 else max(1, R * S // (stride * stride)))
            dy32 = split32(dy, taps_hit * ((C + 127) // 128)) if mfma else None
            dx32, wrote = _image_room(N, H, W, C, dy.device) if (mfma and stride == 1 and X3_EMIT[0] > (0 if emit32 else 1)) else (None, None)
            check(lib().dpig_conv2d_dgrad_x3(ctypes.byref(d), ptr(dy), ptr(dy32), ptr(w), ptr(sh[0]), ptr(sh[2]), ptr(accum),
                                             ptr(mask), ptr(out), ptr(dx32), ctypes.byref(wrote) if wrote is not None else None,
                                             ptr(wsb), wsn, stream_ptr()), "conv2d_dgrad_x3")
            tag_s32(out, dx32 if (wrote is not None and wrote.value) else None)
        else:
            check(lib().dpig_conv2d_dgrad(ctypes.byref(d), ptr(dy), ptr(w), ptr(accum), ptr(mask), ptr(out), ptr(wsb),
                                          wsn, stream_ptr()), "conv2d_dgrad")
    return out


def conv2d_wgrad(x, dy, wshape, stride=1, upsample2x=False, out=None, beta=0.0, split_k=0, db=None, db_beta=0.0):
    """dw[R,S,C,K] = beta*dw + conv_backward_filter(x, dy); with `db` ([K] tensor) the same launch also writes
    the bias gradient db = db_beta*db + sum over pixels of dy."""
    if x.dtype == BF16 or dy.dtype == BF16:
        return _conv2d_wgrad_bf16(x, dy, wshape, stride, upsample2x, out, beta, split_k, db, db_beta)
    _require_gpu(x)
    x, ldx = as_nhwc(x)
    dy, ldy = as_nhwc(dy)
    N, H, W, C = x.shape
    R, S, Cw, K = wshape
    if Cw != C or dy.shape[3] != K:
        raise RuntimeError("conv2d_wgrad: channel mismatch")
    if out is None:
        out = torch.empty(tuple(wshape), dtype=torch.float32, device=x.device)
        beta = 0.0
    d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, upsample2x=upsample2x, split_k=split_k)
    if _WINO[0] and R == 3 and S == 3 and stride == 1 and split_k == 0 and out.is_contiguous() and _al16(x, dy, out) \
            and lib().dpig_conv2d_wgrad_wino_eligible(ctypes.byref(d)):
        nbytes = lib().dpig_conv2d_wgrad_wino_workspace_bytes(ctypes.byref(d))
        wsb, wsn = workspace.get(nbytes, x.device)
        with _Timed("conv_wgrad_wino", 2.0 * N * (H // 2) * (W // 2) * 16 * K * C, (N, H, W, C, K, R, stride, 0)):
            check(lib().dpig_conv2d_wgrad_wino(ctypes.byref(d), ptr(x), ptr(dy), ptr(out), float(beta), ptr(db), float(db_beta),
                                               ptr(wsb), wsn, stream_ptr()), "conv2d_wgrad_wino")
        return out
    wsb, wsn = _ws(d, 2, x.device)
    mfma = (C % 4 == 0 and K % 4 == 0 and C >= 32 and K >= 32)
    with _Timed("conv_wgrad_mfma" if mfma else "conv_wgrad_thin",
                2.0 * N * H * W // (stride * stride) * K * R * S * C, (N, H, W, C, K, R, stride, int(upsample2x))):
        x32 = dy32 = None
        if mfma and _COMPUTE[0] == COMPUTE_BF16X3 and X3_PLANES[0] and K > 32:
            # the images the forward / dgrad launches of this layer left on the tensors (made now if they did not)
            x32 = split32(x, R * S * ((K + 127) // 128))
            dy32 = split32(dy, R * S * ((C + 127) // 128)) if x32 is not None else None
        if x32 is not None and dy32 is not None:
            check(lib().dpig_conv2d_wgrad_x3(ctypes.byref(d), ptr(x), ptr(x32), ptr(dy), ptr(dy32), ptr(out), float(beta), ptr(db),
                                             float(db_beta), ptr(wsb), wsn, stream_ptr()), "conv2d_wgrad_x3")
        else:
            check(lib().dpig_conv2d_wgrad(ctypes.byref(d), ptr(x), ptr(dy), ptr(out), float(beta), ptr(db),
                                          float(db_beta), ptr(wsb), wsn, stream_ptr()), "conv2d_wgrad")
    return out


# ---- bf16-storage convolutions ('bf16' mode) ---------------------------------------------------------------------
def set_large_tile(mode=1, variant=0):
    """Tile family of the bf16-storage forward / dgrad convs (dpig_conv_bf16_set_large_tile): mode 0 = 128 x 128 kernels
    only, 1 = automatic, 2 = the 8-wave large-tile kernels wherever legal; variant 0 = automatic, 1 = 256 x 256, 2 = 512 x 128."""
    check(lib().dpig_conv_bf16_set_large_tile(int(mode), int(variant)), "conv_bf16_set_large_tile")


def set_wave8(mode):
    """Eight-wave forms of the 128 x 128 bf16 kernels (dpig_conv_bf16_set_wave8): bit 0 forward / dgrad, bit 1 filter gradient; identical results."""
    check(lib().dpig_conv_bf16_set_wave8(int(mode)), "conv_bf16_set_wave8")


def set_large_tile_wgrad(mode=1, variant=0):
    """The same switch for the bf16-storage filter gradient (dpig_conv_bf16_set_large_tile_wgrad)."""
    check(lib().dpig_conv_bf16_set_large_tile_wgrad(int(mode), int(variant)), "conv_bf16_set_large_tile_wgrad")


def _f32_mode():
    """Context: run the fp32-tensor entry points with exact fp32 products (the thin-layer fall-back of 'bf16' mode)."""
    class _Ctx(object):
        def __enter__(self):
            self.saved = (_COMPUTE[0], _STORE_BF16[0])
            _COMPUTE[0], _STORE_BF16[0] = COMPUTE_F32, False

        def __exit__(self, *exc):
            _COMPUTE[0], _STORE_BF16[0] = self.saved
            return False
    return _Ctx()


def _ws_bf16(d, which, device):
    nbytes = lib().dpig_conv2d_bf16_workspace_bytes(ctypes.byref(d), which)
    return workspace.get(nbytes, device)


def _conv2d_fwd_bf16(x, w, bias, stride, act, alpha, residual, out, upsample2x, split_k, res_after_act, out_act,
                     res_class):
    _require_dev(x)
    N, H, W, C = x.shape
    R, S, Cw, K = w.shape
    if Cw != C:
        raise RuntimeError("conv2d: filter expects %d input channels, tensor has %d" % (Cw, C))
    Ho, Wo = conv_out_hw(H, W, R, S, stride, upsample2x)
    if _bf16_conv_ok(C, K):
        x = to_bf16(x)
        x, ldx = as_nhwc(x)
        if out is None:
            out = torch.empty((N, Ho, Wo, K), dtype=BF16, device=x.device)
        ldy = nhwc_ld(out)
        if ldy is None or tuple(out.shape) != (N, Ho, Wo, K) or out.dtype != BF16:
            raise RuntimeError("conv2d: bad output tensor")
        ldres, res_b, res_c = 0, None, None
        if residual is not None:
            if res_class:
                res_c = to_f32(residual).contiguous()
                if tuple(res_c.shape) != (N, 9, K):
                    raise RuntimeError("conv2d: class residual must be [N, 9, K]")
                ldres = K
            else:
                res_b, ldres = as_nhwc(to_bf16(residual))
        ldy2 = 0
        if out_act is not None:
            ldy2 = nhwc_ld(out_act)
            if ldy2 is None or tuple(out_act.shape) != (N, Ho, Wo, K) or out_act.dtype != BF16:
                raise RuntimeError("conv2d: bad out_act tensor")
        if _bf16_conv_ok(C, K, ldx, ldy, ldres if res_b is not None else 8, ldy2 if out_act is not None else 8) and \
                _al16(x, out, res_b, out_act):
            _, w_t = filter_shadows(w, want_plain=False)
            if bias is not None:
                bias = bias.contiguous()
            d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, ldres=ldres, act=act, alpha=alpha, upsample2x=upsample2x,
                      split_k=split_k, res_after_act=res_after_act, ldy2=ldy2, res_class=res_class)
            wsb, wsn = _ws_bf16(d, 0, x.device)
            with _Timed("conv_fwd_bf16", 2.0 * N * H * W // (stride * stride) * K * R * S * C,
                        (N, H, W, C, K, R, stride, int(upsample2x))):
                check(lib().dpig_conv2d_fwd_bf16(ctypes.byref(d), ptr(x), ptr(w_t), ptr(bias), ptr(res_b), ptr(res_c),
                                                 ptr(out), ptr(out_act), ptr(wsb), wsn, stream_ptr()), "conv2d_fwd_bf16")
            return out
    # thin layers: 3 output channels (x bf16 -> fp32 image) / 3 input channels (fp32 image -> bf16): vector-ALU kernels
    # that read / write the wide tensor as bf16 directly
    if out is None and out_act is None and residual is None and not upsample2x and (K == 3 or C == 3):
        xt = to_bf16(x) if K == 3 else to_f32(x)
        xt, ldx = as_nhwc(xt)
        y = torch.empty((N, Ho, Wo, K), dtype=F32 if K == 3 else BF16, device=x.device)
        d = _desc(N, H, W, C, K, R, S, stride, ldx, K, act=act, alpha=alpha)
        wc = w.contiguous()
        rc = lib().dpig_conv2d_fwd_thin_bf16(ctypes.byref(d), ptr(xt), ptr(wc), ptr(bias.contiguous() if bias is not None else None),
                                             ptr(y), stream_ptr()) if _thin_ok(C, K, ldx, R, S, stride, xt, y, wc) else -22
        if rc == 0:
            return y
    # anything else (18-channel pose conv, ...): fp32 kernels between conversions
    with _f32_mode():
        o32 = torch.empty((N, Ho, Wo, K), dtype=F32, device=x.device)
        a32 = torch.empty((N, Ho, Wo, K), dtype=F32, device=x.device) if out_act is not None else None
        conv2d_fwd(to_f32(x), w, bias, stride=stride, act=act, alpha=alpha, residual=to_f32(residual), out=o32,
                   upsample2x=upsample2x, split_k=split_k, res_after_act=res_after_act, out_act=a32, res_class=res_class)
    if out_act is not None:
        to_bf16(a32, out=out_act) if out_act.dtype == BF16 else out_act.copy_(a32)
    if out is not None:
        to_bf16(o32, out=out) if out.dtype == BF16 else out.copy_(o32)
        return out
    return to_bf16(o32) if storable_bf16(K) else o32


def _conv2d_dgrad_bf16(dy, w, in_shape, stride, accum, mask, act, alpha, out, upsample2x, split_k):
    _require_dev(dy)
    N, H, W, C = in_shape
    R, S, Cw, K = w.shape
    if Cw != C or dy.shape[3] != K:
        raise RuntimeError("conv2d_dgrad: channel mismatch")
    Ho, Wo = conv_out_hw(H, W, R, S, stride, upsample2x)
    if tuple(dy.shape) != (N, Ho, Wo, K):
        raise RuntimeError("conv2d_dgrad: dy shape %s, expected %s" % (tuple(dy.shape), (N, Ho, Wo, K)))
    if _bf16_conv_ok(C, K):
        dy, ldy = as_nhwc(to_bf16(dy))
        if out is None:
            out = torch.empty((N, H, W, C), dtype=BF16, device=dy.device)
        ldx = nhwc_ld(out)
        if ldx is None or tuple(out.shape) != (N, H, W, C) or out.dtype != BF16:
            raise RuntimeError("conv2d_dgrad: bad output tensor")
        ldres = ldmask = 0
        if accum is not None:
            accum, ldres = as_nhwc(to_bf16(accum))
        if mask is not None:
            mask, ldmask = as_nhwc(to_bf16(mask))
        if _bf16_conv_ok(C, K, ldx, ldy, ldres or 8, ldmask or 8) and _al16(dy, out, accum, mask):
            w_p, _ = filter_shadows(w, want_t=False)
            d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, ldres=ldres, ldmask=ldmask, act=act, alpha=alpha,
                      upsample2x=upsample2x, split_k=split_k)
            wsb, wsn = _ws_bf16(d, 1, dy.device)
            with _Timed("conv_dgrad_bf16", 2.0 * N * H * W // (stride * stride) * K * R * S * C,
                        (N, H, W, C, K, R, stride, int(upsample2x))):
                check(lib().dpig_conv2d_dgrad_bf16(ctypes.byref(d), ptr(dy), ptr(w_p), ptr(accum), ptr(mask), ptr(out),
                                                   ptr(wsb), wsn, stream_ptr()), "conv2d_dgrad_bf16")
            return out
    if out is None and accum is None and mask is None and not upsample2x and (K == 3 or (C == 3 and R == 5 and stride == 2)):
        dyt = to_f32(dy) if K == 3 else to_bf16(dy)
        dyt, ldy = as_nhwc(dyt)
        dx = torch.empty((N, H, W, C), dtype=BF16 if K == 3 else F32, device=dy.device)
        d = _desc(N, H, W, C, K, R, S, stride, C, ldy)
        wc = w.contiguous()
        rc = lib().dpig_conv2d_dgrad_thin_bf16(ctypes.byref(d), ptr(dyt), ptr(wc), ptr(dx), stream_ptr()) \
            if _thin_ok(C, K, C, R, S, stride, dyt, dx, wc) and (K != 3 or ldy == 3) else -22
        if rc == 0:
            return dx
    with _f32_mode():
        o32 = torch.empty((N, H, W, C), dtype=F32, device=dy.device)
        conv2d_dgrad(to_f32(dy), w, in_shape, stride=stride, accum=to_f32(accum), mask=to_f32(mask), act=act,
                     alpha=alpha, out=o32, upsample2x=upsample2x, split_k=split_k)
    if out is not None:
        to_bf16(o32, out=out) if out.dtype == BF16 else out.copy_(o32)
        return out
    return to_bf16(o32) if storable_bf16(C) else o32


def _conv2d_wgrad_bf16(x, dy, wshape, stride, upsample2x, out, beta, split_k, db, db_beta):
    _require_dev(x)
    N, H, W, C = x.shape
    R, S, Cw, K = wshape
    if Cw != C or dy.shape[3] != K:
        raise RuntimeError("conv2d_wgrad: channel mismatch")
    if out is None:
        out = torch.empty(tuple(wshape), dtype=F32, device=x.device)
        beta = 0.0
    if _bf16_conv_ok(C, K):
        xb, ldx = as_nhwc(to_bf16(x))
        dyb, ldy = as_nhwc(to_bf16(dy))
        if _bf16_conv_ok(C, K, ldx, ldy) and _al16(xb, dyb, out) and out.is_contiguous():
            d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy, upsample2x=upsample2x, split_k=split_k)
            wsb, wsn = _ws_bf16(d, 2, x.device)
            with _Timed("conv_wgrad_bf16", 2.0 * N * H * W // (stride * stride) * K * R * S * C,
                        (N, H, W, C, K, R, stride, int(upsample2x))):
                check(lib().dpig_conv2d_wgrad_bf16(ctypes.byref(d), ptr(xb), ptr(dyb), ptr(out), float(beta), ptr(db),
                                                   float(db_beta), ptr(wsb), wsn, stream_ptr()), "conv2d_wgrad_bf16")
            return out
    if not upsample2x and (K == 3 or C == 3) and out.is_contiguous():
        xt = to_bf16(x) if K == 3 else to_f32(x)
        dyt = to_f32(dy) if K == 3 else to_bf16(dy)
        xt, ldx = as_nhwc(xt)
        dyt, ldy = as_nhwc(dyt)
        d = _desc(N, H, W, C, K, R, S, stride, ldx, ldy)
        if _thin_ok(C, K, ldx, R, S, stride, xt, dyt, out) and (K != 3 or ldy == 3):
            wsb, wsn = _ws(d, 2, x.device)
            rc = lib().dpig_conv2d_wgrad_thin_bf16(ctypes.byref(d), ptr(xt), ptr(dyt), ptr(out), float(beta), ptr(db),
                                                   float(db_beta), ptr(wsb), wsn, stream_ptr())
            if rc == 0:
                return out
    with _f32_mode():
        return conv2d_wgrad(to_f32(x), to_f32(dy), wshape, stride=stride, upsample2x=upsample2x, out=out, beta=beta,
                            split_k=split_k, db=db, db_beta=db_beta)


def _rows_ld(t):
    """View an NHWC (or 2-D) tensor as [rows, cols] with row stride ld."""
    if t.dim() == 2:
        if t.stride(1) != 1 and t.shape[1] > 1:
            t = t.contiguous()
        return t, t.shape[0], t.shape[1], (t.stride(0) if t.shape[0] > 1 else t.shape[1])
    t, ld = as_nhwc(t)
    return t, t.shape[0] * t.shape[1] * t.shape[2], t.shape[3], ld


def act_fwd(x, act, alpha=0.2):
    """y = act(x) elementwise."""
    if x.dtype == BF16:
        _require_dev(x)
        x, rows, cols, ldx = _rows_ld(x)
        if cols % 8 == 0 and ldx % 8 == 0 and _al16(x):
            y = torch.empty(x.shape, dtype=BF16, device=x.device)
            check(lib().dpig_act_fwd_bf16(ptr(x), ldx, ptr(y), cols, rows, cols, act, alpha, stream_ptr()), "act_fwd_bf16")
            return y
        return _like_input(act_fwd(to_f32(x), act, alpha), x)
    _require_gpu(x)
    x, rows, cols, ldx = _rows_ld(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(lib().dpig_act_fwd(ptr(x), ldx, ptr(y), cols, rows, cols, act, alpha, stream_ptr()), "act_fwd")
    return y


def pose_stem_fwd(rcv, normalized, e9, cpos, bias, w, E, H, W, act=ACT_RELU, alpha=0.2, bf16_out=False):
    """y [B,H,W,K] = act(e9[b][class] - cpos[class] + bias + sparse disc terms): the generator's first conv fed with keypoints
    (dpig_pose_stem_fwd).  rcv [B,P,3] fp32, e9 [B,9,K], cpos [9,K], w [3,3,E+P,K] fp32 (the whole filter)."""
    _require_gpu(e9)
    B, P = rcv.shape[0], rcv.shape[1]
    K = e9.shape[-1]
    rcv = rcv.contiguous().float()
    y = torch.empty((B, H, W, K), dtype=BF16 if bf16_out else F32, device=e9.device)
    check(lib().dpig_pose_stem_fwd(ptr(rcv), B, P, int(bool(normalized)), ptr(e9.contiguous()), ptr(cpos.contiguous()),
                                   ptr(bias.contiguous() if bias is not None else None), ptr(w.contiguous()), w.shape[2], E, H, W, K, act,
                                   float(alpha), ptr(y), int(bf16_out), stream_ptr()), "pose_stem_fwd")
    return y


def pose_stem_wgrad(rcv, normalized, z9, dz, P):
    """dwp [9,P,K]: gradient of the pose rows of the stem filter (dpig_pose_stem_wgrad); z9 [B,9,K] fp32, dz [B,H,W,K] fp32 / bf16."""
    _require_dev(dz)
    B, H, W, K = dz.shape
    dz = dz.contiguous()
    rcv = rcv.contiguous().float()
    dwp = torch.empty((9, P, K), dtype=F32, device=dz.device)
    check(lib().dpig_pose_stem_wgrad(ptr(rcv), B, P, int(bool(normalized)), ptr(z9.contiguous()), ptr(dz), H, W, K, ptr(dwp),
                                     int(dz.dtype == BF16), stream_ptr()), "pose_stem_wgrad")
    return dwp


def act_bwd_pool2x(dy, y, act, alpha=0.2):
    """sum over 2x2 blocks of dy * act'(y): [N,2H,2W,C] -> [N,H,W,C], the gradient of act(conv1x1(upsample2x(.))) on the
    low-resolution grid the conv runs on (dpig_act_bwd_pool2x).  fp32 or bf16 pairs."""
    _require_dev(dy)
    N, H2, W2, C = dy.shape
    if H2 % 2 or W2 % 2:
        raise RuntimeError("act_bwd_pool2x: odd spatial size")
    lddy = nhwc_ld(dy)
    ldy = nhwc_ld(y) if (y is not None and act != ACT_NONE) else C
    same = y is None or act == ACT_NONE or y.dtype == dy.dtype
    if lddy is None or ldy is None or not same:
        dz = act_bwd(dy, y, act, alpha) if act != ACT_NONE else dy
        return upsample2x_bwd(dz.contiguous())
    dz = torch.empty((N, H2 // 2, W2 // 2, C), dtype=dy.dtype, device=dy.device)
    check(lib().dpig_act_bwd_pool2x(ptr(dy), lddy, ptr(y) if act != ACT_NONE else None, ldy, ptr(dz), N, H2 // 2, W2 // 2, C, act,
                                    float(alpha), int(dy.dtype == BF16), stream_ptr()), "act_bwd_pool2x")
    return dz


def act_bwd(dy, y, act, alpha=0.2, emit32=False):
    """dz = dy * act'(y) with y the activation output.  `emit32`: dz feeds conv kernels next ('bf16x3' mode: also leave its
    split32 image)."""
    if dy.dtype == BF16 or y.dtype == BF16:
        _require_dev(dy)
        if dy.dtype == BF16 and y.dtype == BF16:
            dy, rows, cols, lddy = _rows_ld(dy)
            y, _, _, ldy = _rows_ld(y)
            if cols % 8 == 0 and lddy % 8 == 0 and ldy % 8 == 0 and _al16(dy, y):
                dz = torch.empty(dy.shape, dtype=BF16, device=dy.device)
                check(lib().dpig_act_bwd_bf16(ptr(dy), lddy, ptr(y), ldy, ptr(dz), cols, rows, cols, act, alpha,
                                              stream_ptr()), "act_bwd_bf16")
                return dz
        return _like_input(act_bwd(to_f32(dy), to_f32(y), act, alpha), y)
    _require_gpu(dy)
    dy, rows, cols, lddy = _rows_ld(dy)
    y, _, _, ldy = _rows_ld(y)
    dz = torch.empty(dy.shape, dtype=torch.float32, device=dy.device)
    if (emit32 and _COMPUTE[0] == COMPUTE_BF16X3 and X3_PLANES[0] and X3_EMIT[0] and dz.dim() == 4 and cols % 32 == 0 and
            lddy % 4 == 0 and ldy % 4 == 0 and _al16(dy, y) and rows * cols * 4 < 2 ** 31 - 1):
        # dz feeds split-bf16 convs next (wgrad and dgrad of the layer): leave its split32 image in the same pass
        dz32 = torch.empty(tuple(dz.shape[:3]) + (cols // 32, 64), dtype=BF16, device=dz.device)
        check(lib().dpig_act_bwd_s32(ptr(dy), lddy, ptr(y), ldy, ptr(dz), cols, rows, cols, act, alpha, ptr(dz32), stream_ptr()),
              "act_bwd_s32")
        tag_s32(dz, dz32)
        return dz
    check(lib().dpig_act_bwd(ptr(dy), lddy, ptr(y), ldy, ptr(dz), cols, rows, cols, act, alpha, stream_ptr()),
          "act_bwd")
    return dz


def colsum(a, out=None, beta=0.0):
    """Column sums of an NHWC / 2-D tensor viewed as [rows, C] (bias gradient)."""
    a = to_f32(a)
    _require_gpu(a)
    a, rows, cols, lda = _rows_ld(a)
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=a.device)
        beta = 0.0
    nbytes = lib().dpig_colsum_workspace_bytes(rows, cols)
    wsb, wsn = workspace.get(nbytes, a.device)
    check(lib().dpig_colsum(ptr(a), lda, rows, cols, ptr(out), float(beta), ptr(wsb), wsn, stream_ptr()), "colsum")
    return out


def border_class_sum(a):
    """[N,H,W,C] -> [N,9,C]: per-image sums over the 9 border classes of a SAME 3x3 conv (fp32 result; the input may
    be a bf16 tensor)."""
    _require_dev(a)
    a, lda = as_nhwc(a)
    N, Hh, W, C = a.shape
    out = torch.empty((N, 9, C), dtype=torch.float32, device=a.device)
    wsb, wsn = workspace.get(lib().dpig_border_class_sum_workspace_bytes(N, Hh, W, C), a.device)
    fn = lib().dpig_border_class_sum_bf16 if a.dtype == BF16 else lib().dpig_border_class_sum
    check(fn(ptr(a), lda, N, Hh, W, C, ptr(out), ptr(wsb), wsn, stream_ptr()), "border_class_sum")
    return out


def pad_channels_bf16(x, cols_out):
    """fp32 [N,H,W,C] -> bf16 [N,H,W,cols_out] with zero channels appended (cols_out a multiple of 8)."""
    _require_gpu(x)
    x, ld = as_nhwc(x)
    N, Hh, W, C = x.shape
    out = torch.empty((N, Hh, W, cols_out), dtype=BF16, device=x.device)
    check(lib().dpig_cvt_f32_to_bf16_pad(ptr(x), ld, C, ptr(out), cols_out, cols_out, N * Hh * W, stream_ptr()), "cvt_pad")
    return out


BN_BF16_NATIVE = [os.environ.get('DPIG_BN_BF16_NATIVE', '1') != '0']     # A/B switch: 0 = fp32 batch-norm kernels between conversion passes


def bn_fwd(x, scale, offset, eps=1e-5, act=ACT_NONE, alpha=0.2, stats=None):
    """Training-mode batch norm over all but the last axis (+ fused activation).  `stats` = what `conv2d_fwd_stats` returned
    for x: the statistics passes are replaced by one merge of the producing conv's per-tile partials."""
    if x.dtype == BF16:
        _require_dev(x)
        xb, rows, C, ldx = _rows_ld(x)
        if BN_BF16_NATIVE[0] and C % 8 == 0 and ldx % 8 == 0 and _al16(xb):       # bf16 tensors read / written directly (dpig_bn_*_bf16): no conversion passes
            y = torch.empty(xb.shape, dtype=BF16, device=x.device)
            mean = torch.empty(C, dtype=torch.float32, device=x.device)
            rstd = torch.empty(C, dtype=torch.float32, device=x.device)
            sc, of = scale.contiguous(), offset.contiguous()
            if stats is not None:
                st, rpt = stats
                if st.shape[2] != C or st.shape[0] != (rows + rpt - 1) // rpt:
                    raise RuntimeError("bn_fwd: the statistics do not belong to this tensor")
                check(lib().dpig_bn_stats_finalize(ptr(st), st.shape[0], rows, rpt, C, eps, ptr(mean), ptr(rstd), stream_ptr()),
                      "bn_stats_finalize")
                check(lib().dpig_bn_apply_bf16(ptr(xb), ldx, rows, C, ptr(sc), ptr(of), ptr(mean), ptr(rstd), act, alpha, ptr(y), C,
                                               stream_ptr()), "bn_apply_bf16")
                return y, mean, rstd
            wsb, wsn = workspace.get(lib().dpig_bn_bf16_workspace_bytes(rows, C), x.device)
            check(lib().dpig_bn_fwd_bf16(ptr(xb), ldx, rows, C, ptr(sc), ptr(of), eps, act, alpha, ptr(y), C, ptr(mean), ptr(rstd),
                                         ptr(wsb), wsn, stream_ptr()), "bn_fwd_bf16")
            return y, mean, rstd
        y, mean, rstd = bn_fwd(to_f32(x), scale, offset, eps, act, alpha, stats=stats)
        return to_bf16(y), mean, rstd
    _require_gpu(x)
    x, rows, C, ldx = _rows_ld(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    mean = torch.empty(C, dtype=torch.float32, device=x.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x.device)
    if stats is not None:
        st, rpt = stats
        if st.shape[2] != C or st.shape[0] != (rows + rpt - 1) // rpt:
            raise RuntimeError("bn_fwd: the statistics do not belong to this tensor")
        check(lib().dpig_bn_stats_finalize(ptr(st), st.shape[0], rows, rpt, C, eps, ptr(mean), ptr(rstd), stream_ptr()),
              "bn_stats_finalize")
        check(lib().dpig_bn_apply(ptr(x), ldx, rows, C, ptr(scale.contiguous()), ptr(offset.contiguous()), ptr(mean), ptr(rstd),
                                  act, alpha, ptr(y), C, stream_ptr()), "bn_apply")
        return y, mean, rstd
    wsb, wsn = workspace.get(lib().dpig_bn_workspace_bytes(rows, C), x.device)
    check(lib().dpig_bn_fwd(ptr(x), ldx, rows, C, ptr(scale.contiguous()), ptr(offset.contiguous()), eps, act, alpha,
                            ptr(y), C, ptr(mean), ptr(rstd), ptr(wsb), wsn, stream_ptr()), "bn_fwd")
    return y, mean, rstd


def _param_out(buf, C, device):
    """The [C] fp32 output of a parameter-gradient kernel: the caller's buffer (a slice of a flat gradient buffer) or a new tensor."""
    if buf is not None and buf.dtype == F32 and buf.numel() == C and buf.is_contiguous() and buf.is_cuda:
        return buf.view(C)
    return torch.empty(C, dtype=torch.float32, device=device)


def bn_bwd(dy, x, y, scale, mean, rstd, act=ACT_NONE, alpha=0.2, dscale_out=None, doffset_out=None):
    if BF16 in (dy.dtype, x.dtype) or (y is not None and y.dtype == BF16):
        if BN_BF16_NATIVE[0] and x.dtype == BF16 and x.shape[-1] % 8 == 0:
            dyb, rows, C, lddy = _rows_ld(to_bf16(dy))
            xb, _, _, ldx = _rows_ld(x)
            yb, ldy = None, C
            if y is not None:
                yb, _, _, ldy = _rows_ld(to_bf16(y))
            if lddy % 8 == 0 and ldx % 8 == 0 and ldy % 8 == 0 and _al16(dyb, xb, yb):
                dx = torch.empty(xb.shape, dtype=BF16, device=x.device)
                dscale, doffset = _param_out(dscale_out, C, x.device), _param_out(doffset_out, C, x.device)
                wsb, wsn = workspace.get(lib().dpig_bn_bf16_workspace_bytes(rows, C), x.device)
                check(lib().dpig_bn_bwd_bf16(ptr(dyb), lddy, ptr(xb), ldx, ptr(yb), ldy, rows, C, ptr(scale.contiguous()), ptr(mean), ptr(rstd),
                                             act, alpha, ptr(dx), C, ptr(dscale), ptr(doffset), ptr(wsb), wsn, stream_ptr()), "bn_bwd_bf16")
                return dx, dscale, doffset
        dx, dscale, doffset = bn_bwd(to_f32(dy), to_f32(x), to_f32(y), scale, mean, rstd, act, alpha, dscale_out, doffset_out)
        return _like_input(dx, x), dscale, doffset
    _require_gpu(dy)
    dy, rows, C, lddy = _rows_ld(dy)
    x, _, _, ldx = _rows_ld(x)
    ldy = C
    if y is not None:
        y, _, _, ldy = _rows_ld(y)
    dx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    dscale, doffset = _param_out(dscale_out, C, x.device), _param_out(doffset_out, C, x.device)
    wsb, wsn = workspace.get(lib().dpig_bn_workspace_bytes(rows, C), x.device)
    check(lib().dpig_bn_bwd(ptr(dy), lddy, ptr(x), ldx, ptr(y), ldy, rows, C, ptr(scale.contiguous()), ptr(mean),
                            ptr(rstd), act, alpha, ptr(dx), C, ptr(dscale), ptr(doffset), ptr(wsb), wsn,
                            stream_ptr()), "bn_bwd")
    return dx, dscale, doffset


def bn_sync_fwd(x, scale, offset, eps, act, alpha, allreduce, world):
    """Batch norm with statistics over ALL data-parallel ranks (equal per-rank batches): `allreduce(t)`
    sum-reduces a small device tensor in place.  Same two-pass arithmetic (mean, then centred second moment)
    as the fused single-rank op.  Returns (y, mean, rstd)."""
    if x.dtype == BF16:
        y, mean, rstd = bn_sync_fwd(to_f32(x), scale, offset, eps, act, alpha, allreduce, world)
        return to_bf16(y), mean, rstd
    _require_gpu(x)
    x, rows, C, ldx = _rows_ld(x)
    dev = x.device
    n_total = float(rows) * world
    s1 = colsum(x)
    allreduce(s1)
    mean = s1.mul_(1.0 / n_total)
    sq = torch.empty(C, dtype=torch.float32, device=dev)
    wsb, wsn = workspace.get(lib().dpig_bn_workspace_bytes(rows, C), dev)
    check(lib().dpig_bn_sqdev(ptr(x), ldx, rows, C, ptr(mean), ptr(sq), ptr(wsb), wsn, stream_ptr()), "bn_sqdev")
    allreduce(sq)
    rstd = torch.rsqrt(sq.mul_(1.0 / n_total).add_(eps))
    y = torch.empty(x.shape, dtype=torch.float32, device=dev)
    check(lib().dpig_bn_apply(ptr(x), ldx, rows, C, ptr(scale.contiguous()), ptr(offset.contiguous()), ptr(mean),
                              ptr(rstd), act, alpha, ptr(y), C, stream_ptr()), "bn_apply")
    return y, mean, rstd


def bn_sync_bwd(dy, x, y, scale, mean, rstd, act, alpha, allreduce, world):
    """Returns dx and the GLOBAL (dscale, doffset) sums over all ranks."""
    if BF16 in (dy.dtype, x.dtype) or (y is not None and y.dtype == BF16):
        dx, a, b = bn_sync_bwd(to_f32(dy), to_f32(x), to_f32(y), scale, mean, rstd, act, alpha, allreduce, world)
        return _like_input(dx, x), a, b
    _require_gpu(dy)
    dy, rows, C, lddy = _rows_ld(dy)
    x, _, _, ldx = _rows_ld(x)
    ldy = C
    if y is not None:
        y, _, _, ldy = _rows_ld(y)
    dev = x.device
    sums = torch.empty(2 * C, dtype=torch.float32, device=dev)        # [dscale | doffset]
    wsb, wsn = workspace.get(lib().dpig_bn_workspace_bytes(rows, C), dev)
    check(lib().dpig_bn_bwd_sums(ptr(dy), lddy, ptr(x), ldx, ptr(y), ldy, rows, C, ptr(mean), ptr(rstd), act, alpha,
                                 ptr(sums), ptr(sums) + 4 * C, ptr(wsb), wsn, stream_ptr()), "bn_bwd_sums")
    allreduce(sums)
    dx = torch.empty(x.shape, dtype=torch.float32, device=dev)
    check(lib().dpig_bn_bwd_apply(ptr(dy), lddy, ptr(x), ldx, ptr(y), ldy, rows, C, ptr(scale.contiguous()), ptr(mean),
                                  ptr(rstd), ptr(sums), ptr(sums) + 4 * C, act, alpha, 1.0 / (float(rows) * world),
                                  ptr(dx), C, stream_ptr()), "bn_bwd_apply")
    return dx, sums[:C], sums[C:]


def _ln_same_type(*ts):
    """LayerNorm operands as one storage type: all bf16 when the first one is (and 16-byte channel vectors exist), else all fp32."""
    ts = [t for t in ts]
    bf = ts[0].dtype == BF16
    return bf, [None if t is None else ((to_bf16(t) if bf else to_f32(t)).contiguous()) for t in ts]


def ln_fwd(x, scale, offset, eps=1e-5, act=ACT_NONE, alpha=0.2):
    """Layer norm over (H,W,C) per sample; x NHWC dense, fp32 or bf16 (read / written directly: dpig_ln_fwd_bf16)."""
    _require_dev(x)
    x = x.contiguous()
    N, C = x.shape[0], x.shape[-1]
    P = x.numel() // (N * C)
    y = torch.empty_like(x)
    mean = torch.empty(N, dtype=torch.float32, device=x.device)
    rstd = torch.empty(N, dtype=torch.float32, device=x.device)
    wsb, wsn = workspace.get(lib().dpig_ln_fwd_workspace_bytes(N, P, C), x.device)
    fn = lib().dpig_ln_fwd_bf16 if x.dtype == BF16 else lib().dpig_ln_fwd
    check(fn(ptr(x), N, P, C, ptr(scale.contiguous()), ptr(offset.contiguous()), eps, act, alpha, ptr(y), ptr(mean), ptr(rstd),
             ptr(wsb), wsn, stream_ptr()), "ln_fwd")
    return y, mean, rstd


def ln_bwd(dy, x, y, scale, mean, rstd, act=ACT_NONE, alpha=0.2, want_params=True, dscale_out=None, doffset_out=None):
    """(dx, dscale, doffset); the parameter gradients are skipped (None) with want_params=False."""
    _require_dev(dy)
    bf, (x, dy, y) = _ln_same_type(x, dy, y)
    N, C = x.shape[0], x.shape[-1]
    P = x.numel() // (N * C)
    dx = torch.empty_like(x)
    dscale = _param_out(dscale_out, C, x.device) if want_params else None
    doffset = _param_out(doffset_out, C, x.device) if want_params else None
    wsb, wsn = workspace.get(lib().dpig_ln_workspace_bytes(N, P, C), x.device)
    fn = lib().dpig_ln_bwd_bf16 if bf else lib().dpig_ln_bwd
    check(fn(ptr(dy), ptr(x), ptr(y), N, P, C, ptr(scale.contiguous()), ptr(mean), ptr(rstd), act, alpha, ptr(dx), ptr(dscale),
             ptr(doffset), ptr(wsb), wsn, stream_ptr()), "ln_bwd")
    return dx, dscale, doffset


def ln_bwd2(u, dy, x, y, scale, mean, rstd, act=ACT_NONE, alpha=0.2):
    """Second-order LayerNorm: gradients of ln_bwd's dx w.r.t. (dy, x, scale) given u = dP/d(dx)."""
    _require_dev(u)
    bf, (x, u, dy, y) = _ln_same_type(x, u, dy, y)
    N, C = x.shape[0], x.shape[-1]
    P = x.numel() // (N * C)
    d_dy = torch.empty_like(x)
    d_x = torch.empty_like(x)
    d_scale = torch.empty(C, dtype=torch.float32, device=x.device)
    wsb, wsn = workspace.get(lib().dpig_ln_bwd2_workspace_bytes(N, P, C), x.device)
    fn = lib().dpig_ln_bwd2_bf16 if bf else lib().dpig_ln_bwd2
    check(fn(ptr(u), ptr(dy), ptr(x), ptr(y), N, P, C, ptr(scale.contiguous()), ptr(mean), ptr(rstd), act, alpha, ptr(d_dy), ptr(d_x),
             ptr(d_scale), ptr(wsb), wsn, stream_ptr()), "ln_bwd2")
    return d_dy, d_x, d_scale


def linear_fwd(x, w, bias=None, act=ACT_NONE, alpha=0.2):
    if x.dtype == BF16:        # FC layers: fp32 kernels on fp32 weights; FC-net tensors stay fp32 (a conv that consumes one converts it)
        return linear_fwd(to_f32(x), w, bias, act, alpha)
    _require_gpu(x)
    x = x.contiguous()
    w = w.contiguous()
    M, Kin = x.shape
    Nout = w.shape[1]
    if w.shape[0] != Kin:
        raise RuntimeError("linear: weight is %s, input has %d features" % (tuple(w.shape), Kin))
    y = torch.empty((M, Nout), dtype=torch.float32, device=x.device)
    wsb, wsn = workspace.get(lib().dpig_linear_workspace_bytes(M, Kin, Nout, 0), x.device)
    check(lib().dpig_linear_fwd(ptr(x), ptr(w), ptr(bias.contiguous() if bias is not None else None), ptr(y), M, Kin,
                                Nout, act, alpha, ptr(wsb), wsn, stream_ptr()), "linear_fwd")
    return y


def linear_dgrad(dy, w):
    if dy.dtype == BF16:
        return _like_input(linear_dgrad(to_f32(dy), w), dy)
    _require_gpu(dy)
    dy = dy.contiguous()
    w = w.contiguous()
    M, Nout = dy.shape
    Kin = w.shape[0]
    dx = torch.empty((M, Kin), dtype=torch.float32, device=dy.device)
    wsb, wsn = workspace.get(lib().dpig_linear_workspace_bytes(M, Kin, Nout, 1), dy.device)
    check(lib().dpig_linear_dgrad(ptr(dy), ptr(w), ptr(dx), M, Kin, Nout, ptr(wsb), wsn, stream_ptr()),
          "linear_dgrad")
    return dx


def linear_wgrad(x, dy, out=None, beta=0.0):
    """dw[Kin,Nout] = beta*dw + x^T @ dy."""
    x, dy = to_f32(x), to_f32(dy)
    _require_gpu(x)
    x = x.contiguous()
    dy = dy.contiguous()
    M, Kin = x.shape
    Nout = dy.shape[1]
    if out is None:
        out = torch.empty((Kin, Nout), dtype=torch.float32, device=x.device)
        beta = 0.0
    wsb, wsn = workspace.get(lib().dpig_linear_workspace_bytes(M, Kin, Nout, 2), x.device)
    check(lib().dpig_linear_wgrad(ptr(x), ptr(dy), ptr(out), float(beta), M, Kin, Nout, ptr(wsb), wsn,
                                  stream_ptr()), "linear_wgrad")
    return out


def crop_resize_fwd(img, boxes, box_ind, ch, cw):
    _require_dev(img)
    img = img.contiguous()
    N, H, W, C = img.shape
    boxes = boxes.contiguous().float()
    box_ind = box_ind.contiguous().to(torch.int32)
    nb = boxes.shape[0]
    out = torch.empty((nb, ch, cw, C), dtype=img.dtype, device=img.device)
    fn = lib().dpig_crop_resize_fwd_bf16 if img.dtype == BF16 else lib().dpig_crop_resize_fwd
    check(fn(ptr(img), N, H, W, C, ptr(boxes), ptr(box_ind), nb, ch, cw, ptr(out), stream_ptr()), "crop_resize_fwd")
    return out


def crop_resize_bwd(dout, boxes, box_ind, img_shape):
    _require_dev(dout)
    dout = dout.contiguous()
    N, H, W, C = img_shape
    boxes = boxes.contiguous().float()
    box_ind = box_ind.contiguous().to(torch.int32)
    nb, ch, cw, _ = dout.shape
    dimg = torch.empty(tuple(img_shape), dtype=dout.dtype, device=dout.device)
    wsb, wsn = workspace.get(lib().dpig_crop_resize_bwd_workspace_bytes(W, C, nb, ch), dout.device)
    fn = lib().dpig_crop_resize_bwd_bf16 if dout.dtype == BF16 else lib().dpig_crop_resize_bwd
    check(fn(ptr(dout), N, H, W, C, ptr(boxes), ptr(box_ind), nb, ch, cw, ptr(dimg), ptr(wsb), wsn, stream_ptr()),
          "crop_resize_bwd")
    return dimg


def pose_points(rcv, H, W, keypoint_num=18, is_normalized=True):
    """coord2channel_simple_rcv (utils.py:237-285) on the device: rcv [B, K*3] -> [B,H,W,K]."""
    _require_gpu(rcv)
    B = rcv.shape[0]
    rcv = rcv.reshape(B, keypoint_num, 3).contiguous().float()
    out = torch.empty((B, H, W, keypoint_num), dtype=torch.float32, device=rcv.device)
    check(lib().dpig_pose_points(ptr(rcv), B, keypoint_num, H, W, int(bool(is_normalized)), ptr(out), keypoint_num,
                                 stream_ptr()), "pose_points")
    return out


def pose_inflate(pose):
    """tf_poseInflate (utils.py:287-318) on an NHWC [-1,1] map."""
    _require_gpu(pose)
    pose = pose.contiguous()
    B, H, W, K = pose.shape
    out = torch.empty_like(pose)
    check(lib().dpig_pose_inflate(ptr(pose), K, B, K, H, W, ptr(out), K, stream_ptr()), "pose_inflate")
    return out


def pose_rasterize(rcv, H, W, keypoint_num=18, is_normalized=True):
    """Both steps in one pass from the keypoint coordinates: each visible keypoint becomes a radius-4 disc."""
    _require_gpu(rcv)
    B = rcv.shape[0]
    rcv = rcv.reshape(B, keypoint_num, 3).contiguous().float()
    out = torch.empty((B, H, W, keypoint_num), dtype=torch.float32, device=rcv.device)
    check(lib().dpig_pose_rasterize(ptr(rcv), B, keypoint_num, H, W, int(bool(is_normalized)), ptr(out), keypoint_num,
                                    stream_ptr()), "pose_rasterize")
    return out


def ssim_gray_u8(a255, b255):
    """Per-image SSIM of two [B,H,W,3] batches of 0..255 pixel values, as skimage.compare_ssim gives it on the gray
    uint8 images (trainer.py:516-521): returns [B]."""
    _require_gpu(a255)
    a255, b255 = a255.contiguous().float(), b255.contiguous().float()
    B, Hh, W, C = a255.shape
    if C != 3 or tuple(b255.shape) != tuple(a255.shape):
        raise RuntimeError("ssim_gray_u8 expects two [B,H,W,3] tensors")
    out = torch.empty(B, dtype=torch.float32, device=a255.device)
    wsb, wsn = workspace.get(lib().dpig_ssim_workspace_bytes(B, Hh, W), a255.device)
    check(lib().dpig_ssim_gray_u8(ptr(a255), ptr(b255), B, Hh, W, ptr(out), ptr(wsb), wsn, stream_ptr()), "ssim_gray_u8")
    return out


def gp_interpolate(real, fake, alpha):
    """xhat = real + alpha[b] * (fake - real), alpha: [B]."""
    real, fake = to_f32(real), to_f32(fake)
    _require_gpu(real)
    real, fake = real.contiguous(), fake.contiguous()
    B = real.shape[0]
    out = torch.empty_like(real)
    check(lib().dpig_gp_interpolate(ptr(real), ptr(fake), ptr(alpha.contiguous().reshape(-1).float()), B, real.numel() // B,
                                    ptr(out), stream_ptr()), "gp_interpolate")
    return out


def gp_penalty(g, lam):
    """(penalty [1], dpenalty/dg, slopes [B]) of lambda * mean_b (||g_b|| - 1)^2 for g [B, ...]."""
    g = to_f32(g)
    _require_gpu(g)
    g = g.contiguous()
    B = g.shape[0]
    pen = torch.empty(1, dtype=torch.float32, device=g.device)
    dg = torch.empty_like(g)
    slopes = torch.empty(B, dtype=torch.float32, device=g.device)
    wsb, wsn = workspace.get(lib().dpig_gp_penalty_workspace_bytes(B, g.numel() // B), g.device)
    check(lib().dpig_gp_penalty(ptr(g), B, g.numel() // B, float(lam), ptr(pen), ptr(dg), ptr(slopes), ptr(wsb), wsn, stream_ptr()),
          "gp_penalty")
    return pen, dg, slopes


CRITIC_KEYS = (["Discriminator.%d.Filters" % i for i in range(1, 5)], ["Discriminator.%d.Biases" % i for i in range(1, 5)],
               ["Discriminator.BN%d.scale" % i for i in range(2, 5)], ["Discriminator.BN%d.offset" % i for i in range(2, 5)],
               "Discriminator.Output.W")


def _critic_struct(tensors, prefix):
    from ._lib import DpigCriticParams
    s = DpigCriticParams()
    ws, bs, sc, of, wo = CRITIC_KEYS
    for i in range(4):
        s.w[i] = ptr(tensors[prefix + ws[i]])
        s.b[i] = ptr(tensors[prefix + bs[i]])
    for i in range(3):
        s.ln_scale[i] = ptr(tensors[prefix + sc[i]])
        s.ln_offset[i] = ptr(tensors[prefix + of[i]])
    s.w_out = ptr(tensors[prefix + wo])
    return s


def gp_double_backward(params, real, fake, alpha, lam=10.0, dim=64, grads=None, beta=0.0, prefix="", compute=None,
                       lrelu_alpha=0.2, ln_eps=1e-5):
    """trainer.py:222-236 for Discriminator = DCGANDiscriminator (wgan_gp.py:407-440) in one library call: the penalty value
    [1], the per-sample slopes [B] and -- when `grads` (dict keyed like `params`, or True to allocate) is given -- d penalty /
    d theta for every critic parameter (grads = beta * grads + ...).  `params`: dict name -> fp32 device tensor holding
    `prefix + 'Discriminator.{1..4}.Filters' / '.Biases'`, `'Discriminator.BN{2..4}.scale' / '.offset'` and
    `'Discriminator.Output.W'`; real / fake: NHWC images [B, H, W, Cin]; alpha: [B].  `compute`: a DPIG_COMPUTE_* value; None = the
    current mode -- in 'bf16' storage mode COMPUTE_BF16_STORE: the critic's activations inside the call are bf16 tensors on the
    bf16-storage kernels (images, parameters and gradients stay fp32)."""
    from ._lib import DpigCriticDesc
    real, fake = to_f32(real).contiguous(), to_f32(fake).contiguous()
    _require_gpu(real)
    B, Hh, W, Cin = real.shape
    keys = [prefix + k for grp in CRITIC_KEYS[:4] for k in grp] + [prefix + CRITIC_KEYS[4]]
    for k in keys:
        t = params[k]
        if not (t.is_cuda and t.dtype == F32 and t.is_contiguous()):
            raise RuntimeError("gp_double_backward: parameter %s must be a dense fp32 device tensor" % k)
    d = DpigCriticDesc(B, Hh, W, Cin, int(dim), float(lrelu_alpha), float(ln_eps), float(lam),
                       (COMPUTE_BF16_STORE if _STORE_BF16[0] else _COMPUTE[0]) if compute is None else int(compute))
    P = _critic_struct(params, prefix)
    if grads is True:
        grads = {k: torch.empty_like(params[k]) for k in keys}
        beta = 0.0
    G = _critic_struct(grads, prefix) if grads is not None else None
    pen = torch.empty(1, dtype=F32, device=real.device)
    slopes = torch.empty(B, dtype=F32, device=real.device)
    nbytes = lib().dpig_gp_double_backward_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        raise RuntimeError("gp_double_backward: %s" % lib().dpig_last_error().decode())
    wsb, wsn = workspace.get(nbytes, real.device)
    check(lib().dpig_gp_double_backward(ctypes.byref(d), ctypes.byref(P), ptr(real), ptr(fake),
                                        ptr(alpha.contiguous().reshape(-1).float()), float(beta),
                                        ctypes.byref(G) if G is not None else None, ptr(pen), ptr(slopes), ptr(wsb), wsn,
                                        stream_ptr()), "gp_double_backward")
    return pen, slopes, grads


GP_SLOTS = ("Z", "A", "DA", "DZ", "V", "UB", "ZB", "T", "F", "ZS")


def gp_double_backward_tensors(shape, dim=64, compute=None, device=None):
    """Copies of the tensors the LAST `gp_double_backward` call of this shape / mode left in the workspace (dpig_gp_double_backward_slot):
    {'xhat', 'gin', 'u0'} fp32 images and {'Z2'..'ZS3'}: level tensors [B, H_l, W_l, C_l], fp32 or bf16 by mode.  Inspection / tests."""
    from ._lib import DpigCriticDesc
    B, Hh, W, Cin = shape
    comp = (COMPUTE_BF16_STORE if _STORE_BF16[0] else _COMPUTE[0]) if compute is None else int(compute)
    d = DpigCriticDesc(B, Hh, W, Cin, int(dim), 0.2, 1e-5, 10.0, comp)
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    nbytes = lib().dpig_gp_double_backward_workspace_bytes(ctypes.byref(d))
    wsb, _ = workspace.get(nbytes, device)
    out = {}
    off, nb = ctypes.c_size_t(), ctypes.c_size_t()
    dt = BF16 if comp == COMPUTE_BF16_STORE else F32
    hs, ws_, cs = [Hh], [W], [Cin]
    for l in range(1, 5):
        hs.append((hs[-1] + 1) // 2); ws_.append((ws_[-1] + 1) // 2); cs.append(dim << (l - 1))
    for i, nm in enumerate(("xhat", "gin", "u0")):
        check(lib().dpig_gp_double_backward_slot(ctypes.byref(d), 0, i, ctypes.byref(off), ctypes.byref(nb)), "gp_slot")
        out[nm] = wsb[off.value:off.value + nb.value].view(F32).reshape(B, Hh, W, Cin).clone()
    for l in range(1, 5):
        for i, nm in enumerate(GP_SLOTS):
            check(lib().dpig_gp_double_backward_slot(ctypes.byref(d), l, i, ctypes.byref(off), ctypes.byref(nb)), "gp_slot")
            out["%s%d" % (nm, l)] = wsb[off.value:off.value + nb.value].view(dt).reshape(B, hs[l], ws_[l], cs[l]).clone()
    return out


def upsample2x_fwd(x):
    if x.dtype == BF16:
        return _like_input(upsample2x_fwd(to_f32(x)), x)
    _require_gpu(x)
    x = x.contiguous()
    N, H, W, C = x.shape
    y = torch.empty((N, 2 * H, 2 * W, C), dtype=torch.float32, device=x.device)
    check(lib().dpig_upsample2x_fwd(ptr(x), N, H, W, C, ptr(y), stream_ptr()), "upsample2x_fwd")
    return y


def upsample2x_bwd(dy):
    if dy.dtype == BF16:
        return _like_input(upsample2x_bwd(to_f32(dy)), dy)
    _require_gpu(dy)
    dy = dy.contiguous()
    N, H2, W2, C = dy.shape
    dx = torch.empty((N, H2 // 2, W2 // 2, C), dtype=torch.float32, device=dy.device)
    check(lib().dpig_upsample2x_bwd(ptr(dy), N, H2 // 2, W2 // 2, C, ptr(dx), stream_ptr()), "upsample2x_bwd")
    return dx


def adam_step(p, g, m, v, lr_dev, beta1, beta2, eps, step, grad_scale=1.0):
    """In-place TF Adam on flat fp32 buffers; lr_dev is a 1-element device tensor."""
    _require_gpu(p)
    check(lib().dpig_adam_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(lr_dev), beta1, beta2, eps, step,
                               grad_scale, stream_ptr()), "adam_step")


def adam_step_dev(p, g, m, v, lr_dev, state_dev, beta1, beta2, eps, grad_scale=1.0):
    """TF Adam with the step counter / bias correction in device memory (hipGraph-replayable)."""
    _require_gpu(p)
    check(lib().dpig_adam_step_dev(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), ptr(lr_dev), ptr(state_dev), beta1,
                                   beta2, eps, grad_scale, stream_ptr()), "adam_step_dev")


def rmsprop_step(p, g, ms, mom, lr_dev, decay=0.9, momentum=0.0, eps=1e-10, grad_scale=1.0):
    """In-place tf.train.RMSPropOptimizer update on flat buffers."""
    _require_gpu(p)
    check(lib().dpig_rmsprop_step(ptr(p), ptr(g), ptr(ms), ptr(mom), p.numel(), ptr(lr_dev), decay, momentum, eps,
                                  grad_scale, stream_ptr()), "rmsprop_step")


def clip_(p, lo, hi):
    """In-place clip_by_value (WGAN critic weight clipping, trainer.py:124-128)."""
    _require_gpu(p)
    check(lib().dpig_clip(ptr(p), p.numel(), float(lo), float(hi), stream_ptr()), "clip")


def sce_mean(logits, label, want_grad=False, scale=1.0):
    logits = to_f32(logits)
    _require_gpu(logits)
    logits = logits.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().dpig_sce_mean(ptr(logits), logits.numel(), float(label), ptr(out), ptr(dl), float(scale),
                              stream_ptr()), "sce_mean")
    return out, dl


def logit_mean(logits, squared=False, target=0.0, want_grad=False, scale=1.0):
    """mean(x) or mean((x - target)^2) of a logit vector (+ its gradient): the wgan / lsgan loss terms."""
    logits = to_f32(logits)
    _require_gpu(logits)
    logits = logits.contiguous()
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_grad else None
    check(lib().dpig_logit_mean(ptr(logits), logits.numel(), int(bool(squared)), float(target), ptr(out), ptr(dl),
                                float(scale), stream_ptr()), "logit_mean")
    return out, dl


def l1_mean(a, b, want_grad=False, scale=1.0):
    a, b = to_f32(a), to_f32(b)
    _require_gpu(a)
    a = a.contiguous()
    b = b.contiguous()
    n = a.numel()
    out = torch.empty(1, dtype=torch.float32, device=a.device)
    da = torch.empty_like(a) if want_grad else None
    wsb, wsn = workspace.get(lib().dpig_l1_workspace_bytes(n), a.device)
    check(lib().dpig_l1_mean(ptr(a), ptr(b), n, ptr(out), ptr(da), float(scale), ptr(wsb), wsn, stream_ptr()),
          "l1_mean")
    return out, da


# ---- graph wiring between the convolutions (csrc/dpig_glue.hip) ---------------------------------------------------------
def mask_split_fwd(x, m):
    """(x * m, x * (1 - m)) with m one value per pixel (models.py:402-403); x NHWC fp32 or bf16."""
    _require_dev(x)
    x, ldx = as_nhwc(x)
    N, Hh, W, C = x.shape
    m = m.reshape(-1).to(F32).contiguous()
    if m.numel() != N * Hh * W:
        raise RuntimeError("mask_split: one mask value per pixel expected")
    fg, bg = torch.empty_like(x, memory_format=torch.contiguous_format), torch.empty_like(x, memory_format=torch.contiguous_format)
    check(lib().dpig_mask_split_fwd(ptr(x), ldx, ptr(m), N * Hh * W, C, ptr(fg), C, ptr(bg), C, int(x.dtype == BF16), stream_ptr()),
          "mask_split_fwd")
    return fg, bg, m


def mask_split_bwd(dfg, dbg, m, like):
    """dx = dfg * m + dbg * (1 - m); either gradient may be None.  `like`: shape / dtype donor."""
    N, Hh, W, C = like.shape
    ldf = ldb = 0
    if dfg is not None:
        dfg, ldf = as_nhwc(dfg if dfg.dtype == like.dtype else dfg.to(like.dtype))
    if dbg is not None:
        dbg, ldb = as_nhwc(dbg if dbg.dtype == like.dtype else dbg.to(like.dtype))
    dx = torch.empty((N, Hh, W, C), dtype=like.dtype, device=like.device)
    check(lib().dpig_mask_split_bwd(ptr(dfg), ldf, ptr(dbg), ldb, ptr(m), N * Hh * W, C, ptr(dx), C, int(like.dtype == BF16),
                                    stream_ptr()), "mask_split_bwd")
    return dx


def roi_boxes(bbox, bbox_num, img_H, img_W):
    """models.py:405-413 -> (boxes [P*B, 4] fp32, box_ind [P*B] int32), part-major."""
    if not bbox.is_cuda:
        raise RuntimeError("dpig HIP ops need device tensors (got %s): there is no CPU fallback" % bbox.device)
    if bbox.dtype not in (torch.int32, F32):
        bbox = bbox.to(F32)
    bbox = bbox.contiguous()
    B, P_total, _ = bbox.shape
    boxes = torch.empty((bbox_num * B, 4), dtype=F32, device=bbox.device)
    ind = torch.empty((bbox_num * B,), dtype=torch.int32, device=bbox.device)
    check(lib().dpig_roi_boxes(ptr(bbox), int(bbox.dtype == F32), B, P_total, bbox_num, float(img_H), float(img_W), ptr(boxes),
                               ptr(ind), stream_ptr()), "roi_boxes")
    return boxes, ind


def vis_concat_fwd(fea, vis, bg, B, P, z):
    fea = to_f32(fea).contiguous()
    vis = vis.to(F32)
    vis = vis if vis.stride(1) == 1 else vis.contiguous()
    zbg = 0 if bg is None else bg.shape[1]
    if bg is not None:
        bg = to_f32(bg).contiguous()
    out = torch.empty((B, P * z + zbg), dtype=F32, device=fea.device)
    check(lib().dpig_vis_concat_fwd(ptr(fea), ptr(vis), vis.stride(0), ptr(bg), B, P, z, zbg, ptr(out), stream_ptr()), "vis_concat_fwd")
    return out, vis


def vis_concat_bwd(dall, vis, B, P, z, zbg, want_bg):
    dall = dall.contiguous()
    dfea = torch.empty((P * B, z), dtype=F32, device=dall.device)
    dbg = torch.empty((B, zbg), dtype=F32, device=dall.device) if (want_bg and zbg) else None
    check(lib().dpig_vis_concat_bwd(ptr(dall), ptr(vis), vis.stride(0), B, P, z, zbg, ptr(dfea), ptr(dbg), stream_ptr()), "vis_concat_bwd")
    return dfea, dbg


def emb_class_weights_fwd(w, E):
    w = w.contiguous()
    _, _, C, K = w.shape
    wmat = torch.empty((E, 9 * K), dtype=F32, device=w.device)
    check(lib().dpig_emb_class_weights_fwd(ptr(w), E, C, K, ptr(wmat), stream_ptr()), "emb_class_weights_fwd")
    return wmat


def emb_class_weights_bwd(dwc, E, out, beta):
    """out [3,3,C,K] (contiguous): out[:, :, :E, :] = beta * out[:, :, :E, :] + transpose-of-the-class-sums(dwc [E, 9K])."""
    _, _, C, K = out.shape
    check(lib().dpig_emb_class_weights_bwd(ptr(dwc.contiguous()), E, C, K, ptr(out), float(beta), stream_ptr()), "emb_class_weights_bwd")


def axpby3d(src, dst, beta):
    """dst = beta * dst + src for 3-D views [outer, rows, cols] whose last axis is contiguous (any outer / row strides)."""
    if src.dim() != 3 or tuple(src.shape) != tuple(dst.shape) or src.stride(2) != 1 or dst.stride(2) != 1 or src.dtype != F32 or dst.dtype != F32:
        raise RuntimeError("axpby3d: [outer, rows, cols] fp32 views with a contiguous last axis expected")
    o, r, c = src.shape
    check(lib().dpig_axpby3d(ptr(src), src.stride(0), src.stride(1), ptr(dst), dst.stride(0), dst.stride(1), o, r, c, float(beta),
                             stream_ptr()), "axpby3d")


def transpose12(x):
    """[B, A, C] -> [B, C, A] (contiguous in, contiguous out; fp32 or bf16)."""
    x = x.contiguous()
    B, Aa, C = x.shape
    y = torch.empty((B, C, Aa), dtype=x.dtype, device=x.device)
    check(lib().dpig_transpose12(ptr(x), ptr(y), B, Aa, C, x.element_size(), stream_ptr()), "transpose12")
    return y
