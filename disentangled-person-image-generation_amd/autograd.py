"""torch.autograd plumbing around the HIP kernels (hip_ops).  No arithmetic happens here: every
forward/backward piece is a kernel of libdpig_hip.so; torch only records the graph and owns the
memory.  The gradient definitions are the ones TF autodiff derives for the reference graph
(trainer.py:137-140).
"""
import torch

from . import hip_ops as H
from ._lib import ACT_LRELU, ACT_NONE, ACT_RELU


# The WGAN-GP penalty needs dD(xhat)/dxhat with create_graph=True (trainer.py:233).  torch hands every custom Function of
# that first-level sweep needs_input_grad = True for its parameters whatever `inputs=` says, so without this switch the
# sweep would also deliver d(sum D(xhat))/dtheta into the critic's gradient slices -- a term d_loss does not contain.
_PARAM_GRADS_OFF = [False]


class no_param_grads(object):
    """Context: backward passes run inside deliver no parameter gradients (input gradients only)."""

    def __enter__(self):
        self.saved = _PARAM_GRADS_OFF[0]
        _PARAM_GRADS_OFF[0] = True

    def __exit__(self, *exc):
        _PARAM_GRADS_OFF[0] = self.saved
        return False


# Stage boundaries of the data-parallel backward (trainer: the gradient exchange of a finished stage runs under the next stage's
# kernels).  A trainer sets CUTS to a dict around its forward pass; the model builders record the tensors at which the backward
# graph can be cut and (MARKS) how many `Conv` / `fully_connected` scopes their variable scope had opened at that point -- which
# variables, by TF-slim name, belong to which stage (models.encoder_stage_of).
CUTS = None
MARKS = {}


# ---- hipGraph capture around collectives --------------------------------------------------------------------------------------------
# A collective cannot ride inside a captured graph here (the gloo / RCCL process groups launch from their own machinery), yet the
# cross-rank batch-norm statistics need three of them in the MIDDLE of the critic's forward and backward passes.  `SegmentedCapture`
# captures a function as a CHAIN of graphs: wherever the function reaches an `eager_island(fn)` -- also from the autograd engine's
# worker thread, inside a backward pass -- the running capture ends, `fn` is executed (and remembered), and a new capture begins in
# the same memory pool.  A replay alternates graph launches and the remembered calls.  Outside a capture `eager_island(fn)` is `fn()`.
_SEG = [None]


def eager_island(fn):
    seg = _SEG[0]
    return fn() if seg is None else seg.island(fn)


class SegmentedCapture(object):
    def __init__(self, device, pool=None, stream=None):
        self.device = torch.device(device)
        self.pool = pool
        self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
        self.items = []              # ("graph", CUDAGraph) | ("call", fn)
        self.cur = None

    def _begin(self):
        self.cur = torch.cuda.CUDAGraph()
        # relaxed: a segment may end on another thread than it began (islands reached from the autograd engine's device thread)
        if self.pool is not None:
            self.cur.capture_begin(pool=self.pool, capture_error_mode="relaxed")
        else:
            self.cur.capture_begin(capture_error_mode="relaxed")

    def _end(self):
        self.cur.capture_end()
        self.pool = self.cur.pool()
        self.items.append(("graph", self.cur))
        self.cur = None

    def island(self, fn):
        self._end()
        out = fn()
        self.items.append(("call", fn))
        self._begin()
        return out

    def capture(self, fn):
        import gc
        gc.collect()
        torch.cuda.synchronize(self.device)
        self.stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.stream):
            self._begin()
            _SEG[0] = self
            try:
                out = fn()
            except BaseException:
                _SEG[0] = None
                if self.cur is not None:
                    try:                                  # a broken capture cannot always be ended: keep the ORIGINAL error
                        self._end()
                    except Exception:
                        self.cur = None
                raise
            _SEG[0] = None
            if self.cur is not None:
                self._end()
        torch.cuda.current_stream(self.device).wait_stream(self.stream)
        return out

    def replay(self):
        for kind, x in self.items:
            if kind == "graph":
                x.replay()
            else:
                x()

    @property
    def segments(self):
        return sum(1 for k, _ in self.items if k == "graph")


class _CutFn(torch.autograd.Function):
    """Identity that gives a cut tensor a backward node of its own: two outputs of ONE node (mask_split's x_fg / x_bg) would otherwise
    share their capture point, and a partial backward towards one of them would still run every branch that reaches the node."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g


# A ReLU conv whose output feeds ONE residual block: the block's input-gradient dgrad applies that ReLU's mask in its epilogue
# ((dgrad + skip gradient) * relu'(x0)) and tags the gradient, and the producing conv's backward then skips its own activation-gradient
# pass.  ReLU masks are idempotent, so a gradient that lost its tag (autograd summed it with another consumer's) is simply masked again:
# correctness never depends on the tag.  FUSE_INPUT_MASK[0] = False restores the separate pass (A/B, tests).
FUSE_INPUT_MASK = [__import__('os').environ.get('DPIG_FUSE_INPUT_MASK', '1') != '0']
# Independent towers of one graph on two HIP streams (models.py: the Fg / ROI tower beside the Bg tower of the two-branch
# encoder).  The deep levels of either tower are launches of a few dozen workgroups; side by side they fill the CUs the other
# leaves idle.  Results do not change (no atomics anywhere; every launch has its own per-stream workspace, _lib._Workspace).
TWO_STREAM = [__import__('os').environ.get('DPIG_TWO_STREAM', '1') != '0']
# d_optim in MODE 'dcgan': the critic's pass over the real images (forward AND backward) on a side stream beside the generator's forward
D_OVERLAP = [__import__('os').environ.get('DPIG_D_OVERLAP', '1') != '0']
_SIDE_STREAMS = {}


class side_branch(object):
    """`with side_branch(x) as sb:` runs the block on this device's side stream, ordered after everything launched so far on the
    current stream; `sb.join(t, ...)` afterwards makes the current stream wait for the block and returns the tensors for use on it.
    Inactive (a plain block on the current stream) when TWO_STREAM is off or x is not on a GPU.  Backward: autograd replays each
    node on the stream it was recorded on and orders gradients that cross streams by events; the block's first node (the consumer of
    a current-stream tensor) runs last in backward, so the current stream's wait for ITS input gradient covers every launch of the
    block's backward, including gradients our kernels write straight into the flat buffer."""

    def __init__(self, x, key="tower", enabled=None):
        self.active = bool(TWO_STREAM[0] if enabled is None else enabled) and x.is_cuda
        if self.active:
            key = (x.device.index, key)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = torch.cuda.Stream(device=x.device)
            self.side = _SIDE_STREAMS[key]
            self.main = torch.cuda.current_stream(x.device)
            self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        if self.active:
            self.side.wait_stream(self.main)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.active:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.active:
            self.main.wait_stream(self.side)
            for t in tensors:
                t.record_stream(self.main)          # allocated on the side stream, read on this one
        return tensors[0] if len(tensors) == 1 else tensors


def join_side_streams(device):
    """Make the current stream wait for every side stream of `device`: after a PARTIAL backward pass (torch.autograd.grad towards a
    tensor produced inside a side_branch) the returned gradients may have been written on a side stream."""
    device = torch.device(device)
    # a bare 'cuda' device has index None while the side streams are keyed by the tensors' (always concrete) device index
    want = device.index if device.index is not None else torch.cuda.current_device()
    cur = torch.cuda.current_stream(device)
    for (idx, _), st in _SIDE_STREAMS.items():
        if idx == want:
            cur.wait_stream(st)


def _mark_relu_output(y, act):
    if act == ACT_RELU:
        try:
            y._dpig_relu_out = True
        except Exception:
            pass
    return y


def _premasked(dy, y):
    """dy already carries relu'(y): it is the tagged input gradient of the residual block that consumed y."""
    tag = getattr(dy, "_dpig_masked_for", None)
    # the tag is (data pointer of y, version of dy when it was tagged): an in-place accumulation of another consumer's gradient into
    # the tagged tensor (autograd's InputBuffer may add in place) bumps the version and voids the tag
    return FUSE_INPUT_MASK[0] and tag is not None and tag == (y.data_ptr(), dy._version) and dy.shape == y.shape


def _sink(p, fn):
    """Gradient delivery for a parameter.  If the optimizer registered a persistent gradient slice
    on the parameter (`p._dpig_grad`, a view into its flat gradient buffer -- see trainer.FlatParams)
    the kernel writes (first touch: beta=0) or accumulates (later touches: beta=1) straight into it
    and autograd gets None: no AccumulateGrad copy, no zero-fill pass, and the flat buffer is what
    RCCL all-reduces.  Otherwise the gradient is returned to autograd as usual."""
    buf = getattr(p, "_dpig_grad", None)
    if buf is None:
        return fn(None, 0.0)
    touched = p._dpig_touched
    fn(buf, 1.0 if touched[0] else 0.0)
    touched[0] = True
    return None


# Filter gradients on their own stream: inside `with wgrad_overlap():` (the trainers' backward passes) every conv wgrad that sinks into
# the flat gradient buffer is launched on one side stream, after the stream that produced dz; nothing in the backward chain waits for
# it, so it runs beside the following layers' dgrads and fills their tails.  All contributions to one parameter stay in launch order
# (one wgrad stream); leaving the block makes the current stream wait for the wgrad stream (finalize / all-reduce / Adam come after).
# DPIG_WGRAD_STREAM: 0 = off, 1 = every wgrad, 2 = only the wgrads of SMALL layers (fewer than WGRAD_SMALL_GF GFLOP: launches that do not
# fill the chip -- the deep levels of the towers / the low-resolution decoder stages -- and therefore gain from running beside the next
# layer's dgrad; full-size kernels side by side lose their XCD-local L2 reuse, measured round 3).
WGRAD_STREAM = [int(__import__('os').environ.get('DPIG_WGRAD_STREAM', '0'))]
WGRAD_SMALL_GF = [float(__import__('os').environ.get('DPIG_WGRAD_SMALL_GF', '40'))]
_WG = {"on": 0, "used": False, "streams": {}}


def _wgrad_on_side_stream(x, dz, w, stride):
    if not _WG["on"] or not x.is_cuda or torch.is_grad_enabled():
        return False
    if any(torch.cuda.current_stream(x.device) == st for st in _SIDE_STREAMS.values()):
        return False            # (a third stream forked from a side stream inside a hipGraph capture crashed hipStreamEndCapture on this ROCm)
    if _WG["on"] == 1:
        return True
    gf = 2.0 * dz.shape[0] * dz.shape[1] * dz.shape[2] * dz.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * 1e-9
    return gf < WGRAD_SMALL_GF[0]


class wgrad_overlap(object):
    def __enter__(self):
        self.prev = _WG["on"]
        _WG["on"] = int(WGRAD_STREAM[0])
        return self

    def __exit__(self, *exc):
        _WG["on"] = self.prev
        if not self.prev and _WG["used"]:
            for dev, st in _WG["streams"].items():
                torch.cuda.current_stream(dev).wait_stream(st)
            _WG["used"] = False
        return False


def _wgrad_stream(x, dz):
    """The wgrad stream, ordered after the current one (dz has just been produced on it); x / dz stay allocated for it."""
    dev = x.device
    st = _WG["streams"].get(dev)
    if st is None:
        st = _WG["streams"][dev] = torch.cuda.Stream(device=dev)
    st.wait_stream(torch.cuda.current_stream(dev))
    x.record_stream(st)
    dz.record_stream(st)
    _WG["used"] = True
    return st


def _sink_wgrad_bias(w, b, x, dz, stride=1, upsample2x=False, want_w=True, want_b=True):
    """Filter gradient (+ bias gradient in the same launch when both go to flat sinks)."""
    dw = db = None
    wbuf = getattr(w, "_dpig_grad", None) if want_w else None
    bbuf = getattr(b, "_dpig_grad", None) if (want_b and b is not None) else None
    if wbuf is not None and bbuf is not None:
        wt, bt = w._dpig_touched, b._dpig_touched
        if _wgrad_on_side_stream(x, dz, w, stride):
            with torch.cuda.stream(_wgrad_stream(x, dz)):
                H.conv2d_wgrad(x, dz, tuple(w.shape), stride=stride, upsample2x=upsample2x, out=wbuf,
                               beta=1.0 if wt[0] else 0.0, db=bbuf, db_beta=1.0 if bt[0] else 0.0)
        else:
            H.conv2d_wgrad(x, dz, tuple(w.shape), stride=stride, upsample2x=upsample2x, out=wbuf,
                           beta=1.0 if wt[0] else 0.0, db=bbuf, db_beta=1.0 if bt[0] else 0.0)
        wt[0] = True
        bt[0] = True
        return None, None
    if want_w:
        dw = _sink(w, lambda o, beta: H.conv2d_wgrad(x, dz, tuple(w.shape), stride=stride, upsample2x=upsample2x,
                                                     out=o, beta=beta))
    if want_b and b is not None:
        db = _sink(b, lambda o, beta: H.colsum(dz, out=o, beta=beta))
    return dw, db


def _first_touch_out(p):
    """The flat-gradient slice of `p` when a kernel may write its gradient there directly (registered and not yet touched this step);
    `_sink_small` then only marks it."""
    buf = getattr(p, "_dpig_grad", None)
    return buf if (buf is not None and not p._dpig_touched[0]) else None


def _sink_small(p, g):
    """Same contract as _sink for tiny per-channel gradients that a kernel already produced."""
    buf = getattr(p, "_dpig_grad", None)
    if buf is None:
        return g
    touched = p._dpig_touched
    if touched[0]:
        buf.add_(g.view_as(buf))
    elif g.data_ptr() != buf.data_ptr():           # (the kernel may have written the slice itself: _first_touch_out)
        buf.copy_(g.view_as(buf))
    touched[0] = True
    return None


# ---- second-level functions: the backward passes themselves, differentiable once more -------------
# Used only when a backward pass runs with create_graph=True (the WGAN-GP penalty is a function of
# dD/dx, trainer.py:222-236).  LeakyReLU/ReLU are piecewise linear, so nothing flows through the masks;
# the conv / linear backward-data ops are linear in (dz, w): their adjoints are the forward op and wgrad
# (SURVEY Appendix E "up-sweep"); LayerNorm needs a genuine second-order kernel (dpig_ln_bwd2).
class _ActBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, y, act, alpha):
        ctx.save_for_backward(y)
        ctx.cfg = (act, alpha)
        return H.act_bwd(dy, y, act, alpha)

    @staticmethod
    def backward(ctx, ddz):
        (y,) = ctx.saved_tensors
        return H.act_bwd(ddz, y, *ctx.cfg), None, None, None


class _ConvDgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dz, w, in_shape, stride, upsample2x):
        ctx.save_for_backward(dz, w)
        ctx.cfg = (stride, upsample2x)
        return H.conv2d_dgrad(dz, w, in_shape, stride=stride, upsample2x=upsample2x)

    @staticmethod
    def backward(ctx, ddx):
        dz, w = ctx.saved_tensors
        stride, up = ctx.cfg
        d_dz = H.conv2d_fwd(ddx, w, None, stride=stride, upsample2x=up) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1]:
            d_w = _sink(w, lambda o, beta: H.conv2d_wgrad(ddx, dz, tuple(w.shape), stride=stride, upsample2x=up,
                                                          out=o, beta=beta))
        return d_dz, d_w, None, None, None


class _CastFn(torch.autograd.Function):
    """fp32 <-> bf16 storage conversion that stays on the tape (its gradient is the opposite conversion), for backward passes that are
    differentiated once more: a raw conversion kernel would cut the second-order graph behind it."""

    @staticmethod
    def forward(ctx, x, to_bf16):
        ctx.to_bf16 = to_bf16
        return H.to_bf16(x) if to_bf16 else H.to_f32(x)

    @staticmethod
    def backward(ctx, d):
        if torch.is_grad_enabled():
            return _CastFn.apply(d, not ctx.to_bf16), None
        return (H.to_f32(d) if ctx.to_bf16 else H.to_bf16(d)), None


class _LinearDgradFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dz, w):
        ctx.save_for_backward(dz, w)
        return H.linear_dgrad(dz, w)

    @staticmethod
    def backward(ctx, ddx):
        dz, w = ctx.saved_tensors
        ddx = ddx.contiguous()
        d_dz = H.linear_fwd(ddx, w) if ctx.needs_input_grad[0] else None
        d_w = None
        if ctx.needs_input_grad[1]:
            d_w = _sink(w, lambda o, beta: H.linear_wgrad(ddx, dz, out=o, beta=beta))
        return d_dz, d_w


class _LNBwdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, y, scale, mean, rstd, act, alpha):
        ctx.save_for_backward(dy, x, y, scale, mean, rstd)
        ctx.cfg = (act, alpha)
        dx, _, _ = H.ln_bwd(dy, x, y, scale, mean, rstd, act, alpha, want_params=False)
        return dx

    @staticmethod
    def backward(ctx, u):
        dy, x, y, scale, mean, rstd = ctx.saved_tensors
        act, alpha = ctx.cfg
        d_dy, d_x, d_scale = H.ln_bwd2(u, dy, x, y, scale, mean, rstd, act, alpha)
        ds = _sink_small(scale, d_scale) if ctx.needs_input_grad[3] else None
        return d_dy, d_x, None, ds, None, None, None, None


class _ConvFn(torch.autograd.Function):
    """y = act(conv_SAME(x, w) + b)   [optionally on a nearest-2x-upsampled x, 1x1 only]."""

    @staticmethod
    def forward(ctx, x, w, b, stride, act, alpha, upsample2x, stats_box=None, out=None):
        if stats_box is not None and act == ACT_NONE and not upsample2x:
            # the caller's next op is a batch norm: let the conv epilogue leave the per-tile statistics (None when it cannot)
            y, st = H.conv2d_fwd_stats(x, w, b, stride=stride)
            stats_box.append(st)
        else:
            y = H.conv2d_fwd(x, w, b, stride=stride, act=act, alpha=alpha, upsample2x=upsample2x,
                             out=out.t if out is not None else None, emit32=True)      # (a conv's output mostly feeds a conv)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (stride, act, alpha, upsample2x, b is not None)
        ctx.b_ref = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        stride, act, alpha, up, has_b = ctx.cfg
        second = torch.is_grad_enabled()       # backward under create_graph=True
        if up and not second and w.shape[0] == 1 and w.shape[1] == 1:
            # act(conv1x1(upsample2x(x))): nearest upsampling commutes with the 1x1 conv, so its gradient lives on x's grid:
            # one pass forms sum_{2x2} dy * act'(y) there, then plain 1x1 dgrad / wgrad (no 4-tap gathers of the 4x larger dy)
            dzp = H.act_bwd_pool2x(dy, y, act, alpha)
            dx = H.conv2d_dgrad(dzp, w, tuple(x.shape), stride=1) if ctx.needs_input_grad[0] else None
            if _PARAM_GRADS_OFF[0]:
                return dx, None, None, None, None, None, None, None, None
            dw, db = _sink_wgrad_bias(w, ctx.b_ref, x, dzp, 1, False, ctx.needs_input_grad[1], has_b and ctx.needs_input_grad[2])
            return dx, dw, db, None, None, None, None, None, None
        if second:
            dz = _ActBwdFn.apply(dy, y, act, alpha) if act != ACT_NONE else dy
        elif act == ACT_RELU and _premasked(dy, y):
            dz = dy                                 # the consuming residual block's dgrad already applied relu'(y)
        else:
            dz = H.act_bwd(dy, y, act, alpha, emit32=True) if act != ACT_NONE else dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if second:
                dx = _ConvDgradFn.apply(dz, w, tuple(x.shape), stride, up)
                dz = dz.detach()
            else:
                dx = H.conv2d_dgrad(dz, w, tuple(x.shape), stride=stride, upsample2x=up)
        if _PARAM_GRADS_OFF[0]:
            return dx, None, None, None, None, None, None, None, None
        dw, db = _sink_wgrad_bias(w, ctx.b_ref, x, dz, stride, up, ctx.needs_input_grad[1],
                                  has_b and ctx.needs_input_grad[2])
        return dx, dw, db, None, None, None, None, None, None


def conv2d(x, w, b=None, stride=1, act=ACT_NONE, alpha=0.2, upsample2x=False, bn_stats=False, out=None):
    """`bn_stats=True` (the next op is a training-mode batch norm over this output): the result carries `_dpig_bnstats`, the
    conv epilogue's per-tile statistics, when the launch plan allows it (hip_ops.conv2d_fwd_stats).
    `out`: a channel slice of a wider NHWC buffer to write the result into (see `join_channels`)."""
    if not bn_stats:
        if out is not None:
            return _ConvFn.apply(x, w, b, stride, act, alpha, upsample2x, None, _Out(out))
        return _mark_relu_output(_ConvFn.apply(x, w, b, stride, act, alpha, upsample2x), act)
    box = []
    y = _ConvFn.apply(x, w, b, stride, act, alpha, upsample2x, box)
    if box and box[0] is not None:
        y._dpig_bnstats = box[0]
    return y


class _ResBlockFn(torch.autograd.Function):
    """The reference residual block (models.py:398-400, 425-427, 458-460, 534-536, 564-566):
         c1 = relu(conv3x3(x0)+b1); c2 = relu(conv3x3(c1)+b2); out = c2 + x0
    forward: 2 launches (the add rides in conv2's epilogue); backward: one act_bwd, then
    dgrad(conv2) applies c1's ReLU mask and dgrad(conv1) adds the skip gradient in its epilogue."""

    @staticmethod
    def forward(ctx, x0, w1, b1, w2, b2, out=None):
        ctx.x0_relu = bool(getattr(x0, "_dpig_relu_out", False))
        c1 = H.conv2d_fwd(x0, w1, b1, act=ACT_RELU, emit32=True)
        c2 = torch.empty_like(c1)
        out = torch.empty_like(c1) if out is None else out.t
        H.conv2d_fwd(c1, w2, b2, act=ACT_RELU, residual=x0, res_after_act=True, out=out, out_act=c2, emit32=True)
        ctx.save_for_backward(x0, w1, w2, c1, c2)
        ctx.b_refs = (b1, b2)
        return out

    @staticmethod
    def backward(ctx, dout):
        x0, w1, w2, c1, c2 = ctx.saved_tensors
        b1, b2 = ctx.b_refs
        dz2 = H.act_bwd(dout, c2, ACT_RELU, emit32=True)
        dw2, db2 = _sink_wgrad_bias(w2, b2, c1, dz2, want_w=ctx.needs_input_grad[3], want_b=ctx.needs_input_grad[4])
        dz1 = H.conv2d_dgrad(dz2, w2, tuple(c1.shape), mask=c1, act=ACT_RELU, emit32=True)    # feeds dgrad(conv1)
        dw1, db1 = _sink_wgrad_bias(w1, b1, x0, dz1, want_w=ctx.needs_input_grad[1], want_b=ctx.needs_input_grad[2])
        dx0 = None
        if ctx.needs_input_grad[0]:
            if ctx.x0_relu and FUSE_INPUT_MASK[0] and not torch.is_grad_enabled():
                # x0 is a ReLU conv's output: hand its producer the gradient w.r.t. the PRE-activation (mask in this epilogue)
                dx0 = H.conv2d_dgrad(dz1, w1, tuple(x0.shape), accum=dout, mask=x0, act=ACT_RELU)
                dx0._dpig_masked_for = (x0.data_ptr(), dx0._version)
            else:
                dx0 = H.conv2d_dgrad(dz1, w1, tuple(x0.shape), accum=dout)
        return dx0, dw1, db1, dw2, db2, None


def resblock(x0, w1, b1, w2, b2, out=None):
    """`out`: a channel slice of a wider NHWC buffer to write the block's output into (see `join_channels`)."""
    if out is not None:
        return _ResBlockFn.apply(x0, w1, b1, w2, b2, _Out(out))
    return _ResBlockFn.apply(x0, w1, b1, w2, b2)


class _Out(object):
    """Carrier of an `out=` tensor through Function.apply: not a Tensor argument, so autograd neither treats it as an input
    nor the returned tensor as "an input returned as is"."""

    def __init__(self, t):
        self.t = t


class _PlaceFn(torch.autograd.Function):
    """Copy x into a channel slice of a wider buffer (identity for autograd)."""

    @staticmethod
    def forward(ctx, x, out):
        out.t.copy_(x)
        return out.t

    @staticmethod
    def backward(ctx, d):
        return d, None


def place(x, out):
    return _PlaceFn.apply(x, _Out(out))


def _alias(storage_of, offset, size, stride):
    """A tensor over the same memory that is NOT an autograd view of anything (own version counter, no base)."""
    return torch.empty(0, dtype=storage_of.dtype, device=storage_of.device).set_(storage_of.untyped_storage(), offset, size, stride)


class _JoinFn(torch.autograd.Function):
    """tf.concat([a, b], axis=3) (models.py:560) of two tensors that ALREADY are adjacent channel slices of one NHWC buffer
    (their producers were given `out=` slices): the result is that buffer -- no copy; the gradient splits into two views."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.c1 = a.shape[3]
        N, Hh, W, C1 = a.shape
        return _alias(a, a.storage_offset(), (N, Hh, W, C1 + b.shape[3]), a.stride())

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.c1], d[..., ctx.c1:]


def channel_slices(N, Hh, W, c1, c2, dtype, device):
    """One NHWC buffer of c1 + c2 channels and its two channel slices (for `out=` of the two producers of a concat)."""
    buf = torch.empty((N, Hh, W, c1 + c2), dtype=dtype, device=device)
    C = c1 + c2
    st = (Hh * W * C, W * C, C, 1)
    return _alias(buf, 0, (N, Hh, W, c1), st), _alias(buf, c1, (N, Hh, W, c2), st)


def join_channels(a, b):
    """concat([a, b], channel axis): free when a and b are adjacent slices of one buffer (`channel_slices`), else a copy."""
    adjacent = (a.dim() == 4 and b.dim() == 4 and a.dtype == b.dtype and a.shape[:3] == b.shape[:3] and
                a.stride() == b.stride() and a.stride(3) == 1 and a.stride(2) == a.shape[3] + b.shape[3] and
                a.stride(1) == a.shape[2] * a.stride(2) and a.stride(0) == a.shape[1] * a.stride(1) and
                a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and
                b.storage_offset() == a.storage_offset() + a.shape[3])
    if adjacent:
        return _JoinFn.apply(a, b)
    return torch.cat([a, b], dim=3)


def _valid_taps(c):
    """Filter rows/cols that see real data for border class c of a SAME 3x3 conv (pad 1)."""
    return (slice(1, 3), slice(0, 3), slice(0, 2))[c]


class _TiledEmbConvFn(torch.autograd.Function):
    """First conv of the U-Net generator (models.py:520-528) on concat(tile(emb), pose):

        y = relu(conv3x3_SAME(concat([emb broadcast over H x W, pose], -1), w) + b)

    The reference materialises the [B,H,W,E] tiled embedding (trainer.py:588-590, 184 MB at B=16) and
    convolves it densely (111.7 GFLOP).  Because those E input channels are spatially constant their
    contribution to an output pixel only depends on which filter taps fall inside the image, i.e. on
    the pixel's 3x3 border class:  contribution = emb[b] @ sum_{valid taps} w[tap,:E,:].
    So: one [B,E]x[E,9K] GEMM gives the per-(image, class) vectors, a thin conv over the pose
    channels adds them in its epilogue -- identical result, ~1/20 of the work.  The backward pass is
    the exact transpose: per-class sums of dz, two small GEMMs, and the thin wgrad of the pose part."""

    @staticmethod
    def forward(ctx, emb, pose, w, b):
        E = emb.shape[1]
        K = w.shape[3]
        wmat = H.emb_class_weights_fwd(w, E)                                           # [E, 9K]: per-border-class tap sums
        e9 = H.linear_fwd(H.to_f32(emb), wmat)                                         # [B, 9K]
        P = pose.shape[3]
        ctx.padded = H.get_compute() == "bf16" and K % 8 == 0 and P < 32
        if ctx.padded:
            # 'bf16' mode: the pose channels are widened to 32 (zeros) so that this conv runs on the bf16 matrix-pipe
            # loop like every other layer (the filter rows of the padding channels are zero: same result)
            pose = H.pad_channels_bf16(H.to_f32(pose), 32)
            w_pose = torch.zeros((3, 3, 32, K), dtype=w.dtype, device=w.device)
            w_pose[:, :, :P, :] = w[:, :, E:, :]
        else:
            w_pose = w[:, :, E:, :].contiguous()
        y = H.conv2d_fwd(pose, w_pose, b, act=ACT_RELU, residual=e9.view(-1, 9, K), res_class=True)
        ctx.save_for_backward(emb, pose, w, wmat, y)
        ctx.b_ref = b
        ctx.P = P
        return y

    @staticmethod
    def backward(ctx, dy):
        emb, pose, w, wmat, y = ctx.saved_tensors
        E, K = emb.shape[1], w.shape[3]
        B = emb.shape[0]
        dz = dy.contiguous() if _premasked(dy, y) else H.act_bwd(dy, y, ACT_RELU)
        z9 = H.border_class_sum(dz)                                                    # [B, 9, K] fp32
        db = None
        if ctx.needs_input_grad[3]:            # every pixel belongs to exactly one class: the bias gradient is their sum
            db = _sink(ctx.b_ref, lambda o, beta: H.colsum(z9.view(B * 9, K), out=o, beta=beta))
        z9 = z9.view(B, 9 * K)
        d_emb = H.linear_dgrad(z9, wmat) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[2]:
            dwp = H.conv2d_wgrad(pose, dz, (3, 3, pose.shape[3], K))                   # [3, 3, P (padded), K]
            dwc = H.linear_wgrad(H.to_f32(emb), z9)                                    # [E, 9K]: gradient of the class sums
            P = ctx.P

            def put(out, beta):
                if out is None:
                    out = torch.empty((3, 3, E + P, K), dtype=torch.float32, device=dwc.device)
                    beta = 0.0
                H.emb_class_weights_bwd(dwc, E, out, beta)                             # the transposed class sums -> taps
                H.axpby3d(dwp.view(9, dwp.shape[2], K)[:, :P, :], out.view(9, E + P, K)[:, E:, :], beta)
                return out
            dw = _sink(w, put)
        if d_emb is not None and d_emb.dtype != emb.dtype:
            d_emb = H.to_bf16(d_emb) if emb.dtype == H.BF16 else H.to_f32(d_emb)
        return d_emb, None, dw, db


class PoseKeypoints(object):
    """Stand-in for the [B,H,W,P] pose target map when the batch carries the keypoints it is built from (trainer.py:556-560:
    `coord2channel_simple_rcv(pose_rcv) -> tf_poseInflate`): the generator's first conv consumes the keypoints directly
    (`_TiledEmbKeypointConvFn`), nothing else on the hot path reads the map.  `dense()` rasterises it on demand."""

    def __init__(self, rcv, img_H, img_W, keypoint_num=18, is_normalized=False):
        B = rcv.shape[0]
        self.rcv = rcv.reshape(B, keypoint_num, 3)
        self.shape = (B, img_H, img_W, keypoint_num)
        self.is_normalized = is_normalized
        self.device = rcv.device

    def dim(self):
        return 4

    def dense(self):
        from . import utils
        B, Hh, W, K = self.shape
        return utils.pose_target_from_rcv(self.rcv.reshape(B, -1).contiguous(), K, self.is_normalized, Hh, W)


class _TiledEmbKeypointConvFn(torch.autograd.Function):
    """`_TiledEmbConvFn` with the pose channels given as keypoints: their -1 background is one more border-class constant, their
    discs a sparse sum (csrc/dpig_glue.hip::pose_stem_*): y = relu(conv3x3(concat([tile(emb), pose_map]), w) + b) without the map and
    without the dense 18-channel conv; the filter gradient of the pose rows likewise (class sums of dz + a gather over the discs)."""

    @staticmethod
    def forward(ctx, emb, rcv, w, b, Hh, W, normalized):
        E, K = emb.shape[1], w.shape[3]
        P = w.shape[2] - E
        wmat = H.emb_class_weights_fwd(w, E + P)                                       # [E + P, 9K]
        e9 = H.linear_fwd(H.to_f32(emb), wmat[:E])                                     # [B, 9K]
        cpos = H.colsum(wmat[E:])                                                      # [9K]: what the -1 background removes
        y = H.pose_stem_fwd(rcv, normalized, e9.view(-1, 9, K), cpos.view(9, K), b, w, E, Hh, W, ACT_RELU, 0.2,
                            bf16_out=H.get_compute() == "bf16" and K % 8 == 0)
        ctx.save_for_backward(emb, rcv, w, wmat, y)
        ctx.b_ref = b
        ctx.cfg = (P, normalized)
        return y

    @staticmethod
    def backward(ctx, dy):
        emb, rcv, w, wmat, y = ctx.saved_tensors
        P, normalized = ctx.cfg
        E, K = emb.shape[1], w.shape[3]
        B = emb.shape[0]
        dz = dy.contiguous() if _premasked(dy, y) else H.act_bwd(dy, y, ACT_RELU)
        z9 = H.border_class_sum(dz)                                                    # [B, 9, K] fp32
        db = None
        if ctx.needs_input_grad[3]:
            db = _sink(ctx.b_ref, lambda o, beta: H.colsum(z9.view(B * 9, K), out=o, beta=beta))
        d_emb = H.linear_dgrad(z9.view(B, 9 * K), wmat[:E]) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[2]:
            dwp = H.pose_stem_wgrad(rcv, normalized, z9, dz, P)                        # [9, P, K]
            dwc = H.linear_wgrad(H.to_f32(emb), z9.view(B, 9 * K))                     # [E, 9K]

            def put(out, beta):
                if out is None:
                    out = torch.empty((3, 3, E + P, K), dtype=torch.float32, device=dwc.device)
                    beta = 0.0
                H.emb_class_weights_bwd(dwc, E, out, beta)
                H.axpby3d(dwp, out.view(9, E + P, K)[:, E:, :], beta)
                return out
            dw = _sink(w, put)
        return d_emb, None, dw, db, None, None, None


def _keypoint_stem_ok(pose, w):
    """What dpig_pose_stem_fwd / _wgrad take: at most 32 keypoints, output channels in quads whose count divides 256."""
    K, P = w.shape[3], pose.shape[3]
    return P <= 32 and K % 4 == 0 and 256 % (K // 4) == 0 and w.data_ptr() % 16 == 0


def tiled_emb_conv(emb, pose, w, b):
    if isinstance(pose, PoseKeypoints):
        if not _keypoint_stem_ok(pose, w):                  # (e.g. conv_hidden_num = 96): the dense map and the dense conv
            return _mark_relu_output(_TiledEmbConvFn.apply(emb, pose.dense(), w, b), ACT_RELU)
        return _mark_relu_output(_TiledEmbKeypointConvFn.apply(emb, pose.rcv, w, b, pose.shape[1], pose.shape[2], pose.is_normalized),
                                 ACT_RELU)
    return _mark_relu_output(_TiledEmbConvFn.apply(emb, pose, w, b), ACT_RELU)


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, alpha):
        y = H.act_fwd(x, act, alpha)
        ctx.save_for_backward(y)
        ctx.cfg = (act, alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        if torch.is_grad_enabled():
            return _ActBwdFn.apply(dy, y, *ctx.cfg), None, None
        return H.act_bwd(dy, y, *ctx.cfg), None, None


def activation(x, act, alpha=0.2):
    return _ActFn.apply(x, act, alpha)


class _DeconvFn(torch.autograd.Function):
    """tf.nn.conv2d_transpose, stride 2, SAME (tflib/ops/deconv2d.py:97-103): y = F^T(x) where F is
    the stride-2 SAME conv [N,2H,2W,Cout] -> [N,H,W,Cin] with the (k,k,Cout,Cin) filter read as HWIO."""

    @staticmethod
    def forward(ctx, x, w, b):
        N, Hh, W, _ = x.shape
        out_shape = (N, 2 * Hh, 2 * W, w.shape[2])
        y = H.conv2d_dgrad(x, w, out_shape, stride=2)
        if b is not None:
            y = y + b          # bias_add on the small generator-side tensor (dormant path)
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = H.conv2d_fwd(dy, w, None, stride=2) if ctx.needs_input_grad[0] else None
        dw = H.conv2d_wgrad(dy, x, tuple(w.shape), stride=2) if ctx.needs_input_grad[1] else None
        db = H.colsum(dy) if ctx.has_b and ctx.needs_input_grad[2] else None
        return dx, dw, db


def conv2d_transpose(x, w, b=None):
    return _DeconvFn.apply(x, w, b)


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act, alpha):
        y = H.linear_fwd(x, w, b, act=act, alpha=alpha)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (act, alpha, b is not None)
        ctx.b_ref = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        act, alpha, has_b = ctx.cfg
        if torch.is_grad_enabled():            # backward under create_graph=True
            dz = _ActBwdFn.apply(dy, y, act, alpha) if act != ACT_NONE else dy.contiguous()
            dx = _LinearDgradFn.apply(dz, w) if ctx.needs_input_grad[0] else None
            dz = dz.detach()
        else:
            dz = H.act_bwd(dy, y, act, alpha) if act != ACT_NONE else dy.contiguous()
            dx = H.linear_dgrad(dz, w) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] and not _PARAM_GRADS_OFF[0]:
            dw = _sink(w, lambda o, beta: H.linear_wgrad(x, dz, out=o, beta=beta))
        if has_b and ctx.needs_input_grad[2] and not _PARAM_GRADS_OFF[0]:
            db = _sink(ctx.b_ref, lambda o, beta: H.colsum(dz, out=o, beta=beta))
        if dx is not None and dx.dtype != x.dtype:
            if torch.is_grad_enabled():        # (on the tape: the penalty's gradient w.r.t. `w` flows back through this conversion)
                dx = _CastFn.apply(dx, x.dtype == H.BF16)
            else:
                dx = H.to_bf16(dx) if x.dtype == H.BF16 else H.to_f32(dx)
        return dx, dw, db, None, None


def linear(x, w, b=None, act=ACT_NONE, alpha=0.2):
    return _LinearFn.apply(x, w, b, act, alpha)


class _BatchNormFn(torch.autograd.Function):
    """Training-mode BN over (N,H,W) of an NHWC tensor + fused activation (batchnorm.py:30)."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha, stats=None):
        y, mean, rstd = H.bn_fwd(x, scale, offset, eps, act, alpha, stats=stats)
        ctx.save_for_backward(x, scale, mean, rstd, y if act != ACT_NONE else None)
        ctx.cfg = (act, alpha)
        ctx.offset_ref = offset
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, mean, rstd, y = ctx.saved_tensors
        act, alpha = ctx.cfg
        off = _PARAM_GRADS_OFF[0]
        dx, dscale, doffset = H.bn_bwd(dy, x, y, scale, mean, rstd, act, alpha,
                                       dscale_out=None if off or not ctx.needs_input_grad[1] else _first_touch_out(scale),
                                       doffset_out=None if off or not ctx.needs_input_grad[2] else _first_touch_out(ctx.offset_ref))
        if off:
            return dx, None, None, None, None, None, None
        ds = _sink_small(scale, dscale) if ctx.needs_input_grad[1] else None
        do = _sink_small(ctx.offset_ref, doffset) if ctx.needs_input_grad[2] else None
        return dx, ds, do, None, None, None, None


class _SyncBatchNormFn(torch.autograd.Function):
    """The same op with batch statistics over every data-parallel rank ("offer SyncBN"): three tiny
    sum-all-reduces forward ([C] sums, [C] centred squares) / one backward ([2C]).  The returned scale / offset
    gradients are the global sums divided by the world size, so that the trainer's gradient all-reduce
    (sum over ranks, 1/world folded into Adam) reproduces exactly the single-process gradient."""

    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha, group):
        import torch.distributed as dist
        world = dist.get_world_size(group)
        ar = lambda t: eager_island(lambda: dist.all_reduce(t, group=group))     # (between two captured graphs under SegmentedCapture)
        y, mean, rstd = H.bn_sync_fwd(x, scale, offset, eps, act, alpha, ar, world)
        ctx.save_for_backward(x, scale, mean, rstd, y if act != ACT_NONE else None)
        ctx.cfg = (act, alpha, group, world)
        ctx.offset_ref = offset
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        x, scale, mean, rstd, y = ctx.saved_tensors
        act, alpha, group, world = ctx.cfg
        ar = lambda t: eager_island(lambda: dist.all_reduce(t, group=group))
        dx, dscale, doffset = H.bn_sync_bwd(dy, x, y, scale, mean, rstd, act, alpha, ar, world)
        # every rank back-propagates its LOCAL mean loss, i.e. world x its share of the global mean loss; the
        # trainer then averages parameter gradients over ranks.  dx is consistent with that convention as it is;
        # the scale / offset sums are already global, so each rank contributes sum / world.
        dscale = dscale * (1.0 / world)
        doffset = doffset * (1.0 / world)
        ds = _sink_small(scale, dscale) if ctx.needs_input_grad[1] else None
        do = _sink_small(ctx.offset_ref, doffset) if ctx.needs_input_grad[2] else None
        return dx, ds, do, None, None, None, None


class _GPPenaltyFn(torch.autograd.Function):
    """LAMBDA * mean_b (||g_b||_2 - 1)^2 of the critic's input gradient g (trainer.py:233-236): the value and its
    derivative w.r.t. g -- the seed of the double-backward sweep -- come out of one pass over g."""

    @staticmethod
    def forward(ctx, g, lam):
        pen, dg, _ = H.gp_penalty(g, lam)
        ctx.save_for_backward(dg)
        return pen.reshape(())

    @staticmethod
    def backward(ctx, dout):
        (dg,) = ctx.saved_tensors
        return dg * dout, None


def gp_penalty(g, lam):
    return _GPPenaltyFn.apply(g, lam)


class _FusedGPFn(torch.autograd.Function):
    """The whole penalty term of the DCGAN critic (trainer.py:222-236 over wgan_gp.py:407-440) through the library's single
    entry point `dpig_gp_double_backward`: value and every parameter gradient in one call, no second-level autograd tape."""

    @staticmethod
    def forward(ctx, real, fake, alpha, lam, dim, keys, *params):
        want = any(ctx.needs_input_grad[6:])
        pd = {k: p.detach() for k, p in zip(keys, params)}
        pen, _, grads = H.gp_double_backward(pd, real, fake, alpha, lam, dim=dim, grads=True if want else None)
        ctx.grads = [grads[k] for k in keys] if want else None
        ctx.params = params
        return pen.reshape(())

    @staticmethod
    def backward(ctx, dout):
        out = [None] * 6
        for i, (p, g) in enumerate(zip(ctx.params, ctx.grads)):
            out.append(_sink_small(p, g * dout) if ctx.needs_input_grad[6 + i] else None)
        return tuple(out)


def gp_fused(real, fake, alpha, lam, dim, named_params):
    """named_params: ordered {critic variable name: parameter} with the keys of `hip_ops.CRITIC_KEYS`."""
    keys = tuple(named_params.keys())
    return _FusedGPFn.apply(real, fake, alpha, float(lam), int(dim), keys, *[named_params[k] for k in keys])


_SYNC_BN_GROUP = [False, None]      # (enabled, process group)


def set_sync_batchnorm(enabled, group=None):
    """Use cross-rank batch statistics in every subsequent batchnorm() call (no-op without torch.distributed)."""
    _SYNC_BN_GROUP[0], _SYNC_BN_GROUP[1] = bool(enabled), group


def batchnorm(x, scale, offset, eps=1e-5, act=ACT_NONE, alpha=0.2, stats=None):
    """`stats`: the `_dpig_bnstats` a `conv2d(..., bn_stats=True)` attached to x (ignored with cross-rank statistics)."""
    if _SYNC_BN_GROUP[0]:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(_SYNC_BN_GROUP[1]) > 1:
            return _SyncBatchNormFn.apply(x, scale, offset, eps, act, alpha, _SYNC_BN_GROUP[1])
    return _BatchNormFn.apply(x, scale, offset, eps, act, alpha, stats)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, offset, eps, act, alpha):
        y, mean, rstd = H.ln_fwd(x, scale, offset, eps, act, alpha)
        ctx.save_for_backward(x, scale, mean, rstd, y if act != ACT_NONE else None)
        ctx.cfg = (act, alpha)
        ctx.offset_ref = offset
        return y

    @staticmethod
    def backward(ctx, dy):
        x, scale, mean, rstd, y = ctx.saved_tensors
        act, alpha = ctx.cfg
        if torch.is_grad_enabled():            # backward under create_graph=True
            dx = _LNBwdFn.apply(dy, x, y, scale, mean, rstd, act, alpha)
            if _PARAM_GRADS_OFF[0] or not (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                return dx, None, None, None, None, None
            _, dscale, doffset = H.ln_bwd(dy.detach(), x, y, scale, mean, rstd, act, alpha)
        else:
            dx, dscale, doffset = H.ln_bwd(dy, x, y, scale, mean, rstd, act, alpha,
                                           dscale_out=_first_touch_out(scale) if ctx.needs_input_grad[1] else None,
                                           doffset_out=_first_touch_out(ctx.offset_ref) if ctx.needs_input_grad[2] else None)
        ds = _sink_small(scale, dscale) if ctx.needs_input_grad[1] else None
        do = _sink_small(ctx.offset_ref, doffset) if ctx.needs_input_grad[2] else None
        return dx, ds, do, None, None, None


def layernorm(x, scale, offset, eps=1e-5, act=ACT_NONE, alpha=0.2):
    return _LayerNormFn.apply(x, scale, offset, eps, act, alpha)


class _CropResizeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, boxes, box_ind, ch, cw):
        out = H.crop_resize_fwd(img, boxes, box_ind, ch, cw)
        ctx.save_for_backward(boxes, box_ind)
        ctx.img_shape = tuple(img.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        boxes, box_ind = ctx.saved_tensors
        return H.crop_resize_bwd(dout, boxes, box_ind, ctx.img_shape), None, None, None, None


def crop_and_resize(img, boxes, box_ind, ch, cw):
    return _CropResizeFn.apply(img, boxes, box_ind, ch, cw)


class _Upsample2xFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return H.upsample2x_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        return H.upsample2x_bwd(dy)


def upsample2x(x):
    return _Upsample2xFn.apply(x)


class _SceMeanFn(torch.autograd.Function):
    """mean(sigmoid_cross_entropy_with_logits(logits, label)) (trainer.py:239-243)."""

    @staticmethod
    def forward(ctx, logits, label):
        out, dl = H.sce_mean(logits, label, want_grad=True, scale=1.0)
        ctx.save_for_backward(dl)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def sce_mean(logits, label):
    return _SceMeanFn.apply(logits, float(label))


class _LogitMeanFn(torch.autograd.Function):
    """mean(x) / mean((x - target)^2) over a logit vector (trainer.py:218-220 wgan, :246-248 lsgan)."""

    @staticmethod
    def forward(ctx, logits, squared, target):
        out, dl = H.logit_mean(logits, squared, target, want_grad=True, scale=1.0)
        ctx.save_for_backward(dl)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def logit_mean(logits):
    return _LogitMeanFn.apply(logits, False, 0.0)


def logit_sq_mean(logits, target):
    return _LogitMeanFn.apply(logits, True, float(target))


class _L1MeanFn(torch.autograd.Function):
    """mean(|a - b|) with gradient to a only (trainer.py:607: tf.reduce_mean(tf.abs(G - x)))."""

    @staticmethod
    def forward(ctx, a, b):
        out, da = H.l1_mean(a, b, want_grad=True, scale=1.0)
        ctx.save_for_backward(da)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (da,) = ctx.saved_tensors
        return da * g, None


def l1_mean(a, b):
    return _L1MeanFn.apply(a, b)



class _Like(object):
    """shape / dtype / device donor for an output allocation (no tensor kept alive)"""

    def __init__(self, t):
        self.shape, self.dtype, self.device = tuple(t.shape), t.dtype, t.device


class _MaskSplitFn(torch.autograd.Function):
    """x_fg = x * m, x_bg = x * (1 - m) (models.py:402-403) in one launch; the gradient dx = dfg * m + dbg * (1 - m) in another."""

    @staticmethod
    def forward(ctx, x, m):
        fg, bg, mflat = H.mask_split_fwd(x, m)
        ctx.save_for_backward(mflat)
        ctx.like = _Like(x)
        return fg, bg

    @staticmethod
    def backward(ctx, dfg, dbg):
        (mflat,) = ctx.saved_tensors
        return H.mask_split_bwd(dfg, dbg, mflat, ctx.like), None


def mask_split(x, m):
    return _MaskSplitFn.apply(x, m)


class _VisConcatFn(torch.autograd.Function):
    """models.py:433-442 + 467-468: per-part features times the visibility flag, concatenated (with the background feature)."""

    @staticmethod
    def forward(ctx, fea, vis, bg, P, z):
        B = vis.shape[0]
        out, visf = H.vis_concat_fwd(fea, vis, bg, B, P, z)
        ctx.save_for_backward(visf)
        ctx.cfg = (B, P, z, 0 if bg is None else bg.shape[1])
        return out

    @staticmethod
    def backward(ctx, dall):
        (visf,) = ctx.saved_tensors
        B, P, z, zbg = ctx.cfg
        dfea, dbg = H.vis_concat_bwd(dall, visf, B, P, z, zbg, zbg > 0 and ctx.needs_input_grad[2])
        return dfea, None, dbg, None, None


def vis_concat(fea, vis, bg, P, z):
    return _VisConcatFn.apply(fea, vis, bg, P, z)


class _Transpose12Fn(torch.autograd.Function):
    """[B, A, C] -> [B, C, A] in one kernel; its gradient is the same op on the gradient, so it differentiates any number of
    times (the WGAN-GP sweep differentiates the critic's backward pass once more)."""

    @staticmethod
    def forward(ctx, x):
        return H.transpose12(x)

    @staticmethod
    def backward(ctx, dy):
        return _Transpose12Fn.apply(dy.contiguous()) if torch.is_grad_enabled() else H.transpose12(dy)


def nchw_flatten(x_nchw_view, n):
    """tf.reshape(output, [-1, n]) of the critic's logical NCHW tensor (wgan_gp.py:433).  When the data is physically NHWC (a
    permuted view) this is one [B, HW, C] -> [B, C, HW] transpose kernel instead of a strided torch copy."""
    if x_nchw_view.dim() == 4:
        nhwc = x_nchw_view.permute(0, 2, 3, 1)
        B, Hh, W, C = nhwc.shape
        if nhwc.is_contiguous() and (Hh * W * C) % n == 0:
            return _Transpose12Fn.apply(nhwc.reshape(B, Hh * W, C)).reshape(-1, n)
    return x_nchw_view.reshape(-1, n)
