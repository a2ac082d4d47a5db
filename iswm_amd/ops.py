"""Tensor-level wrappers over the C ABI (include/iswm_hip.h).

torch supplies device memory, the current HIP stream and nothing else: every
function here validates its tensors, allocates outputs through torch's caching
allocator and enqueues hand-written gfx950 kernels from libiswm_hip.so on
``torch.cuda.current_stream()``.  There is no fallback path.

Activations are NHWC: a 4-D fp32 tensor ``[N, H, W, C]`` whose last dim is
contiguous and whose pixel pitch ``ld = t.stride(2)`` may exceed C (a channel
slice of a wider buffer -- that is how torch.cat disappears from the graph).
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, call

BN_EPS = 1e-5


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def geom(t, dtype=torch.float32):
    """(N, H, W, C, ld) of an NHWC activation view; raises on any other layout."""
    if isinstance(t, Planes):
        raise ValueError("this operation takes an fp32 NHWC tensor, not Planes (use .f32())")
    if t.dim() != 4 or t.dtype != dtype or not t.is_cuda:
        raise ValueError("expected a 4-D fp32 CUDA NHWC tensor, got %s %s %s" % (tuple(t.shape), t.dtype, t.device))
    n, h, w, c = t.shape
    if w > 1:
        ld = t.stride(2)
    elif h > 1:
        ld = t.stride(1)
    elif n > 1:
        ld = t.stride(0)
    else:
        ld = c
    ok = (c == 1 or t.stride(3) == 1) and (w == 1 or t.stride(2) == ld) and \
        (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld)
    if not ok or ld < c:
        raise ValueError("tensor is not a pitched NHWC view: shape %s strides %s" % (tuple(t.shape), t.stride()))
    return n, h, w, c, ld


def new_act(n, h, w, c, device):
    return torch.empty((n, h, w, c), dtype=torch.float32, device=device)


# ---- "planes": activations stored pre-split for the bf16x6 convolution kernels (csrc/planes.h) -------------
# ISWM_PLANES=0 keeps every activation fp32 (the round-1 data path); conv math f32 / bf16 do that too.
_PLANES_ENV = os.environ.get("ISWM_PLANES", "1") != "0"
_WGRAD_PLANES = os.environ.get("ISWM_WGRAD_PLANES", "1") != "0"     # tuning switch: 0 joins the planes and runs the fp32-input weight gradient


# conv math "bf16" (BASELINE configs[4], mixed precision): activations between convolutions are STORED as one bf16 plane
# (round to nearest even) -- 2 bytes per element through every memory-bound pass instead of 4 (fp32) or 6 (bf16x6 planes).
# ISWM_BF16_STORE=0 keeps them fp32 and rounds inside the conv kernels (the round-2 form of this mode).
_BF16_STORE = os.environ.get("ISWM_BF16_STORE", "1") != "0"


def nplanes():
    """bf16 planes an activation is stored as under the current conv math: 3 (bf16x6: exact split), 1 (bf16: rounded), 0 (fp32)"""
    math = _lib.load().iswm_get_conv_math()
    if not _PLANES_ENV:
        return 0
    return 3 if math == 1 else (1 if (math == 2 and _BF16_STORE) else 0)


def planes_on():
    return nplanes() > 0


class Planes(object):
    """An NHWC fp32 activation held as its exact 3-way bf16 split: `t` is a bf16 tensor [3, N, H, W, C]
    (plane, then a pitched NHWC view), hi + mid + lo == the fp32 value bit for bit.  Producers (BatchNorm / pooling /
    resize passes) write it, the convolution kernels stage it into LDS by DMA; anything else asks for `.f32()`.
    Under conv math "bf16" there is ONE plane, [1, N, H, W, C]: the value rounded to nearest bf16 (C-ABI plane stride -1)."""

    __slots__ = ("t",)

    def __init__(self, t):
        assert t.dim() == 5 and t.dtype == torch.bfloat16 and t.shape[0] in (1, 3)
        self.t = t

    @property
    def shape(self):
        return self.t.shape[1:]

    @property
    def device(self):
        return self.t.device

    def dim(self):
        return 4

    def __getitem__(self, idx):
        if not (isinstance(idx, tuple) and len(idx) == 2 and idx[0] is Ellipsis and isinstance(idx[1], slice)):
            raise IndexError("Planes supports channel slices x[..., a:b] only")
        return Planes(self.t[..., idx[1]])

    def f32(self):
        n, h, w, c, ld, ps = pgeom(self)
        out = new_act(n, h, w, c, self.t.device)
        call("iswm_join_planes", _p(self.t), ld, ps, n * h * w, c, _p(out), c, _stream())
        return out

    def zero_(self):
        self.t.zero_()
        return self


def is_planes(x):
    return isinstance(x, Planes)


def as_f32(x):
    return x.f32() if isinstance(x, Planes) else x


def new_planes(n, h, w, c, device, zero=False):
    mk = torch.zeros if zero else torch.empty
    return Planes(mk((max(1, nplanes()), n, h, w, c), dtype=torch.bfloat16, device=device))


def pgeom(x):
    """(N, H, W, C, ld, plane stride) of a Planes tensor, both in bf16 elements"""
    n, h, w, c, ld = geom(x.t[0], torch.bfloat16)
    return n, h, w, c, ld, (x.t.stride(0) if x.t.shape[0] == 3 else -1)


def xgeom(x):
    """(pointer tensor, N, H, W, C, ld, ps) of an fp32 NHWC tensor (ps = 0) or a Planes tensor"""
    if isinstance(x, Planes):
        return (x.t,) + pgeom(x)
    return (x,) + geom(x) + (0,)


def pad_weights(w, cout_p, cin_p):
    """zero-padded OHWI [cout_p, KH, KW, cin_p] copy of an OIHW parameter (any strides)"""
    cout, cin, kh, kw = w.shape
    out = torch.empty((cout_p, kh, kw, cin_p), dtype=torch.float32, device=w.device)
    st = (ctypes.c_int64 * 4)(*w.stride())
    call("iswm_pad_weights", _p(w), cout, cin, kh, kw, st, cout_p, cin_p, _p(out), _stream())
    return out


def unpad_weights(dw_ohwi, grad):
    """grad (OIHW view, any strides) <- the leading [cout, :, :, cin] block of a padded OHWI gradient"""
    cout_p, kh, kw, cin_p = dw_ohwi.shape
    cout, cin = grad.shape[0], grad.shape[1]
    assert dw_ohwi.is_contiguous() and tuple(grad.shape[2:]) == (kh, kw)
    st = (ctypes.c_int64 * 4)(*grad.stride())
    call("iswm_unpad_weights", _p(dw_ohwi), cout_p, cin_p, cout, cin, kh, kw, _p(grad), st, _stream())
    return grad


def zero_channels(t, c0):
    """zero channels [c0, C) of an NHWC activation (fp32 tensor or Planes): the padding channels of a concatenation buffer"""
    if isinstance(t, Planes):
        n, h, w, c, ld, ps = pgeom(t)
        call("iswm_zero_cols", _p(t.t), n * h * w, ld * 2, c0 * 2, c * 2, (ps if ps > 0 else 0) * 2, t.t.shape[0], _stream())
    else:
        n, h, w, c, ld = geom(t)
        call("iswm_zero_cols", _p(t), n * h * w, ld * 4, c0 * 4, c * 4, 0, 1, _stream())
    return t


def split_planes(x, out=None):
    """fp32 NHWC -> Planes (one extra pass; producers normally write planes themselves)"""
    n, h, w, c, ld = geom(x)
    if out is None:
        out = new_planes(n, h, w, c, x.device)
    _, _, _, _, ldp, ps = pgeom(out)
    call("iswm_split_planes", _p(x), n * h * w, c, ld, _p(out.t), ldp, ps if ps > 0 else n * h * w * ldp, _stream())
    return out


class KernelProfile:
    """Optional per-launch timing of the conv kernels with HIP events recorded on the stream the
    kernels are launched on (torch's current stream).  bench.py turns it on for the timed region to
    report the MFMA roofline fraction; it is off (None) otherwise and costs nothing."""

    def __init__(self, only=None):
        self.records = []   # (kernel name, start event, end event, algorithmic flops, geometry tag)
        self.only = only    # when set: time launches of this kernel only (keeps the timed region light)
        self.scope = None   # free-form label stamped on the records added while it is set (bench: "head_fwd")

    def begin(self):
        return True

    def end(self, name, start, flops, tag=None):
        # the bracketing events are recorded by _timed(); kept for API symmetry
        raise NotImplementedError

    def wants(self, name):
        return self.only is None or name == self.only or name + "+reduce" == self.only

    def add(self, name, a, b, flops, tag):
        self.records.append((name, a, b, flops, tag, self.scope))

    def scope_total(self, scope):
        """dict(launches, ms, flops) over the records stamped with this scope"""
        d = dict(launches=0, ms=0.0, flops=0.0)
        for name, a, b, fl, _, sc in self.records:
            if sc == scope:
                d["launches"] += 1
                d["ms"] += a.elapsed_time(b)
                d["flops"] += fl
        return d

    def summary(self):
        """{name: dict(launches, ms, flops)} -- call after torch.cuda.synchronize()"""
        out = {}
        for name, a, b, fl, _, _sc in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += fl
        return out

    def by_geometry(self):
        """{(name, geometry tag): dict(launches, ms, flops)}"""
        out = {}
        for name, a, b, fl, tag, _sc in self.records:
            d = out.setdefault((name, tag), dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += a.elapsed_time(b)
            d["flops"] += fl
        return out


KPROF = None


def _valid_taps(h, ho, k, stride, pad, dil):
    """number of (output index, tap) pairs whose input index is in bounds"""
    n = 0
    for t in range(k):
        off = t * dil - pad
        # count o in [0, ho) with 0 <= o*stride + off < h
        o_min = 0 if off >= 0 else (-off + stride - 1) // stride
        o_max = min(ho - 1, (h - 1 - off) // stride) if h - 1 - off >= 0 else -1
        n += max(0, o_max - o_min + 1)
    return n


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


class ConvGeom:
    """Static geometry of one conv call (wraps iswm_conv_desc)."""

    def __init__(self, x, cout, kh, kw, stride, pad, dil):
        n, h, w, cin = x.shape
        self.n, self.h, self.w, self.cin, self.cout = n, h, w, cin, cout
        self.kh, self.kw, self.stride, self.pad, self.dil = kh, kw, stride, pad, dil
        self.ho = conv_out_size(h, kh, stride, pad, dil)
        self.wo = conv_out_size(w, kw, stride, pad, dil)
        self.alg_cin, self.alg_cout = cin, cout      # un-padded channel counts (set by the module)
        self._flops = None

    def tag(self):
        return "n%d %dx%d c%d->%d k%d s%d d%d" % (self.n, self.h, self.w, self.alg_cin, self.alg_cout, self.kh,
                                                  self.stride, self.dil)

    def flops(self):
        """algorithmic FLOPs = 2 x MACs over IN-BOUNDS taps only; identical for the
        forward, data-gradient and weight-gradient passes"""
        if self._flops is None:
            vh = _valid_taps(self.h, self.ho, self.kh, self.stride, self.pad, self.dil)
            vw = _valid_taps(self.w, self.wo, self.kw, self.stride, self.pad, self.dil)
            self._flops = 2.0 * self.n * vh * vw * self.alg_cin * self.alg_cout
        return self._flops

    def desc(self, ldx, ldy):
        return ConvDesc(self.n, self.h, self.w, self.cin, self.ho, self.wo, self.cout, self.kh, self.kw,
                        self.stride, self.pad, self.dil, ldx, ldy)


_NAME_CACHE = {}


def _kernel_name(d, kind):
    """device kernel symbol the library launches for this geometry (labels profile records)"""
    lib = _lib.load()
    key = (kind, lib.iswm_get_conv_math(), d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.stride, d.pad, d.dil)
    name = _NAME_CACHE.get(key)
    if name is None:
        buf = ctypes.create_string_buffer(64)
        lib.iswm_conv2d_kernel_name(ctypes.byref(d), kind, buf, 64)
        name = _NAME_CACHE[key] = buf.value.decode()
    return name


class _timed:
    """bracket one conv launch with HIP events on the launch stream when a KernelProfile is active"""

    def __init__(self, d, kind, g, suffix=""):
        self.on = False
        if KPROF is not None:
            name = _kernel_name(d, kind) + suffix
            if KPROF.wants(name):
                self.on, self.name, self.g = True, name, g

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            KPROF.add(self.name, self.a, b, self.g.flops(), self.g.tag())
        return False


def _check_w(w_ohwi, g):
    if tuple(w_ohwi.shape) != (g.cout, g.kh, g.kw, g.cin) or not w_ohwi.is_contiguous():
        raise ValueError("weight must be contiguous OHWI %s, got %s strides %s" %
                         ((g.cout, g.kh, g.kw, g.cin), tuple(w_ohwi.shape), w_ohwi.stride()))


WEIGHTS_EPOCH = 0     # bumped by the fused optimizers: packed-weight buffers made before a step are stale after it


def weights_changed():
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


def conv2d_fwd(x, w_ohwi, g, bias=None, out=None, want_stats=False, wpk=None, wpk2=None):
    """y = conv(x, w) [+ bias]; returns (y, partials|None, (tiles, tile_rows)).  wpk: this weight already packed
    for the forward kernel (network._hip.WeightPacker), else it is packed here."""
    _check_w(w_ohwi, g)
    if out is None:
        out = new_act(g.n, g.ho, g.wo, g.cout, x.device)
    on, oh, ow, oc, ldy = geom(out)
    assert (on, oh, ow, oc) == (g.n, g.ho, g.wo, g.cout)
    if isinstance(x, Planes):
        _, _, _, _, ldp, ps = pgeom(x)
        d = g.desc(ldp, ldy)
        nb2 = _pl2_bytes(d, 0)
        if nb2:
            partials, tiles = None, (0, 0)
            if want_stats:
                tr = _lib.load().iswm_conv2d_pl2_tile_rows(ctypes.byref(d), 0)
                tiles = ((g.n * g.ho * g.wo + tr - 1) // tr, tr)
                partials = torch.empty((2, tiles[0], g.cout), dtype=torch.float32, device=out.device)
            if wpk2 is None:
                wpk2 = torch.empty((nb2 // 4,), dtype=torch.float32, device=out.device)
                call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d), 0, _p(w_ohwi), _p(wpk2), _stream())
            with _timed(d, 5, g):
                call("iswm_conv2d_fwd_pl2", ctypes.byref(d), _p(x.t), ps, _p(wpk2), _p(bias), _p(out), _p(partials), _stream())
            return out, partials, tiles
        x = x.f32()
    ldx = geom(x)[4]
    d = g.desc(ldx, ldy)
    partials, tiles = None, (0, 0)
    if want_stats:
        lib = _lib.load()
        tiles = (lib.iswm_conv2d_stat_tiles(ctypes.byref(d)), lib.iswm_conv2d_stat_tile_rows(ctypes.byref(d)))
        partials = torch.empty((2, tiles[0], g.cout), dtype=torch.float32, device=x.device)
    nb = _packed_bytes(d, 0)
    if nb:
        if want_stats:
            nt, tr = ctypes.c_int(0), ctypes.c_int(0)
            call("iswm_conv2d_fwd_packed_stat_layout", ctypes.byref(d), ctypes.byref(nt), ctypes.byref(tr))
            tiles = (nt.value, tr.value)
            flat = torch.empty((2 * nt.value * g.cout + nt.value,), dtype=torch.float32, device=x.device)
            partials = flat[:2 * nt.value * g.cout].view(2, nt.value, g.cout)   # per-tile row counts follow
        if wpk is None:
            wpk = torch.empty((nb // 4,), dtype=torch.float32, device=x.device)
            call("iswm_conv2d_pack_weights", ctypes.byref(d), 0, _p(w_ohwi), _p(wpk), _stream())
        with _timed(d, 3, g):
            call("iswm_conv2d_fwd_packed", ctypes.byref(d), _p(x), _p(wpk), _p(bias), _p(out), _p(partials), _stream())
        return out, partials, tiles
    with _timed(d, 0, g):
        call("iswm_conv2d_fwd", ctypes.byref(d), _p(x), _p(w_ohwi), _p(bias), _p(out), _p(partials), _stream())
    return out, partials, tiles


_USE_PACKED = os.environ.get("ISWM_X6_PK", "1") != "0"


def _packed_bytes(d, kind):
    return _lib.load().iswm_conv2d_packed_weight_bytes(ctypes.byref(d), kind) if _USE_PACKED else 0


def _pl2_bytes(d, kind):
    """packed-weight bytes of the planes kernels for this geometry, 0 when they do not apply"""
    return _lib.load().iswm_conv2d_pl2_weight_bytes(ctypes.byref(d), kind) if planes_on() else 0


def planes_conv_ok(cin, cout, kind):
    """will a conv with these channel counts take a Planes operand (kind 0: x forward, 1: dy data gradient)?"""
    return planes_on() and (cout if kind else cin) % 64 == 0


class BnStats(object):
    """Request / result of the BatchNorm-backward statistics a planes data gradient can take in its epilogue: `y`, `coef`
    (scale, shift, mean, invstd) and `relu` describe the stage that PRODUCED the conv's input (whose BatchNorm backward will
    consume dx); conv2d_dgrad fills `partials`, `tiles` when its kernel supports the fusion (they stay None otherwise)."""

    def __init__(self, y, coef, relu, mask=None):
        self.y, self.coef, self.relu = y, coef, relu
        self.partials, self.tiles = None, 0
        # a RESIDUAL producer stage (out = relu(bn(y) + identity)): `mask` = its saved output (Planes); the data gradient then
        # stores dx MASKED by (out > 0) and sets `masked` -- that tensor is the stage's dout (relu already applied) AND the
        # gradient of its identity branch
        self.mask, self.masked = mask, False


_BN_FUSE = os.environ.get("ISWM_BN_FUSE", "1") != "0"      # tuning switch: 0 = BatchNorm backward always reduces itself


def conv2d_dgrad(dy, w_ohwi, g, x_like_shape, dx=None, accumulate=False, wpk=None, wpk2=None, bn_stats=None):
    """dx (=|+=) conv^T(dy, w).  x_like_shape = (N,H,W,Cin) of the conv input.  wpk: weight already packed for the
    data-gradient kernel (wpk2: for the planes kernel).  bn_stats: a BnStats to fill (planes kernel only)."""
    _check_w(w_ohwi, g)
    if dx is None:
        assert not accumulate
        dx = new_act(*x_like_shape, dy.device)
    ldx = geom(dx)[4]
    if isinstance(dy, Planes):
        _, _, _, _, ldp, ps = pgeom(dy)
        d = g.desc(ldx, ldp)
        nb2 = _pl2_bytes(d, 1)
        if nb2:
            if wpk2 is None:
                wpk2 = torch.empty((nb2 // 4,), dtype=torch.float32, device=dx.device)
                call("iswm_conv2d_pl2_pack_weights", ctypes.byref(d), 1, _p(w_ohwi), _p(wpk2), _stream())
            if bn_stats is not None and _BN_FUSE and g.cin % 4 == 0:
                b = bn_stats
                tiles = _lib.load().iswm_conv2d_dgrad_pl2_stat_tiles(ctypes.byref(d))
                part = torch.empty((2, tiles, g.cin), dtype=torch.float64, device=dx.device)
                masky = _relu_code(b.relu) == 1 and b.mask is None
                pm, ldm, code = None, 0, 2 if masky else 0
                if b.mask is not None:
                    pm, _, _, ldm, _ = xrows(b.mask)            # plane 0 = hi
                    code = 3
                with _timed(d, 6, g):
                    call("iswm_conv2d_dgrad_pl2_bn", ctypes.byref(d), _p(dy.t), ps, _p(wpk2), _p(dx), int(bool(accumulate)),
                         _p(b.y), rows(b.y)[2], _p(b.coef[2]), _p(b.coef[3]), _p(b.coef[0]) if masky else None,
                         _p(b.coef[1]) if masky else None, code, _p(pm), ldm, _p(part), tiles, _stream())
                b.partials, b.tiles, b.masked = part, tiles, b.mask is not None
                return dx
            with _timed(d, 6, g):
                call("iswm_conv2d_dgrad_pl2", ctypes.byref(d), _p(dy.t), ps, _p(wpk2), _p(dx), int(bool(accumulate)), _stream())
            return dx
        dy = dy.f32()
    ldy = geom(dy)[4]
    d = g.desc(ldx, ldy)
    nb = _packed_bytes(d, 1)
    if nb:
        if wpk is None:
            wpk = torch.empty((nb // 4,), dtype=torch.float32, device=dy.device)
            call("iswm_conv2d_pack_weights", ctypes.byref(d), 1, _p(w_ohwi), _p(wpk), _stream())
        with _timed(d, 4, g):
            call("iswm_conv2d_dgrad_packed", ctypes.byref(d), _p(dy), _p(wpk), _p(dx), int(bool(accumulate)), _stream())
        return dx
    if _lib.load().iswm_conv2d_dgrad_wants_wt(ctypes.byref(d)):
        # bf16x6 math: the matrix cores want the K axis (tap, cout) contiguous -> transposed weights
        wt = torch.empty((g.cin, g.kh, g.kw, g.cout), dtype=torch.float32, device=dy.device)
        call("iswm_transpose_weights", ctypes.byref(d), _p(w_ohwi), _p(wt), _stream())
        with _timed(d, 1, g):
            call("iswm_conv2d_dgrad_wt", ctypes.byref(d), _p(dy), _p(wt), _p(dx), int(bool(accumulate)), _stream())
        return dx
    with _timed(d, 1, g):
        call("iswm_conv2d_dgrad", ctypes.byref(d), _p(dy), _p(w_ohwi), _p(dx), int(bool(accumulate)), _stream())
    return dx


def _wgrad_planes_ok(x, dy, g, c8):
    """iswm_conv2d_wgrad_planes_ok on the descriptor the planes weight gradient would get (every precondition of the
    kernel: channel / pitch alignment, the 2^30-element indexing bound); False -> the fp32-input weight gradient runs"""
    ldx = pgeom(x)[4]
    ldy = pgeom(dy)[4] if (isinstance(dy, Planes) and c8 == g.cout) else c8
    d = ConvDesc(g.n, g.h, g.w, g.cin, g.ho, g.wo, c8, g.kh, g.kw, g.stride, g.pad, g.dil, ldx, ldy)
    return bool(_lib.load().iswm_conv2d_wgrad_planes_ok(ctypes.byref(d)))


def conv2d_wgrad(x, dy, g, dw_ohwi=None):
    """dw[Cout,KH,KW,Cin] = sum_pixels dy (x) gathered x.  With a pre-split x the planes kernel runs (dy is split
    here when the producer did not: the few-channel classifier gradient)."""
    if dw_ohwi is None:
        dw_ohwi = torch.empty((g.cout, g.kh, g.kw, g.cin), dtype=torch.float32, device=x.device)
    _check_w(dw_ohwi, g)
    c8 = (g.cout + 7) // 8 * 8
    if isinstance(x, Planes) and planes_on() and _WGRAD_PLANES and _wgrad_planes_ok(x, dy, g, c8):
        if not isinstance(dy, Planes) or c8 != g.cout:
            dyf = as_f32(dy)
            dy = new_planes(g.n, g.ho, g.wo, c8, x.device)
            if c8 != g.cout:
                zero_channels(dy, g.cout)
            split_planes(dyf, out=dy[..., :g.cout] if c8 != g.cout else dy)
        _, _, _, _, ldx, psx = pgeom(x)
        _, _, _, _, ldy, psy = pgeom(dy)
        d = ConvDesc(g.n, g.h, g.w, g.cin, g.ho, g.wo, c8, g.kh, g.kw, g.stride, g.pad, g.dil, ldx, ldy)
        need = _lib.load().iswm_conv2d_wgrad_planes_workspace(ctypes.byref(d))
        ws = torch.empty((need // 4,), dtype=torch.float32, device=x.device) if need else None
        tgt = dw_ohwi if c8 == g.cout else torch.empty((c8, g.kh, g.kw, g.cin), dtype=torch.float32, device=x.device)
        with _timed(d, 7, g, "+reduce"):
            call("iswm_conv2d_wgrad_planes", ctypes.byref(d), _p(x.t), psx, _p(dy.t), psy, _p(tgt), _p(ws), need, _stream())
        if tgt is not dw_ohwi:
            unpad_weights(tgt, dw_ohwi.permute(0, 3, 1, 2))
        return dw_ohwi
    x, dy = as_f32(x), as_f32(dy)
    ldx, ldy = geom(x)[4], geom(dy)[4]
    d = g.desc(ldx, ldy)
    need = _lib.load().iswm_conv2d_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty((need // 4,), dtype=torch.float32, device=x.device) if need else None
    with _timed(d, 2, g, "+reduce"):
        call("iswm_conv2d_wgrad", ctypes.byref(d), _p(x), _p(dy), _p(dw_ohwi), _p(ws), need, _stream())
    return dw_ohwi


# ---- ASPP: the parallel branch convolutions as one launch (csrc/conv_mfma_pl2t.hip) ---------------------------------
ASPP_TILE_ROWS = 144
_ASPP_PLANS = {}
_ASPP_FUSED = os.environ.get("ISWM_ASPP_FUSED", "1") != "0"      # tuning switch: 0 = one launch per branch


def _int_array(v):
    return (ctypes.c_int * len(v))(*[int(i) for i in v])


def _ptr_array(ts):
    return (ctypes.c_void_p * len(ts))(*[(t.data_ptr() if t is not None else None) for t in ts])


def aspp_desc(n, h, w, cin, cout, ldx, ldy):
    return ConvDesc(n, h, w, cin, h, w, cout, 1, 1, 1, 0, 1, ldx, ldy)


def aspp_plan(n, h, w, cin, cout, ksize, dil, kind, device):
    """device copy of the tile plan of iswm_aspp_fwd (kind 0) / iswm_aspp_bwd (kind 1) for this geometry (built on the host once
    and cached), or None when the fused kernel does not cover it"""
    if not (_ASPP_FUSED and nplanes() == 3):          # bf16x6 only (checked per call: the conv math can change at run time)
        return None
    key = (n, h, w, cin, cout, tuple(ksize), tuple(dil), kind, str(device))
    plan = _ASPP_PLANS.get(key)
    if plan is None:
        lib = _lib.load()
        d = aspp_desc(n, h, w, cin, cout, cin, cout)
        ks, dl = _int_array(ksize), _int_array(dil)
        nb = lib.iswm_aspp_plan_bytes(ctypes.byref(d), len(ksize), ks, dl, kind)
        if nb == 0:
            _ASPP_PLANS[key] = False
            return None
        host = torch.empty((nb,), dtype=torch.uint8)
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        call("iswm_aspp_plan", ctypes.byref(d), len(ksize), ks, dl, kind, ctypes.c_void_p(host.data_ptr()), cus)
        plan = _ASPP_PLANS[key] = host.to(device)
    return None if plan is False else plan


def aspp_fwd(x, ksize, dil, cout, wpks, want_stats):
    """ys[b] = conv(x, w_b) for the parallel branches in one launch; x Planes; wpks[b] packed by iswm_conv2d_pl2_pack_weights
    (kind 0).  Returns (ys, partials | None, tiles) or None when the geometry is not covered."""
    _, n, h, w, cin, ldx, ps = xgeom(x)
    plan = aspp_plan(n, h, w, cin, cout, ksize, dil, 0, x.device)
    if plan is None:
        return None
    nb = len(ksize)
    tiles = (n * h * w + ASPP_TILE_ROWS - 1) // ASPP_TILE_ROWS
    ys = [new_act(n, h, w, cout, x.device) for _ in range(nb)]
    parts = [torch.empty((2, tiles, cout), dtype=torch.float32, device=x.device) for _ in range(nb)] if want_stats else None
    d = aspp_desc(n, h, w, cin, cout, ldx, cout)
    gs = [ConvGeom(x, cout, k, k, 1, dl * (k - 1) // 2, dl) for k, dl in zip(ksize, dil)]
    with _timed_multi("k_conv_pl2t<false>", gs):
        call("iswm_aspp_fwd", ctypes.byref(d), nb, _int_array(ksize), _int_array(dil), _p(plan), _p(x.t), ps, _ptr_array(wpks),
             _ptr_array(ys), _ptr_array(parts) if parts else None, _stream())
    return ys, parts, tiles


def aspp_dgrad(dyc, ksize, dil, cin, cout, wpks, dx=None, accumulate=False, x=None, dws=None):
    """dx (=|+=) sum_b conv^T(dyc[..., b*cout:(b+1)*cout], w_b) in one launch; dyc Planes [N,H,W,nb*cout]; wpks[b] packed by
    iswm_conv2d_pl2_pack_weights (kind 1).  With x (Planes) and dws (OHWI tensors per branch) the same call also runs the
    branches' weight gradients (iswm_aspp_bwd's full form).  Returns dx or None when not covered."""
    _, n, h, w, ctot, ld, ps = xgeom(dyc)
    nb = len(ksize)
    assert ctot >= nb * cout
    plan = aspp_plan(n, h, w, cin, cout, ksize, dil, 1, dyc.device)
    if plan is None:
        return None
    if dx is None:
        assert not accumulate
        dx = new_act(n, h, w, cin, dyc.device)
    d = aspp_desc(n, h, w, cin, cout, geom(dx)[4], cout)
    gs = []
    for k, dl in zip(ksize, dil):
        g = ConvGeom(dx, cout, k, k, 1, dl * (k - 1) // 2, dl)
        g.alg_cin, g.alg_cout = cin, cout
        gs.append(g)
    px, xps, pdw, ws, need = None, 0, None, None, 0
    if dws is not None:
        px, xps = x.t, pgeom(x)[5]
        pdw = _ptr_array(dws)
        lib = _lib.load()
        for k, dl in zip(ksize, dil):
            db = ConvDesc(n, h, w, cin, h, w, cout, k, k, 1, dl * (k - 1) // 2, dl, pgeom(x)[4], ld)
            need = max(need, lib.iswm_conv2d_wgrad_planes_workspace(ctypes.byref(db)))
        ws = torch.empty((max(need, 16) // 4,), dtype=torch.float32, device=dyc.device)
        d = aspp_desc(n, h, w, cin, cout, pgeom(x)[4], cout)
        assert geom(dx)[4] == pgeom(x)[4], "iswm_aspp_bwd takes ONE pitch for x and dx"
    with _timed_multi("k_conv_pl2t<true>", gs):
        call("iswm_aspp_bwd", ctypes.byref(d), nb, _int_array(ksize), _int_array(dil), _p(plan), _p(dyc.t), ps, ld, _ptr_array(wpks),
             _p(dx), int(bool(accumulate)), _p(px), xps, pdw, _p(ws), need, _stream())
    return dx


class _timed_multi:
    """_timed for a launch that computes several conv geometries at once (flops add up)"""

    def __init__(self, name, geoms):
        self.on = KPROF is not None and KPROF.wants(name)
        self.name, self.geoms = name, geoms

    def __enter__(self):
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            g0 = self.geoms[0]
            tag = "n%d %dx%d c%d->%dx%d aspp d%s" % (g0.n, g0.h, g0.w, g0.alg_cin, len(self.geoms), g0.alg_cout,
                                                    "/".join(str(g.dil) for g in self.geoms if g.kh > 1))
            KPROF.add(self.name, self.a, b, sum(g.flops() for g in self.geoms), tag)
        return False


# ---- depthwise conv (groups == channels) -------------------------------------------------------------
def dwconv2d_fwd(x, w, g, bias=None, out=None):
    """x NHWC [N,H,W,C]; w the parameter [Cw,1,KH,KW] (contiguous), Cw <= C"""
    x = as_f32(x)
    ldx = geom(x)[4]
    if out is None:
        out = new_act(g.n, g.ho, g.wo, g.cin, x.device)
    d = g.desc(ldx, geom(out)[4])
    call("iswm_dwconv2d_fwd", ctypes.byref(d), _p(x), _p(w), w.shape[0], _p(bias), _p(out), _stream())
    return out


def dwconv2d_dgrad(dy, w, g, x_like_shape, dx=None, accumulate=False):
    if dx is None:
        assert not accumulate
        dx = new_act(*x_like_shape, dy.device)
    d = g.desc(geom(dx)[4], geom(dy)[4])
    call("iswm_dwconv2d_dgrad", ctypes.byref(d), _p(dy), _p(w), w.shape[0], _p(dx), int(bool(accumulate)), _stream())
    return dx


def dwconv2d_wgrad(x, dy, g, cw, dw=None):
    """dw [Cw,1,KH,KW]"""
    x, dy = as_f32(x), as_f32(dy)
    if dw is None:
        dw = torch.empty((cw, 1, g.kh, g.kw), dtype=torch.float32, device=x.device)
    d = g.desc(geom(x)[4], geom(dy)[4])
    need = _lib.load().iswm_dwconv2d_wgrad_workspace(ctypes.byref(d))
    ws = torch.empty((need // 8,), dtype=torch.float64, device=x.device)
    call("iswm_dwconv2d_wgrad", ctypes.byref(d), _p(x), _p(dy), cw, _p(dw), _p(ws), need, _stream())
    return dw


def rows(t):
    n, h, w, c, ld = geom(t)
    return n * h * w, c, ld


def xrows(t):
    """(pointer tensor, rows, C, ld, ps) of an fp32 or Planes activation"""
    p, n, h, w, c, ld, ps = xgeom(t)
    return p, n * h * w, c, ld, ps


def colstat(x):
    """per-tile column statistics {S_t, M2_t}; returns (partials, tiles, tile_rows)"""
    m, c, ld = rows(x)
    tile_rows = _lib.load().iswm_colstat_tile_rows(m)
    tiles = (m + tile_rows - 1) // tile_rows
    partials = torch.empty((2, tiles, c), dtype=torch.float32, device=x.device)
    call("iswm_colstat", _p(x), m, c, ld, _p(partials), _stream())
    return partials, tiles, tile_rows


def bn_finalize(partials, tiles, count, tile_rows, gamma, beta, running_mean, running_var, momentum, eps=BN_EPS):
    c = partials.shape[2]
    coef = torch.empty((4, c), dtype=torch.float32, device=partials.device)  # scale, beta, mean, invstd
    call("iswm_bn_finalize", _p(partials), tiles, c, count, tile_rows, _p(gamma), _p(beta), _p(running_mean),
         _p(running_var), float(momentum), float(eps), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _stream())
    return coef


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps=BN_EPS):
    c = running_mean.numel()
    coef = torch.empty((4, c), dtype=torch.float32, device=running_mean.device)
    call("iswm_bn_eval_coeffs", c, _p(gamma), _p(beta), _p(running_mean), _p(running_var), float(eps), _p(coef[0]),
         _p(coef[1]), _p(coef[2]), _p(coef[3]), _stream())
    return coef


def _relu_code(relu):
    """activation after a BatchNorm: False / 0 none, True / 1 ReLU, 6 ReLU6"""
    return 6 if (relu == 6 and relu is not True) else int(bool(relu))


def bn_apply(y, coef, relu, residual=None, out=None, planes=False):
    """out = act((y - mean) * scale + beta (+ residual)); `out` (fp32 or Planes, possibly a channel slice) decides the
    output format, else `planes` does"""
    m, c, ldy = rows(y)
    if out is None:
        out = new_planes(*y.shape, y.device) if planes else torch.empty(y.shape, dtype=torch.float32, device=y.device)
    po, mo, co, ldo, pso = xrows(out)
    assert (mo, co) == (m, c)
    pr, ldr, psr = None, 0, 0
    if residual is not None:
        pr, mr, cr, ldr, psr = xrows(residual)
        assert (mr, cr) == (m, c)
    call("iswm_bn_apply_pl", _p(y), m, c, ldy, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(pr), ldr, psr, _relu_code(relu),
         _p(po), ldo, pso, _stream())
    return out


_BN_MASK_FROM_Y = os.environ.get("ISWM_BN_MASKY", "1") != "0"     # tuning switch


def bn_backward(dout, out, y, coef, gamma, relu, training, dgamma, dbeta, want_dres=False, dy=None, dy_planes=False,
                stats=None):
    """Returns (dy, dres|None); writes dgamma / dbeta (length-C fp32 tensors).  `out` (the saved activation, for the
    ReLU pattern) may be Planes; dy is written as Planes when dy_planes (the conv's data / weight gradient kernels
    take it pre-split).  stats: a filled BnStats from the data gradient that produced dout -- the reduction pass is skipped."""
    m, c, ldy = rows(y)
    _, _, ldd = rows(dout)
    po, ldo, pso = None, 0, 0
    if out is not None:
        po, _, _, ldo, pso = xrows(out)
    need = _lib.load().iswm_bn_bwd_workspace(m, c)
    ws = torch.empty((need // 8,), dtype=torch.float64, device=y.device)
    if dy is None:
        dy = new_planes(*y.shape, y.device) if dy_planes else torch.empty(y.shape, dtype=torch.float32, device=y.device)
    pdy, _, _, lddy, psdy = xrows(dy)
    dres = torch.empty(y.shape, dtype=torch.float32, device=y.device) if want_dres else None
    # ReLU without a residual: hand over the forward's scale / shift so the sign pattern is recomputed from y and the
    # saved output is never read (a residual stage's pattern depends on the identity tensor: read `out` there)
    masky = _relu_code(relu) == 1 and not want_dres and _BN_MASK_FROM_Y
    if stats is not None and stats.partials is not None:
        call("iswm_bn_backward_stats_pl", _p(dout), ldd, _p(po), ldo, pso, _p(y), ldy, m, c, _p(coef[2]), _p(coef[3]), _p(gamma),
             _p(coef[0]) if masky else None, _p(coef[1]) if masky else None,
             _relu_code(relu), int(bool(training)), _p(dgamma), _p(dbeta), _p(pdy), lddy, psdy, _p(dres),
             rows(dres)[2] if dres is not None else 0, _p(stats.partials), stats.tiles, _p(ws), need, _stream())
        return dy, dres
    call("iswm_bn_backward_pl", _p(dout), ldd, _p(po), ldo, pso, _p(y), ldy, m, c, _p(coef[2]), _p(coef[3]), _p(gamma),
         _p(coef[0]) if masky else None, _p(coef[1]) if masky else None,
         _relu_code(relu), int(bool(training)), _p(dgamma), _p(dbeta), _p(pdy), lddy, psdy, _p(dres),
         rows(dres)[2] if dres is not None else 0, _p(ws), need, _stream())
    return dy, dres


def bn_apply_classify(y, coef, wc4, bias4):
    """logits [N,H,W,4] = bias4 + wc4 . relu(bn(y)) in one pass over the raw conv output y (csrc/bn_classify.hip); wc4 = the 1x1
    classifier's weight zero-padded to [4, C] (pad_weights), C == 256"""
    m, c, ldy = rows(y)
    out = torch.empty(tuple(y.shape[:3]) + (4,), dtype=torch.float32, device=y.device)
    call("iswm_bn_apply_classify", _p(y), m, c, ldy, _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(wc4), _p(bias4), _p(out), 4, _stream())
    return out


def bn_backward_classify(dlogit, wc4, y, coef, gamma, training, dgamma, dbeta, dy_planes):
    """BatchNorm + ReLU backward of the stage whose activation fed the folded classifier: returns (dy, dwc4) -- dy the gradient of
    the raw conv output (Planes when dy_planes), dwc4 [4, C] the classifier's weight gradient"""
    m, c, ldy = rows(y)
    _, cl, ldl = rows(dlogit)
    assert cl == 4
    need = _lib.load().iswm_bn_classify_bwd_workspace(m, c)
    ws = torch.empty((need // 8,), dtype=torch.float64, device=y.device)
    dy = new_planes(*y.shape, y.device) if dy_planes else torch.empty(y.shape, dtype=torch.float32, device=y.device)
    pdy, _, _, lddy, psdy = xrows(dy)
    dwc4 = torch.empty((4, c), dtype=torch.float32, device=y.device)
    call("iswm_bn_backward_classify", _p(dlogit), ldl, _p(wc4), _p(y), ldy, m, c, _p(coef[2]), _p(coef[3]), _p(gamma), _p(coef[0]),
         _p(coef[1]), int(bool(training)), _p(dgamma), _p(dbeta), _p(dwc4), _p(pdy), lddy, psdy, _p(ws), need, _stream())
    return dy, dwc4


def colsum(x):
    """per-channel sum over all pixels (bias gradient)"""
    x = as_f32(x)
    partials, tiles, _ = colstat(x)
    c = x.shape[3]
    out = torch.empty((2, c), dtype=torch.float32, device=x.device)
    call("iswm_colsum_finalize", _p(partials), tiles, c, _p(out[0]), _p(out[1]), _stream())
    return out[0]


def maxpool_fwd(x, planes=False):
    x = as_f32(x)
    n, h, w, c, ld = geom(x)
    assert ld == c
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    y = new_planes(n, ho, wo, c, x.device) if planes else new_act(n, ho, wo, c, x.device)
    py, _, _, _, _, _, ps = xgeom(y)
    idx = torch.empty((n, ho, wo, c), dtype=torch.uint8, device=x.device)
    call("iswm_maxpool3x3s2_fwd_pl", _p(x), n, h, w, c, _p(py), ps, _p(idx), ho, wo, _stream())
    return y, idx


def maxpool_bwd(dy, idx, in_shape):
    n, h, w, c = in_shape
    _, ho, wo, _, ld = geom(dy)
    assert ld == c
    dx = new_act(n, h, w, c, dy.device)
    call("iswm_maxpool3x3s2_bwd", _p(dy), _p(idx), n, h, w, c, ho, wo, _p(dx), _stream())
    return dx


def gap_fwd(x):
    px, n, h, w, c, ld, ps = xgeom(x)
    y = new_act(n, 1, 1, c, x.device)
    call("iswm_gap_fwd_pl", _p(px), ps, n, h * w, c, ld, _p(y), _stream())
    return y


def gap_bwd(dy, dx, accumulate):
    n, h, w, c, ld = geom(dx)
    call("iswm_gap_bwd", _p(dy), n, h * w, c, _p(dx), ld, int(bool(accumulate)), _stream())
    return dx


def bcast_fwd(v, out):
    po, n, h, w, c, ld, ps = xgeom(out)
    call("iswm_bcast_fwd_pl", _p(as_f32(v)), n, h * w, c, _p(po), ld, ps, _stream())
    return out


def bcast_bwd(dy):
    n, h, w, c, ld = geom(dy)
    dv = new_act(n, 1, 1, c, dy.device)
    call("iswm_bcast_bwd", _p(dy), ld, n, h * w, c, _p(dv), _stream())
    return dv


def bilinear_fwd(x, ho, wo, out=None):
    x = as_f32(x)
    n, hi, wi, c, ldx = geom(x)
    if out is None:
        out = new_act(n, ho, wo, c, x.device)
    po, _, _, _, _, ldy, ps = xgeom(out)
    call("iswm_bilinear_fwd_pl", _p(x), n, hi, wi, c, ldx, _p(po), ps, ho, wo, ldy, _stream())
    return out


def bilinear_bwd(dy, hi, wi):
    n, ho, wo, c, lddy = geom(dy)
    dx = new_act(n, hi, wi, c, dy.device)
    call("iswm_bilinear_bwd", _p(dy), n, hi, wi, c, lddy, ho, wo, _p(dx), c, _stream())
    return dx


def bilinear_to_nchw_fwd(x, c, ho, wo):
    """NHWC low-res logits (first c channels) -> NCHW [N,c,ho,wo]."""
    x = as_f32(x)
    n, hi, wi, cp, ldx = geom(x)
    y = torch.empty((n, c, ho, wo), dtype=torch.float32, device=x.device)
    call("iswm_bilinear_nhwc_to_nchw_fwd", _p(x), n, hi, wi, c, ldx, _p(y), ho, wo, _stream())
    return y


def bilinear_to_nchw_bwd(dy, hi, wi, cp):
    n, c, ho, wo = dy.shape
    assert dy.is_contiguous() and dy.dtype == torch.float32
    dx = new_act(n, hi, wi, cp, dy.device)
    call("iswm_bilinear_nhwc_to_nchw_bwd", _p(dy), n, hi, wi, c, cp, ho, wo, _p(dx), _stream())
    return dx


def nchw_to_nhwc(x, cp=None):
    assert x.dim() == 4 and x.dtype == torch.float32 and x.is_cuda
    x = x.contiguous()
    n, c, h, w = x.shape
    cp = cp or (c + 3) // 4 * 4
    y = new_act(n, h, w, cp, x.device)
    call("iswm_nchw_to_nhwc", _p(x), n, c, h * w, _p(y), cp, _stream())
    return y


def nhwc_to_nchw(x, c=None):
    x = as_f32(x)
    n, h, w, cc, ld = geom(x)
    c = c or cc
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    call("iswm_nhwc_to_nchw", _p(x), n, c, h * w, ld, _p(y), _stream())
    return y


def copy_channels(src, dst):
    src = as_f32(src)
    m, c, lds = rows(src)
    md, cd, ldd = rows(dst)
    assert (m, c) == (md, cd)
    call("iswm_copy_channels", _p(src), lds, _p(dst), ldd, m, c, _stream())
    return dst


def add_inplace(dst, src):
    assert dst.is_contiguous() and src.is_contiguous() and dst.numel() == src.numel()
    call("iswm_add_inplace", _p(dst), _p(src), dst.numel(), _stream())
    return dst


def dropout_fwd(x, p, seed, offset):
    x = as_f32(x)
    assert x.is_contiguous()
    y = torch.empty_like(x)
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    call("iswm_dropout_fwd", _p(x), _p(y), _p(mask), x.numel(), float(p), int(seed), int(offset), _stream())
    return y, mask


def dropout_bwd(dy, mask, p):
    assert dy.is_contiguous()
    dx = torch.empty_like(dy)
    call("iswm_dropout_bwd", _p(dy), _p(mask), _p(dx), dy.numel(), float(p), _stream())
    return dx


MODE_WCE, MODE_FOCAL_MEAN, MODE_FOCAL_SUM = 0, 1, 2


def loss_fwd(logits, labels, weight, ignore_index, alpha, gamma, mode):
    """Returns (loss[1], sums[2], grad_unnorm) -- one pass over the logits."""
    assert logits.dim() == 4 and logits.dtype == torch.float32 and logits.is_cuda
    logits = logits.contiguous()
    b, c, h, w = logits.shape
    assert labels.shape == (b, h, w) and labels.dtype in (torch.uint8, torch.int64)
    labels = labels.contiguous()
    npix = b * h * w
    blocks = _lib.load().iswm_loss_blocks(npix)
    grad = torch.empty_like(logits)
    partials = torch.empty((2, blocks), dtype=torch.float32, device=logits.device)
    out = torch.empty((3,), dtype=torch.float32, device=logits.device)  # sums[2], loss
    call("iswm_loss_fwd", _p(logits), _p(labels), labels.element_size(), b, c, h * w, _p(weight), int(ignore_index),
         float(alpha), float(gamma), int(mode), _p(grad), _p(partials), _stream())
    call("iswm_loss_finalize", _p(partials), blocks, int(mode), npix, _p(out), _p(out[2:]), _stream())
    return out[2:], out[:2], grad


def loss_bwd_scale(grad, sums, upstream, mode, npix):
    call("iswm_loss_bwd_scale", _p(grad), grad.numel(), _p(sums), _p(upstream), int(mode), npix, _stream())
    return grad


def argmax_nchw(logits):
    logits = logits.contiguous()
    b, c, h, w = logits.shape
    out = torch.empty((b, h, w), dtype=torch.int64, device=logits.device)
    call("iswm_argmax_nchw", _p(logits), b, c, h * w, _p(out), _stream())
    return out


def _int_code(t):
    if t.dtype == torch.uint8:
        return 0
    if t.dtype == torch.int64:
        return 1
    raise TypeError("expected a uint8 or int64 tensor, got %s" % t.dtype)


def confusion_matrix(labels, preds, n_classes, hist=None):
    """hist[n_classes, n_classes] (int64, device) += bincount(n_classes*label + pred) over valid labels"""
    labels, preds = labels.contiguous(), preds.contiguous()
    if labels.numel() != preds.numel():
        raise ValueError("labels and predictions differ in size: %d vs %d" % (labels.numel(), preds.numel()))
    if hist is None:
        hist = torch.zeros((n_classes, n_classes), dtype=torch.int64, device=labels.device)
    call("iswm_confusion_matrix", _p(labels), _int_code(labels), _p(preds), _int_code(preds), labels.numel(), n_classes,
         _p(hist), _stream())
    return hist


def confusion_matrix_logits(labels, logits, n_classes, hist=None):
    """the same with pred = logits.max(1)[1] computed on the fly (logits NCHW fp32)"""
    labels, logits = labels.contiguous(), logits.contiguous()
    b, c, h, w = logits.shape
    if labels.numel() != b * h * w:
        raise ValueError("labels %s do not match logits %s" % (tuple(labels.shape), tuple(logits.shape)))
    if hist is None:
        hist = torch.zeros((n_classes, n_classes), dtype=torch.int64, device=labels.device)
    call("iswm_confusion_matrix_logits", _p(labels), _int_code(labels), _p(logits), b, c, h * w, n_classes, _p(hist),
         _stream())
    return hist


def sgd_step(p, g, buf, lr_dev, momentum, weight_decay, nesterov):
    call("iswm_sgd_step", _p(p), _p(g), _p(buf), p.numel(), _p(lr_dev), float(momentum), float(weight_decay),
         int(bool(nesterov)), _stream())


def adam_step(p, g, m, v, hyper_dev, beta1, beta2, eps, weight_decay, decoupled):
    call("iswm_adam_step", _p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper_dev), float(beta1), float(beta2),
         float(eps), float(weight_decay), int(bool(decoupled)), _stream())
