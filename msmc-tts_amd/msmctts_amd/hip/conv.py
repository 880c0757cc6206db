"""Channels-last implicit-GEMM convolution ops over csrc/conv.hip (C ABI: msmc_conv_* in include/msmc_hip.h).

Geometry helpers turn a PyTorch-style convolution (kernel, stride, dilation, padding, zero/reflect) into the
lattice + tap-table descriptors the gather kernel consumes, for
  * the forward of a (strided / dilated) convolution,
  * its data gradient (one launch per stride phase),
  * the forward of a transposed convolution (== the data gradient of a strided convolution),
  * the data gradient of a transposed convolution (== a strided convolution).
Activations are channels-last ``[B, H, W, C]`` tensors (1-D signals use H = 1); dtype float32 or bfloat16.
Weights are "slices" ``[n_taps, C_out, C_in]`` in the activation dtype.
"""
import ctypes
import os

import torch

from . import lib

_DT = {torch.float32: 0, torch.bfloat16: 1}



class Geometry(object):
    """Forward geometry of conv(kernel (kh,kw), stride, dilation, padding) on an (Hin, Win) image."""

    def __init__(self, Hin, Win, kernel, stride=(1, 1), dilation=(1, 1), padding=(0, 0), reflect=False):
        self.Hin, self.Win = Hin, Win
        self.kh, self.kw = kernel
        self.sy, self.sx = stride
        self.dy, self.dx = dilation
        self.py, self.px = padding
        self.reflect = reflect
        self.Hout = (Hin + 2 * self.py - self.dy * (self.kh - 1) - 1) // self.sy + 1
        self.Wout = (Win + 2 * self.px - self.dx * (self.kw - 1) - 1) // self.sx + 1

        self.ntaps = self.kh * self.kw
        # launch tables, built once per geometry
        self.fwd_taps = tuple((ky * self.dy, kx * self.dx, ky * self.kw + kx)
                              for ky in range(self.kh) for kx in range(self.kw))
        self.fwd_lattice = (self.Hout, self.Wout, 0, 1, 0, 1, self.sy, self.sx, -self.py, -self.px)
        self._dgrad_plan = None
        self.plans = {}

    def dgrad_plan(self):
        """[(lattice, taps)] per stride phase of the data gradient (None taps: no kernel tap reaches the phase)."""
        if self._dgrad_plan is None:
            if self.reflect:
                Hx, Wx, py, px = self.Hin + 2 * self.py, self.Win + 2 * self.px, 0, 0
            else:
                Hx, Wx, py, px = self.Hin, self.Win, self.py, self.px
            plan = []
            for ry, ny, ty in _phases(Hx, self.kh, self.sy, self.dy, py):
                for rx, nx, tx in _phases(Wx, self.kw, self.sx, self.dx, px):
                    if ny == 0 or nx == 0:
                        continue
                    taps = tuple((oy, ox, ky * self.kw + kx) for oy, ky in ty for ox, kx in tx)
                    plan.append(((ny, nx, ry, self.sy, rx, self.sx, 1, 1, 0, 0), taps, ry, rx))
            self._dgrad_plan = (Hx, Wx, plan)
        return self._dgrad_plan


_PLANS = {}

# Per-layer-shape kernel selection: the first launch of a descriptor on the GPU times the candidate kernels
# (msmc_conv_desc.variant / .split_shift) once and keeps the fastest -- tile heuristics cannot see L2 / LDS effects
# that differ by 2x between layers of equal arithmetic.  Off inside hipGraph capture and on the interpreter.
AUTOTUNE = os.environ.get('MSMC_AUTOTUNE', '1') != '0'
# (round 6: 22, 23, 30 and 43 left the candidates and the library -- never chosen by the tuner over configurations 1-5)
_GATHER_CANDIDATES = tuple((v, 0) for v in (1, 2, 3, 4, 5, 8, 9, 16, 17, 18, 19, 20, 21, 24, 25, 26, 27, 28, 29, 31, 32, 34, 35,
                                             40, 41, 42, 44, 45, 46, 47, 50, 56, 59, 60, 61, 63))
# (round 6: (3, 0), (3, -2), (3, 1) and the first generation (1, 0) left the candidates: never chosen over configurations 1-5;
#  the third generation stays reachable through its one chosen split and as wgrad5's fallback)
_WGRAD_CANDIDATES = ((4, 0), (4, -1), (9, 0), (9, -1), (7, 0), (8, 0), (3, -1), (2, 0), (2, -1), (2, 1))
TUNED = {}                                    # (kind, shape signature) -> (variant, split_shift, {candidate: ms})
TUNE_CACHE = os.environ.get('MSMC_TUNE_CACHE', os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                            'tuned_gfx950.json'))


def load_tuned(path=None):
    """Choices measured earlier on this GPU model (tools/tune_bench_shapes.py writes them): shapes found here skip
    the timing launches; unseen shapes are still tuned on first use.  A missing or unreadable file is ignored.  Two files:
    the DECISIONS (``tuned_gfx950.json``: one row per line, sorted by key -- a re-tune shows up in ``git diff`` as exactly the
    decisions that changed) and, beside it, the candidates' timings the decisions were taken from
    (``tuned_gfx950_timings.json``: optional, merged by ``tools/tune_bench_shapes.py`` when it re-times one kernel family)."""
    import json
    path = path or TUNE_CACHE
    try:
        with open(path) as f:
            rows = json.load(f)['choices']
    except Exception:
        return
    timings = {}
    try:
        with open(_timings_path(path)) as f:
            timings = {repr(_thaw(r['key'])): r['ms'] for r in json.load(f)['timings']}
    except Exception:
        pass
    for row in rows:
        try:
            key = tuple(_thaw(row['key']))
            ms = row.get('ms', timings.get(repr(key), []))          # ('ms' inside the row: the single-file format of rounds 1-4)
            TUNED[key] = (int(row['variant']), int(row['split_shift']), {tuple(_thaw(k)): v for k, v in ms})
        except Exception:
            continue


def _timings_path(path):
    root, ext = os.path.splitext(path)
    return root + '_timings' + ext


def save_tuned(path=None):
    import json
    path = path or TUNE_CACHE
    items = sorted(TUNED.items(), key=lambda kv: repr(kv[0]))
    enc = lambda o: json.dumps(o, separators=(',', ':'))
    head = dict(device='gfx950', note='per-layer-shape kernel choices (msmc_conv_desc.variant / split_shift) timed on MI355X; '
                'regenerate with tools/tune_bench_shapes.py; one decision per line, sorted by key; candidate timings in '
                + os.path.basename(_timings_path(path)))
    with open(path, 'w') as f:
        f.write('{"device":%s,"note":%s,"choices":[\n' % (enc(head['device']), enc(head['note'])))
        f.write(',\n'.join(enc(dict(key=_freeze(k), variant=v[0], split_shift=v[1])) for k, v in items))
        f.write('\n]}\n')
    with open(_timings_path(path), 'w') as f:
        f.write('{"device":"gfx950","note":"milliseconds per candidate (variant, split_shift) behind the decisions of %s","timings":[\n'
                % os.path.basename(path))
        f.write(',\n'.join(enc(dict(key=_freeze(k), ms=[[_freeze(c), round(t, 5)] for c, t in sorted(v[2].items(), key=lambda ct: repr(ct[0]))]))
                            for k, v in items))
        f.write('\n]}\n')


def _freeze(x):
    return [_freeze(v) for v in x] if isinstance(x, (tuple, list)) else x


def _thaw(x):
    return tuple(_thaw(v) for v in x) if isinstance(x, list) else x


load_tuned()


def _signature(desc):
    return (desc.dtype, desc.B, desc.Hin, desc.Win, desc.Cin, desc.Hout, desc.Wout, desc.Cout, desc.QH, desc.QW,
            desc.osy, desc.osx, desc.isy, desc.isx, desc.ntaps, tuple(desc.tap_dy[:desc.ntaps]),
            tuple(desc.tap_dx[:desc.ntaps]), desc.pad_mode, bool(desc.res), bool(desc.res2), bool(desc.mask_src))


# Variable-length batches: every new padded length T is a new signature.  Timing launches are a per-process BUDGET of
# new shapes (MSMC_TUNE_BUDGET); a shape outside the cache first borrows the choice of the nearest tuned shape of its
# CLASS (everything but batch, spatial extents and epilogue operands: channels, taps, strides, dilation, padding rule) --
# kernel preferences follow the channel / tap configuration far more than the length -- and is timed only when its class
# has no entry and budget is left.  Ranks of a data-parallel job therefore stop issuing timing launches after the same
# bounded number of shapes instead of stalling each other at the collectives whenever one of them meets a new length.
TUNE_BUDGET = [int(os.environ.get('MSMC_TUNE_BUDGET', '256'))]
TUNE_BORROW = os.environ.get('MSMC_TUNE_BORROW', '1') != '0'       # 0: time every shape (tools/tune_bench_shapes.py)
_CLASS_INDEX = {}                             # (kind, class) -> [(pixels, signature)]
_CLASS_INDEXED = [0]


def _class_of(sig):
    """signature (kind, dtype, B, Hin, Win, Cin, Hout, Wout, Cout, QH, QW, ...) -> (class key, output pixels)"""
    return (sig[0], sig[1], sig[5], sig[8]) + tuple(sig[11:-3]), sig[2] * sig[9] * sig[10]


def _nearest_tuned(sig):
    if not TUNE_BORROW:
        return None
    if _CLASS_INDEXED[0] != len(TUNED):      # (re)index lazily: TUNED only grows
        _CLASS_INDEX.clear()
        for k in TUNED:
            if k[0] not in ('gather', 'wgrad') or len(k) < 12:     # (grouping decisions are keyed by member lists)
                continue
            c, px = _class_of(k)
            _CLASS_INDEX.setdefault(c, []).append((px, k))
        _CLASS_INDEXED[0] = len(TUNED)
    if sig[0] not in ('gather', 'wgrad') or len(sig) < 12:
        return None
    c, px = _class_of(sig)
    rows = _CLASS_INDEX.get(c)
    if not rows:
        return None
    import math
    k = min(rows, key=lambda r: abs(math.log(max(1, r[0])) - math.log(max(1, px))))[1]
    return TUNED[k]


def _bounded(cache, limit=4096):
    """plan / geometry caches are keyed by shape: a long run over variable-length batches must not grow them for ever"""
    if len(cache) > limit:
        cache.clear()


def _tune(kind, desc, launch, candidates):
    """Time ``launch()`` under every candidate (variant, split_shift); leave the fastest in the descriptor."""
    desc._tuned = True
    if lib._host_pointers_ok:
        return
    sig = (kind,) + _signature(desc)
    hit = TUNED.get(sig)
    if hit is None:
        near = _nearest_tuned(sig)
        if near is not None:        # borrowed from the nearest tuned shape of the same class (validated below)
            keep = (desc.variant, desc.split_shift)
            desc.variant, desc.split_shift = near[0], near[1]
            if launch() == 0:
                return
            desc.variant, desc.split_shift = keep
    if hit is None and (not AUTOTUNE or TUNE_BUDGET[0] <= 0 or torch.cuda.is_current_stream_capturing()):
        return                      # no timing launches now: library heuristic for this shape
    if hit is None:
        TUNE_BUDGET[0] -= 1
        times = {}
        for variant, shift in candidates:
            desc.variant, desc.split_shift = variant, shift
            if launch() != 0:
                continue
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                launch()
            e.record()
            e.synchronize()
            times[(variant, shift)] = s.elapsed_time(e) / 3.0
        best = min(times, key=times.get) if times else (0, 0)
        hit = TUNED[sig] = (best[0], best[1], times)
    desc.variant, desc.split_shift = hit[0], hit[1]


def _gather(desc, stream, what):
    fn = lib.get().msmc_conv_gather
    if not getattr(desc, '_tuned', False):
        _tune('gather', desc, lambda: fn(ctypes.byref(desc), stream), _GATHER_CANDIDATES)
    lib.check(fn(ctypes.byref(desc), stream), what)


# -- deferred second stage of the no-atomics weight gradients ------------------------------------------------------------
# A backward pass issues ~40 weight-gradient calls per network whose second stage (the fixed-order sum of the pixel
# splits' partial results) used to be one or two launches EACH (77 launches, 1 ms per training step).  With a
# ``DeferredReduce`` armed (``DEFER_TO``; hip/convnet.py arms the ConvBank's around its backward launches) the calls
# run their first stage into an arena that lives until the end of the backward pass, and ``flush`` adds everything up in
# ceil(records / 16) launches (msmc_conv_wgrad_reduce_pending) -- same sums, same order, bit-identical results.
DEFER = True                 # (module switch for A/B runs from tools/: False reduces every weight gradient at once)
DEFER_TO = None             # the DeferredReduce the running weight-gradient calls record into (None: reduce at once)


class DeferredReduce(object):
    CAPACITY = 1024                       # records per backward pass
    CHUNK = 256 << 20                     # largest arena growth step (bytes); chunks persist and are reused every pass
    FIRST = 16 << 20                      # first chunk of a bank (small banks -- an FFT stack, the quantiser -- never need more)

    def __init__(self):
        self.records = (lib.WgPending * self.CAPACITY)()
        self.n = 0
        self.chunks = []                  # [tensor, bytes used]

    def take(self, device, nbytes):
        """``nbytes`` of partial-result storage that stays untouched until ``flush``"""
        nbytes = (nbytes + 255) & ~255
        for c in self.chunks:
            if c[0].device == device and c[0].numel() * 4 - c[1] >= nbytes:
                p = c[0].data_ptr() + c[1]
                c[1] += nbytes
                return p
        # grow geometrically from FIRST to CHUNK: a bank pins what its passes need, not 256 MiB each (round-3 advice)
        have = sum(c[0].numel() * 4 for c in self.chunks)
        step = min(self.CHUNK, max(self.FIRST, have))
        t = torch.empty(max(nbytes, step) // 4, dtype=torch.float32, device=device)
        self.chunks.append([t, nbytes])
        return t.data_ptr()

    def begin(self):
        room = self.CAPACITY - self.n
        lib.get().msmc_conv_wgrad_defer_begin(
            ctypes.cast(ctypes.byref(self.records, self.n * ctypes.sizeof(lib.WgPending)), ctypes.POINTER(lib.WgPending)), room)

    def end(self):
        self.n += lib.get().msmc_conv_wgrad_defer_end()

    def flush(self, stream):
        """the merged second stage of everything recorded since the last flush (on ``stream``: the caller has ordered it
        after the launches that wrote the partial results); the arena is free again afterwards"""
        if self.n:
            lib.check(lib.get().msmc_conv_wgrad_reduce_pending(self.records, self.n, stream), 'msmc_conv_wgrad_reduce_pending')
        self.reset()

    def reset(self):
        """forget the recorded second stages and hand the arena back (also the way out of a backward pass that raised)"""
        self.n = 0
        for c in self.chunks:
            c[1] = 0


_WORKSPACES = {}            # (device, stream) -> fp32 scratch of the third-generation weight gradient (split partials)


def _workspace(device, stream, nbytes):
    """stream-private scratch, grown geometrically; launches on one stream are ordered, so one buffer serves them all"""
    if nbytes <= 0:
        return None, 0
    key = (device.index, stream.value)
    t = _WORKSPACES.get(key)
    if t is None or t.numel() * 4 < nbytes:
        grow = max(nbytes, 2 * (t.numel() * 4 if t is not None else 0), 8 << 20)
        t = _WORKSPACES[key] = torch.empty((grow + 3) // 4, dtype=torch.float32, device=device)
    return t.data_ptr(), t.numel() * 4


def _wgrad(desc, g_ptr, dw, db, stream, what, seen=None):
    L = lib.get()

    def fn(dref, gp, dwp, dbp, st, defer=None):
        need = L.msmc_conv_wgrad_workspace(dref, gp)
        if defer is not None and need and defer.n < defer.CAPACITY:
            wsp = defer.take(dw.device, need)
            defer.begin()
            try:
                return L.msmc_conv_wgrad_ws(dref, gp, dwp, dbp, wsp, need, st)
            finally:
                defer.end()
        wsp, wsb = _workspace(dw.device, st, need) if need else (None, 0)
        return L.msmc_conv_wgrad_ws(dref, gp, dwp, dbp, wsp, wsb, st)
    dbp = db.data_ptr() if db is not None else None
    defer = DEFER_TO if DEFER else None
    if not getattr(desc, '_tuned', False):
        cached = TUNED.get(('wgrad',) + _signature(desc)) if not lib._host_pointers_ok else None
        if cached is None and not lib._host_pointers_ok:
            cached = _nearest_tuned(('wgrad',) + _signature(desc))
            if cached is not None and cached[0] >= 3 and desc.dtype != 1:
                cached = None                 # (the third and fourth generations are bf16-only)
        if cached is not None:
            desc._borrowed = ('wgrad',) + _signature(desc) not in TUNED
            desc.variant, desc.split_shift, desc._tuned = cached[0], cached[1], True
        elif AUTOTUNE and TUNE_BUDGET[0] > 0 and not lib._host_pointers_ok and not torch.cuda.is_current_stream_capturing():
            R = max(1, desc.dw_copies)                       # candidates accumulate into scratch, not into dW
            sdw = torch.zeros(R * desc.ntaps * desc.Cout * desc.Cin, dtype=torch.float32, device=dw.device)
            sdb = torch.zeros(R * desc.Cout, dtype=torch.float32, device=dw.device) if db is not None else None
            sdbp = sdb.data_ptr() if sdb is not None else None
            _tune('wgrad', desc, lambda: fn(ctypes.byref(desc), g_ptr, sdw.data_ptr(), sdbp, stream), _WGRAD_CANDIDATES)
        else:
            desc._tuned = True
    rc = fn(ctypes.byref(desc), g_ptr, dw.data_ptr(), dbp, stream, defer)
    if rc != 0 and getattr(desc, '_borrowed', False):        # a neighbour's choice this shape cannot run: library heuristic
        desc.variant, desc.split_shift, desc._borrowed = 0, 0, False
        rc = fn(ctypes.byref(desc), g_ptr, dw.data_ptr(), dbp, stream, defer)
    if seen is not None:
        seen.add(int(desc.variant))           # (ConvBank: layers whose weight gradients never use atomics need no privatised copies)
    if rc != 0:
        raise RuntimeError('%s failed with code %d (dtype %d variant %d split_shift %d copies %d, x %dx%dx%dx%d -> %dx%dx%d, '
                           '%d taps)' % (what, rc, desc.dtype, desc.variant, desc.split_shift, desc.dw_copies, desc.B, desc.Hin,
                                         desc.Win, desc.Cin, desc.Hout, desc.Wout, desc.Cout, desc.ntaps))



def _ptr(t):
    """data pointer of a tensor the caller has validated (dtype, contiguity); GPU-only unless the kernel
    interpreter is bound."""
    if t is None:
        return None
    if not (t.is_cuda or lib._host_pointers_ok):
        raise RuntimeError('msmc HIP ops run on the GPU only (got a %s tensor); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise ValueError('msmc HIP ops take contiguous tensors')
    if t.data_ptr() % 16:
        raise ValueError('msmc HIP ops take 16-byte aligned operands (got an offset view)')
    return t.data_ptr()


def _fill(desc_unused, x, w, out, B, Hin, Win, Cin, Hout, Wout, Cout, lattice, taps, pad_mode, bias=None, mask_src=None,
          res=None, res2=None, in_slope=1.0, mask_slope=1.0, out_div=1.0, out_slope=1.0, kind='gather'):
    """Descriptor for one launch.  The geometry part is built once per distinct (shape, lattice, taps, flags) and
    cached -- per call only the seven pointers change (the step issues ~2000 convolution launches, so the
    host cost of a launch matters as much as its GPU time)."""
    # ``kind``: a weight gradient and a data gradient of one transposed convolution share every geometry field, but
    # their descriptors carry DIFFERENT kernel choices (desc.variant means another thing to msmc_conv_wgrad)
    key = (kind, x.dtype, B, Hin, Win, Cin, Hout, Wout, Cout, lattice, tuple(taps), pad_mode, in_slope, mask_slope, out_div,
           out_slope)
    desc = _PLANS.get(key)
    if desc is None:
        _bounded(_PLANS)
        desc = _PLANS[key] = _build_desc(x.dtype, B, Hin, Win, Cin, Hout, Wout, Cout, lattice, taps, pad_mode,
                                         in_slope, mask_slope, out_div, out_slope)
    desc.x, desc.w, desc.out = _ptr(x), _ptr(w), _ptr(out)
    desc.bias, desc.mask_src, desc.res, desc.res2 = _ptr(bias), _ptr(mask_src), _ptr(res), _ptr(res2)
    return desc


def _build_desc(dtype, B, Hin, Win, Cin, Hout, Wout, Cout, lattice, taps, pad_mode, in_slope, mask_slope, out_div,
                out_slope):
    desc = lib.ConvDesc()
    desc.dtype = _DT[dtype]
    desc.B, desc.Hin, desc.Win, desc.Cin = B, Hin, Win, Cin
    desc.Hout, desc.Wout, desc.Cout = Hout, Wout, Cout
    (desc.QH, desc.QW, desc.oy0, desc.osy, desc.ox0, desc.osx, desc.isy, desc.isx, desc.iy0, desc.ix0) = lattice
    assert 0 < len(taps) <= lib.MAX_TAPS, len(taps)
    desc.ntaps = len(taps)
    for t, (dy, dx, ws) in enumerate(taps):
        desc.tap_dy[t], desc.tap_dx[t], desc.tap_w[t] = dy, dx, ws
    desc.pad_mode = pad_mode
    desc.in_slope, desc.mask_slope, desc.out_div = float(in_slope), float(mask_slope), float(out_div)
    desc.out_slope = float(out_slope)
    return desc


def _check(x, w, *others):
    assert x.dtype in _DT and w.dtype == x.dtype, (x.dtype, w.dtype)
    for t in others:
        assert t is None or (t.dtype == x.dtype and t.is_contiguous()), 'epilogue operands share the activation dtype'


def _dev_ok(t):
    if not (t.is_cuda or lib._host_pointers_ok):
        raise RuntimeError('msmc HIP ops run on the GPU only (got a %s tensor); there is no CPU path' % t.device)
    if not t.is_contiguous():
        raise ValueError('msmc HIP ops take contiguous tensors')
    if t.data_ptr() % 16:
        raise ValueError('msmc HIP ops take 16-byte aligned operands (got an offset view)')


def _opt_ptr(t, like):
    if t is None:
        return None
    if t.dtype != like.dtype or not t.is_contiguous() or t.data_ptr() % 16:
        raise ValueError('epilogue operands share the activation dtype, are contiguous and 16-byte aligned')
    return t.data_ptr()


def _forward_desc(x, w, geom, bias=None, in_slope=1.0, res=None, res2=None, out_div=1.0, out_slope=1.0):
    """descriptor (pointers filled in) and output tensor of one forward convolution; nothing is launched"""
    key = (x.dtype, x.shape[0], w.shape[1], w.shape[2], in_slope, out_div, out_slope)
    plan = geom.plans.get(key)
    if plan is None:
        _check(x, w, res, res2)
        B, Hin, Win, Cin = x.shape
        T, Cout, _ = w.shape
        assert (Hin, Win) == (geom.Hin, geom.Win) and T == geom.ntaps and w.shape[2] == Cin
        desc = _build_desc(x.dtype, B, Hin, Win, Cin, geom.Hout, geom.Wout, Cout, geom.fwd_lattice, geom.fwd_taps,
                           1 if geom.reflect else 0, in_slope, 1.0, out_div, out_slope)
        plan = geom.plans[key] = (desc, (B, geom.Hout, geom.Wout, Cout))
    desc, oshape = plan
    _dev_ok(x)
    out = torch.empty(oshape, dtype=x.dtype, device=x.device)
    desc.x, desc.w, desc.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    desc.bias = bias.data_ptr() if bias is not None else None
    desc.res, desc.res2, desc.mask_src = _opt_ptr(res, x), _opt_ptr(res2, x), None
    return desc, out


def conv_forward(x, w, geom, bias=None, in_slope=1.0, res=None, res2=None, out_div=1.0, out_slope=1.0):
    """x [B,Hin,Win,Cin] -> [B,Hout,Wout,Cout];  w [kh*kw, Cout, Cin]."""
    desc, out = _forward_desc(x, w, geom, bias, in_slope, res, res2, out_div, out_slope)
    _gather(desc, lib.stream(x), 'msmc_conv_gather')
    return out


_SPLIT_CANDIDATES = ((36, 0), (37, 0))


def const_gemm_split(x, wimg, cout):
    """x [B, 1, T, Cin] fp32 times a CONSTANT matrix given as its pre-split bf16 image (hip/spectral.py split_image:
    [cout][chunks of 32 values][hi 32 | lo 32]) -> [B, 1, T, cout] fp32: three bf16 matrix-core products with fp32
    accumulation (csrc/gemm1.inc conv_gemm1s_kernel, variants 36 / 37 -- 128- or 64-row tiles, timed once per shape)."""
    B, _, T, Cin = x.shape
    assert x.dtype == torch.float32 and wimg.dtype == torch.bfloat16 and wimg.shape[0] == cout and wimg.is_contiguous()
    assert wimg.shape[1] * 32 >= Cin and wimg.shape[2] == 64, (tuple(wimg.shape), Cin)
    key = ('g1s', B, T, Cin, cout)
    desc = _PLANS.get(key)
    if desc is None:
        _bounded(_PLANS)
        desc = _PLANS[key] = _build_desc(torch.float32, B, 1, T, Cin, 1, T, cout, (1, T, 0, 1, 0, 1, 1, 1, 0, 0), ((0, 0, 0),), 0,
                                         1.0, 1.0, 1.0, 1.0)
        desc.variant = 37 if ((B * T + 127) // 128) * ((cout + 127) // 128) < 192 else 36
    _dev_ok(x)
    _dev_ok(wimg)
    out = torch.empty((B, 1, T, cout), dtype=torch.float32, device=x.device)
    desc.x, desc.w, desc.out = x.data_ptr(), wimg.data_ptr(), out.data_ptr()
    desc.bias = desc.res = desc.res2 = desc.mask_src = None
    stream = lib.stream(x)
    fn = lib.get().msmc_conv_gather
    if not getattr(desc, '_tuned', False):
        _tune('gather-split', desc, lambda: fn(ctypes.byref(desc), stream), _SPLIT_CANDIDATES)
    lib.check(fn(ctypes.byref(desc), stream), 'msmc_conv_gather(split constant GEMM)')
    return out


def const_gemm_split_group(xs, wimgs, couts):
    """``const_gemm_split`` of several independent (x, matrix image) pairs as ONE grouped call: members of one tile width share a
    grid (csrc/gemm1.inc conv_gemm1s_group_kernel) -- the five resolution front-ends' DFT (or filter-bank) GEMMs in one launch"""
    assert 0 < len(xs) <= 16
    snaps, outs = [], []
    stream = lib.stream(xs[0])
    for x, wimg, cout in zip(xs, wimgs, couts):
        B, _, T, Cin = x.shape
        assert x.dtype == torch.float32 and wimg.dtype == torch.bfloat16 and wimg.shape[0] == cout and wimg.is_contiguous()
        assert wimg.shape[1] * 32 >= Cin and wimg.shape[2] == 64, (tuple(wimg.shape), Cin)
        key = ('g1s', B, T, Cin, cout)
        desc = _PLANS.get(key)
        if desc is None:
            _bounded(_PLANS)
            desc = _PLANS[key] = _build_desc(torch.float32, B, 1, T, Cin, 1, T, cout, (1, T, 0, 1, 0, 1, 1, 1, 0, 0), ((0, 0, 0),), 0,
                                             1.0, 1.0, 1.0, 1.0)
            desc.variant = 37 if ((B * T + 127) // 128) * ((cout + 127) // 128) < 192 else 36
        _dev_ok(x)
        _dev_ok(wimg)
        out = torch.empty((B, 1, T, cout), dtype=torch.float32, device=x.device)
        desc.x, desc.w, desc.out = x.data_ptr(), wimg.data_ptr(), out.data_ptr()
        desc.bias = desc.res = desc.res2 = desc.mask_src = None
        if not getattr(desc, '_tuned', False):
            fn = lib.get().msmc_conv_gather
            _tune('gather-split', desc, lambda: fn(ctypes.byref(desc), stream), _SPLIT_CANDIDATES)
        snaps.append(lib.ConvDesc.from_buffer_copy(desc))
        outs.append(out)
    arr = (lib.ConvDesc * len(snaps))(*snaps)
    lib.check(lib.get().msmc_conv_gather_group(arr, len(snaps), stream), 'msmc_conv_gather_group(split constant GEMMs)')
    return outs


def _snapshot(desc, stream):
    """by-value copy of a (tuned) descriptor: grouped calls may meet the same cached descriptor twice"""
    if not getattr(desc, '_tuned', False):
        fn = lib.get().msmc_conv_gather
        _tune('gather', desc, lambda: fn(ctypes.byref(desc), stream), _GATHER_CANDIDATES)
    return lib.ConvDesc.from_buffer_copy(desc)


_GROUP_CODES = {'single': 0, 'group': 1, 'group4': 2, 'uniform': 3}
# grouped forward / data-gradient calls whose members chose different kernel families become several launches; the tuner
# also times the call with ONE variant imposed on every member (where all of them accept it): a single grid
_UNIFORM_CANDIDATES = (2, 3, 4, 5, 8, 9, 16, 17, 20, 21, 24, 25, 26, 27, 28, 29, 31, 40, 41, 42, 44, 45, 46, 47, 50, 56, 59, 60, 61, 63)


def _group_choice(kind, snaps, grouped_fn, single_fn, group4_fn=None, uniform_fn=None):
    """1: issue the members as one grouped call, 0: one by one, 2: grouped with the fourth-generation weight-gradient
    members on grids of their own (``group4_fn``, weight gradients only), (3, v): grouped with variant v imposed on every
    member (``uniform_fn(v)`` returns the launcher or None when a member refuses v; forward / data gradients only).
    Timed once per member-shape combination (grouping fills the chip for small grids but imposes one kernel
    instantiation on all members).  Returns (code, variant)."""
    if len(snaps) == 1:
        return 0, 0
    if lib._host_pointers_ok:
        return 1, 0
    sig = (kind,) + tuple(_signature(d) + (d.variant, d.split_shift) for d in snaps)
    hit = TUNED.get(sig)
    if hit is None and (not AUTOTUNE or torch.cuda.is_current_stream_capturing()):
        return 1, 0
    if hit is None:
        times = {}
        options = [(('group', 0), grouped_fn), (('single', 0), single_fn), (('group4', 0), group4_fn)]
        if uniform_fn is not None and len(set(d.variant for d in snaps)) > 1:
            options += [(('uniform', v), uniform_fn(v)) for v in _UNIFORM_CANDIDATES]
        for name, fn in options:
            if fn is None:
                continue
            fn()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                fn()
            e.record()
            e.synchronize()
            times[name] = s.elapsed_time(e) / 3.0
        best = min(times, key=times.get)
        hit = TUNED[sig] = (_GROUP_CODES[best[0]], best[1], times)
    return hit[0], hit[1]


# Measured and NOT kept (round 6, profiles/r06_group_fork_ab.txt): a grouped forward / data-gradient call whose members chose
# different kernel families is several grids issued back to back, and so is a call the tuner keeps as single launches --
# independent convolutions, each alone on the chip.  Issuing those grids on side streams (parallel branches of the captured
# step, joined before the call returns) made the step SLOWER, 15.57 -> 15.78 ms with only the generator's calls forked: a
# dependency that crosses streams costs 5-15 us in a replayed hipGraph (tools/step_timeline.sh: the gap in front of every
# kernel that waits for another branch) against ~0 between two kernels of one stream, so a fork + join around 15-30 us kernels
# loses what the overlap wins.  From INSIDE a side branch (the discriminator's families) such a fork crashes hipStreamEndCapture
# on this runtime.  Branches pay off for CHAINS of launches (one fork and one join per chain), not for single launches.
GROUP_PHASES = os.environ.get('MSMC_GROUP_PHASES', '1') != '0'            # stride phases of a transposed convolution as one grouped call


def _gather_group(snaps, stream, what, device=None):
    """independent launches issued together (msmc_conv_gather_group) when that is the faster way for these shapes"""
    L = lib.get()

    def single():
        for d in snaps:
            lib.check(L.msmc_conv_gather(ctypes.byref(d), stream), what)

    if len(snaps) == 1:
        return single()
    arr = (lib.ConvDesc * len(snaps))(*snaps)

    def grouped():
        lib.check(L.msmc_conv_gather_group(arr, len(snaps), stream), what)

    def uniform(v):
        """launcher of the call with variant v on every member, None when the library refuses it for one of them"""
        forced = (lib.ConvDesc * len(snaps))(*[lib.ConvDesc.from_buffer_copy(d) for d in snaps])
        for d in forced:
            d.variant, d.split_shift = v, 0
        if L.msmc_conv_gather_group(forced, len(snaps), stream) != 0:
            return None
        return lambda: lib.check(L.msmc_conv_gather_group(forced, len(snaps), stream), what)

    code, v = _group_choice('gather-group', snaps, grouped, single, uniform_fn=uniform)
    if code == 3:
        fn = uniform(v)             # (validating launch included: the outputs are simply written twice)
        if fn is not None:
            return
        code = 1
    (grouped if code else single)()


def conv_forward_group(items):
    """``items``: list of dicts with the arguments of ``conv_forward`` -- independent convolutions (the parallel
    ResBlocks of a generator stage, one layer of several sub-discriminators) issued as one grouped launch where their
    kernel choices coincide.  Returns the outputs in order."""
    stream = lib.stream(items[0]['x'])
    snaps, outs = [], []
    for it in items:
        desc, out = _forward_desc(**it)
        snaps.append(_snapshot(desc, stream))
        outs.append(out)
    _gather_group(snaps, stream, 'msmc_conv_gather_group', items[0]['x'].device)
    return outs


def _phases(size, k, stride, dil, pad):
    """Per output phase r (0..stride-1) of a data gradient / transposed convolution along one axis:
    (r, n_points, [(offset, tap_index)]) with source index = q + offset for point r + q*stride."""
    out = []
    for r in range(stride):
        n = (size - r + stride - 1) // stride if size > r else 0
        taps = []
        for kk in range(k):
            num = r + pad - kk * dil
            if num % stride == 0:
                taps.append((num // stride, kk))
        out.append((r, n, taps))
    return out


def _dgrad_descs(g, wb, geom, mask_src=None, mask_slope=1.0, res=None, out_div=1.0):
    """per-phase descriptors (pointers filled in; None = phase no tap reaches, already written) and the output"""
    key = ('d', g.dtype, g.shape[0], wb.shape[1], wb.shape[2], mask_slope, out_div)
    plan = geom.plans.get(key)
    if plan is None:
        _check(g, wb, mask_src, res)
        B, Hout, Wout, Cout = g.shape
        T, Cin, _ = wb.shape
        assert (Hout, Wout) == (geom.Hout, geom.Wout) and T == geom.ntaps and wb.shape[2] == Cout
        Hx, Wx, phases = geom.dgrad_plan()
        descs = []
        for lattice, taps, ry, rx in phases:
            descs.append((None, ry, rx) if not taps else
                         (_build_desc(g.dtype, B, Hout, Wout, Cout, Hx, Wx, Cin, lattice, taps, 0, 1.0, mask_slope, out_div,
                                      1.0), ry, rx))
        plan = geom.plans[key] = (descs, (B, Hx, Wx, Cin))
    descs, oshape = plan
    _dev_ok(g)
    gx = torch.empty(oshape, dtype=g.dtype, device=g.device)
    gp, wp, op = g.data_ptr(), wb.data_ptr(), gx.data_ptr()
    mp, rp = _opt_ptr(mask_src, g), _opt_ptr(res, g)
    live = []
    for desc, ry, rx in descs:
        if desc is None:                      # phase that no kernel tap reaches: gradient is the epilogue of zero
            gx[:, ry::geom.sy, rx::geom.sx] = 0 if res is None else res[:, ry::geom.sy, rx::geom.sx] / out_div
            continue
        desc.x, desc.w, desc.out, desc.mask_src, desc.res = gp, wp, op, mp, rp
        desc.bias = desc.res2 = None
        live.append(desc)
    return live, gx


def conv_dgrad(g, wb, geom, mask_src=None, mask_slope=1.0, res=None, out_div=1.0):
    """Data gradient of ``conv_forward``: g [B,Hout,Wout,Cout] -> gx [B,Hin(+2p),Win(+2p),Cin].

    wb [kh*kw, Cin, Cout] (channel roles swapped).  For reflect-padded convolutions the gradient is
    returned on the PADDED grid (Hin+2py, Win+2px); the caller folds the border back.
    Epilogue: gx = (gx * lrelu'(mask_src) + res) / out_div.
    """
    descs, gx = _dgrad_descs(g, wb, geom, mask_src, mask_slope, res, out_div)
    stream = lib.stream(g)
    for desc in descs:
        _gather(desc, stream, 'msmc_conv_gather(dgrad)')
    return gx


def conv_dgrad_group(items):
    """``items``: list of dicts with the arguments of ``conv_dgrad``; all phases of all members in one grouped call"""
    stream = lib.stream(items[0]['g'])
    snaps, outs = [], []
    for it in items:
        live, gx = _dgrad_descs(**it)
        snaps.extend(_snapshot(d, stream) for d in live)
        outs.append(gx)
    for i in range(0, len(snaps), 16):                   # msmc_conv_gather_group carries at most 16 members
        _gather_group(snaps[i:i + 16], stream, 'msmc_conv_gather_group(dgrad)', items[0]['g'].device)
    return outs


def conv_transpose1d_forward(x, w, k, stride, padding, bias=None, in_slope=1.0):
    """x [B,1,Lin,Cin] -> [B,1,Lout,Cout];  w [k, Cout, Cin]   (torch ConvTranspose1d semantics)."""
    _check(x, w)
    B, _, Lin, Cin = x.shape
    Cout = w.shape[1]
    Lout = (Lin - 1) * stride - 2 * padding + k
    out = torch.empty((B, 1, Lout, Cout), dtype=x.dtype, device=x.device)
    stream = lib.stream(x)
    snaps = []
    for r, n, taps1 in _phases(Lout, k, stride, 1, padding):
        assert taps1, 'kernel_size >= stride expected'
        taps = [(0, off, kk) for off, kk in taps1]
        lattice = (1, n, 0, 1, r, stride, 1, 1, 0, 0)
        d = _fill(None, x, w, out, B, 1, Lin, Cin, 1, Lout, Cout, lattice, taps, 0, bias=bias,
                  in_slope=in_slope)
        if not GROUP_PHASES:
            _gather(d, stream, 'msmc_conv_gather(convT)')
        else:
            snaps.append(_snapshot(d, stream))
    # the stride phases write disjoint output positions: independent launches, issued like the members of a grouped call
    # (one grid where their kernel choices coincide, parallel branches otherwise) instead of back to back
    for i in range(0, len(snaps), 16):
        _gather_group(snaps[i:i + 16], stream, 'msmc_conv_gather_group(convT)', x.device)
    return out


def conv_transpose1d_dgrad(g, wb, k, stride, padding, Lin, mask_src=None, mask_slope=1.0, res=None, out_div=1.0):
    """g [B,1,Lout,Cout] -> gx [B,1,Lin,Cin];  wb [k, Cin, Cout]:  gx[q] = sum_k g[q*stride + k - padding] wb[k]
    (epilogue: (* lrelu'(mask_src) + res) / out_div)."""
    _check(g, wb, mask_src, res)
    B, _, Lout, Cout = g.shape
    Cin = wb.shape[1]
    gx = torch.empty((B, 1, Lin, Cin), dtype=g.dtype, device=g.device)
    taps = [(0, kk, kk) for kk in range(k)]
    lattice = (1, Lin, 0, 1, 0, 1, 1, stride, 0, -padding)
    d = _fill(None, g, wb, gx, B, 1, Lout, Cout, 1, Lin, Cin, lattice, taps, 0, mask_src=mask_src, res=res,
              mask_slope=mask_slope, out_div=out_div)
    _gather(d, lib.stream(g), 'msmc_conv_gather(convT dgrad)')
    return gx


def conv_wgrad(x, g, geom, n_slices, in_slope=1.0, dw=None, db=None, copies=1, seen=None):
    """dW [kh*kw, Cout, Cin] fp32 of ``conv_forward`` (x [B,Hin,Win,Cin] pre-activation, g [B,Hout,Wout,Cout]);
    when ``db`` (fp32 [Cout]) is given the bias gradient is accumulated into it by the same launch."""
    key = ('w', x.dtype, x.shape[0], x.shape[3], g.shape[3], in_slope)
    desc = geom.plans.get(key)
    if desc is None:
        _check(x, g)
        B, Hin, Win, Cin = x.shape
        assert g.shape[1:3] == (geom.Hout, geom.Wout)
        desc = geom.plans[key] = _build_desc(x.dtype, B, Hin, Win, Cin, geom.Hout, geom.Wout, g.shape[3],
                                             geom.fwd_lattice, geom.fwd_taps, 1 if geom.reflect else 0, in_slope, 1.0,
                                             1.0, 1.0)
    _dev_ok(x)
    _dev_ok(g)
    if dw is None:
        dw = torch.zeros((n_slices, g.shape[3], x.shape[3]), dtype=torch.float32, device=x.device)
    desc.x = desc.w = desc.out = x.data_ptr()
    desc.bias = desc.mask_src = desc.res = desc.res2 = None
    desc.dw_copies = copies              # dw / db then hold ``copies`` privatised accumulators back to back
    _wgrad(desc, g.data_ptr(), dw, db, lib.stream(x), 'msmc_conv_wgrad', seen)
    return dw


def conv_wgrad_group(items):
    """``items``: list of dicts with the arguments of ``conv_wgrad`` (``dw`` required): independent weight gradients
    issued as grouped launches (msmc_conv_wgrad_group)."""
    stream = lib.stream(items[0]['x'])
    snaps, gs, dws, dbs = [], [], [], []
    for it in items:
        x, g, geom = it['x'], it['g'], it['geom']
        in_slope, dw, db, copies = it.get('in_slope', 1.0), it['dw'], it.get('db'), it.get('copies', 1)
        key = ('w', x.dtype, x.shape[0], x.shape[3], g.shape[3], in_slope)
        desc = geom.plans.get(key)
        if desc is None or not getattr(desc, '_tuned', False):
            conv_wgrad(x, g, geom, it['n_slices'], in_slope=in_slope, dw=dw, db=db, copies=copies, seen=it.get('seen'))   # builds + tunes
            continue
        _dev_ok(x)
        _dev_ok(g)
        desc.x = desc.w = desc.out = x.data_ptr()
        desc.bias = desc.mask_src = desc.res = desc.res2 = None
        desc.dw_copies = copies
        if it.get('seen') is not None:
            it['seen'].add(int(desc.variant))
        snaps.append(lib.ConvDesc.from_buffer_copy(desc))
        gs.append(g.data_ptr())
        dws.append(dw.data_ptr())
        dbs.append(db.data_ptr() if db is not None else None)
    L = lib.get()
    for i in range(0, len(snaps), 16):
        part = snaps[i:i + 16]
        n = len(part)
        arr = (lib.ConvDesc * n)(*part)
        vp = ctypes.c_void_p * n
        ga, dwa, dba = vp(*gs[i:i + 16]), vp(*dws[i:i + 16]), vp(*dbs[i:i + 16])

        needs = [L.msmc_conv_wgrad_workspace(ctypes.byref(part[k]), ga[k]) for k in range(n)]
        need = sum(needs)
        wsp, wsb = _workspace(items[0]['x'].device, stream, need) if need else (None, 0)
        defer = DEFER_TO if DEFER else None

        def deferred(call):
            """the real (not a timing) issue of this group: first stage only, partial results into the arena"""
            nonlocal wsp, wsb
            if defer is None or not need or defer.n + n > defer.CAPACITY:
                return call()
            keep = (wsp, wsb)
            wsp, wsb = defer.take(items[0]['x'].device, need), need
            defer.begin()
            try:
                return call()
            finally:
                defer.end()
                wsp, wsb = keep

        def grouped():
            lib.check(L.msmc_conv_wgrad_group_ws(arr, ga, dwa, dba, n, wsp, wsb, stream), 'msmc_conv_wgrad_group_ws')

        def grouped4():
            lib.check(L.msmc_conv_wgrad_group_ws4(arr, ga, dwa, dba, n, wsp, wsb, stream, 1), 'msmc_conv_wgrad_group_ws4')

        g4 = grouped4 if sum(1 for d in part if d.variant >= 4) > 1 else None

        def single():
            off = 0                                   # (members get regions of their own: a deferred second stage reads
            for k in range(n):                        #  them all at the end of the backward pass)
                lib.check(L.msmc_conv_wgrad_ws(ctypes.byref(part[k]), ga[k], dwa[k], dba[k],
                                               (wsp + off) if needs[k] else None, needs[k], stream), 'msmc_conv_wgrad_ws')
                off += needs[k]

        if n == 1:
            deferred(single)
            continue
        # the timing launches accumulate into the real dW / db: harmless only on scratch, so time on copies
        if AUTOTUNE and not lib._host_pointers_ok and not torch.cuda.is_current_stream_capturing():
            sig = ('wgrad-group',) + tuple(_signature(d) + (d.variant, d.split_shift) for d in part)
            if sig not in TUNED:
                keep = (dwa, dba)
                scratch = [torch.zeros(max(1, d.dw_copies) * d.ntaps * d.Cout * d.Cin, dtype=torch.float32,
                                       device=items[0]['x'].device) for d in part]
                sb = [torch.zeros(max(1, d.dw_copies) * d.Cout, dtype=torch.float32, device=items[0]['x'].device)
                      for d in part]
                dwa, dba = vp(*[t.data_ptr() for t in scratch]), vp(*[t.data_ptr() for t in sb])
                _group_choice('wgrad-group', part, grouped, single, g4)
                dwa, dba = keep
        deferred((single, grouped, g4 or grouped)[_group_choice('wgrad-group', part, grouped, single, g4)[0]])


def conv_transpose1d_wgrad(x, g, k, stride, padding, in_slope=1.0, dw=None, copies=1, seen=None):
    """dW [k, Cin, Cout] fp32 of ``conv_transpose1d_forward`` (x [B,1,Lin,Cin] pre-activation, g [B,1,Lout,Cout]):
    dW[k][ci][co] = sum_q act(x[q][ci]) * g[q*stride + k - padding][co]  -- the weight gradient of the strided
    convolution fine -> coarse with the operand roles swapped (the activation rides on the 'gradient' operand)."""
    _check(x, g)
    B, _, Lin, Cin = x.shape
    Lout, Cout = g.shape[2], g.shape[3]
    if dw is None:
        dw = torch.zeros((k, Cin, Cout), dtype=torch.float32, device=x.device)
    taps = [(0, kk, kk) for kk in range(k)]
    lattice = (1, Lin, 0, 1, 0, 1, 1, stride, 0, -padding)
    # kernel roles: "x" = g (fine, channels Cout), "g" = x (coarse, channels Cin) -> dw[k][Cin][Cout]
    d = _fill(None, g, g, g, B, 1, Lout, Cout, 1, Lin, Cin, lattice, taps, 0, in_slope=1.0,
              mask_slope=in_slope, kind='wgrad')
    lib.ptr(dw, torch.float32)
    d.dw_copies = copies
    _wgrad(d, lib.ptr(x), dw, None, lib.stream(x), 'msmc_conv_wgrad(convT)', seen)
    return dw


def colsum(g2d, out=None):
    """g [rows, C] (fp32 / bf16) -> fp32 [C] column sums (bias gradient), added to ``out`` when given (no atomics: per-block
    partial sums + a fixed-order second stage, msmc_colsum_ws)."""
    rows, C = g2d.shape
    L = lib.get()
    acc = out is not None
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=g2d.device)
    need = int(L.msmc_colsum_workspace(rows, C))
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=g2d.device)
    lib.check(L.msmc_colsum_ws(lib.ptr(g2d), lib.ptr(out, torch.float32), rows, C, _DT[g2d.dtype], int(acc), lib.ptr(ws), need,
                               lib.stream(g2d)), 'msmc_colsum_ws')
    return out



def reflect_fold(gp, H, W, p=1, mask_src=None, slope=1.0):
    """Backward of ReflectionPad2d(p) (+ leaky-ReLU' mask): gp [B,H+2p,W+2p,C] -> gx [B,H,W,C]."""
    B, C = gp.shape[0], gp.shape[3]
    assert gp.shape[1:3] == (H + 2 * p, W + 2 * p) and gp.is_contiguous()
    gx = torch.empty((B, H, W, C), dtype=gp.dtype, device=gp.device)
    lib.check(lib.get().msmc_reflect_fold(lib.ptr(gp), lib.ptr(mask_src) if mask_src is not None else None,
                                          lib.ptr(gx), B, H, W, C, p, float(slope), _DT[gp.dtype], lib.stream(gp)),
              'msmc_reflect_fold')
    return gx


def lrelu_bwd(g, y, slope):
    """g * (y > 0 ? 1 : slope) -- leaky-ReLU backward from the activation's output."""
    assert g.shape == y.shape and g.dtype == y.dtype and g.is_contiguous() and y.is_contiguous()
    gx = torch.empty_like(g)
    lib.check(lib.get().msmc_lrelu_bwd(lib.ptr(g), lib.ptr(y), lib.ptr(gx), g.numel(), float(slope), _DT[g.dtype],
                                       lib.stream(g)), 'msmc_lrelu_bwd')
    return gx


def lrelu_bwd_group(pairs, slope):
    """[(g, y), ...] -> [g * (y > 0 ? 1 : slope), ...] in launches of up to six tensors (msmc_lrelu_bwd_multi)."""
    outs = []
    L = lib.get()
    for i in range(0, len(pairs), 6):
        part = pairs[i:i + 6]
        n = len(part)
        gxs = []
        for g, y in part:
            assert g.shape == y.shape and g.dtype == y.dtype == part[0][0].dtype and g.is_contiguous() and y.is_contiguous()
            _dev_ok(g)
            _dev_ok(y)
            gxs.append(torch.empty_like(g))
        vp = ctypes.c_void_p * n
        lib.check(L.msmc_lrelu_bwd_multi(vp(*[g.data_ptr() for g, _ in part]), vp(*[y.data_ptr() for _, y in part]),
                                         vp(*[t.data_ptr() for t in gxs]), (ctypes.c_long * n)(*[g.numel() for g, _ in part]),
                                         n, float(slope), _DT[part[0][0].dtype], lib.stream(part[0][0])),
                  'msmc_lrelu_bwd_multi')
        outs.extend(gxs)
    return outs


def reflect_fold_group(items, p=1, slope=1.0, tap_first=False):
    """[(gp, H, W, mask_src or None[, res or None]), ...] -> folded gradients (* lrelu'(mask_src) + res), up to six tensors
    per launch (msmc_reflect_fold_multi_res).  ``tap_first``: (fold + res) * lrelu'(mask_src) instead -- the input is an
    activated map and res the gradient of its other reader (msmc_reflect_fold_multi_tap)."""
    items = [tuple(it) + (None,) * (5 - len(it)) for it in items]
    outs = []
    L = lib.get()
    for i in range(0, len(items), 6):
        part = items[i:i + 6]
        n = len(part)
        gxs, masks, ress = [], [], []
        for gp, H, W, mask, res in part:
            assert gp.shape[1:3] == (H + 2 * p, W + 2 * p) and gp.is_contiguous() and gp.dtype == part[0][0].dtype
            _dev_ok(gp)
            gx = torch.empty((gp.shape[0], H, W, gp.shape[3]), dtype=gp.dtype, device=gp.device)
            for aux in (mask, res):
                if aux is not None:
                    _dev_ok(aux)
                    assert aux.shape == gx.shape and aux.dtype == gp.dtype and aux.is_contiguous()
            gxs.append(gx)
            masks.append(mask.data_ptr() if mask is not None else None)
            ress.append(res.data_ptr() if res is not None else None)
        vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
        entry = L.msmc_reflect_fold_multi_tap if tap_first else L.msmc_reflect_fold_multi_res
        lib.check(entry(vp(*[t[0].data_ptr() for t in part]), vp(*masks), vp(*ress),
                        vp(*[t.data_ptr() for t in gxs]), ip(*[t[0].shape[0] for t in part]),
                        ip(*[t[1] for t in part]), ip(*[t[2] for t in part]),
                        ip(*[t[0].shape[3] for t in part]), n, p, float(slope), _DT[part[0][0].dtype],
                        lib.stream(part[0][0])), 'msmc_reflect_fold_multi_tap' if tap_first else 'msmc_reflect_fold_multi_res')
        outs.extend(gxs)
    return outs

