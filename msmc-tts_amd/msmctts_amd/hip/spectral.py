"""Spectral front-ends on the gfx950 kernels: framed-DFT GEMMs (csrc/conv.hip, exact-fp32 MFMA, one tap) between
the framing / magnitude / image / log kernels of csrc/spectral.hip.

Replaces ``torch.stft`` + glue of ``TorchSTFT.transform`` / ``MelScale.forward``
(reference msmctts/utils/audio.py:398-419, 348-376) and ``MelLoss.mel_spectrogram``
(reference msmctts/trainers/criterions/stft_loss.py:76-108).  Everything is fp32 (also in bf16 runs).
"""
import math

import torch

from . import conv as K
from . import lib


def _pad4(n):
    return (n + 3) // 4 * 4


# the constant-matrix GEMMs (framed DFT, mel / filter-bank projections) of the bf16 training configuration run as three
# bf16 matrix-core products from two-piece splits (csrc/gemm1.inc conv_gemm1s_kernel); MSMC_SPECTRAL_SPLIT=0: exact fp32 (A/B)
import os as _os
SPLIT_BF16 = _os.environ.get('MSMC_SPECTRAL_SPLIT', '1') != '0'


class _Frames(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, T, n_fft, NP, hop, pad):
        B, L = x.shape
        xc = x.contiguous().float()
        fr = torch.empty((B, 1, T, NP), dtype=torch.float32, device=x.device)
        lib.check(lib.get().msmc_stft_frames_fwd(lib.ptr(xc), lib.ptr(fr), B, L, T, n_fft, NP, hop, pad, lib.stream(xc)),
                  'msmc_stft_frames_fwd')
        ctx.args = (B, L, T, n_fft, NP, hop, pad)
        return fr

    @staticmethod
    def backward(ctx, g):
        B, L, T, n_fft, NP, hop, pad = ctx.args
        g = g.contiguous()
        gx = torch.empty((B, L), dtype=torch.float32, device=g.device)
        lib.check(lib.get().msmc_stft_frames_bwd(lib.ptr(g), lib.ptr(gx), B, L, T, n_fft, NP, hop, pad, lib.stream(g)),
                  'msmc_stft_frames_bwd')
        return gx, None, None, None, None, None


class _ConstGemm(torch.autograd.Function):
    """y[b,1,t,:] = W x[b,1,t,:] with a constant matrix (DFT basis / filter bank): one-tap conv on the fp32 MFMA."""

    @staticmethod
    def forward(ctx, x, wf, wb, split=False):
        geom = _geom1(x.shape[2])
        ctx.wb = wb                       # (a constant, not a graph tensor)
        ctx.geom, ctx.split = geom, bool(split)
        return _const_gemm(x, wf, geom, ctx.split)

    @staticmethod
    def backward(ctx, g):
        return _const_gemm(g.contiguous(), ctx.wb, ctx.geom, ctx.split, dgrad=True), None, None, None


class _SpecMag(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, F, FP, lo, clamp_mode):
        CP = spec.shape[-1]
        R = spec.numel() // CP
        mag = torch.empty(spec.shape[:-1] + (FP,), dtype=torch.float32, device=spec.device)
        lib.check(lib.get().msmc_spec_mag_fwd(lib.ptr(spec), lib.ptr(mag), R, F, CP, FP, lo, clamp_mode, lib.stream(spec)),
                  'msmc_spec_mag_fwd')
        ctx.save_for_backward(spec, mag)
        ctx.args = (R, F, CP, FP, lo, clamp_mode)
        return mag

    @staticmethod
    def backward(ctx, g):
        spec, mag = ctx.saved_tensors
        R, F, CP, FP, lo, clamp_mode = ctx.args
        g = g.contiguous()
        gs = torch.empty_like(spec)
        lib.check(lib.get().msmc_spec_mag_bwd(lib.ptr(spec), lib.ptr(mag), lib.ptr(g), lib.ptr(gs), R, F, CP, FP, lo,
                                              clamp_mode, lib.stream(g)), 'msmc_spec_mag_bwd')
        return gs, None, None, None, None


_IMG_DT = {torch.float32: 0, torch.bfloat16: 1}


class _MrdImage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mel, F, dtype=torch.float32):
        B, _, T, FP = mel.shape
        img = torch.empty((B, F, T, 2), dtype=dtype, device=mel.device)
        lib.check(lib.get().msmc_mrd_image_fwd_dt(lib.ptr(mel), lib.ptr(img), B, T, F, FP, _IMG_DT[dtype], lib.stream(mel)),
                  'msmc_mrd_image_fwd_dt')
        ctx.save_for_backward(mel)
        ctx.F, ctx.dtype = F, dtype
        return img

    @staticmethod
    def backward(ctx, g):
        (mel,) = ctx.saved_tensors
        B, _, T, FP = mel.shape
        g = g.contiguous()
        if g.dtype != ctx.dtype:
            g = g.to(ctx.dtype)
        gm = torch.empty_like(mel)
        lib.check(lib.get().msmc_mrd_image_bwd_dt(lib.ptr(mel), lib.ptr(g), lib.ptr(gm), B, T, ctx.F, FP, _IMG_DT[ctx.dtype],
                                                  lib.stream(g)), 'msmc_mrd_image_bwd_dt')
        return gm, None, None


class _LogClamp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo):
        xc = x.contiguous()
        y = torch.empty_like(xc)
        lib.check(lib.get().msmc_log_clamp_fwd(lib.ptr(xc), lib.ptr(y), xc.numel(), lo, lib.stream(xc)),
                  'msmc_log_clamp_fwd')
        ctx.save_for_backward(xc)
        ctx.lo = lo
        return y

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(xc)
        lib.check(lib.get().msmc_log_clamp_bwd(lib.ptr(xc), lib.ptr(g), lib.ptr(gx), xc.numel(), ctx.lo, lib.stream(g)),
                  'msmc_log_clamp_bwd')
        return gx, None


def dft_basis(n_fft, win, normalized, device):
    """Windowed real-DFT basis as one-tap conv slices: forward [1, CP, NP] (rows: re 0..F-1 | im F..2F-1),
    data-gradient [1, NP, CP].  ``win`` is the analysis window already centred / zero-padded to n_fft; samples the
    window zeroes (``win_length < n_fft``: MelLoss frames 1200 of 2048) are dropped from the reduction, so the
    basis covers only ``n_eff`` samples starting ``lo`` into each frame.  Returns (forward, backward, lo, n_eff)."""
    F = n_fft // 2 + 1
    nz = torch.nonzero(win.detach().cpu() != 0).flatten()
    lo, hi = (int(nz[0]), int(nz[-1]) + 1) if nz.numel() else (0, n_fft)
    n_eff = hi - lo
    NP, CP = _pad4(n_eff), _pad4(2 * F)
    j = torch.arange(lo, hi, dtype=torch.float64)
    f = torch.arange(F, dtype=torch.float64)
    ang = 2.0 * math.pi * torch.outer(f, j) / n_fft
    scale = (1.0 / math.sqrt(n_fft)) if normalized else 1.0
    w = win.double().cpu()[lo:hi] * scale
    W = torch.zeros((CP, NP), dtype=torch.float64)
    W[:F, :n_eff] = torch.cos(ang) * w
    W[F:2 * F, :n_eff] = -torch.sin(ang) * w
    W = W.float()
    return W.unsqueeze(0).contiguous().to(device), W.t().unsqueeze(0).contiguous().to(device), lo, n_eff


def projection(mat, device):
    """out[:, o] = sum_i in[:, i] * mat[i, o] as one-tap conv slices with both sides padded to multiples of 4."""
    I, O = mat.shape
    W = torch.zeros((_pad4(O), _pad4(I)), dtype=torch.float32)
    W[:O, :I] = mat.t().float().cpu()
    return W.unsqueeze(0).contiguous().to(device), W.t().unsqueeze(0).contiguous().to(device)


_SPLIT_IMAGES = {}


def split_image(w):
    """[1, N, K] fp32 constant one-tap slices -> their split-bf16 image [N][ceil(K / 32)][hi 32 | lo 32] (hi = bf16(w), lo =
    bf16(w - hi)), built once per matrix (csrc/gemm1.inc conv_gemm1s_kernel)"""
    hit = _SPLIT_IMAGES.get(w.data_ptr())
    if hit is not None and hit[0] is w:
        return hit[1]
    K._bounded(_SPLIT_IMAGES, 256)
    W = w[0]
    N, Kc = W.shape
    nch = (Kc + 31) // 32
    Wp = torch.zeros((N, nch * 32), dtype=torch.float32, device=W.device)
    Wp[:, :Kc] = W
    hi = Wp.to(torch.bfloat16)
    lo = (Wp - hi.float()).to(torch.bfloat16)
    img = torch.cat((hi.view(N, nch, 32), lo.view(N, nch, 32)), dim=2).contiguous()
    _SPLIT_IMAGES[w.data_ptr()] = (w, img)
    return img


def _const_gemm(x, w, geom, split, dgrad=False):
    """x @ constant matrix ``w`` ([1, N, K] forward slices; ``dgrad``: w is the transposed slice set) -- exact fp32 on the
    matrix cores, or (``split``: the bf16 training configuration) three bf16 products from two-piece splits"""
    if split:
        return K.const_gemm_split(x, split_image(w), w.shape[1])
    return K.conv_dgrad(x, w, geom) if dgrad else K.conv_forward(x, w, geom)


_GEOMS = {}


def _geom1(T):
    """geometry of a one-tap GEMM over [B, 1, T, C] rows (cached: its descriptors and tuner choices live on it)"""
    g = _GEOMS.get(T)
    if g is None:
        K._bounded(_GEOMS, 256)
        g = _GEOMS[T] = K.Geometry(1, T, (1, 1))
    return g


class MrdFront(object):
    """Every tensor of one ``mrd_image`` evaluation (frames -> spectrum -> magnitude -> mel-scaled magnitude -> image),
    kept so that (a) the backward pass of the chain is five launches on the saved tensors and (b) a LATER consumer of some
    rows of the same waveforms -- the generator step, which shows the discriminator the batch the D step already framed and
    transformed: the front-end has no parameters -- reuses the image and back-propagates through its rows only
    (``image_rows``) instead of recomputing 25 launches per pass."""

    def __init__(self, x, n_fft, hop, dft, fb, dtype, split=None):
        B, L = x.shape
        self.B, self.L, self.n_fft, self.hop, self.dft, self.fb, self.dtype = B, L, n_fft, hop, dft, fb, dtype
        # constant-matrix GEMMs in split bf16 (2^-16 relative) when the stack computes in bf16, exact fp32 otherwise
        self.split = split = (dtype == torch.bfloat16 and SPLIT_BF16) if split is None else bool(split)
        F = self.F = n_fft // 2 + 1
        T = self.T = L // hop + 1
        lo, n_eff = dft[2], dft[3]
        self.frame_args = (T, n_eff, _pad4(n_eff), hop, n_fft // 2 - lo)
        Lb = lib.get()
        xc = x.detach().contiguous().float()
        fr = torch.empty((B, 1, T, _pad4(n_eff)), dtype=torch.float32, device=x.device)
        lib.check(Lb.msmc_stft_frames_fwd(lib.ptr(xc), lib.ptr(fr), B, L, T, n_eff, _pad4(n_eff), hop, n_fft // 2 - lo,
                                          lib.stream(xc)), 'msmc_stft_frames_fwd')
        geom = _geom1(T)
        self.spec = _const_gemm(fr, dft[0], geom, split)
        CP, FP = self.spec.shape[-1], _pad4(F)
        self.mag = torch.empty((B, 1, T, FP), dtype=torch.float32, device=x.device)
        lib.check(Lb.msmc_spec_mag_fwd(lib.ptr(self.spec), lib.ptr(self.mag), B * T, F, CP, FP, 1e-7, 1, lib.stream(xc)),
                  'msmc_spec_mag_fwd')
        self.mel = _const_gemm(self.mag, fb[0], geom, split) if fb is not None else self.mag
        self.img = torch.empty((B, F, T, 2), dtype=dtype, device=x.device)
        lib.check(Lb.msmc_mrd_image_fwd_dt(lib.ptr(self.mel), lib.ptr(self.img), B, T, F, FP, _IMG_DT[dtype], lib.stream(xc)),
                  'msmc_mrd_image_fwd_dt')

    def image(self, r0, r1):
        """rows r0 .. r1-1 of the image: a view where the kernels can take it (16-byte aligned -- any row offset that is a
        multiple of 8: the training batches), a copy otherwise"""
        v = self.img[r0:r1]
        return v if v.data_ptr() % 16 == 0 else v.clone()

    def backward_rows(self, g, r0, r1):
        """gradient of the waveform rows r0 .. r1-1 from the gradient ``g`` of their image rows"""
        b, T, F = r1 - r0, self.T, self.F
        FP, CP = self.mag.shape[-1], self.spec.shape[-1]
        Lb = lib.get()
        g = g.contiguous()
        if g.dtype != self.dtype:
            g = g.to(self.dtype)
        mel, mag, spec = self.mel[r0:r1], self.mag[r0:r1], self.spec[r0:r1]
        gm = torch.empty_like(mel)
        lib.check(Lb.msmc_mrd_image_bwd_dt(lib.ptr(mel), lib.ptr(g), lib.ptr(gm), b, T, F, FP, _IMG_DT[self.dtype],
                                           lib.stream(g)), 'msmc_mrd_image_bwd_dt')
        geom = _geom1(T)
        if self.fb is not None:
            gm = _const_gemm(gm, self.fb[1], geom, self.split, dgrad=True)
        gs = torch.empty_like(spec)
        lib.check(Lb.msmc_spec_mag_bwd(lib.ptr(spec), lib.ptr(mag), lib.ptr(gm), lib.ptr(gs), b * T, F, CP, FP, 1e-7, 1,
                                       lib.stream(g)), 'msmc_spec_mag_bwd')
        gfr = _const_gemm(gs, self.dft[1], geom, self.split, dgrad=True)
        T_, n_eff, NP, hop, pad = self.frame_args
        gx = torch.empty((b, self.L), dtype=torch.float32, device=g.device)
        lib.check(Lb.msmc_stft_frames_bwd(lib.ptr(gfr), lib.ptr(gx), b, self.L, T_, n_eff, NP, hop, pad, lib.stream(g)),
                  'msmc_stft_frames_bwd')
        return gx


# ---- the front-ends of SEVERAL hop lengths in lock step (round 6) -------------------------------------------------------------
# The five resolution discriminators' chains frames -> DFT -> magnitude -> filter bank -> image are independent and identical in
# shape; one after the other they were 25 launches of 5-10 us forward and 25 backward on the step's critical chain.  Here every
# stage of all chains is ONE launch (msmc_spectral_multi for the element-wise stages, a grouped call for the constant-matrix
# GEMMs): 5 + 5.  MSMC_FRONTS_LOCKSTEP=0: chain by chain (A/B).
FRONTS_LOCKSTEP = _os.environ.get('MSMC_FRONTS_LOCKSTEP', '1') != '0'


def _multi(ops):
    L = lib.get()
    for i in range(0, len(ops), lib.SPECTRAL_MULTI_MAX):
        part = ops[i:i + lib.SPECTRAL_MULTI_MAX]
        arr = (lib.SpectralOp * len(part))(*part)
        lib.check(L.msmc_spectral_multi(arr, len(part), part[0]._stream), 'msmc_spectral_multi')


def _op(kind, stream, a, out, b=None, c=None, dtype=0, **kw):
    o = lib.SpectralOp()
    o.kind, o.dtype = kind, dtype
    o.a, o.b, o.c, o.out = a.data_ptr(), (b.data_ptr() if b is not None else None), (c.data_ptr() if c is not None else None), out.data_ptr()
    for k, v in kw.items():
        setattr(o, k, v)
    o._stream = stream
    o._keep = (a, b, c, out)
    return o


def _gemms(xs, ws, split, dgrad=False):
    """one constant-matrix GEMM per chain: a grouped call in the split-bf16 form, chain by chain in exact fp32"""
    if split:
        return K.const_gemm_split_group(xs, [split_image(w) for w in ws], [w.shape[1] for w in ws])
    return [_const_gemm(x, w, _geom1(x.shape[2]), False, dgrad=dgrad) for x, w in zip(xs, ws)]


def mrd_fronts(x, specs, dtype, split=None):
    """``MrdFront(x, n_fft, hop, dft, fb, dtype)`` for every (n_fft, hop, dft, fb) of ``specs``, all chains advancing together"""
    fronts = [MrdFront.__new__(MrdFront) for _ in specs]
    B, L = x.shape
    xc = x.detach().contiguous().float()
    st = lib.stream(xc)
    split = (dtype == torch.bfloat16 and SPLIT_BF16) if split is None else bool(split)
    frs, ops = [], []
    for f, (n_fft, hop, dft, fb) in zip(fronts, specs):
        f.B, f.L, f.n_fft, f.hop, f.dft, f.fb, f.dtype, f.split = B, L, n_fft, hop, dft, fb, dtype, split
        f.F, f.T = n_fft // 2 + 1, L // hop + 1
        lo, n_eff = dft[2], dft[3]
        f.frame_args = (f.T, n_eff, _pad4(n_eff), hop, n_fft // 2 - lo)
        fr = torch.empty((B, 1, f.T, _pad4(n_eff)), dtype=torch.float32, device=x.device)
        frs.append(fr)
        ops.append(_op(0, st, xc, fr, B=B, L=L, T=f.T, n_fft=n_eff, NP=_pad4(n_eff), hop=hop, pad=n_fft // 2 - lo))
    _multi(ops)
    specs_t = _gemms(frs, [f.dft[0] for f in fronts], split)
    ops = []
    for f, sp in zip(fronts, specs_t):
        f.spec = sp
        f.mag = torch.empty((B, 1, f.T, _pad4(f.F)), dtype=torch.float32, device=x.device)
        ops.append(_op(2, st, sp, f.mag, R=B * f.T, F=f.F, CP=sp.shape[-1], FP=_pad4(f.F), lo=1e-7, clamp_mode=1))
    _multi(ops)
    with_fb = [f for f in fronts if f.fb is not None]
    for f, mel in zip(with_fb, _gemms([f.mag for f in with_fb], [f.fb[0] for f in with_fb], split) if with_fb else []):
        f.mel = mel
    ops = []
    for f in fronts:
        if f.fb is None:
            f.mel = f.mag
        f.img = torch.empty((B, f.F, f.T, 2), dtype=dtype, device=x.device)
        ops.append(_op(4, st, f.mel, f.img, dtype=_IMG_DT[dtype], B=B, T=f.T, F=f.F, FP=_pad4(f.F)))
    _multi(ops)
    return fronts


def backward_rows_lockstep(fronts, gs, r0, r1):
    """``MrdFront.backward_rows(g, r0, r1)`` of several fronts, every stage of all chains in one launch; ``gs[i]`` None: no gradient"""
    live = [(f, g) for f, g in zip(fronts, gs) if g is not None]
    out = [None] * len(fronts)
    if not live:
        return out
    b = r1 - r0
    ops, gms, st = [], [], None
    for f, g in live:
        g = g.contiguous()
        if g.dtype != f.dtype:
            g = g.to(f.dtype)
        st = lib.stream(g)
        mel = f.mel[r0:r1]
        gm = torch.empty_like(mel)
        gms.append(gm)
        ops.append(_op(5, st, mel, gm, b=g, dtype=_IMG_DT[f.dtype], B=b, T=f.T, F=f.F, FP=f.mag.shape[-1]))
    _multi(ops)
    with_fb = [i for i, (f, _) in enumerate(live) if f.fb is not None]
    if with_fb:
        for i, gm in zip(with_fb, _gemms([gms[i] for i in with_fb], [live[i][0].fb[1] for i in with_fb], live[0][0].split, dgrad=True)):
            gms[i] = gm
    ops, gss = [], []
    for (f, _), gm in zip(live, gms):
        spec, mag = f.spec[r0:r1], f.mag[r0:r1]
        gsp = torch.empty_like(spec)
        gss.append(gsp)
        ops.append(_op(3, st, spec, gsp, b=mag, c=gm, R=b * f.T, F=f.F, CP=spec.shape[-1], FP=mag.shape[-1], lo=1e-7, clamp_mode=1))
    _multi(ops)
    gfrs = _gemms(gss, [f.dft[1] for f, _ in live], live[0][0].split, dgrad=True)
    ops, gxs = [], []
    for (f, _), gfr in zip(live, gfrs):
        T_, n_eff, NP, hop, pad = f.frame_args
        gx = torch.empty((b, f.L), dtype=torch.float32, device=gfr.device)
        gxs.append(gx)
        ops.append(_op(1, st, gfr, gx, B=b, L=f.L, T=T_, n_fft=n_eff, NP=NP, hop=hop, pad=pad))
    _multi(ops)
    it = iter(gxs)
    for i, g in enumerate(gs):
        if g is not None:
            out[i] = next(it)
    return out


class _MrdImageRowsMulti(torch.autograd.Function):
    """``_MrdImageRows`` of several fronts as ONE node: its backward runs all chains in lock step.  ``xs``: one alias of the
    waveform rows per front (hip/spectral.py wave_fan: their gradients are summed by the fan's backward launch)"""

    @staticmethod
    def forward(ctx, fronts, r0, r1, *xs):
        ctx.fronts, ctx.rows = fronts, (r0, r1)
        return tuple(f.image(r0, r1) for f in fronts)

    @staticmethod
    def backward(ctx, *gs):
        return (None, None, None) + tuple(backward_rows_lockstep(ctx.fronts, gs, *ctx.rows))


def mrd_image_rows_multi(xs, fronts, r0, r1):
    """image rows r0 .. r1-1 of every front as functions of the waveform aliases ``xs`` (one backward node for all)"""
    if not (FRONTS_LOCKSTEP and any(x.requires_grad for x in xs)):
        return [mrd_image_rows(x, f, r0, r1) for x, f in zip(xs, fronts)]
    return list(_MrdImageRowsMulti.apply(list(fronts), r0, r1, *[x.contiguous() for x in xs]))


class _MrdImage2(torch.autograd.Function):
    """the whole chain as ONE autograd node: forward = MrdFront(x), backward = MrdFront.backward_rows over all rows"""

    @staticmethod
    def forward(ctx, x, n_fft, hop, dft, fb, dtype):
        ctx.front = MrdFront(x, n_fft, hop, dft, fb, dtype)
        img, ctx.front.img = ctx.front.img, None         # (the node must not hold its own output: a reference cycle)
        return img

    @staticmethod
    def backward(ctx, g):
        f = ctx.front
        return f.backward_rows(g, 0, f.B), None, None, None, None, None


class _MrdImageRows(torch.autograd.Function):
    """rows r0 .. r1-1 of an image an earlier ``MrdFront`` computed from the same waveform rows; ``x`` (those rows of the
    waveform, possibly with a gradient history the earlier evaluation did not have) only ties the node into the graph"""

    @staticmethod
    def forward(ctx, x, front, r0, r1):
        assert x.shape == (r1 - r0, front.L), (x.shape, r0, r1, front.L)
        ctx.front, ctx.rows = front, (r0, r1)
        return front.image(r0, r1)

    @staticmethod
    def backward(ctx, g):
        return ctx.front.backward_rows(g, *ctx.rows), None, None, None


def mrd_image(x, n_fft, hop, dft, fb, dtype=torch.float32):
    """x (B, L) -> MRD input image, channels-last [B, F, T', 2] (ch0 mel-scaled magnitude, ch1 normalised log), written
    in ``dtype`` (the discriminator stack's compute dtype: the spectra themselves stay fp32)."""
    return _MrdImage2.apply(x, n_fft, hop, dft, fb, dtype)


def mrd_front(x, n_fft, hop, dft, fb, dtype=torch.float32):
    """the same evaluation as an object whose image (``.img``) and intermediates outlive the call (no autograd: for
    waveforms without gradient history -- the D step's detached batch)"""
    return MrdFront(x, n_fft, hop, dft, fb, dtype)


def mrd_image_rows(x, front, r0, r1):
    """image rows r0 .. r1-1 of ``front`` as a function of ``x`` (= the waveform rows they were computed from)"""
    if not x.requires_grad:
        return front.image(r0, r1)
    return _MrdImageRows.apply(x.contiguous(), front, r0, r1)


def stft_magnitude(x, n_fft, hop, dft, lo, split=False):
    """x (B, L) -> (B, T', F) magnitude sqrt(clamp(re^2 + im^2, lo)) of the centred STFT (torch.stft defaults:
    reflect padding n_fft // 2, window already folded into ``dft``)."""
    B, L = x.shape
    F = n_fft // 2 + 1
    T = L // hop + 1
    lo_, n_eff = dft[2], dft[3]
    fr = _Frames.apply(x, T, n_eff, _pad4(n_eff), hop, n_fft // 2 - lo_)
    spec = _ConstGemm.apply(fr, dft[0], dft[1], split)
    mag = _SpecMag.apply(spec, F, _pad4(F), lo, 1)
    return mag[:, 0, :, :F]


def log_mel(y, n_fft, hop, dft, mel, num_mels, split=False):
    """y (B, L) -> log-mel [B, 1, T', pad4(num_mels)] per MelLoss.mel_spectrogram (manual reflect pad, no centring)."""
    B, L = y.shape
    F = n_fft // 2 + 1
    pad = int((n_fft - hop) / 2)
    T = (L + 2 * pad - n_fft) // hop + 1
    lo, n_eff = dft[2], dft[3]
    fr = _Frames.apply(y, T, n_eff, _pad4(n_eff), hop, pad - lo)
    spec = _ConstGemm.apply(fr, dft[0], dft[1], split)
    mag = _SpecMag.apply(spec, F, _pad4(F), 1e-9, 0)
    m = _ConstGemm.apply(mag, mel[0], mel[1], split)
    return _LogClamp.apply(m, 1e-5)[..., :num_mels]


class _LogMelPair(torch.autograd.Function):
    """``log_mel`` of a prediction (with gradient) and of its target (without) -- MelLoss.forward's two chains, identical in shape
    -- with every stage of both chains in one launch (msmc_spectral_multi / a grouped call for the two constant-matrix GEMMs): five
    launches instead of ten; the backward pass is the prediction's chain alone (five launches on the saved tensors)."""

    @staticmethod
    def forward(ctx, y, t, n_fft, hop, dft, mel, num_mels, split):
        B, L = y.shape
        F = n_fft // 2 + 1
        pad = int((n_fft - hop) / 2)
        T = (L + 2 * pad - n_fft) // hop + 1
        lo, n_eff = dft[2], dft[3]
        NP, FP = _pad4(n_eff), _pad4(F)
        ys = [y.contiguous().float(), t.detach().contiguous().float()]
        st = lib.stream(ys[0])
        frs = [torch.empty((B, 1, T, NP), dtype=torch.float32, device=y.device) for _ in ys]
        _multi([_op(0, st, x, fr, B=B, L=L, T=T, n_fft=n_eff, NP=NP, hop=hop, pad=pad - lo) for x, fr in zip(ys, frs)])
        specs = _gemms(frs, [dft[0], dft[0]], split)
        mags = [torch.empty((B, 1, T, FP), dtype=torch.float32, device=y.device) for _ in ys]
        _multi([_op(2, st, sp, mg, R=B * T, F=F, CP=sp.shape[-1], FP=FP, lo=1e-9, clamp_mode=0) for sp, mg in zip(specs, mags)])
        ms = _gemms(mags, [mel[0], mel[0]], split)
        outs = [torch.empty_like(m) for m in ms]
        _multi([_op(6, st, m, o, R=m.numel(), lo=1e-5) for m, o in zip(ms, outs)])
        ctx.save_for_backward(specs[0], mags[0], ms[0])
        ctx.args = (B, L, T, n_eff, NP, hop, pad - lo, F, FP, dft, mel, split, num_mels)
        return outs[0][..., :num_mels], outs[1][..., :num_mels]

    @staticmethod
    def backward(ctx, g, g_unused):
        spec, mag, m = ctx.saved_tensors
        B, L, T, n_eff, NP, hop, pad, F, FP, dft, mel, split, num_mels = ctx.args
        Lb = lib.get()
        geom = _geom1(T)
        gfull = torch.zeros_like(m) if m.shape[-1] != num_mels else None
        if gfull is not None:
            gfull[..., :num_mels] = g
            g = gfull
        g = g.contiguous()
        gm = torch.empty_like(m)
        lib.check(Lb.msmc_log_clamp_bwd(lib.ptr(m), lib.ptr(g), lib.ptr(gm), m.numel(), 1e-5, lib.stream(g)), 'msmc_log_clamp_bwd')
        gmag = _const_gemm(gm, mel[1], geom, split, dgrad=True)
        gs = torch.empty_like(spec)
        lib.check(Lb.msmc_spec_mag_bwd(lib.ptr(spec), lib.ptr(mag), lib.ptr(gmag), lib.ptr(gs), B * T, F, spec.shape[-1], FP, 1e-9, 0,
                                       lib.stream(g)), 'msmc_spec_mag_bwd')
        gfr = _const_gemm(gs, dft[1], geom, split, dgrad=True)
        gy = torch.empty((B, L), dtype=torch.float32, device=g.device)
        lib.check(Lb.msmc_stft_frames_bwd(lib.ptr(gfr), lib.ptr(gy), B, L, T, n_eff, NP, hop, pad, lib.stream(g)), 'msmc_stft_frames_bwd')
        return gy, None, None, None, None, None, None, None


def log_mel_pair(y, t, n_fft, hop, dft, mel, num_mels, split=False):
    """(log_mel(y), log_mel(t)) with t taken as a constant: both chains in lock step (see _LogMelPair)"""
    return _LogMelPair.apply(y, t, n_fft, hop, dft, mel, num_mels, bool(split))


class _WaveFan(torch.autograd.Function):
    """y (B, L) fp32 -> ``n_alias`` aliases of y (for consumers that read it in fp32: the resolution sub-discriminators'
    front-ends) followed by one copy per entry of ``padded`` in ``dtype``, reflection-padded on the right to that length (the
    period sub-discriminators' inputs).  The backward pass is ONE launch summing every consumer's gradient (msmc_wave_fan_bwd)."""

    @staticmethod
    def forward(ctx, y, n_alias, padded, dtype):
        import ctypes
        B, L = y.shape
        yc = y.contiguous().float()
        copies = [torch.empty((B, lp), dtype=dtype, device=y.device) for lp in padded]
        n = len(copies)
        if n:
            vp, ip = ctypes.c_void_p * n, ctypes.c_int * n
            lib.check(lib.get().msmc_wave_fan_fwd(lib.ptr(yc), vp(*[lib.ptr(c).value for c in copies]), ip(*padded), n, B, L,
                                                  _IMG_DT[dtype], lib.stream(yc)), 'msmc_wave_fan_fwd')
        ctx.args = (B, L, int(n_alias), tuple(padded), dtype)
        return tuple(yc.view_as(yc) for _ in range(n_alias)) + tuple(copies)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        B, L, n_alias, padded, dtype = ctx.args
        g32 = [None if g is None else g.contiguous().float() for g in grads[:n_alias]]
        g16 = [None if g is None else (g if g.dtype == dtype else g.to(dtype)).contiguous() for g in grads[n_alias:]]
        dev = next(g.device for g in list(g32) + list(g16) if g is not None)
        gy = torch.empty((B, L), dtype=torch.float32, device=dev)
        n32, n = len(g32), len(g16)
        vp32, vp16, ip = ctypes.c_void_p * max(1, n32), ctypes.c_void_p * max(1, n), ctypes.c_int * max(1, n)
        p32 = vp32(*([None if g is None else lib.ptr(g).value for g in g32] or [None]))
        p16 = vp16(*([None if g is None else lib.ptr(g).value for g in g16] or [None]))
        lib.check(lib.get().msmc_wave_fan_bwd(p32, n32, p16, ip(*(list(padded) or [L])), n, lib.ptr(gy), B, L, _IMG_DT[dtype],
                                              lib.stream(gy)), 'msmc_wave_fan_bwd')
        return gy, None, None, None


def wave_fan(y, n_alias, padded, dtype):
    """-> (aliases of y for fp32 readers, reflection-padded copies of y in ``dtype``): see _WaveFan"""
    outs = _WaveFan.apply(y, int(n_alias), tuple(int(v) for v in padded), dtype)
    return list(outs[:n_alias]), list(outs[n_alias:])


def window_gather(starts, wav, nframes, hop):
    """vocoder windows from per-utterance start frames (int64 [B], on the device) -> (frame indices [B, nframes] int64, waveform
    windows [B, nframes * hop] fp32) in one launch (msmc_window_gather); ``wav`` [B, L] fp32 contiguous"""
    B, L = wav.shape
    assert starts.dtype == torch.int64 and starts.numel() == B and wav.dtype == torch.float32 and wav.is_contiguous()
    frames = torch.empty((B, nframes), dtype=torch.int64, device=wav.device)
    target = torch.empty((B, nframes * hop), dtype=torch.float32, device=wav.device)
    lib.check(lib.get().msmc_window_gather(lib.ptr(starts), lib.ptr(wav), lib.ptr(frames), lib.ptr(target), B, int(nframes), int(hop),
                                           int(L), lib.stream(wav)), 'msmc_window_gather')
    return frames, target
