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

    @property
    def ntaps(self):
        return self.kh * self.kw


def _fill(desc, x, w, out, B, Hin, Win, Cin, Hout, Wout, Cout, lattice, taps, pad_mode, bias=None, mask_src=None,
          res=None, res2=None, in_slope=1.0, mask_slope=1.0, out_div=1.0, out_slope=1.0):
    desc.x, desc.w, desc.out = lib.ptr(x), lib.ptr(w), lib.ptr(out)
    desc.bias = lib.ptr(bias, torch.float32) if bias is not None else None
    desc.mask_src = lib.ptr(mask_src) if mask_src is not None else None
    desc.res = lib.ptr(res) if res is not None else None
    desc.res2 = lib.ptr(res2) if res2 is not None else None
    desc.dtype = _DT[x.dtype]
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


def conv_forward(x, w, geom, bias=None, in_slope=1.0, res=None, res2=None, out_div=1.0, out_slope=1.0):
    """x [B,Hin,Win,Cin] -> [B,Hout,Wout,Cout];  w [kh*kw, Cout, Cin]."""
    _check(x, w, res, res2)
    B, Hin, Win, Cin = x.shape
    T, Cout, _ = w.shape
    assert (Hin, Win) == (geom.Hin, geom.Win) and T == geom.ntaps and w.shape[2] == Cin
    out = torch.empty((B, geom.Hout, geom.Wout, Cout), dtype=x.dtype, device=x.device)
    taps = [(ky * geom.dy, kx * geom.dx, ky * geom.kw + kx) for ky in range(geom.kh) for kx in range(geom.kw)]
    lattice = (geom.Hout, geom.Wout, 0, 1, 0, 1, geom.sy, geom.sx, -geom.py, -geom.px)
    d = _fill(lib.ConvDesc(), x, w, out, B, Hin, Win, Cin, geom.Hout, geom.Wout, Cout, lattice, taps,
              1 if geom.reflect else 0, bias=bias, res=res, res2=res2, in_slope=in_slope, out_div=out_div,
              out_slope=out_slope)
    lib.check(lib.get().msmc_conv_gather(ctypes.byref(d), lib.stream(x)), 'msmc_conv_gather')
    return out


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


def conv_dgrad(g, wb, geom, mask_src=None, mask_slope=1.0, res=None):
    """Data gradient of ``conv_forward``: g [B,Hout,Wout,Cout] -> gx [B,Hin(+2p),Win(+2p),Cin].

    wb [kh*kw, Cin, Cout] (channel roles swapped).  For reflect-padded convolutions the gradient is
    returned on the PADDED grid (Hin+2py, Win+2px); the caller folds the border back.
    Epilogue: gx = gx * lrelu'(mask_src) + res.
    """
    _check(g, wb, mask_src, res)
    B, Hout, Wout, Cout = g.shape
    T, Cin, _ = wb.shape
    assert (Hout, Wout) == (geom.Hout, geom.Wout) and T == geom.ntaps and wb.shape[2] == Cout
    if geom.reflect:
        Hx, Wx, py, px = geom.Hin + 2 * geom.py, geom.Win + 2 * geom.px, 0, 0
    else:
        Hx, Wx, py, px = geom.Hin, geom.Win, geom.py, geom.px
    gx = torch.empty((B, Hx, Wx, Cin), dtype=g.dtype, device=g.device)
    L = lib.get()
    for ry, ny, ty in _phases(Hx, geom.kh, geom.sy, geom.dy, py):
        for rx, nx, tx in _phases(Wx, geom.kw, geom.sx, geom.dx, px):
            if ny == 0 or nx == 0:
                continue
            taps = [(oy, ox, ky * geom.kw + kx) for oy, ky in ty for ox, kx in tx]
            lattice = (ny, nx, ry, geom.sy, rx, geom.sx, 1, 1, 0, 0)
            if not taps:                      # phase that no kernel tap reaches: gradient is the epilogue of zero
                gx[:, ry::geom.sy, rx::geom.sx] = 0 if res is None else res[:, ry::geom.sy, rx::geom.sx]
                continue
            d = _fill(lib.ConvDesc(), g, wb, gx, B, Hout, Wout, Cout, Hx, Wx, Cin, lattice, taps, 0,
                      mask_src=mask_src, mask_slope=mask_slope, res=res)
            lib.check(L.msmc_conv_gather(ctypes.byref(d), lib.stream(g)), 'msmc_conv_gather(dgrad)')
    return gx


def conv_transpose1d_forward(x, w, k, stride, padding, bias=None, in_slope=1.0):
    """x [B,1,Lin,Cin] -> [B,1,Lout,Cout];  w [k, Cout, Cin]   (torch ConvTranspose1d semantics)."""
    _check(x, w)
    B, _, Lin, Cin = x.shape
    Cout = w.shape[1]
    Lout = (Lin - 1) * stride - 2 * padding + k
    out = torch.empty((B, 1, Lout, Cout), dtype=x.dtype, device=x.device)
    L = lib.get()
    for r, n, taps1 in _phases(Lout, k, stride, 1, padding):
        assert taps1, 'kernel_size >= stride expected'
        taps = [(0, off, kk) for off, kk in taps1]
        lattice = (1, n, 0, 1, r, stride, 1, 1, 0, 0)
        d = _fill(lib.ConvDesc(), x, w, out, B, 1, Lin, Cin, 1, Lout, Cout, lattice, taps, 0, bias=bias,
                  in_slope=in_slope)
        lib.check(L.msmc_conv_gather(ctypes.byref(d), lib.stream(x)), 'msmc_conv_gather(convT)')
    return out


def conv_transpose1d_dgrad(g, wb, k, stride, padding, Lin, mask_src=None, mask_slope=1.0):
    """g [B,1,Lout,Cout] -> gx [B,1,Lin,Cin];  wb [k, Cin, Cout]:  gx[q] = sum_k g[q*stride + k - padding] wb[k]."""
    _check(g, wb, mask_src)
    B, _, Lout, Cout = g.shape
    Cin = wb.shape[1]
    gx = torch.empty((B, 1, Lin, Cin), dtype=g.dtype, device=g.device)
    taps = [(0, kk, kk) for kk in range(k)]
    lattice = (1, Lin, 0, 1, 0, 1, 1, stride, 0, -padding)
    d = _fill(lib.ConvDesc(), g, wb, gx, B, 1, Lout, Cout, 1, Lin, Cin, lattice, taps, 0, mask_src=mask_src,
              mask_slope=mask_slope)
    lib.check(lib.get().msmc_conv_gather(ctypes.byref(d), lib.stream(g)), 'msmc_conv_gather(convT dgrad)')
    return gx


def conv_wgrad(x, g, geom, n_slices, in_slope=1.0, dw=None, db=None):
    """dW [kh*kw, Cout, Cin] fp32 of ``conv_forward`` (x [B,Hin,Win,Cin] pre-activation, g [B,Hout,Wout,Cout]);
    when ``db`` (fp32 [Cout]) is given the bias gradient is accumulated into it by the same launch."""
    _check(x, g)
    B, Hin, Win, Cin = x.shape
    Cout = g.shape[3]
    assert g.shape[1:3] == (geom.Hout, geom.Wout)
    if dw is None:
        dw = torch.zeros((n_slices, Cout, Cin), dtype=torch.float32, device=x.device)
    taps = [(ky * geom.dy, kx * geom.dx, ky * geom.kw + kx) for ky in range(geom.kh) for kx in range(geom.kw)]
    lattice = (geom.Hout, geom.Wout, 0, 1, 0, 1, geom.sy, geom.sx, -geom.py, -geom.px)
    d = _fill(lib.ConvDesc(), x, x, x, B, Hin, Win, Cin, geom.Hout, geom.Wout, Cout, lattice, taps,
              1 if geom.reflect else 0, in_slope=in_slope)
    lib.check(lib.get().msmc_conv_wgrad(ctypes.byref(d), lib.ptr(g), lib.ptr(dw, torch.float32),
                                        lib.ptr(db, torch.float32) if db is not None else None, lib.stream(x)),
              'msmc_conv_wgrad')
    return dw


def conv_transpose1d_wgrad(x, g, k, stride, padding, in_slope=1.0, dw=None):
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
    d = _fill(lib.ConvDesc(), g, g, g, B, 1, Lout, Cout, 1, Lin, Cin, lattice, taps, 0, in_slope=1.0,
              mask_slope=in_slope)
    lib.check(lib.get().msmc_conv_wgrad(ctypes.byref(d), lib.ptr(x), lib.ptr(dw, torch.float32), None, lib.stream(x)),
              'msmc_conv_wgrad(convT)')
    return dw


def colsum(g2d):
    """g [rows, C] (fp32 / bf16) -> fp32 [C] column sums (bias gradient)."""
    rows, C = g2d.shape
    out = torch.empty(C, dtype=torch.float32, device=g2d.device)
    lib.check(lib.get().msmc_colsum(lib.ptr(g2d), lib.ptr(out), rows, C, _DT[g2d.dtype], lib.stream(g2d)),
              'msmc_colsum')
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
