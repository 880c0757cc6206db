"""Multi-tensor GAN loss terms over csrc/losses.hip: the feature-matching L1 sum and the LSGAN
MSE-to-constant sum of ``VQGANTrainer.train_step`` (reference msmctts/trainers/msmctts_trainer.py:165-171,
:187-193) as one forward and one backward launch each, instead of ~10 small kernels per tensor."""
import ctypes

import torch

from . import lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _dense(t):
    """True when the tensor's elements occupy one gap-free block (any permutation of a contiguous layout)."""
    if t.is_contiguous():
        return True
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1), reverse=True)
    expect = 1
    for st, sz in reversed(sizes_strides):
        if st != expect:
            return False
        expect *= sz
    return True


def _table(a_list, b_list=None, ga_list=None):
    tab = lib.TensorTable()
    assert 0 < len(a_list) <= lib.MAX_TENSORS
    tab.count = len(a_list)
    tab.dtype = _DT[a_list[0].dtype]
    for i, a in enumerate(a_list):
        assert a.dtype == a_list[0].dtype and _dense(a), 'loss operands must be dense and share one dtype'
        if not a.is_cuda and not lib._host_pointers_ok:
            raise RuntimeError('msmc HIP ops run on the GPU only; there is no CPU path')
        tab.a[i] = a.data_ptr()
        tab.n[i] = a.numel()
        if b_list is not None:
            b = b_list[i]
            assert b.dtype == a.dtype and b.shape == a.shape and b.stride() == a.stride()
            tab.b[i] = b.data_ptr()
        if ga_list is not None:
            tab.ga[i] = ga_list[i].data_ptr()
    return tab


class _L1Sum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, k, *tensors):
        a, b = list(tensors[:k]), list(tensors[k:])
        out = torch.empty(1, dtype=torch.float32, device=a[0].device)
        tab = _table(a, b)
        lib.check(lib.get().msmc_l1_multi_fwd(ctypes.byref(tab), lib.ptr(out), lib.stream(out)), 'msmc_l1_multi_fwd')
        ctx.k = k
        ctx.save_for_backward(*tensors)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        k = ctx.k
        tensors = ctx.saved_tensors
        a, b = list(tensors[:k]), list(tensors[k:])
        ga = [torch.empty_like(t) for t in a]          # preserve_format keeps the dense (permuted) strides
        tab = _table(a, b, ga)
        g = gout.reshape(1).float().contiguous()
        lib.check(lib.get().msmc_l1_multi_bwd(ctypes.byref(tab), lib.ptr(g), lib.stream(g)), 'msmc_l1_multi_bwd')
        return (None,) + tuple(ga) + (None,) * k


class _MseConstSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, target, *tensors):
        a = list(tensors)
        out = torch.empty(1, dtype=torch.float32, device=a[0].device)
        tab = _table(a)
        lib.check(lib.get().msmc_mse_const_multi_fwd(ctypes.byref(tab), float(target), lib.ptr(out), lib.stream(out)),
                  'msmc_mse_const_multi_fwd')
        ctx.target = float(target)
        ctx.save_for_backward(*tensors)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        a = list(ctx.saved_tensors)
        ga = [torch.empty_like(t) for t in a]
        tab = _table(a, None, ga)
        g = gout.reshape(1).float().contiguous()
        lib.check(lib.get().msmc_mse_const_multi_bwd(ctypes.byref(tab), ctx.target, lib.ptr(g), lib.stream(g)),
                  'msmc_mse_const_multi_bwd')
        return (None,) + tuple(ga)


def l1_sum(fake, real):
    """sum_i mean|fake_i - real_i| over two equally structured tensor lists (gradient flows to ``fake``)."""
    return _L1Sum.apply(len(fake), *fake, *[r.detach() for r in real])


def mse_const_sum(tensors, target):
    """sum_i mean((t_i - target)^2)  -- the LSGAN terms of the D and G steps."""
    return _MseConstSum.apply(float(target), *tensors)
