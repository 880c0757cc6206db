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
        part = torch.empty(lib.get().msmc_loss_multi_parts(), dtype=torch.float32, device=a[0].device)
        lib.check(lib.get().msmc_l1_multi_fwd_ws(ctypes.byref(tab), lib.ptr(part), lib.ptr(out), lib.stream(out)),
                  'msmc_l1_multi_fwd_ws')
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
        part = torch.empty(lib.get().msmc_loss_multi_parts(), dtype=torch.float32, device=a[0].device)
        lib.check(lib.get().msmc_mse_const_multi_fwd_ws(ctypes.byref(tab), float(target), lib.ptr(part), lib.ptr(out),
                                                        lib.stream(out)), 'msmc_mse_const_multi_fwd_ws')
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


class _MseConstHalves(torch.autograd.Function):
    """LSGAN terms of a discriminator pass over a concatenated batch [first; second]: (sum_i mean((t_i[:B] - c0)^2),
    sum_i mean((t_i[B:] - c1)^2)) straight from the full tensors -- the halves are contiguous row ranges, so the tensor tables
    point into them and the backward pass writes both halves of ONE full-size gradient per tensor (slicing each score tensor
    first cost a concatenation per tensor in the backward pass: 10 launches per step)."""

    @staticmethod
    def forward(ctx, B, c0, c1, *tensors):
        outs = []
        L = lib.get()
        for half, target in ((0, c0), (1, c1)):
            part = [t[:B] if half == 0 else t[B:] for t in tensors]
            out = torch.empty(1, dtype=torch.float32, device=tensors[0].device)
            ws = torch.empty(L.msmc_loss_multi_parts(), dtype=torch.float32, device=tensors[0].device)
            tab = _table(part)
            lib.check(L.msmc_mse_const_multi_fwd_ws(ctypes.byref(tab), float(target), lib.ptr(ws), lib.ptr(out), lib.stream(out)),
                      'msmc_mse_const_multi_fwd_ws')
            outs.append(out.reshape(()))
        ctx.args = (B, float(c0), float(c1))
        ctx.save_for_backward(*tensors)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g0, g1):
        B, c0, c1 = ctx.args
        tensors = ctx.saved_tensors
        ga = [torch.empty_like(t) for t in tensors]
        L = lib.get()
        for half, target, g in ((0, c0, g0), (1, c1, g1)):
            rows = (lambda t: t[:B]) if half == 0 else (lambda t: t[B:])
            if g is None:
                for t in ga:
                    rows(t).zero_()
                continue
            tab = _table([rows(t) for t in tensors], None, [rows(t) for t in ga])
            gs = g.reshape(1).float().contiguous()
            lib.check(L.msmc_mse_const_multi_bwd(ctypes.byref(tab), target, lib.ptr(gs), lib.stream(gs)), 'msmc_mse_const_multi_bwd')
        return (None, None, None) + tuple(ga)


def mse_const_halves(tensors, B, c0, c1):
    """(sum_i mean((t_i[:B] - c0)^2), sum_i mean((t_i[B:] - c1)^2)) for score tensors of a [2B, ...] batch"""
    return _MseConstHalves.apply(int(B), float(c0), float(c1), *tensors)


def l1_sum(fake, real):
    """sum_i mean|fake_i - real_i| over two equally structured tensor lists (gradient flows to ``fake``)."""
    return _L1Sum.apply(len(fake), *fake, *[r.detach() for r in real])


def mse_const_sum(tensors, target):
    """sum_i mean((t_i - target)^2)  -- the LSGAN terms of the D and G steps."""
    return _MseConstSum.apply(float(target), *tensors)


class _MaskedMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, lengths, mode):
        assert a.dim() == 3 and a.dtype in _DT and a.is_contiguous(), 'masked_mean takes contiguous [B, T, C] tensors'
        assert b is None or (b.shape == a.shape and b.dtype in _DT and b.is_contiguous())
        assert lengths.dtype in (torch.int32, torch.int64) and lengths.is_contiguous() and lengths.numel() == a.shape[0]
        B, T, C = a.shape
        L = lib.get()
        part = torch.empty(L.msmc_masked_mean_parts(B), dtype=torch.float32, device=a.device)
        out = torch.empty(2, dtype=torch.float32, device=a.device)
        lib.check(L.msmc_masked_mean_fwd(lib.ptr(a), lib.ptr(b), lib.ptr(lengths), int(lengths.dtype == torch.int64), B, T, C,
                                         _DT[a.dtype], _DT[b.dtype] if b is not None else 0, mode, lib.ptr(part), lib.ptr(out),
                                         lib.stream(a)), 'msmc_masked_mean_fwd')
        ctx.save_for_backward(a, b, lengths, out)
        ctx.mode = mode
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        a, b, lengths, out = ctx.saved_tensors
        B, T, C = a.shape
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if (b is not None and ctx.needs_input_grad[1]) else None
        g = gout.reshape(1).float().contiguous()
        lib.check(lib.get().msmc_masked_mean_bwd(lib.ptr(a), lib.ptr(b), lib.ptr(lengths), int(lengths.dtype == torch.int64), B, T,
                                                 C, _DT[a.dtype], _DT[b.dtype] if b is not None else 0, ctx.mode, lib.ptr(out),
                                                 lib.ptr(g), lib.ptr(ga), lib.ptr(gb), lib.stream(a)), 'msmc_masked_mean_bwd')
        return ga, gb, None, None


def masked_mean(a, lengths, b=None):
    """sum over the valid rows (t < lengths[b]) of ``a`` -- or of ``(a - b)^2`` -- divided by ``lengths.sum() * C``:
    the length-masked scalar terms of the step (QuantizerLoss, frame loss, 'mse' embedding loss) in two launches."""
    return _MaskedMean.apply(a.contiguous(), None if b is None else b.contiguous(), lengths.contiguous(), 0 if b is None else 1)


class _TripleLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, trg, embed_t, enorm, margin, mean):
        N, D = p.shape
        H, K = enorm.shape
        lossh = torch.empty(N, H, dtype=torch.float32, device=p.device)
        gp = torch.empty(N, D, dtype=torch.float32, device=p.device)
        lib.check(lib.get().msmc_triple_loss(lib.ptr(p, torch.float32), lib.ptr(trg, torch.int64), lib.ptr(embed_t, torch.float32),
                                             lib.ptr(enorm, torch.float32), lib.ptr(lossh), lib.ptr(gp), N, D, H, K, float(margin),
                                             int(mean), lib.stream(p)), 'msmc_triple_loss')
        ctx.save_for_backward(gp)
        ctx.heads = H
        return lossh

    @staticmethod
    def backward(ctx, glossh):
        gp, = ctx.saved_tensors
        N, D = gp.shape
        H = ctx.heads
        g = glossh.reshape(N, H, 1).to(gp.dtype)
        return (gp.view(N, H, D // H) * g).view(N, D), None, None, None, None, None


def triple_loss(p, trg, embed_t, enorm, reduction='sum', margin=1e-6):
    """per-(frame, head) triple loss of predictions ``p`` [N, D] against target indices ``trg`` [N, H] and the prepared codebook
    (``hip/vq.py vq_prepare``: embed_t [H, K, d], enorm [H, K]) in one launch (msmc_triple_loss) -> [N, H]"""
    return _TripleLoss.apply(p.contiguous().float(), trg.contiguous().long(), embed_t, enorm, float(margin), reduction == 'mean')


def usable(*tensors):
    """the fused loss ops run on the GPU (or on the kernel interpreter in the CPU tests)"""
    return all(t is None or t.is_cuda or lib._host_pointers_ok for t in tensors)


class _WeightedSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weights, *terms):
        n = len(terms)
        ts = [t.reshape(1) if t.dtype == torch.float32 else t.float().reshape(1) for t in terms]
        out = torch.empty(1, dtype=torch.float32, device=ts[0].device)
        vp, fp = ctypes.c_void_p * n, ctypes.c_float * n
        lib.check(lib.get().msmc_scalar_wsum_fwd(vp(*[lib.ptr(t, torch.float32).value for t in ts]), fp(*weights), n, lib.ptr(out),
                                                 lib.stream(out)), 'msmc_scalar_wsum_fwd')
        ctx.weights = tuple(weights)
        return out.reshape(())

    @staticmethod
    def backward(ctx, gout):
        n = len(ctx.weights)
        g = gout.reshape(1).float().contiguous()
        gvec = torch.empty(n, dtype=torch.float32, device=g.device)
        lib.check(lib.get().msmc_scalar_wsum_bwd(lib.ptr(g), (ctypes.c_float * n)(*ctx.weights), n, lib.ptr(gvec), lib.stream(g)),
                  'msmc_scalar_wsum_bwd')
        return (None,) + tuple(gvec[i] for i in range(n))


def weighted_sum(terms, weights=None):
    """sum_i weights[i] * terms[i] for 0-dim loss tensors (weights: python floats, default 1) -- one launch (msmc_scalar_wsum_fwd)
    where the tensors live on the GPU / the interpreter is bound, the stock multiply-and-add chain otherwise"""
    terms = list(terms)
    weights = [1.0] * len(terms) if weights is None else [float(w) for w in weights]
    if (0 < len(terms) <= lib.MAX_TENSORS and all(torch.is_tensor(t) and t.numel() == 1 and t.dtype.is_floating_point for t in terms)
            and usable(*terms) and len(set(t.device for t in terms)) == 1):
        return _WeightedSum.apply(tuple(weights), *terms)
    out = 0
    for t, w in zip(terms, weights):
        out = out + (t if w == 1.0 else w * t)
    return out
