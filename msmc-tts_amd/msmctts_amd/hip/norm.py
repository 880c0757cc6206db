"""Fused element-wise / row-normalisation ops over csrc/norm.hip: the tails of the FFT-block sub-layers
(``layer_norm(dropout(h) + residual) * non_pad_mask``, reference acoustic_models/transformer.py:262-266, 318-323,
352-356), the WaveNet gate (vqgantts/modules.py:172-179) and the Tanh of the quantiser's 1x1 stacks
(vqgantts/msmc_vqgan.py:115-136) -- one launch per pass each instead of a chain of stock kernels.

Dropout masks are never stored: both passes derive them from (a seed word on the device, a per-call salt, the element
index).  ``advance_seed(device)`` bumps the seed word with one tiny kernel -- inside a captured hipGraph too, so every
replay draws fresh masks.
"""
import itertools

import ctypes
import os

import torch

from . import lib

_DT = {torch.float32: 0, torch.bfloat16: 1}
_SEEDS = {}
_SALT = itertools.count(1)


def seed_word(device):
    key = (device.type, device.index)
    t = _SEEDS.get(key)
    if t is None:
        t = _SEEDS[key] = torch.zeros(1, dtype=torch.int64, device=device)
        t.fill_(int(torch.initial_seed() & 0x7fffffff))
    return t


def advance_seed(device):
    """once per training step, before the first dropout of the step (captured into the graph in graph mode)"""
    seed_word(device).add_(1)


def new_salt():
    """a process-unique call-site id: two dropouts of one step never share a mask"""
    return next(_SALT)


def _rows(x):
    C = x.shape[-1]
    return x.numel() // C, C


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, gamma, beta, keep_row, p_drop, salt, eps, fc=None):
        assert x.dtype in _DT and x.is_contiguous() and (res is None or (res.dtype == x.dtype and res.is_contiguous()))
        N, C = _rows(x)
        y, v = torch.empty_like(x), torch.empty_like(x)
        mean = torch.empty(N, dtype=torch.float32, device=x.device)
        rstd = torch.empty(N, dtype=torch.float32, device=x.device)
        seed = seed_word(x.device) if p_drop > 0 else None
        if fc is not None:
            # x is the placeholder output of a deferred 1-tap projection (hip/convnet.py hip_conv_add_ln): this launch computes
            # the product a W^T + bias itself and never reads x
            a, w, bias = fc
            K = a.shape[-1]
            assert x.dtype == torch.bfloat16 and res is not None and a.is_contiguous() and a.dtype == x.dtype
            assert a.numel() == N * K and w.dtype == x.dtype and w.is_contiguous() and w.numel() == C * K
            lib.check(lib.get().msmc_fc_add_ln_fwd(lib.ptr(a), lib.ptr(w), lib.ptr(bias, torch.float32), lib.ptr(res),
                                                   lib.ptr(gamma, torch.float32), lib.ptr(beta, torch.float32),
                                                   lib.ptr(keep_row, torch.uint8), lib.ptr(y), lib.ptr(v), lib.ptr(mean),
                                                   lib.ptr(rstd), N, C, K, float(eps), float(p_drop), lib.ptr(seed), salt,
                                                   lib.stream(x)), 'msmc_fc_add_ln_fwd')
        else:
            lib.check(lib.get().msmc_add_ln_fwd(lib.ptr(x), lib.ptr(res), lib.ptr(gamma, torch.float32), lib.ptr(beta, torch.float32),
                                                lib.ptr(keep_row, torch.uint8), lib.ptr(y), lib.ptr(v), lib.ptr(mean), lib.ptr(rstd),
                                                N, C, float(eps), float(p_drop), lib.ptr(seed), salt, _DT[x.dtype], lib.stream(x)),
                      'msmc_add_ln_fwd')
        ctx.save_for_backward(v, mean, rstd, gamma, keep_row)
        ctx.p_drop, ctx.salt, ctx.has_res = float(p_drop), salt, res is not None
        # leaf parameters: their gradients can leave in the pass's ONE parameter-gradient launch (_ln_flush) instead of a launch
        # behind every LayerNorm backward; anything else (a non-leaf gamma) keeps autograd's own route
        leaf = lambda t: t.requires_grad and t.is_leaf and t.dtype == torch.float32 and t.is_contiguous()
        ctx.params = (gamma, beta) if (LN_PARAM_DEFER and leaf(gamma) and leaf(beta)) else None
        return y

    @staticmethod
    def backward(ctx, g):
        v, mean, rstd, gamma, keep_row = ctx.saved_tensors
        g = g.contiguous()
        N, C = _rows(v)
        L = lib.get()
        gx = torch.empty_like(v)
        gres = torch.empty_like(v) if ctx.has_res else None
        # the parameter gradients: both accumulated into .grad by this pass and deferrable -> partial sums now, ONE reduction
        # launch at the end of the pass (_ln_flush writes .grad); a pass restricted to other inputs (backward(inputs=[...]))
        # -> none delivered; torch.autograd.grad() (gradients travel on the edges, nothing is accumulated) or one of the two
        # only -> autograd's own route
        want_g, want_b, on_edges = ctx.needs_input_grad[2], ctx.needs_input_grad[3], False
        if ctx.params is not None:
            want_g, want_b = _ln_wanted(ctx.params[0], want_g), _ln_wanted(ctx.params[1], want_b)
        if want_g is None or want_b is None:
            want_g, want_b, on_edges = ctx.needs_input_grad[2], ctx.needs_input_grad[3], True
        defer = ctx.params is not None and want_g and want_b and not on_edges and _ln_defer_ok()
        dgamma = dbeta = None
        if not defer:
            dgamma = torch.empty(C, dtype=torch.float32, device=v.device)
            dbeta = torch.empty(C, dtype=torch.float32, device=v.device)
        nbytes = int(L.msmc_add_ln_bwd_workspace(N, C))
        ws = torch.empty(max(1, (nbytes + 3) // 4), dtype=torch.float32, device=v.device)
        seed = seed_word(v.device) if ctx.p_drop > 0 else None
        lib.check(L.msmc_add_ln_bwd(lib.ptr(g), lib.ptr(v), lib.ptr(mean), lib.ptr(rstd), lib.ptr(gamma, torch.float32),
                                    lib.ptr(keep_row, torch.uint8), lib.ptr(gx), lib.ptr(gres), lib.ptr(dgamma), lib.ptr(dbeta),
                                    lib.ptr(ws), ws.numel() * 4, N, C, ctx.p_drop, lib.ptr(seed), ctx.salt, 0, _DT[v.dtype],
                                    lib.stream(v)), 'msmc_add_ln_bwd')
        if defer:
            _ln_queue(ws, (N + 15) // 16, C, ctx.params[0], ctx.params[1])
        return gx, gres, (dgamma if want_g else None), (dbeta if want_b else None), None, None, None, None, None


# ---- parameter gradients of all LayerNorms of a backward pass in one launch --------------------------------------------
# Every LayerNorm backward leaves per-workgroup partial sums of dgamma / dbeta in its workspace; their reduction is a 10 us
# launch of 32 workgroups, and behind each of the 24 LayerNorms of the FFT stacks it sits on the critical path of the block
# chain.  With LN_PARAM_DEFER the workspaces wait until the end of the backward pass (an autograd-engine callback, as
# ConvBank's weight gradients do) and ONE msmc_add_ln_param_multi launch reduces them all, in the same fixed order; the
# gradients are then delivered to ``.grad`` directly (torch semantics: a live gradient is added to, in the kernel).
# ``.backward()`` and ``.backward(inputs=[...])`` see exactly what autograd's route gives them (a pass restricted to other
# inputs delivers nothing for the LayerNorm parameters); a ``torch.autograd.grad()`` pass accumulates nothing into ``.grad``
# -- its gradients are captured on the edges -- so there the parameter gradients take autograd's own route (``_ln_wanted``;
# tests/test_product_emu.py::test_layernorm_parameter_gradients_under_restricted_passes).
LN_PARAM_DEFER = os.environ.get('MSMC_LN_PARAM_DEFER', '1') != '0'
_LN_PENDING = {'task': None, 'items': []}


def _ln_wanted(p, needs):
    """will the running backward pass accumulate into ``p.grad``?  None: it is a torch.autograd.grad() pass (the engine refuses
    the question for leaves there: gradients are captured on the edges instead)"""
    if not needs:
        return False
    probe = getattr(torch._C, '_will_engine_execute_node', None)
    if probe is None or not p.is_leaf:
        return True
    try:
        return bool(probe(torch.autograd.graph.get_gradient_edge(p).node))
    except RuntimeError:
        return None


def _ln_defer_ok():
    return hasattr(torch._C, '_current_graph_task_id') and torch._C._current_graph_task_id() >= 0


def _ln_queue(ws, nblocks, C, gamma, beta):
    from torch.autograd import Variable
    task = torch._C._current_graph_task_id()
    if _LN_PENDING['task'] != task:          # (a pass that raised before its callback leaves nothing behind for the next)
        _LN_PENDING['task'], _LN_PENDING['items'] = task, []
        Variable._execution_engine.queue_callback(_ln_flush)
    _LN_PENDING['items'].append((ws, nblocks, C, gamma, beta))


def _ln_flush():
    from . import convnet
    items, _LN_PENDING['items'], _LN_PENDING['task'] = _LN_PENDING['items'], [], None
    while items:
        batch, rest, seen = [], [], set()
        for it in items:                      # a LayerNorm applied twice in one pass: its second reduction in a later launch
            key = it[3].data_ptr()
            (rest if key in seen else batch).append(it)
            seen.add(key)
        items = rest
        arr = (lib.LnParamItem * len(batch))()
        fresh = []
        with torch.no_grad():
            for slot, (ws, nblocks, C, gamma, beta) in zip(arr, batch):
                targets = []
                for p in (gamma, beta):
                    live = p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and \
                        p.grad.device == ws.device
                    t = p.grad if live else torch.empty(C, dtype=torch.float32, device=ws.device)
                    targets.append((p, t, live))
                slot.part, slot.nblocks, slot.C = ws.data_ptr(), nblocks, C
                slot.dgamma, slot.dbeta = targets[0][1].data_ptr(), targets[1][1].data_ptr()
                slot.accumulate = 1 if (targets[0][2] and targets[1][2]) else 0
                if targets[0][2] != targets[1][2]:        # (one of the two live: never in practice -- route both through buffers)
                    targets = [(p, torch.empty(C, dtype=torch.float32, device=ws.device), False) for p, _, _ in targets]
                    slot.dgamma, slot.dbeta, slot.accumulate = targets[0][1].data_ptr(), targets[1][1].data_ptr(), 0
                fresh.append(targets)
            dev = batch[0][0]
            lib.check(lib.get().msmc_add_ln_param_multi(arr, len(batch), lib.stream(dev)),
                      'msmc_add_ln_param_multi')
            for targets in fresh:
                for p, t, live in targets:
                    if not live:
                        if p.grad is None:
                            p.grad = t
                        else:
                            p.grad.add_(t.to(p.grad.dtype))
                    if convnet.GRAD_READY_HOOK is not None:
                        convnet.GRAD_READY_HOOK(p)


def add_layer_norm(x, res, gamma, beta, keep_row=None, p_drop=0.0, salt=0, eps=1e-5, fc=None):
    """LayerNorm(dropout(x) + res) * gamma + beta over the last axis, rows with ``keep_row == 0`` zeroed.
    x / res: same dtype (fp32 or bf16), contiguous; gamma / beta fp32; keep_row uint8 [rows] or None.
    ``fc = (a, w, bias)``: x is the not-yet-computed a w^T + bias of a deferred projection (hip/convnet.py hip_conv_add_ln)."""
    return _AddLayerNorm.apply(x, res, gamma, beta, keep_row, p_drop, salt, eps, fc)


def sum_n(tensors):
    """((t0 + t1) + t2) + t3 of two to four same-shaped contiguous tensors in ONE launch (fp32 sums, one rounding)"""
    ts = [t.contiguous() for t in tensors]
    assert 2 <= len(ts) <= 4 and all(t.shape == ts[0].shape and t.dtype == ts[0].dtype for t in ts)
    a = ts[0]
    if a.dtype not in _DT or a.numel() % 4 or any(t.data_ptr() % 16 for t in ts):
        out = ts[0] + ts[1]
        for t in ts[2:]:
            out = out + t
        return out
    out = torch.empty_like(a)
    ts = ts + [None] * (4 - len(ts))
    lib.check(lib.get().msmc_sum_n(lib.ptr(ts[0]), lib.ptr(ts[1]), lib.ptr(ts[2]), lib.ptr(ts[3]), lib.ptr(out), a.numel(),
                                   _DT[a.dtype], lib.stream(a)), 'msmc_sum_n')
    return out


class _DropoutAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, res, p_drop, salt):
        assert x.dtype in _DT and x.is_contiguous() and (res is None or (res.dtype == x.dtype and res.is_contiguous() and res.shape == x.shape))
        y = torch.empty_like(x)
        seed = seed_word(x.device) if p_drop > 0 else None
        lib.check(lib.get().msmc_dropout_add_fwd(lib.ptr(x), lib.ptr(res), lib.ptr(y), x.numel(), float(p_drop), lib.ptr(seed), salt,
                                                 _DT[x.dtype], lib.stream(x)), 'msmc_dropout_add_fwd')
        ctx.p_drop, ctx.salt, ctx.has_res = float(p_drop), salt, res is not None
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None, None, None
        g = g.contiguous()
        gx = g
        if ctx.p_drop > 0 and ctx.needs_input_grad[0]:
            gx = torch.empty_like(g)
            lib.check(lib.get().msmc_dropout_bwd(lib.ptr(g), lib.ptr(gx), g.numel(), ctx.p_drop, lib.ptr(seed_word(g.device)), ctx.salt,
                                                 _DT[g.dtype], lib.stream(g)), 'msmc_dropout_bwd')
        return (gx if ctx.needs_input_grad[0] else None), (g if ctx.has_res else None), None, None


def dropout_add_usable(x, res=None):
    return (x.dtype in _DT and x.numel() % 4 == 0 and (x.is_cuda or lib._host_pointers_ok) and
            (res is None or (res.dtype == x.dtype and res.shape == x.shape)))


def dropout_add(x, res, p_drop, salt):
    """dropout(x) + res (res may be None) with the counter-hash masks of this module: one launch forward, one backward"""
    return _DropoutAdd.apply(x.contiguous(), None if res is None else res.contiguous(), float(p_drop), salt)


def row_mask(lengths, T, dtype):
    """keep [B, T] in ``dtype``: 1 where t < lengths[b], 0 on padding (one launch; no gradient)"""
    assert dtype in _DT and lengths.dtype in (torch.int32, torch.int64)
    lengths = lengths.contiguous()
    B = lengths.numel()
    keep = torch.empty((B, T), dtype=dtype, device=lengths.device)
    lib.check(lib.get().msmc_row_mask(lib.ptr(lengths, lengths.dtype), int(lengths.dtype == torch.int64), lib.ptr(keep), B, int(T),
                                      _DT[dtype], lib.stream(lengths)), 'msmc_row_mask')
    return keep


class _Gate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, salt):
        assert x.dtype in _DT and x.is_contiguous() and x.shape[-1] % 2 == 0
        C = x.shape[-1] // 2
        N = x.numel() // (2 * C)
        y = torch.empty(x.shape[:-1] + (C,), dtype=x.dtype, device=x.device)
        seed = seed_word(x.device) if p_drop > 0 else None
        lib.check(lib.get().msmc_gate_fwd(lib.ptr(x), lib.ptr(y), N, C, float(p_drop), lib.ptr(seed), salt, _DT[x.dtype],
                                          lib.stream(x)), 'msmc_gate_fwd')
        ctx.save_for_backward(x)
        ctx.p_drop, ctx.salt = float(p_drop), salt
        return y

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        C = x.shape[-1] // 2
        N = x.numel() // (2 * C)
        gx = torch.empty_like(x)
        g = g.contiguous()                       # (bound to a name: a temporary would be freed before the launch reads it)
        seed = seed_word(x.device) if ctx.p_drop > 0 else None
        lib.check(lib.get().msmc_gate_bwd(lib.ptr(x), lib.ptr(g), lib.ptr(gx), N, C, ctx.p_drop, lib.ptr(seed),
                                          ctx.salt, _DT[x.dtype], lib.stream(x)), 'msmc_gate_bwd')
        return gx, None, None


def gate(x, p_drop=0.0, salt=0):
    """x [..., 2C] -> dropout(tanh(x[..., :C]) * sigmoid(x[..., C:]))"""
    return _Gate.apply(x, p_drop, salt)


class _FftPrologue(torch.autograd.Function):
    @staticmethod
    def forward(ctx, seq, lengths, table, out_dtype, Tp):
        assert seq.dtype in _DT and out_dtype in _DT and seq.is_contiguous() and table.dtype == torch.float32
        assert lengths.dtype in (torch.int32, torch.int64) and lengths.is_contiguous() and table.is_contiguous()
        B, T, C = seq.shape
        out = torch.empty((B, T, C), dtype=out_dtype, device=seq.device)
        keep_row = torch.empty(B * T, dtype=torch.uint8, device=seq.device)
        bias = torch.empty((B, Tp), dtype=torch.float32, device=seq.device) if Tp else None
        lib.check(lib.get().msmc_fft_prologue(lib.ptr(seq), lib.ptr(lengths, lengths.dtype), int(lengths.dtype == torch.int64),
                                              lib.ptr(table), table.shape[0], lib.ptr(out), lib.ptr(keep_row, torch.uint8),
                                              lib.ptr(bias), B, T, C, int(Tp), _DT[seq.dtype], _DT[out_dtype],
                                              lib.stream(seq)), 'msmc_fft_prologue')
        ctx.in_dtype = seq.dtype
        ctx.mark_non_differentiable(keep_row)
        ctx.set_materialize_grads(False)      # (the engine otherwise zero-fills a gradient for the mask and the bias: two launches)
        if bias is None:
            return out, keep_row
        ctx.mark_non_differentiable(bias)
        return out, keep_row, bias

    @staticmethod
    def backward(ctx, g, *_):
        if g is None:
            return None, None, None, None, None
        return (g if g.dtype == ctx.in_dtype else g.to(ctx.in_dtype)), None, None, None, None


def fft_prologue(seq, lengths, table, out_dtype, key_bias_width=0):
    """seq [B, T, C] + sinusoid-table rows of the positions 1 .. len (0 on padding), in ``out_dtype``; the row mask
    (uint8 [B T]) and, with ``key_bias_width`` = Tp > 0, the additive key-padding bias [B, Tp] of hip/attn.py -- one launch
    (msmc_fft_prologue) for the head of FFTBlocks.forward and the positions its callers build."""
    return _FftPrologue.apply(seq.contiguous(), lengths.contiguous(), table, out_dtype, int(key_bias_width))


class _Tanh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        assert x.dtype in _DT and x.is_contiguous()
        y = torch.empty_like(x)
        lib.check(lib.get().msmc_tanh_fwd(lib.ptr(x), lib.ptr(y), x.numel(), _DT[x.dtype], lib.stream(x)), 'msmc_tanh_fwd')
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        gx = torch.empty_like(y)
        g = g.contiguous()
        lib.check(lib.get().msmc_tanh_bwd(lib.ptr(y), lib.ptr(g), lib.ptr(gx), y.numel(), _DT[y.dtype],
                                          lib.stream(y)), 'msmc_tanh_bwd')
        return gx


def tanh(x):
    return _Tanh.apply(x)


class _TanhF32(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        assert x.dtype in _DT and x.is_contiguous()
        y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        lib.check(lib.get().msmc_tanh_f32_fwd(lib.ptr(x), lib.ptr(y), x.numel(), _DT[x.dtype], lib.stream(x)), 'msmc_tanh_f32_fwd')
        ctx.save_for_backward(y)
        ctx.dtype = x.dtype
        return y

    @staticmethod
    def backward(ctx, g):
        y, = ctx.saved_tensors
        gx = torch.empty(y.shape, dtype=ctx.dtype, device=y.device)
        g = g.contiguous().float()
        lib.check(lib.get().msmc_tanh_f32_bwd(lib.ptr(y), lib.ptr(g), lib.ptr(gx), y.numel(), _DT[ctx.dtype], lib.stream(y)),
                  'msmc_tanh_f32_bwd')
        return gx


def tanh_f32(x):
    """tanh(x) in fp32 for x in the compute dtype (the cast rides in the kernel, the gradient comes back in x's dtype)"""
    return _TanhF32.apply(x)
