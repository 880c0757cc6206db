#!/usr/bin/env python
"""bench.py -- MSMC-VQ-GAN GAN-phase train step throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by ``torch.distributed.run`` (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment) or directly -- with WORLD_SIZE unset, ``--gpus N`` starts its own N ranks through ``torch.distributed.run`` on
127.0.0.1 (``self_spawn``), forwards their output and exit code, and kills the whole job after ``--job-timeout`` seconds.
Every rank runs a watchdog (``--stall-timeout``): a step or collective that does not finish turns into exit code 5, not a hang.
``--dry --backend gloo`` checks the launcher / process group / gradient reducer on CPU with a toy model (no GPU, not a benchmark).

One "step" = one full ``VQGANTrainer.train_step`` in the GAN phase (autoencoder forward, mel/STFT
loss, D step, G step, clipping, both optimizer steps, gradient all-reduces when N>1) on one
synthetic batch per rank (B=16/GPU, T=400, hop 300, 40-frame vocoder window).
Workload = BASELINE configs[1]: CSMSC msmc_vq_gan.yaml with 4 heads x 256 codewords, bf16 autocast
for the GEMM/conv bodies, fp32 VQ search (indices must be bit-exact).  Weak scaling: per-GPU batch
fixed.  Rank 0 prints ONE JSON line.

Besides the contract fields the line carries
  roofline      the dominant hand-written kernel inside the timed region (HIP events on the launch
                stream), algorithmic bytes / duration against the gfx950 HBM peak
  vq_microbench the "VQ argmin GB/s" half of the BASELINE metric: msmc_vq_search at N=2^20 frames
  cpu_baseline  the oracle (oracle/step.py, plain PyTorch fp32) timed on this box's host cores on a
                bounded sample of the same workload (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'msmc-tts_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import msmctts_amd  # noqa: E402,F401  (first: sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before any HIP call, see its docstring)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}     # dense peaks, same guide
FLOP_PER_STEP = 3.006e12       # reference GAN step at B=16, T=400 (torch FlopCounter)
FLOP_PER_STEP_ELIDED = 2.65e12  # without the discarded D weight-gradients of the G step


LINE_LIMIT = 8192             # the driver parses ONE JSON line from stdout; round 2's 36 KB line came back unparsed


def emit_line(out, kernels_path=None):
    """The one JSON line of the contract, at most LINE_LIMIT characters.  The per-kernel-symbol table (about 30 KB) goes
    to ``kernels_path`` (a JSON side file; the line only names it); should the line still be too long, the explanatory
    notes are dropped first, then the optional blocks, never the contract keys / roofline / cpu_baseline."""
    out = dict(out)
    kernels = out.pop('kernels', None)
    if kernels is not None:
        out['kernels_file'] = None
        if kernels_path:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(kernels_path)), exist_ok=True)
                with open(kernels_path, 'w') as f:
                    json.dump({'ms_per_step': out.get('ms_per_step'), 'kernels': kernels}, f, indent=1, sort_keys=True)
                out['kernels_file'] = os.path.relpath(kernels_path, ROOT)
            except OSError as e:
                sys.stderr.write('bench.py: could not write %s: %s\n' % (kernels_path, e))
    line = json.dumps(out)
    for drop in (('roofline', 'note'), ('roofline_step', 'note'), ('runtime', 'note'), ('step_flop_model',), ('warmup_phase', 'note'),
                 ('cpu_baseline', 'sample'), ('runtime',), ('losses',), ('vq_microbench',), ('warmup_phase',)):
        if len(line) <= LINE_LIMIT:
            break
        node = out
        for k in drop[:-1]:
            node = node.get(k) if isinstance(node, dict) else None
        if isinstance(node, dict) and drop[-1] in node:
            del node[drop[-1]]
            line = json.dumps(out)
    assert len(line) <= LINE_LIMIT, len(line)
    return line


def vq_bytes_per_frame(D, H):
    """read x, write quant, int64 indices, head-averaged diff."""
    return 4 * D + 4 * D + 8 * H + 4 * D // H


class KernelTimer(object):
    """Per-kernel timing of the hand-written launches through the library's own launch log (msmc_prof_*): every launch
    is bracketed by a HIP event pair recorded on the stream it is launched on and logged under the symbol rocprofv3
    prints.  The Python-level calls are wrapped only to attach their algorithmic flops / bytes to the launches they issue:
    a call's work goes to its convolution / search kernels (helper launches such as the partial-sum reduce are listed with
    zero work); when one grouped call resolves to several kernel instantiations its work is split in proportion to their
    measured durations."""
    HELPERS = ('conv_wgrad_reduce_kernel', 'colsum_kernel', 'reflect_fold', 'lrelu_bwd')

    def __init__(self):
        self.calls = []
        self.shapes = []                 # one description per entry of ``calls`` (layer shape: --calls-out)
        self.enabled = False
        self.lib = None

    def start(self, L):
        self.lib = L
        L.msmc_prof_enable(1)
        self.enabled = True

    def stop(self):
        self.enabled = False
        self.lib.msmc_prof_enable(0)

    def wrap(self, module, fn_name, label, work):
        inner = getattr(module, fn_name)
        timer = self

        def timed(*args, **kw):
            if not timer.enabled:
                return inner(*args, **kw)
            i0 = timer.lib.msmc_prof_count()
            out = inner(*args, **kw)
            timer.calls.append((i0, timer.lib.msmc_prof_count()) + tuple(work(*args, **kw)) + (True,))
            timer.shapes.append(fn_name + ' ' + describe_call(args, kw))
            return out

        setattr(module, fn_name, timed)

    def wrap_abi(self, L, name, work):
        """wrap one C-ABI entry point of the loaded library (the helper families: normalisation, attention, losses,
        spectral glue, VQ statistics ...): ``work(*args) -> (flops, bytes)`` from the call's own size arguments"""
        inner = getattr(L, name)
        timer = self

        def timed(*args):
            if not timer.enabled:
                return inner(*args)
            i0 = L.msmc_prof_count()
            rc = inner(*args)
            timer.calls.append((i0, L.msmc_prof_count()) + tuple(work(*args)) + (False,))
            timer.shapes.append(name)
            return rc

        setattr(L, name, timed)

    def note(self, name, i0, flops, byts):
        """work of the launches issued since record ``i0`` by a host-level call (bank / optimizer passes)"""
        if self.enabled:
            self.calls.append((i0, self.lib.msmc_prof_count(), flops, byts, False))
            self.shapes.append(name)

    def records(self):
        import ctypes
        n = self.lib.msmc_prof_count()
        buf, ms = ctypes.create_string_buffer(128), ctypes.c_float()
        recs = []
        for i in range(n):
            if self.lib.msmc_prof_read(i, buf, 128, ctypes.byref(ms)) != 0:
                raise RuntimeError('msmc_prof_read(%d) failed' % i)
            recs.append([buf.value.decode(), float(ms.value), 0.0, 0.0])
        return recs

    def by_call(self, steps):
        """per (entry point, layer shape): calls per step, summed launch time per step, kernels, work -- the table the
        per-symbol summary cannot give (which LAYERS a symbol's time belongs to)"""
        if self.lib is None:
            return []
        recs = self.records()
        out = {}
        for (i0, i1, flops, byts, _), shape in zip(self.calls, self.shapes):
            o = out.setdefault(shape, dict(call=shape, calls=0, ms=0.0, kernels={}, gflop=0.0, mbytes=0.0))
            o['calls'] += 1
            o['gflop'], o['mbytes'] = flops / 1e9, byts / 1e6
            for name, t, _, _ in recs[i0:i1]:
                o['ms'] += t
                o['kernels'][name] = o['kernels'].get(name, 0) + 1
        rows = sorted(out.values(), key=lambda o: -o['ms'])
        for o in rows:
            o['us_per_call'] = o['ms'] * 1e3 / o['calls']
            o['ms_per_step'] = o.pop('ms') / steps
            o['calls_per_step'] = o.pop('calls') / float(steps)
        return rows

    def summary(self):
        if self.lib is None:
            return {}
        recs = self.records()
        for i0, i1, flops, byts, conv_call in self.calls:
            # (inside a convolution entry point, helper launches -- partial-sum reduce, bias column sums -- take none of the
            # convolution's work; a helper family's own call shares its work among all of its launches)
            main = [r for r in recs[i0:i1] if not (conv_call and r[0].startswith(self.HELPERS))] or recs[i0:i1]
            tot = sum(r[1] for r in main)
            for r in main:
                share = r[1] / tot if tot > 0 else 1.0 / len(main)
                r[2] += flops * share
                r[3] += byts * share
        out = {}
        for name, t, f, b in recs:
            o = out.setdefault(name, dict(launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
            o['launches'] += 1
            o['total_ms'] += t
            o['flops'] += f
            o['bytes'] += b
        return out


def _val(v):
    return getattr(v, 'value', v)


def abi_work_models():
    """{C-ABI entry point: work(*args) -> (algorithmic flops, algorithmic bytes)} for the helper families of the step, from
    each call's own size arguments (include/msmc_hip.h gives the positions).  Bytes = every operand read or written once;
    element sizes from the call's dtype code (0 fp32, 1 bf16)."""
    E = lambda code: 4 if int(_val(code)) == 0 else 2
    HEAD = 64

    def spectral_multi_work(ops, n, st):
        """msmc_spectral_multi: the sum of its members' work, each priced like the single-stage entry point"""
        total = 0.0
        for i in range(int(_val(n))):
            o = ops[i]
            if o.kind in (0, 1):
                total += 4.0 * o.B * (o.L + o.T * o.NP)
            elif o.kind == 2:
                total += 4.0 * o.R * 3 * o.F
            elif o.kind == 3:
                total += 4.0 * o.R * 6 * o.F
            else:
                total += float(o.B * o.T * o.F * ((4 if o.kind == 4 else 8) + 2 * E(o.dtype)))
        return 0.0, total

    def table_bytes(tab_ref, passes):
        tab = tab_ref._obj
        e = E(tab.dtype)
        return sum(int(tab.n[i]) for i in range(tab.count)) * e * passes

    def fold(gp, mask, res, gx, B, H, W, C, n, p, slope, dt, st):
        e, tot = E(dt), 0
        for k in range(int(n)):
            b, h, w, c = int(B[k]), int(H[k]), int(W[k]), int(C[k])
            tot += b * c * ((h + 2 * p) * (w + 2 * p) + h * w * (1 + (1 if mask[k] else 0) + (1 if res is not None and res[k] else 0))) * e
        return 0.0, float(tot)

    def pending(items, n, st):          # second stage of the no-atomics weight gradients: 4 (splits + 1) |dW| bytes per record
        tot = 0
        for i in range(int(n)):
            r = items[i]
            tot += 4 * (int(r.nsplit) + 2) * (int(r.n_dw) + int(r.n_db))       # partials in, dW in and out
        return 0.0, float(tot)

    return {
        'msmc_add_ln_fwd': lambda x, res, g, b, keep, y, v, mean, rstd, N, C, eps, p, seed, salt, dt, st:
            (0.0, float(N * C * E(dt) * (3 + (1 if res else 0)) + 8 * N)),
        'msmc_sum_n': lambda a, b, c, d, out, n, dt, st: (0.0, float(n * E(dt) * (3 + (1 if c else 0) + (1 if d else 0)))),
        'msmc_dropout_add_fwd': lambda x, res, y, n, p, seed, salt, dt, st: (0.0, float(n * E(dt) * (2 + (1 if res else 0)))),
        'msmc_dropout_bwd': lambda g, gx, n, p, seed, salt, dt, st: (0.0, float(n * E(dt) * 2)),
        'msmc_row_mask': lambda lens, is64, keep, B, T, dt, st: (0.0, float(B * T * E(dt))),
        'msmc_fc_add_ln_fwd': lambda a, w, bias, res, g, b, keep, y, v, mean, rstd, N, C, K, eps, p, seed, salt, st:
            (2.0 * N * C * K, float(2 * (N * K + C * K + 3 * N * C) + 8 * N)),
        'msmc_add_ln_bwd': lambda g, v, mean, rstd, gamma, keep, gx, gres, dga, dbe, ws, wsb, N, C, p, seed, salt, acc, dt, st:
            (0.0, float(N * C * E(dt) * (3 + (1 if gres else 0)) + 8 * N)),
        'msmc_add_ln_param_multi': lambda items, n, st:
            (0.0, float(sum(items[i].nblocks * 2 * items[i].C * 4 + 2 * items[i].C * 4 for i in range(n)))),
        'msmc_attn_fwd': lambda qkv, bias, out, lse, B, T, H, Tp, *a:
            (4.0 * T * T * HEAD * B * H, float(B * T * H * (3 * HEAD + HEAD) * 2)),
        'msmc_attn_bwd': lambda qkv, bias, out, lse, dout, dqkv, dsum, B, T, H, Tp, *a:
            (10.0 * T * T * HEAD * B * H, float(B * T * H * (3 * HEAD + HEAD) * 2 * 2)),
        'msmc_fft_prologue': lambda seq, ln, l64, table, rows, out, keep, bias, B, T, C, Tp, idt, odt, st:
            (0.0, float(B * T * C * (E(idt) + E(odt) + 4))),
        'msmc_gate_fwd': lambda x, y, N, C, p, seed, salt, dt, st: (0.0, float(N * C * 3 * E(dt))),
        'msmc_gate_bwd': lambda x, g, gx, N, C, p, seed, salt, dt, st: (0.0, float(N * C * 5 * E(dt))),
        'msmc_tanh_fwd': lambda x, y, n, dt, st: (0.0, float(n * 2 * E(dt))),
        'msmc_tanh_bwd': lambda y, g, gx, n, dt, st: (0.0, float(n * 3 * E(dt))),
        'msmc_tanh_f32_fwd': lambda x, y, n, dt, st: (0.0, float(n * (4 + E(dt)))),
        'msmc_tanh_f32_bwd': lambda y, g, gx, n, dt, st: (0.0, float(n * (8 + E(dt)))),
        'msmc_window_gather': lambda s_, w, f, t, B, nf, hop, L, st: (0.0, float(B * nf * hop * 8 + B * nf * 8)),
        'msmc_lrelu_bwd': lambda g, y, gx, n, slope, dt, st: (0.0, float(n * 3 * E(dt))),
        'msmc_lrelu_bwd_multi': lambda g, y, gx, nelem, n, slope, dt, st:
            (0.0, float(sum(int(nelem[k]) for k in range(int(n))) * 3 * E(dt))),
        'msmc_reflect_fold_multi_res': fold,
        'msmc_reflect_fold_multi_tap': fold,
        'msmc_reflect_fold': lambda gp, mask, gx, B, H, W, C, p, slope, dt, st:
            (0.0, float(B * C * ((H + 2 * p) * (W + 2 * p) + H * W * (2 if mask else 1)) * E(dt))),
        'msmc_colsum': lambda g, out, rows, C, dt, st: (0.0, float(rows * C * E(dt) + 4 * C)),
        'msmc_colsum_ws': lambda g, out, rows, C, dt, acc, ws, wsb, st: (0.0, float(rows * C * E(dt) + 4 * C)),
        'msmc_triple_loss': lambda p, trg, et, en, lossh, gp, N, D, H, K, margin, mean, st: (4.0 * _val(N) * _val(D) * _val(K), 8.0 * _val(N) * _val(D)),
        'msmc_masked_mean_fwd': lambda a, b, ln, l64, B, T, C, adt, bdt, mode, part, out, st:
            (0.0, float(B * T * C * (E(adt) + (E(bdt) if mode else 0)))),
        'msmc_masked_mean_bwd': lambda a, b, ln, l64, B, T, C, adt, bdt, mode, out, gout, ga, gb, st:
            (0.0, float(B * T * C * ((E(adt) + E(bdt)) * (1 if mode else 0) + (E(adt) if ga else 0) + (E(bdt) if gb else 0)))),
        'msmc_conv_wgrad_reduce_pending': pending,
        'msmc_spectral_multi': spectral_multi_work,
        'msmc_stft_frames_fwd': lambda x, fr, B, L, T, n_fft, NP, hop, pad, st: (0.0, 4.0 * B * (L + T * NP)),
        'msmc_stft_frames_bwd': lambda gf, gx, B, L, T, n_fft, NP, hop, pad, st: (0.0, 4.0 * B * (L + T * NP)),
        'msmc_spec_mag_fwd': lambda spec, mag, R, F, CP, FP, lo, mode, st: (0.0, 4.0 * R * (2 * F + F)),
        'msmc_spec_mag_bwd': lambda spec, mag, gmag, gspec, R, F, CP, FP, lo, mode, st: (0.0, 4.0 * R * (2 * F + F + F + 2 * F)),
        'msmc_mrd_image_fwd_dt': lambda mel, img, B, T, F, FP, dt, st: (0.0, float(B * T * F * (4 + 2 * E(dt)))),
        'msmc_mrd_image_bwd_dt': lambda mel, gimg, gmel, B, T, F, FP, dt, st: (0.0, float(B * T * F * (8 + 2 * E(dt)))),
        'msmc_mrd_image_fwd': lambda mel, img, B, T, F, FP, st: (0.0, 12.0 * B * T * F),
        'msmc_mrd_image_bwd': lambda mel, gimg, gmel, B, T, F, FP, st: (0.0, 16.0 * B * T * F),
        'msmc_wave_fan_fwd': lambda y, copies, plen, n, B, L, dt, st:
            (0.0, float(B * (4 * L + E(dt) * sum(int(plen[k]) for k in range(int(n)))))),
        'msmc_wave_fan_bwd': lambda g32, n32, gc, plen, n, gy, B, L, dt, st:
            (0.0, float(B * (4 * L * (1 + sum(1 for k in range(int(n32)) if g32[k])) +
                             E(dt) * sum(int(plen[k]) for k in range(int(n)) if gc[k])))),
        'msmc_scalar_wsum_fwd': lambda terms, w, n, out, st: (0.0, 4.0 * (n + 1)),
        'msmc_scalar_wsum_bwd': lambda gout, w, n, gvec, st: (0.0, 4.0 * (n + 1)),
        'msmc_log_clamp_fwd': lambda x, y, n, lo, st: (0.0, 8.0 * n),
        'msmc_log_clamp_bwd': lambda x, g, gx, n, lo, st: (0.0, 12.0 * n),
        'msmc_l1_multi_fwd': lambda tab, out, st: (0.0, float(table_bytes(tab, 2))),
        'msmc_l1_multi_bwd': lambda tab, gout, st: (0.0, float(table_bytes(tab, 3))),
        'msmc_l1_multi_fwd_ws': lambda tab, part, out, st: (0.0, float(table_bytes(tab, 2))),
        'msmc_mse_const_multi_fwd_ws': lambda tab, target, part, out, st: (0.0, float(table_bytes(tab, 1))),
        'msmc_mse_const_multi_fwd': lambda tab, target, out, st: (0.0, float(table_bytes(tab, 1))),
        'msmc_mse_const_multi_bwd': lambda tab, target, gout, st: (0.0, float(table_bytes(tab, 2))),
        'msmc_vq_prepare': lambda e, et, en, H, d, K, st: (0.0, 4.0 * H * d * K * 2 + 4.0 * H * K),
        'msmc_vq_prepare_shortlist': lambda et, en, img, H, d, K, st: (0.0, 4.0 * H * d * K + 4.0 * H * d * K),
        'msmc_vq_ema_update': lambda x, ind, ln, emb, cs, ea, ws, wsb, B, T, D, H, K, decay, eps, st:
            (2.0 * B * T * D, float(B * T * (4 * D + 8 * H) + 4 * D * K * 4)),
        'msmc_vq_ema_stats': lambda x, ind, ln, stats, ws, wsb, B, T, D, H, K, st:
            (2.0 * B * T * D, float(B * T * (4 * D + 8 * H) + 4 * D * K)),
        'msmc_vq_ema_apply': lambda stats, emb, cs, ea, D, H, K, decay, eps, st: (0.0, 4.0 * D * K * 4),
        'msmc_vq_backward': lambda gq, gd, x, q, gx, N, D, H, st: (0.0, 16.0 * N * D + (4.0 * N * D / H if gd else 0.0)),
    }


def describe_call(args, kw):
    """shapes of a convolution entry point's operands as one short string (tensors: shape, geometry objects: their
    lattice, grouped calls: every member)"""
    def one(v):
        if torch.is_tensor(v):
            return 'x'.join(str(int(n)) for n in v.shape)
        if isinstance(v, (list, tuple)) and v and isinstance(v[0], dict):
            return '[' + ' | '.join(describe_call((), it) for it in v) + ']'
        if isinstance(v, (int, float, bool)) or v is None:
            return str(v)
        fields = [f for f in ('kh', 'kw', 'sy', 'sx', 'dy', 'dx', 'Hout', 'Wout') if hasattr(v, f)]
        if fields:
            return 'geom(' + ','.join('%s=%s' % (f, getattr(v, f)) for f in fields) + ')'
        return type(v).__name__
    parts = [one(v) for v in args]
    parts += ['%s=%s' % (k, one(v)) for k, v in kw.items() if k in ('x', 'w', 'g', 'wb', 'geom', 'n_slices') or
              isinstance(v, (int, float))]
    return ' '.join(parts)


def build(args, device, rank, world):
    from msmctts_amd.configs import csmsc_config
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    model_kw = getattr(args, 'model_kw', None) or dict(n_heads=args.heads, embedding_sizes=args.codewords)    # (tools/ pass bare namespaces)
    cfg = Config(csmsc_config(batch_size=args.batch, warmup_steps=0, **model_kw))
    torch.manual_seed(cfg.seed)
    task = build_task(cfg, mode='train')
    trainer = build_trainer(cfg, task, num_gpus=world, rank=rank)      # moves to GPU; arms RCCL reducer if world>1
    trainer.optimizer = build_optimizer(trainer.model, cfg.optimizer, capturable=args.graph)
    trainer.use_graphs = args.graph
    trainer.graph_exchange = getattr(args, 'exchange', 'serial')
    trainer.amp_dtype = torch.bfloat16 if args.dtype == 'bf16' else None
    trainer.amp_autocast = not args.no_autocast
    trainer.model.train()
    return cfg, trainer


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def say(msg):
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.perf_counter() - T_START, msg))
    sys.stderr.flush()


T_START = time.perf_counter()


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except Exception:
        pass
    return 'unknown'


def cpu_baseline(cfg, state_dict, batch, windows, steps, warmups, threads, sample):
    """Oracle train step on host cores: same weights, the first ``sample`` utterances of the same batch
    (default: all of it), same windows; ``warmups`` untimed + ``steps`` timed steps, median, for the GAN phase and the
    warm-up phase (no vocoder / discriminator).  Returns {phase: (seconds per step, mel frames per step)}."""
    from oracle.step import OracleTrainer
    torch.set_num_threads(threads)
    task = cfg.task.to_dict()
    cb = {k: v.detach().cpu()[:sample] for k, v in batch.items() if torch.is_tensor(v)}
    T = int(cb['mel_length'].max())
    cb['mel'], cb['wav'] = cb['mel'][:, :T], cb['wav'][:, :T * 300]
    windows = (windows[0][:sample], windows[1][:sample])
    out = {}
    for phase, wsteps in (('gan', 0), ('warmup', 10 ** 9)):
        tcfg = {k: v for k, v in cfg.trainer.to_dict().items() if k != '_name'}
        tcfg['warmup_steps'] = wsteps
        tr = OracleTrainer({k: v.detach().float().cpu() for k, v in state_dict.items()}, task, tcfg)
        times = []
        for i in range(warmups + steps):
            t0 = time.perf_counter()
            tr.train_step(cb, 10 + i, windows=windows)
            times.append(time.perf_counter() - t0)
            say('cpu oracle %s step %d: %.2f s' % (phase, i, times[-1]))
        timed = sorted(times[warmups:])
        out[phase] = (timed[len(timed) // 2], float(cb['mel_length'].sum()))
    return out


def vq_microbench(device, H, K, D=256, N=1 << 20, iters=20, shortlist=None):
    """msmc_vq_search[_shortlist] at N = 2^20 frames.  ``shortlist``: None = what the product runs for this
    shape (the bf16-shortlist kernel where it applies, bit-identical results), False = the exact register-resident kernel."""
    from msmctts_amd.hip import lib, vq
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(N, D, generator=g).to(device)
    embed = torch.randn(H, D // H, K, generator=g).to(device)
    et, en = vq.vq_prepare(embed)
    for _ in range(3):
        vq.vq_search(x, et, en, shortlist=shortlist)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        vq.vq_search(x, et, en, shortlist=shortlist)
    e.record()
    torch.cuda.synchronize()
    # exact-path counters in an untimed launch of their own (the timed launches run as the product does: no counters --
    # tens of thousands of atomics on one address would be the slowest thing in them)
    vq.SLOW_COUNT = torch.zeros(2, dtype=torch.int64, device=device)
    try:
        vq.vq_search(x, et, en, shortlist=shortlist)
        torch.cuda.synchronize()
        slow = [float(v) for v in vq.SLOW_COUNT.tolist()]
    finally:
        vq.SLOW_COUNT = None
    ms = s.elapsed_time(e) / iters          # includes three small output allocations per call (cached allocator)
    byts = N * vq_bytes_per_frame(D, H)
    r4 = lambda v: float('%.4g' % v)
    out = dict(kernel=lib.get().msmc_vq_last_kernel().decode(), N=N, D=D, H=H, K=K, ms=r4(ms), GBps=r4(byts / ms / 1e6),
               frac_hbm=r4(byts / ms / 1e6 / HBM_PEAK_GBS), Mframes_per_s=r4(N / ms / 1e3),
               fp32_TFLOPs=r4(2.0 * K * D * N / ms / 1e9))
    if out['kernel'] == 'vq_search_sl_kernel':
        tiles = (N + 15) // 16 * H          # fractions of the 16-frame tiles (per head) that left the shortlist path
        out.update(exact_rerank_frac=r4(slow[0] / tiles), exact_research_frac=r4(slow[1] / tiles))
    return out


def make_timer():
    """KernelTimer with a work model attached to every entry point of the library a train step goes through (algorithmic
    flops / bytes per call from the call's own arguments); ``register_banks()`` must run once the model's banks exist"""
    from msmctts_amd.hip import lib, vq as hipvq
    timer = KernelTimer()
    from msmctts_amd.hip import conv as hipconv
    def esz(t):
        return t.element_size()

    def vq_work(x, et, en):
        n = x.numel() // x.shape[-1]
        return 2.0 * n * x.shape[-1] * et.shape[1], n * vq_bytes_per_frame(x.shape[-1], et.shape[0])

    timer.wrap(hipvq, 'vq_search', lambda: 'vq_search_reg_kernel', lambda x, et, en, shortlist=None: vq_work(x, et, en))

    # algorithmic work of one call: flops = 2 * output points * Cout * Cin * taps; bytes = every operand once
    def conv_work(x, w, geom, bias=None, in_slope=1.0, res=None, res2=None, **k):
        pts = x.shape[0] * geom.Hout * geom.Wout
        extra = sum(1 for t in (res, res2) if t is not None)
        return (2.0 * pts * w.shape[1] * w.shape[2] * w.shape[0],
                (x.numel() + pts * w.shape[1] * (1 + extra) + w.numel()) * esz(x))

    def dgrad_work(g, wb, geom, mask_src=None, mask_slope=1.0, res=None, **k):
        pts = g.shape[0] * geom.Hout * geom.Wout
        Hx, Wx = geom.dgrad_plan()[:2]
        nx = g.shape[0] * Hx * Wx * wb.shape[1]
        return (2.0 * pts * wb.shape[1] * wb.shape[2] * wb.shape[0],
                (g.numel() + nx * (2 if mask_src is not None else 1) + wb.numel()) * esz(g))

    def wgrad_work(x, g, geom, n_slices, *a, **k):       # (also called with the keyword items of conv_wgrad_group)
        pts = x.shape[0] * geom.Hout * geom.Wout
        return (2.0 * pts * g.shape[3] * x.shape[3] * n_slices,
                (x.numel() + g.numel()) * esz(x) + 8.0 * n_slices * g.shape[3] * x.shape[3])

    def convt_work(x, w, kk, stride, padding, *a, **k):   # every input pixel meets every tap once
        Lout = (x.shape[2] - 1) * stride - 2 * padding + kk
        return (2.0 * x.shape[0] * x.shape[2] * w.shape[1] * w.shape[2] * kk,
                (x.numel() + x.shape[0] * Lout * w.shape[1] + w.numel()) * esz(x))

    def convt_dgrad_work(g, wb, kk, stride, padding, Lin, mask_src=None, **k):
        nx = g.shape[0] * Lin * wb.shape[1]
        return (2.0 * g.shape[0] * Lin * wb.shape[1] * wb.shape[2] * kk,
                (g.numel() + nx * (2 if mask_src is not None else 1) + wb.numel()) * esz(g))

    def convt_wgrad_work(x, g, kk, stride, padding, *a, **k):
        return (2.0 * x.shape[0] * x.shape[2] * x.shape[3] * g.shape[3] * kk,
                (x.numel() + g.numel()) * esz(x) + 8.0 * kk * x.shape[3] * g.shape[3])

    def last_kernel():
        return lib.get().msmc_conv_last_kernel().decode()

    def group_work(one):
        def work(items):
            f = b = 0.0
            for it in items:
                df, db_ = one(**it)
                f, b = f + df, b + db_
            return f, b
        return work

    def split_gemm_work(x, wimg, cout):                   # the THREE bf16 products it executes (priced against the bf16 peak);
        m = x.numel() // x.shape[-1]                      # x, out and the matrix image once
        return 6.0 * m * x.shape[-1] * cout, 4.0 * (x.numel() + m * cout) + 2.0 * wimg.numel()

    timer.wrap(hipconv, 'const_gemm_split', last_kernel, split_gemm_work)

    def split_gemm_group_work(xs, wimgs, couts):          # the members of one grouped call (round 6: the resolution front-ends)
        works = [split_gemm_work(x, w, c) for x, w, c in zip(xs, wimgs, couts)]
        return sum(f for f, _ in works), sum(b for _, b in works)

    timer.wrap(hipconv, 'const_gemm_split_group', last_kernel, split_gemm_group_work)
    for fn, work in (('conv_forward', conv_work), ('conv_dgrad', dgrad_work), ('conv_wgrad', wgrad_work),
                     ('conv_transpose1d_forward', convt_work), ('conv_transpose1d_dgrad', convt_dgrad_work),
                     ('conv_transpose1d_wgrad', convt_wgrad_work)):
        timer.wrap(hipconv, fn, last_kernel, work)
    # grouped calls (several members per launch) are attributed to the symbol of their last launch
    for fn, one in (('conv_forward_group', conv_work), ('conv_dgrad_group', dgrad_work), ('conv_wgrad_group', wgrad_work)):
        timer.wrap(hipconv, fn, last_kernel, group_work(one))

    # every other hand-written kernel: C-ABI level (sizes are the call's own arguments)
    L0 = lib.get()
    for name, work in abi_work_models().items():
        timer.wrap_abi(L0, name, work)
    # weight-norm passes and the fused optimizer take device tables: their sizes come from the objects that own the tables
    wn_elems, chunk = {}, L0.msmc_opt_chunk()

    def wn_bytes(per_elem_of):
        def work(items, nitems, *rest):
            n, e = wn_elems.get(_val(items), (0, 2))
            return 0.0, float(n * per_elem_of(e))
        return work
    # prepare: v read once, both kernel layouts written; backward: dW read and re-zeroed, v read, gradient written
    timer.wrap_abi(L0, 'msmc_wn_prepare_multi_tiles', wn_bytes(lambda e: 4 + 2 * e))
    timer.wrap_abi(L0, 'msmc_wn_backward_multi_rows', wn_bytes(lambda e: 16))
    timer.wrap_abi(L0, 'msmc_opt_clip_adamw', lambda table, nt, nblocks, max_norm, *rest:
                   (0.0, float(nblocks) * chunk * (28 + (4 if max_norm > 0 else 0))))

    def register_banks():
        import gc
        from msmctts_amd.hip.convnet import ConvBank
        for o in gc.get_objects():
            if isinstance(o, ConvBank) and getattr(o, 'items_dev', None) is not None:
                wn_elems[o.items_dev.data_ptr()] = (o.w1.numel(), o.w1.element_size())
        from msmctts_amd.hip import convnet as _cn
        for hit in _cn._TOGETHER.values():           # (several banks refreshed by one call: hip/convnet.py prepare_together)
            wn_elems[hit[0].data_ptr()] = (hit[6], hit[7])

    return timer, register_banks


def summarize_kernels(timer, dtype, nsteps, ms_per_step):
    """per-symbol table, the dominant kernel's roofline object and the step-level figure from the instrumented steps"""
    ks = timer.summary()
    roof, kernels = None, {}
    mfma_peak = MFMA_PEAK_TFLOPS[dtype]
    nst = max(1, nsteps)
    for label, rec in ks.items():
        # fp32 kernels (spectral DFT projections, VQ search) are priced against the fp32 MFMA peak
        peak = MFMA_PEAK_TFLOPS['fp32'] if ('float' in label or label.startswith('vq_')) else mfma_peak
        t_mfma, t_hbm = rec['flops'] / (peak * 1e12), rec['bytes'] / (HBM_PEAK_GBS * 1e9)
        sec = max(rec['total_ms'] * 1e-3, 1e-12)
        kernels[label] = dict(launches=rec['launches'], avg_us=rec['total_ms'] * 1e3 / rec['launches'],
                              ms_per_step=rec['total_ms'] / nst, tflops=rec['flops'] / sec / 1e12,
                              gbps=rec['bytes'] / sec / 1e9, bound='mfma' if t_mfma >= t_hbm else 'hbm',
                              mfma_peak_tflops=peak, frac_mfma=t_mfma / sec, frac_hbm=t_hbm / sec,
                              bytes_per_launch=rec['bytes'] / rec['launches'],
                              flops_per_launch=rec['flops'] / rec['launches'])
    step_roof = None
    if kernels:
        # step-level figure: the time the step's launches would take at their rooflines / the time they took
        t_roof = sum(max(rec['flops'] / ((MFMA_PEAK_TFLOPS['fp32'] if ('float' in label or label.startswith('vq_')) else mfma_peak) * 1e12),
                         rec['bytes'] / (HBM_PEAK_GBS * 1e9)) for label, rec in ks.items()) / nst * 1e3
        t_all = sum(rec['total_ms'] for rec in ks.values()) / nst
        t_attr = sum(rec['total_ms'] for rec in ks.values() if rec['flops'] > 0 or rec['bytes'] > 0) / nst
        step_roof = dict(roofline_ms_per_step=t_roof, kernel_ms_per_step=t_all, frac=t_roof / max(t_all, 1e-9),
                         attributed_time_frac=t_attr / max(t_all, 1e-9),
                         launches_per_step=sum(rec['launches'] for rec in ks.values()) / float(nst),
                         frac_of_step_time=t_roof / max(ms_per_step, 1e-9),
                         note='sum over every hand-written launch of max(flops / dense MFMA peak of its dtype, bytes / 8 TB/s) '
                              'divided by the summed launch durations (HIP events, instrumented single-stream steps); '
                              'frac_of_step_time divides by the timed step instead (shorter than the sum: the step runs parallel branches)')
        label = max((k for k in kernels if kernels[k]['flops_per_launch'] > 0), key=lambda k: kernels[k]['ms_per_step'])
        k = kernels[label]
        traffic = mfma_util = None
        try:
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')))
            traffic = pmc['kernels'][label]['hbm_bytes_per_launch']
            mfma_util = pmc['kernels'][label].get('mfma_util_percent')
        except Exception:
            pass
        if k['bound'] == 'mfma':
            roof = dict(bound='mfma', achieved=k['tflops'], peak=k['mfma_peak_tflops'], unit='TFLOP/s',
                        frac=k['frac_mfma'], traffic=traffic)
        else:
            roof = dict(bound='hbm', achieved=k['gbps'], peak=HBM_PEAK_GBS, unit='GB/s', frac=k['frac_hbm'],
                        traffic=traffic)
        roof.update(mfma_util_percent_pmc=mfma_util, kernel=label, launches=k['launches'], avg_us=k['avg_us'], ms_per_step=k['ms_per_step'],
                    bytes_per_launch=k['bytes_per_launch'], flops_per_launch=k['flops_per_launch'])
        roof['note'] = ('dominant hand-written kernel by summed HIP-event time (one event pair per launch, recorded on the '
                        'launch stream by the library: msmc_prof_*) over %d instrumented single-stream steps; achieved = '
                        'algorithmic flops (or bytes) of its launches / their summed durations; bound = the roofline '
                        'that prices those launches higher; traffic = PMC HBM bytes per launch from '
                        'profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command), null if not collected'
                        % nst)
    return kernels, roof, step_roof


def build_predictor(args, device):
    """trainer, predictor task, frozen autoencoder task, their configurations and one synthetic batch (device / host copies) of
    BASELINE configuration #4 (also used by tools/tune_bench_shapes.py CONFIG=4)"""
    from msmctts_amd.configs import am_config, csmsc_config
    from msmctts_amd.synthetic import make_batch, make_text_batch
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    model_kw = getattr(args, 'model_kw', None) or dict(n_heads=args.heads, embedding_sizes=args.codewords)
    acfg = Config(csmsc_config(batch_size=args.batch, warmup_steps=0, **model_kw))
    torch.manual_seed(acfg.seed)
    atask = build_task(acfg, mode='train').to(device).eval()
    cfg = Config(am_config(batch_size=args.batch))
    torch.manual_seed(cfg.seed + 1)
    task = build_task(cfg, mode='train').to(device).train()
    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    tr.amp_dtype = torch.bfloat16 if args.dtype == 'bf16' else None
    tr.use_graphs = bool(getattr(args, 'graph', False))
    for m in atask.modules():                       # (the frozen analysis pass computes in the same type)
        if hasattr(m, 'hip_dtype'):
            m.hip_dtype = tr.amp_dtype or torch.float32
    cpu_batch = make_batch(args.batch, args.frames, 80, 300, seed=1234, rank=0, device='cpu')
    cpu_batch.pop('wav')
    cpu_batch.pop('wav_length')
    cpu_batch.update(make_text_batch(cpu_batch['mel_length'].tolist()))
    batch = {k: v.to(device) for k, v in cpu_batch.items()}
    return tr, task, atask, cfg, acfg, batch, cpu_batch


def bench_predictor(args, device, wd):
    """--config 4 (BASELINE.json: CSMSC msmc_vq_gan_am.yaml, predictor training, one GPU): one step = one
    ``PredictorTrainer.train_step`` (reference msmctts_trainer.py:222-286) -- analysis of the mel batch by the frozen autoencoder
    (configuration #2's architecture, random weights, eval mode), text -> per-stage predictions through the 600-wide FFT stacks,
    'mse' + 'triple_sum' embedding losses and the duration loss, backward, clipping, Adam.  Same line layout as the headline:
    value = mel frames of the batch per second, roofline of the dominant hand-written kernel from the instrumented steps,
    cpu_baseline = oracle/predictor.py (plain PyTorch fp32) on the first utterances of the same batch."""
    from msmctts_amd.configs import BASELINE_CONFIGS, am_config
    from msmctts_amd.hip import convnet, lib
    preset = BASELINE_CONFIGS[4]
    tr, task, atask, cfg, acfg, batch, cpu_batch = build_predictor(args, device)
    frames_per_step = float(cpu_batch['mel_length'].sum())
    timer, register_banks = make_timer()

    def step(i):
        if not tr.use_graphs:
            task.zero_grad()
        return tr.train_step(batch, i)
    say('built predictor + frozen autoencoder; starting warm-up')
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        wd.beat('predictor warm-up %d' % i)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        log = step(args.warmup + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    wd.beat('predictor timed steps')
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_per_step = elapsed / args.steps * 1e3
    ms_instr = None
    graphs_were = tr.use_graphs
    if args.kernel_timing_steps > 0:
        tr.use_graphs = False                       # per-launch HIP events need the eager path ...
        convnet.STREAMS_ENABLED = False             # ... and one stream: an event pair brackets exactly one kernel
        step(0)
        register_banks()
        timer.start(lib.get())
        t1 = time.perf_counter()
        for i in range(args.kernel_timing_steps):
            torch.cuda._sleep(int(2.0e8))
            step(1 + i)
        torch.cuda.synchronize()
        ms_instr = (time.perf_counter() - t1) / args.kernel_timing_steps * 1e3
        timer.stop()
        convnet.STREAMS_ENABLED = True
        tr.use_graphs = graphs_were
        wd.beat('predictor instrumented steps')
    kernels, roof, step_roof = summarize_kernels(timer, args.dtype, args.kernel_timing_steps, ms_per_step)
    out = {
        'metric': 'mel-frames/sec multi-stage predictor train step (BASELINE config #4)', 'value': frames_per_step / (elapsed / args.steps),
        'unit': 'mel-frames/s', 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'ms_per_step_median': per_step[len(per_step) // 2], 'ms_per_step_min': per_step[0], 'ms_per_step_max': per_step[-1],
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': '%s: text -> 2-stage predictions, mse + triple_sum + duration losses; frozen autoencoder %d heads x %d codewords'
                               % (preset['name'], args.heads, args.codewords),
                   'baseline_config': 4, 'per_gpu_batch': args.batch, 'global_batch': args.batch, 'frames': args.frames,
                   'mel_frames_per_step': frames_per_step, 'phonemes_per_step': int(cpu_batch['text_length'].sum()),
                   'parallelism': 'dp1',
                   'execution': 'hipGraph replay (forward + backward | clip + update)' if graphs_were else 'eager, host-paced'},
        'roofline': roof, 'roofline_step': step_roof, 'kernels': kernels, 'ms_per_step_instrumented': ms_instr,
        'losses': {k: float(v) for k, v in log['loss'].items()} if isinstance(log, dict) and 'loss' in log else None,
    }
    if out['losses'] and any(v != v or v in (float('inf'), float('-inf')) for v in out['losses'].values()):
        sys.stderr.write('bench.py: non-finite losses after the timed steps: %s\n' % out['losses'])
        sys.exit(3)
    if args.cpu_steps > 0:
        from oracle import predictor as op
        from oracle.step import prepare_params
        cores = host_cores()
        threads = args.cpu_threads or min(cores, 32)
        torch.set_num_threads(threads)
        nsample = args.cpu_batch or min(args.batch, 8)
        sub = {k: v[:nsample].clone() for k, v in cpu_batch.items()}
        tl = int(sub['text_length'].max())
        sub['text'], sub['dur'] = sub['text'][:, :tl], sub['dur'][:, :tl]
        fr = int(sub['mel_length'].max())
        sub['mel'] = sub['mel'][:, :fr]
        P = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
        for k, v in P.items():
            if not k.endswith('position.weight'):
                v.requires_grad_(True)
        P_ae = prepare_params({k: v.detach().float().cpu() for k, v in atask.state_dict().items()})
        am = cfg.to_dict() if hasattr(cfg, 'to_dict') else am_config(args.batch)
        times = []
        for i in range(args.cpu_warmup + args.cpu_steps):
            for v in P.values():
                v.grad = None
            t1 = time.perf_counter()
            op.predictor_step(P, am['task']['predictor'], P_ae, acfg.task.to_dict()['autoencoder'], sub,
                              am['trainer']['training_methods'], am['trainer']['loss_weights'], am['trainer']['lambda_dur'],
                              am['trainer']['grad_clip_thresh'])
            times.append(time.perf_counter() - t1)
            wd.beat('oracle predictor step %d' % i)
        times = sorted(times[args.cpu_warmup:])
        sec = times[len(times) // 2]
        sample_frames = float(sub['mel_length'].sum())
        out['cpu_baseline'] = dict(value=sample_frames / sec, unit='mel-frames/s', cores=threads, kind='port',
                                   sample='oracle/predictor.py (plain PyTorch fp32, dropout as configured) predictor train step on the first %d '
                                          'utterances (%d mel frames) of the same batch with the same weights; median of %d timed steps after '
                                          '%d warm-ups; %d of %d visible cores; %s; torch %s'
                                          % (nsample, sample_frames, args.cpu_steps, args.cpu_warmup, threads, cores, cpu_model(), torch.__version__),
                                   s_per_step=sec, cpu_model=cpu_model())
        out['speedup_vs_cpu'] = out['value'] / out['cpu_baseline']['value']
    print(emit_line(out, args.kernels_out))


def self_spawn(argv, n, job_timeout, capture=False):
    """``python bench.py --gpus N`` with WORLD_SIZE unset: start the N ranks ourselves (reference launcher train_dist.py:14-36
    starts one process per GPU the same way).  Returns the job's exit code; a job that outlives ``job_timeout`` seconds is
    killed (its own process group) and reported as 124."""
    import signal
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(('127.0.0.1', 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    # (a rank that starts a job of its own -- the overlapped exchange mode as a CHILD job -- must not hand its own rendezvous on)
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK',
                                                              'ROLE_RANK', 'ROLE_NAME', 'ROLE_WORLD_SIZE', 'GROUP_WORLD_SIZE',
                                                              'MASTER_ADDR', 'MASTER_PORT') and not k.startswith('TORCHELASTIC_')}
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC only on this driver (RCCL across processes)
    env.setdefault('OMP_NUM_THREADS', str(max(1, host_cores() // n)))
    say('starting %d ranks: %s' % (n, ' '.join(cmd[1:])))
    proc = subprocess.Popen(cmd, env=env, start_new_session=True, stdout=subprocess.PIPE if capture else None, text=capture or None)
    try:
        if capture:
            text, _ = proc.communicate(timeout=job_timeout)
            return proc.returncode, text
        return proc.wait(timeout=job_timeout)
    except subprocess.TimeoutExpired:
        sys.stderr.write('bench.py: the %d-rank job did not finish within %d s: killing it\n' % (n, job_timeout))
        # exactly the processes we started: torch.distributed.run (a session of its own) and its descendants -- the ranks
        # sit in sessions the elastic agent opened for them, so the process-group signal alone would leave them running
        pids = [proc.pid]
        try:
            import psutil
            pids += [c.pid for c in psutil.Process(proc.pid).children(recursive=True)]
        except Exception:
            pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        for pid in pids:
            try:
                os.kill(pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        proc.wait()
        return (124, '') if capture else 124


class Watchdog(object):
    """Turns a wedged step / collective into a non-zero exit: the main thread reports progress (``beat``), a daemon thread
    ends the process with code 5 when nothing was reported for ``limit`` seconds (a collective stuck inside a replayed graph
    cannot be recovered in-process; torch.distributed.run then stops the other ranks)."""

    def __init__(self, limit, rank=0):
        import threading
        self.limit, self.rank, self.last, self.what = float(limit), rank, time.perf_counter(), 'start'
        self.on_stall = None
        if self.limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def beat(self, what):
        self.last, self.what = time.perf_counter(), what

    def guard(self, on_stall, limit):
        """SECONDARY measurements (the exchange modes timed after the headline on N > 1 ranks): a stall from here on calls
        ``on_stall(what)`` -- rank 0 prints the line with what it has -- and ends the rank with exit code 0: a collective that
        wedges in an untried mode must not cost the run its headline"""
        self.on_stall, self.limit, self.last = on_stall, min(self.limit, float(limit)) if self.limit > 0 else float(limit), time.perf_counter()

    def _run(self):
        while True:
            time.sleep(1.0)
            idle = time.perf_counter() - self.last
            if idle > self.limit:
                sys.stderr.write('bench.py: rank %d made no progress for %.0f s (last: %s) -- giving up\n' % (self.rank, idle, self.what))
                sys.stderr.flush()
                if self.on_stall is not None:
                    try:
                        self.on_stall(self.what)
                    finally:
                        sys.stdout.flush()
                        os._exit(0)
                os._exit(5)


def dry_run(args, rank, world, wd):
    """--dry: the launcher, the process group and the gradient reducer on a TOY model (two children of three linear layers, the
    way the trainer steps them: D backward, exchange, G backward, exchange) -- runs on CPU over gloo.  Not a benchmark: the
    line says so in ``data``; what it proves is that N ranks start, exchange gradients in buckets, stay bit-identical and
    that rank 0 prints one line carrying the world size it saw."""
    import torch.nn as nn
    from msmctts_amd.distributed.distributed import apply_gradient_allreduce
    torch.manual_seed(7)
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if args.backend == 'nccl' else torch.device('cpu')
    mk = lambda: nn.Sequential(nn.Linear(64, 128), nn.Tanh(), nn.Linear(128, 128), nn.Tanh(), nn.Linear(128, 64))
    model = nn.ModuleDict(dict(autoencoder=mk(), discriminator=mk())).to(dev)
    if rank != 0:                                            # (start-up broadcast must repair this)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    if world > 1:
        apply_gradient_allreduce(model, bucket_bytes=32 * 1024)
    reducer = getattr(model, 'grad_reducer', None)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    gen = torch.Generator().manual_seed(100 + rank)

    def step():
        x = torch.randn(args.batch, 64, generator=gen).to(dev)
        opt.zero_grad()
        model['discriminator'](model['autoencoder'](x).detach()).pow(2).mean().backward()
        if reducer is not None:
            reducer.finish()
        (model['discriminator'](model['autoencoder'](x)) - x).pow(2).mean().backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
    for i in range(args.warmup):
        step()
        wd.beat('dry warm-up %d' % i)
    if rank == world - 1 and os.environ.get('MSMC_BENCH_TEST_STALL'):      # (tests: a rank that wedges must end the job, not hang it)
        time.sleep(float(os.environ['MSMC_BENCH_TEST_STALL']))
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step()
        wd.beat('dry step %d' % i)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double()
    spread, per_rank = 0.0, [elapsed / args.steps * 1e3]
    if world > 1:
        lo, hi = flat.clone(), flat.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        spread = float((hi - lo).abs().max())
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        per_rank = [float(t) / args.steps * 1e3 for t in every]
        elapsed = max(float(t) for t in every)
    if rank != 0:
        return 0
    out = {'metric': 'dry run: toy-model steps/s (launcher + process group + gradient reducer check, NOT the benchmark)',
           'value': args.steps * world / elapsed, 'unit': 'rank-steps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
           'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
           'dtype': 'f32', 'data': 'synthetic (toy model, dry run)', 'dry': True, 'backend': args.backend,
           'world_size_seen': dist.get_world_size() if world > 1 else 1, 'per_rank_ms_per_step': per_rank,
           'config': {'workload': 'dry run', 'gradient_exchange': 'bucketed from hooks' if world > 1 else None,
                      'buckets': len(reducer.buckets) if reducer is not None else 0},
           'ranks_identical': spread == 0.0, 'max_parameter_spread': spread}
    print(json.dumps(out))
    return 0 if spread == 0.0 else 6


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=2, choices=[1, 2, 3, 4, 5],
                    help='BASELINE.json configuration (msmctts_amd/configs.py BASELINE_CONFIGS): 2 = the headline (CSMSC, 2 stages, '
                         '4 heads x 256, B=16, one GPU); 1 = 1 stage, 1 head x 64, B=4; 3 = the LJSpeech-named copy of #2 meant for '
                         '--gpus 8; 4 = predictor training (msmc_vq_gan_am.yaml, B=64) against a frozen autoencoder of #2; 5 = 1024-wide '
                         'input, 8 heads x 512, meant for --gpus 8.  --batch/--heads/--codewords override.')
    ap.add_argument('--batch', type=int, default=None)
    ap.add_argument('--frames', type=int, default=400)
    ap.add_argument('--heads', type=int, default=None)
    ap.add_argument('--codewords', type=int, default=None)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
    ap.add_argument('--cpu-steps', type=int, default=5, help='timed oracle steps per phase for cpu_baseline (0 = skip)')
    ap.add_argument('--cpu-warmup', type=int, default=2, help='untimed oracle steps per phase')
    ap.add_argument('--cpu-batch', type=int, default=0, help='utterances of the batch in the cpu_baseline sample (0 = all)')
    ap.add_argument('--warmup-phase-steps', type=int, default=10,
                    help='extra eager steps of the warm-up phase (no vocoder / discriminator) timed after the headline')
    ap.add_argument('--cpu-threads', type=int, default=0, help='0 = min(available cores, 32)')
    ap.add_argument('--no-microbench', action='store_true')
    ap.add_argument('--microbench-only', action='store_true',
                    help='only the VQ argmin micro-benchmarks at N = 2^20 frames (the PMC passes of tools/profile_round.sh)')
    ap.add_argument('--no-autocast', action='store_true',
                    help='bf16 only inside the HIP conv stacks; the transformer encoder/decoder stay fp32')
    ap.add_argument('--exec', dest='exec_mode', default='auto', choices=['auto', 'graph', 'eager'],
                    help='graph (= auto): replay the step from three hipGraphs, gradients all-reduced between the '
                         'segments (no host launch cost: the eager step is paced by the host); eager: multi-stream eager '
                         'step with bucketed all-reduce overlapped with backward')
    ap.add_argument('--graph', action='store_true', help='same as --exec graph')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend (gloo: --dry only)')
    ap.add_argument('--share-gpu', action='store_true',
                    help='TEST MODE (tests/test_gpu_parity.py): the N ranks share the visible GPUs round robin and exchange over gloo '
                         '(--backend gloo) -- the whole N > 1 code path of this file on a one-GPU box; its numbers are not a measurement')
    ap.add_argument('--dry', action='store_true',
                    help='launcher / process-group / gradient-reducer check on a toy model (runs on CPU with --backend gloo); not a benchmark')
    ap.add_argument('--job-timeout', type=int, default=int(os.environ.get('MSMC_BENCH_JOB_TIMEOUT', '1500')),
                    help='self-spawned N > 1 job: seconds before the launcher kills it (exit 124)')
    ap.add_argument('--stall-timeout', type=int, default=int(os.environ.get('MSMC_BENCH_STALL_TIMEOUT', '300')),
                    help='seconds without progress before a rank gives up (exit 5); also the process-group timeout')
    ap.add_argument('--exchange', default=os.environ.get('MSMC_GRAPH_EXCHANGE', 'auto'),
                    choices=['auto', 'serial', 'overlap', 'both', 'all'],
                    help='graph mode, N > 1: serial = one flat all-reduce per child between the replayed segments (the headline); '
                         'overlap = bucketed all-reduces captured into the segments on the RCCL stream (DESIGN.md section 6: '
                         'only ever run on one GPU); both = serial is the headline, the overlap mode is timed after it in the same '
                         'run and reported under exchange_modes; all = serial, then serial with bf16 on the wire, then overlap.  '
                         'auto (default) = all when N > 1: the secondary modes run under a guard (--secondary-timeout) that prints '
                         'the line with the headline and exits 0 if one of them wedges')
    ap.add_argument('--overlap-in-process', action='store_true',
                    help='time the overlapped exchange inside this job (default for an explicit --exchange both / overlap); '
                         'auto / all run it as a CHILD job of rank 0 after the other ranks have left, so that nothing it does '
                         '-- RCCL captured into hipGraphs has only ever met one rank -- can cost this job its line or exit code')
    ap.add_argument('--secondary-timeout', type=int, default=int(os.environ.get('MSMC_BENCH_SECONDARY_TIMEOUT', '90')),
                    help='N > 1: seconds without progress in a SECONDARY exchange mode before the line is printed without it')
    ap.add_argument('--kernel-timing-steps', type=int, default=3, help='extra steps timed kernel by kernel (rank 0)')
    ap.add_argument('--kernels-out', default=os.path.join(ROOT, 'gpurun_out', 'bench_kernels.json'),
                    help='JSON side file for the per-kernel-symbol table (the printed line only names it)')
    ap.add_argument('--calls-out', default=None,
                    help='JSON side file: the instrumented steps per (entry point, layer shape) -- which layers a '
                         "symbol's time belongs to")
    ap.add_argument('--fp32-steps', type=int, default=3,
                    help='eager fp32 steps (the parity configuration) timed after the headline, 0 = skip')
    args = ap.parse_args()
    from msmctts_amd.configs import BASELINE_CONFIGS
    preset = BASELINE_CONFIGS[args.config]
    args.model_kw = dict(preset['model'])
    if args.heads is not None:
        args.model_kw['n_heads'] = args.heads
    if args.codewords is not None:
        args.model_kw['embedding_sizes'] = args.codewords
    args.heads, args.codewords = args.model_kw['n_heads'], args.model_kw['embedding_sizes']
    args.batch = args.batch if args.batch is not None else preset['per_gpu_batch']
    args.in_dim = args.model_kw.get('in_dim', 80)

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # launched the way the one-GPU line is launched: start the ranks ourselves and hand their line / exit code on
        sys.exit(self_spawn(sys.argv[1:], args.gpus, args.job_timeout))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    args.graph = args.graph or args.exec_mode in ('graph', 'auto')
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d (torch.distributed.run --nproc-per-node must match)' % (args.gpus, world))
    wd = Watchdog(args.stall_timeout, rank)
    if args.backend == 'gloo' and not (args.dry or args.share_gpu):
        raise SystemExit('bench.py: --backend gloo is for --dry (the product path has no CPU execution path) or --share-gpu (test mode)')
    if args.share_gpu and (args.backend != 'gloo' or args.exchange not in ('serial', 'auto', 'all')):
        raise SystemExit('bench.py: --share-gpu needs --backend gloo and the serial exchange (RCCL refuses two ranks on one device; '
                         'gloo collectives cannot be captured)')
    if args.dry and args.backend == 'gloo':
        if world > 1:
            import datetime
            dist.init_process_group('gloo', init_method='env://', world_size=world, rank=rank,
                                    timeout=datetime.timedelta(seconds=max(10, args.stall_timeout)))
        rc = dry_run(args, rank, world, wd)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(rc)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the product path has no CPU fallback)')
    if args.share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if world > 1:
        import datetime
        dist.init_process_group(args.backend, init_method='env://', world_size=world, rank=rank,
                                timeout=datetime.timedelta(seconds=max(10, args.stall_timeout)))
    if args.dry:
        rc = dry_run(args, rank, world, wd)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(rc)
    if args.exchange == 'auto':
        args.exchange = 'all' if (world > 1 and args.backend == 'nccl' and not args.share_gpu) else 'serial'
    secondary = {'both': ['overlap'], 'all': ['serial_bf16_wire', 'overlap']}.get(args.exchange, [])
    # (MSMC_BENCH_CHILD_ARGS: TEST HOOK -- extra arguments for the child job, so that its orchestration can run on a one-GPU box)
    child_extra = os.environ.get('MSMC_BENCH_CHILD_ARGS', '').split()
    child_overlap = args.exchange == 'all' and not args.overlap_in_process and (not args.share_gpu or bool(child_extra))
    if child_overlap:
        secondary = [m for m in secondary if m != 'overlap']
    if args.share_gpu:
        secondary = [m for m in secondary if m != 'overlap']          # (gloo collectives cannot be captured)
    if secondary:
        args.exchange = 'serial'

    from msmctts_amd.hip import lib, vq as hipvq
    from msmctts_amd.synthetic import make_batch
    assert lib.backend() == 'gfx950'
    if args.config == 4:
        if world != 1:
            raise SystemExit('bench.py --config 4 is a one-GPU line')
        bench_predictor(args, device, wd)
        return
    if args.microbench_only:
        print(json.dumps({'vq_microbench': [vq_microbench(device, args.heads, args.codewords, iters=5),
                                            vq_microbench(device, 4, 64, iters=5), vq_microbench(device, 8, 512, iters=5)]}))
        return
    cfg, trainer = build(args, device, rank, world)
    state0 = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()} if rank == 0 and world == 1 else None
    batch = make_batch(args.batch, args.frames, args.in_dim, 300, seed=1234, rank=rank, device='cpu')
    lengths_host = batch['mel_length'].tolist()
    batch = {k: v.to(device) for k, v in batch.items()}
    batch['mel_length_host'] = lengths_host
    import random
    trainer.rng = random.Random(1234 + rank)

    timer, register_banks = make_timer()

    def step(i):
        if not trainer.use_graphs:
            trainer.model.zero_grad()
            trainer.optimizer.zero_grad()
        return trainer.train_step(batch, 10 + i)

    say('built model; starting warm-up')
    for i in range(args.warmup):
        step(i)
        torch.cuda.synchronize()
        wd.beat('warm-up step %d' % i)
        say('warm-up step %d done' % i)
    def headline(elapsed_max, per_rank, frames, modes):
        """the contract fields of the line from the headline measurement (all a stalled secondary mode leaves us with)"""
        return {
            'metric': 'mel-frames/sec MSMC-VQ-GAN train step (GAN phase)', 'value': frames / (elapsed_max / args.steps),
            'unit': 'mel-frames/s', 'n_gpus': world, 'world_size_seen': dist.get_world_size() if world > 1 else 1,
            'backend': ('nccl (RCCL)' if args.backend == 'nccl' else 'gloo, ranks sharing a GPU (test mode: not a measurement)') if world > 1 else None,
            'per_rank_ms_per_step': per_rank, 'exchange_modes': modes,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': elapsed_max / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': '%s: %d-stage %d-head x %d-codeword VQ + HifiGAN + MPD/MRD, GAN phase'
                                   % (preset['name'], len(args.model_kw.get('downsample_scales', (1, 4))), args.heads, args.codewords),
                       'baseline_config': args.config, 'in_dim': args.in_dim,
                       'per_gpu_batch': args.batch, 'global_batch': args.batch * world, 'frames': args.frames,
                       'mel_frames_per_step': frames, 'parallelism': 'dp%d' % world,
                       'vq_search': 'fp32 (bit-exact indices)',
                       'execution': 'hipGraph replay (3 segments/step)' if args.graph else 'eager, multi-stream',
                       'gradient_exchange': None if world == 1 else (args.exchange if args.graph else 'bucketed from hooks')},
        }

    def timed_steps(first):
        '''exactly K steps between barrier + synchronize on both sides; per-step durations from events recorded on the step's
        stream between the steps (no host synchronisation inside the timed region: the headline stays the wall clock over all
        K steps; the events give the median SURVEY 8d asks for)'''
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            log_ = step(first + i)
            marks[i + 1].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        wd.beat('timed steps')
        return dt, sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)), log_

    def over_ranks(dt):
        '''(max over ranks, per-rank ms/step)'''
        if world == 1:
            return dt, [dt / args.steps * 1e3]
        tt = torch.tensor([dt], device=device if args.backend == 'nccl' else 'cpu', dtype=torch.float64)
        every = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(every, tt)
        return max(float(t) for t in every), [float(t) / args.steps * 1e3 for t in every]

    elapsed, per_step, log = timed_steps(args.warmup)
    exchange_modes = None
    core = {}
    if (secondary or child_overlap) and world > 1 and args.graph:
        # Everything the line needs from the HEADLINE (serial exchange, fp32 on the wire) first -- its collectives included --,
        # then the secondary modes under the guard: each is the same K steps, the riskiest (RCCL captured into the graphs: only
        # ever run on one GPU) last.  A mode that raises or wedges is reported as such; the headline stands.
        e_serial, ranks_serial = over_ranks(elapsed)
        fr = torch.tensor([float(sum(lengths_host))], device=device, dtype=torch.float64)
        dist.all_reduce(fr)
        core = dict(elapsed=e_serial, per_rank=ranks_serial, frames=float(fr.item()))
        exchange_modes = {'serial': dict(ms_per_step=e_serial / args.steps * 1e3, per_rank_ms_per_step=ranks_serial)}

        def emergency(what):
            if rank == 0:
                exchange_modes['stalled'] = what
                print(emit_line(headline(core['elapsed'], core['per_rank'], core['frames'], exchange_modes), None))
        wd.guard(emergency, args.secondary_timeout)
        reducer = trainer.model.grad_reducer
        for mode in secondary:
            wd.beat('exchange mode %s' % mode)
            try:
                if mode == 'serial_bf16_wire':
                    reducer.exchange_dtype = torch.bfloat16          # (the flat buffers are rebuilt for the new wire type)
                else:
                    trainer._graphs, trainer.graph_exchange = None, 'overlap'
                for i in range(max(2, min(args.warmup, 4))):
                    step(i)
                    torch.cuda.synchronize()
                    wd.beat('%s warm-up step %d' % (mode, i))
                e_m, _, _ = timed_steps(args.warmup)
                e_m, ranks_m = over_ranks(e_m)
                exchange_modes[mode] = dict(ms_per_step=e_m / args.steps * 1e3, per_rank_ms_per_step=ranks_m)
                say('exchange mode %s: %.2f ms/step' % (mode, e_m / args.steps * 1e3))
            except Exception as e:                                    # (the process group may be unusable now: stop here)
                exchange_modes[mode] = dict(error='%s: %s' % (type(e).__name__, str(e)[:200]))
                say('exchange mode %s failed: %s' % (mode, exchange_modes[mode]['error']))
                break
            finally:
                reducer.exchange_dtype = torch.float32
        trainer.graph_exchange = 'serial'
        if child_overlap:
            # The overlapped exchange -- RCCL all-reduces captured INTO the hipGraphs -- has only ever met one rank.  It runs as a
            # job of its own, started by rank 0 once every other rank of THIS job has left (their GPUs are free again): whatever
            # it does (raise, wedge, take the runtime down with it) costs this job neither its line nor its exit code.
            if rank != 0:
                sys.stdout.flush()
                os._exit(0)
            wd.guard(emergency, args.stall_timeout + 120)
            wd.beat('child job: overlapped exchange')
            child = ['--gpus', str(world), '--exchange', 'overlap', '--config', str(args.config), '--batch', str(args.batch),
                     '--frames', str(args.frames), '--steps', str(args.steps), '--warmup', str(args.warmup), '--dtype', args.dtype,
                     '--no-microbench', '--cpu-steps', '0', '--fp32-steps', '0', '--kernel-timing-steps', '0',
                     '--warmup-phase-steps', '0', '--stall-timeout', str(min(args.stall_timeout, 120))] + child_extra
            try:
                rc, text = self_spawn(child, world, min(args.stall_timeout + 60, 420), capture=True)
                lines = [l for l in (text or '').splitlines() if l.startswith('{')]
                if rc == 0 and lines:
                    got = json.loads(lines[-1])
                    exchange_modes['overlap'] = dict(ms_per_step=got['ms_per_step'], per_rank_ms_per_step=got['per_rank_ms_per_step'],
                                                     note='a job of its own (same ranks, same batch), started after this one')
                else:
                    exchange_modes['overlap'] = dict(error='the child job ended with exit code %s and %d result lines' % (rc, len(lines)))
            except Exception as e:
                exchange_modes['overlap'] = dict(error='%s: %s' % (type(e).__name__, str(e)[:200]))
            say('exchange mode overlap (child job): %s' % exchange_modes['overlap'])
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    # second pass over the same steps with HIP events around every hand-written launch (the events cost a
    # few percent of host time, so the headline value above is taken without them)
    ms_instr = None
    if rank == 0 and world == 1 and args.kernel_timing_steps > 0:
        graphs_were = trainer.use_graphs
        trainer.use_graphs = False               # per-launch HIP events need the eager path ...
        from msmctts_amd.hip import convnet
        convnet.STREAMS_ENABLED = False          # ... and one stream, so that an event pair brackets exactly one kernel
        trainer.model.zero_grad()
        step(args.warmup + args.steps)           # (untimed: the eager banks / tables of this mode exist afterwards)
        register_banks()
        timer.start(lib.get())
        t1 = time.perf_counter()
        for i in range(args.kernel_timing_steps):
            # park the GPU while the host enqueues the step: with the queue full, an event pair brackets kernel
            # execution only (an eager step is host-paced; without this the pairs would also time the GPU waiting
            # for the next launch packet and disagree with rocprofv3's per-kernel durations)
            torch.cuda._sleep(int(2.0e8))
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        ms_instr = (time.perf_counter() - t1) / args.kernel_timing_steps * 1e3
        timer.stop()
        wd.beat('instrumented steps')
        if args.calls_out:
            with open(args.calls_out, 'w') as f:
                json.dump(timer.by_call(args.kernel_timing_steps), f, indent=0)
        convnet.STREAMS_ENABLED = True
        trainer.use_graphs = graphs_were
    # warm-up phase (iteration < warmup_steps: autoencoder + frame decoder only), SURVEY.md 8d asks for it separately
    warm_ms = None
    if rank == 0 and world == 1 and args.warmup_phase_steps > 0:
        keep = trainer.warmup_steps
        trainer.warmup_steps = 10 ** 9
        for i in range(3):
            step(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.warmup_phase_steps):
            step(i)
        torch.cuda.synchronize()
        warm_ms = (time.perf_counter() - t1) / args.warmup_phase_steps * 1e3
        trainer.warmup_steps = keep
        wd.beat('warm-up phase steps')
    # the same step in fp32 end to end (the configuration the parity tests prove): eager, after two untimed steps
    fp32_ms = fp32_err = None
    if rank == 0 and world == 1 and args.fp32_steps > 0 and args.dtype != 'fp32':
        keep = (trainer.use_graphs, trainer.amp_dtype)
        try:
            trainer.use_graphs, trainer.amp_dtype = False, None
            for i in range(2):
                step(i)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.fp32_steps):
                step(i)
            torch.cuda.synchronize()
            fp32_ms = (time.perf_counter() - t1) / args.fp32_steps * 1e3
            say('fp32 eager: %.2f ms/step' % fp32_ms)
        except Exception as e:                     # the headline must not die with the side measurement
            fp32_err = '%s: %s' % (type(e).__name__, str(e)[:200])
            say('fp32 eager steps failed: ' + fp32_err)
        finally:
            trainer.use_graphs, trainer.amp_dtype = keep
        wd.beat('fp32 steps')
    if core:                                   # (N > 1 with secondary modes: reduced over ranks before they ran)
        elapsed, per_rank_ms, frames_per_step = core['elapsed'], core['per_rank'], core['frames']
    else:
        if world > 1:
            dist.barrier()
        elapsed, per_rank_ms = over_ranks(elapsed)
        if world > 1:
            fr = torch.tensor([float(sum(lengths_host))], device=device, dtype=torch.float64)
            dist.all_reduce(fr)
            frames_per_step = float(fr.item())
        else:
            frames_per_step = float(sum(lengths_host))
    wd.beat('reductions over ranks')
    ms_per_step = elapsed / args.steps * 1e3
    value = frames_per_step / (elapsed / args.steps)
    say('timed %d steps: %.2f ms/step' % (args.steps, ms_per_step))

    if rank != 0:
        if core:
            sys.stdout.flush()
            os._exit(0)                          # (no further collective after a secondary mode: the group may be unusable)
        return
    kernels, roof, step_roof = summarize_kernels(timer, args.dtype, args.kernel_timing_steps, ms_per_step)
    out = headline(elapsed, per_rank_ms, frames_per_step, exchange_modes)
    out.update({
        'ms_per_step_median': ms_median, 'ms_per_step_min': per_step[0], 'ms_per_step_max': per_step[-1],
        'step_tflops': (FLOP_PER_STEP_ELIDED * (args.batch / 16.0) * world / (elapsed / args.steps) / 1e12) if args.config in (2, 3) else None,
        'step_flop_model': 'SURVEY 8d: 3.006 TFLOP/step at B=16,T=400 minus the elided D weight-grads of the G step '
                           '= 2.65 TFLOP',
        'roofline': roof,
        'roofline_step': step_roof,
        'kernels': kernels,
        'ms_per_step_instrumented': ms_instr,
        'fp32_ms_per_step': fp32_ms if fp32_err is None else fp32_err,
        'losses': {k: float(v) for k, v in log['loss'].items()},
        'warmup_phase': None if warm_ms is None else dict(
            ms_per_step=warm_ms, value=frames_per_step / (warm_ms * 1e-3), unit='mel-frames/s',
            note='iteration < warmup_steps (no vocoder, no discriminator), %s, %d steps'
                 % ('hipGraph replay (forward | backward | update)' if (args.graph and trainer.graph_warmup) else 'eager', args.warmup_phase_steps)),
        'runtime': {'DEBUG_CLR_GRAPH_PACKET_CAPTURE': os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE'),
                    'note': 'hipGraph memset nodes are mis-ordered on the AQL packet-capture path of ROCm 7.2 '
                            '(tools/repro_graph_memset.py); the package disables that path before the first HIP call'},
    })
    bad = [k for k, v in out['losses'].items() if v != v or v in (float('inf'), float('-inf'))]
    if bad:
        sys.stderr.write('bench.py: non-finite losses after the timed steps: %s -- the step does not train; '
                         'no result line\n' % bad)
        sys.exit(3)
    if not args.no_microbench:
        out['vq_microbench'] = [vq_microbench(device, args.heads, args.codewords), vq_microbench(device, 4, 64),
                                vq_microbench(device, 8, 512), vq_microbench(device, 1, 64),
                                vq_microbench(device, args.heads, args.codewords, shortlist=False),
                                vq_microbench(device, 4, 64, shortlist=False)]
        say('vq microbench done')
        wd.beat('vq microbench')
        wd.limit = max(wd.limit, 900.0)        # (the oracle's CPU steps below run without reporting)
    if world == 1 and args.cpu_steps > 0:
        cores = host_cores()
        r = random.Random(99)
        fw = []
        for n in lengths_host:
            s = r.randrange(max(1, n - 40))
            fw.append((s, s + 40))
        sw = [(s * 300, e * 300) for s, e in fw]
        threads = args.cpu_threads or min(cores, 32)
        nsample = args.cpu_batch or args.batch
        res = cpu_baseline(cfg, state0, batch, (fw, sw), args.cpu_steps, args.cpu_warmup, threads, nsample)
        sec, sample_frames = res['gan']
        wsec, _ = res['warmup']
        out['cpu_baseline'] = dict(value=sample_frames / sec, unit='mel-frames/s', cores=threads, kind='port',
                                   sample='oracle (plain PyTorch fp32) GAN-phase train step on %s utterances '
                                          '(%d mel frames) of the same batch with the same weights; median of %d timed '
                                          'steps after %d warm-ups; %d of %d visible cores; %s; torch %s'
                                          % ('all %d' % nsample if nsample == args.batch else 'the first %d' % nsample,
                                             sample_frames, args.cpu_steps, args.cpu_warmup, threads, cores, cpu_model(),
                                             torch.__version__),
                                   s_per_step=sec, cpu_model=cpu_model(),
                                   warmup_phase=dict(value=sample_frames / wsec, unit='mel-frames/s', s_per_step=wsec))
        out['speedup_vs_cpu'] = value / out['cpu_baseline']['value']
    print(emit_line(out, args.kernels_out))
    sys.stdout.flush()
    if world > 1:
        if core:
            os._exit(0)                          # (the line is out; tearing down a group that ran untried modes may wedge)
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
