#!/usr/bin/env python
"""Runs ON the GPU box: time the kernel candidates of every convolution shape of the bench configuration (and of the
GPU test-suite's layer cases) and write the choices to msmctts_amd/hip/tuned_gfx950.json (copied back via gpurun_out).
RETUNE=wgrad4 keeps the committed table and re-times only the single-launch bf16 weight gradients the fourth-generation
kernel can serve (one pass).  RETUNE=gather3x keeps it too and times only the LDS-DMA halo variants (24..31) of every bf16
forward / data-gradient shape, merging them with the committed timings of the other candidates; RETUNE=gather4 does the
same for the persistent thin-layer kernel (variant 32), RETUNE=gemm1 for the 1-tap GEMM kernel (variant 34), RETUNE=gather5 for the sixteen-wave staged-tap kernel (variants 40..47), RETUNE=gather7 for the eight-wave 64 x 64-per-wave kernel (variants 56, 59, 60, 61, 63), RETUNE=wgrad6 for the direct thin-layer weight gradient (variant 8) and RETUNE=wgrad5 for the general-lattice LDS-DMA weight gradient
(variant 7; the grouped calls whose members change are re-timed by the run itself).  RETUNE=new keeps every decision
on file and only adds the shapes and grouped calls the run meets for the first time (after a change of how the step
groups its launches, e.g. MSMC_WGRAD_BATCH).  RETUNE=all times every decision the run meets again (two passes, the faster
measurement of each candidate) and replaces those entries, keeping the entries of shapes it does not meet.
RETUNE=tails drops the decisions of bf16 layers whose channel counts are multiples of 8 but not of 64 (and of the grouped calls
with such members) and times them afresh.  CONFIG=4 (with RETUNE=new or tails) runs the predictor step of BASELINE configuration #4 (600 / 1536-wide FFT stacks at B = 64) instead of
the GAN-phase step and adds its shapes."""
import os, sys, random, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
RETUNE = os.environ.get('RETUNE', '')
if not RETUNE:
    os.environ['MSMC_TUNE_CACHE'] = '/nonexistent'        # start from scratch
os.environ['MSMC_TUNE_BUDGET'] = '1000000'                # ... and time every shape itself (no borrowing from neighbours)
os.environ['MSMC_TUNE_BORROW'] = '0'
import torch
import bench
from msmctts_amd.hip import conv
from msmctts_amd.synthetic import make_batch


class A(object):
    codewords, heads, batch, frames, graph, dtype, no_autocast = 256, 4, 16, 400, False, 'bf16', False


def _wgrad4_scope(key):
    """signature of a single-launch weight gradient inside wgrad4.inc's scope (see wg4_plan)"""
    if key[0] != 'wgrad':
        return False
    (dtype, B, Hin, Win, Cin, Hout, Wout, Cout, QH, QW, osy, osx, isy, isx, ntaps, dys, dxs, pad_mode) = key[1:19]
    if dtype != 1 or Cin % 64 or Cout % 64 or pad_mode != 0 or (osy, osx, isy, isx) != (1, 1, 1, 1):
        return False
    if (QH, QW) != (Hout, Wout):
        return False
    return (Hin == 1 and Hout == 1 and not any(dys)) or (Win == Wout and not any(dxs))


dev = torch.device('cuda:0')
torch.cuda.set_device(0)
if RETUNE == 'wgrad4':
    dropped = [k for k in conv.TUNED if _wgrad4_scope(k)]
    # ... and the grouped calls with two or more fourth-generation members (grids of their own: a third alternative)
    dropped += [k for k in conv.TUNED if k[0] == 'wgrad-group' and sum(1 for m in k[1:] if m[21] >= 4) > 1]
    for k in dropped:
        del conv.TUNED[k]
    print('re-timing %d weight-gradient shapes' % len(dropped))
kept = {}
if RETUNE == 'gather3x':
    conv._GATHER_CANDIDATES = tuple((v, 0) for v in range(24, 32))
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED) if k[0] == 'gather' and k[1] == 1}
    print('timing the LDS-DMA halo variants of %d forward / data-gradient shapes' % len(kept))
if RETUNE == 'gather4':                                   # the persistent thin-layer kernel: Cin, Cout in {32, 64}
    conv._GATHER_CANDIDATES = ((32, 0),)
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'gather' and k[1] == 1 and k[5] in (32, 64) and k[8] in (32, 64)}
    print('timing the persistent thin-layer kernel on %d forward / data-gradient shapes' % len(kept))
if RETUNE == 'gemm1':                                     # 1-tap layers as a plain channel GEMM (variant 34)
    conv._GATHER_CANDIDATES = ((34, 0), (35, 0))
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'gather' and k[1] in (0, 1) and k[15] == 1 and tuple(k[16]) == (0,) and tuple(k[17]) == (0,)}
    print('timing the 1-tap GEMM kernel on %d forward / data-gradient shapes' % len(kept))
if RETUNE == 'gather5':                                   # sixteen-wave staged-tap kernel (variants 40..44): every bf16
    conv._GATHER_CANDIDATES = tuple((v, 0) for v in range(40, 48))      # shape with 64-multiple input channels and >= 2 taps
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'gather' and k[1] == 1 and k[5] % 64 == 0 and k[15] >= 2}
    print('timing the fifth-generation forward / data-gradient kernel on %d shapes' % len(kept))
if RETUNE == 'gather7':                                   # eight-wave 64 x 64-per-wave kernel (variants 56..63): the same scope
    conv._GATHER_CANDIDATES = tuple((v, 0) for v in (56, 59, 60, 61, 63))
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'gather' and k[1] == 1 and k[5] % 64 == 0 and k[15] >= 2}
    print('timing the seventh-generation forward / data-gradient kernel on %d shapes' % len(kept))
if RETUNE == 'gather6':                                   # thin-channel kernel (variant 50): bf16 shapes with 8 / 16 / 32 input channels
    conv._GATHER_CANDIDATES = ((50, 0),)
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED) if k[0] == 'gather' and k[1] == 1 and k[5] in (2, 4, 8, 16, 32, 64)}
    print('timing the thin-channel kernel on %d forward / data-gradient shapes' % len(kept))
if RETUNE == 'wgrad6':                                    # direct thin-layer weight gradient (variant 8): every bf16 shape with a
    conv._WGRAD_CANDIDATES = ((8, 0),)                    # channel count of at most 16 on one side
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'wgrad' and k[1] == 1 and min(k[5], k[8]) <= 16}
    print('timing the direct thin-layer weight gradient on %d shapes' % len(kept))
if RETUNE == 'tails':                                     # channel counts that end inside a tile of 64 (the predictor's 600-wide
    def _tail(m):                                         # FFT blocks: gather7.inc / wgrad4.inc read zeros past them): time those
        return m[0] == 1 and min(m[4], m[7]) >= 64 and (m[4] % 64 or m[7] % 64) and m[4] % 8 == 0 and m[7] % 8 == 0       # shapes and their groups afresh
    dropped = [k for k in conv.TUNED if (k[0] in ('gather', 'wgrad') and _tail(k[1:])) or
               (k[0].endswith('-group') and any(_tail(m) for m in k[1:] if isinstance(m, tuple)))]
    for k in dropped:
        del conv.TUNED[k]
    print('re-timing %d decisions on layers whose channel counts end inside a tile of 64' % len(dropped))
old_groups = {}
if RETUNE == 'groups':                                    # grouped-versus-single decisions only (every single shape stays cached)
    old_groups = {k: conv.TUNED.pop(k) for k in list(conv.TUNED) if k[0].endswith('-group')}
    print('re-timing the grouped calls the run meets (%d decisions on file)' % len(old_groups))
if RETUNE == 'wgrad5':                                    # general-lattice LDS-DMA weight gradient (variant 7): time it on
    conv._WGRAD_CANDIDATES = ((7, 0), (7, -1))            # every bf16 shape with 64-multiple channels, merge with the table
    kept = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)
            if k[0] == 'wgrad' and k[1] == 1 and k[5] % 64 == 0 and k[8] % 64 == 0}
    print('timing the general-lattice LDS-DMA weight gradient on %d shapes' % len(kept))
fresh = {}
if RETUNE == 'all':                                       # after a change that moves every kernel (epilogues): the shapes and grouped
    fresh = {k: conv.TUNED.pop(k) for k in list(conv.TUNED)}      # calls this run meets are timed again and REPLACE their entries;
    print('re-timing every decision the run meets (%d on file)' % len(fresh))     # the rest of the table stays
for rep in range(1 if (RETUNE and RETUNE != 'all') else 2):                     # two passes: keep the faster measurement of each candidate
    saved = dict(conv.TUNED)
    if not RETUNE or RETUNE == 'all':
        conv.TUNED.clear()
    if os.environ.get('CONFIG', '2') == '4':                # BASELINE configuration #4: the predictor step (bench.py --config 4)
        A.batch = 64
        trainer, task4, _, _, _, batch, _ = bench.build_predictor(A, dev)
        for i in range(2):
            task4.zero_grad()
            trainer.train_step(batch, i)
    else:
        cfg, trainer = bench.build(A, dev, 0, 1)
        batch = make_batch(A.batch, A.frames, 80, 300, seed=1234, rank=0, device='cpu')
        lengths = batch['mel_length'].tolist()
        batch = {k: v.to(dev) for k, v in batch.items()}
        batch['mel_length_host'] = lengths
        trainer.rng = random.Random(1234)
        for i in range(2):
            trainer.model.zero_grad()
            trainer.optimizer.zero_grad()
            trainer.train_step(batch, 10 + i)
    torch.cuda.synchronize()
    for k, v in saved.items():
        if k in conv.TUNED:
            times = {c: min(t, v[2].get(c, t)) for c, t in conv.TUNED[k][2].items()}
            best = min(times, key=times.get)
            conv.TUNED[k] = (conv._GROUP_CODES[best[0]] if isinstance(best[0], str) else best[0], best[1], times)
        else:
            conv.TUNED[k] = v
    del trainer
for k, v in kept.items():                                 # committed candidates + the new ones; shapes the run did not meet stay
    times = dict(v[2])
    for c, t in (conv.TUNED[k][2] if k in conv.TUNED else {}).items():
        times[c] = min(t, times.get(c, t))
    best = min(times, key=times.get) if times else (v[0], v[1])
    conv.TUNED[k] = (best[0], best[1], times)
for k, v in fresh.items():                                # (decisions of other configurations and of the test-suite's cases)
    conv.TUNED.setdefault(k, v)
for k, v in old_groups.items():                           # decisions of calls this run did not meet (other configurations, tests)
    conv.TUNED.setdefault(k, v)
out = os.path.join(ROOT, 'gpurun_out', 'tuned_gfx950.json')
conv.save_tuned(out)
print('shapes tuned:', len(conv.TUNED))
