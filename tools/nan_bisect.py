#!/usr/bin/env python
"""GPU diagnostic: run the bench configuration for a few steps and print every loss per step, the first
non-finite parameter / gradient / buffer by name.  Switches: --exec, --dtype, --batch, env MSMC_GROUPED.

    MSMC_GROUPED=0 python tools/nan_bisect.py --exec eager --dtype bf16 --steps 4
"""
import argparse
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
import torch  # noqa: E402

import bench  # noqa: E402


def first_bad(model):
    bad = []
    for n, p in model.named_parameters():
        if not torch.isfinite(p).all():
            bad.append('param ' + n)
        if p.grad is not None and not torch.isfinite(p.grad).all():
            bad.append('grad ' + n)
    for n, b in model.named_buffers():
        if b.is_floating_point() and not torch.isfinite(b).all():
            bad.append('buffer ' + n)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--frames', type=int, default=400)
    ap.add_argument('--heads', type=int, default=4)
    ap.add_argument('--codewords', type=int, default=256)
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--exec', dest='exec_mode', default='graph')
    ap.add_argument('--no-autocast', action='store_true')
    args = ap.parse_args()
    args.graph = args.exec_mode == 'graph'
    dev = torch.device('cuda:0')
    from msmctts_amd.synthetic import make_batch
    cfg, trainer = bench.build(args, dev, 0, 1)
    batch = make_batch(args.batch, args.frames, 80, 300, seed=1234, rank=0, device='cpu')
    lengths = batch['mel_length'].tolist()
    batch = {k: v.to(dev) for k, v in batch.items()}
    batch['mel_length_host'] = lengths
    trainer.rng = random.Random(1234)
    tag = 'exec=%s dtype=%s grouped=%s B=%d %s' % (args.exec_mode, args.dtype, os.environ.get('MSMC_GROUPED', '1'), args.batch,
                                                  ' '.join('%s=%s' % (k[5:], v) for k, v in sorted(os.environ.items()) if k.startswith('MSMC_G_') or k == 'MSMC_CGROUP'))
    if os.environ.get('MSMC_CGROUP', '1') == '0':
        from msmctts_amd.hip import lib
        lib.get().msmc_conv_set_grouping(0)
    for i in range(args.steps):
        if not trainer.use_graphs:
            trainer.model.zero_grad()
            trainer.optimizer.zero_grad()
        if trainer.use_graphs and trainer._graphs is not None and os.environ.get('NAN_SEGMENTS', '1') != '0':
            # replay segment by segment: name the first non-finite gradients BEFORE clipping spreads them
            g = trainer._graphs
            starts = [trainer.rng.randrange(max(1, int(n) - trainer.frame_lengths)) for n in lengths]
            g['starts'].copy_(torch.tensor(starts, dtype=torch.int64))
            for seg in 'ab':
                g[seg].replay()
                torch.cuda.synchronize()
                bad = [n for n, p in trainer.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
                if bad and seg == 'b':
                    print('[%s] step %d after segment %s: %d non-finite grads: %s' % (tag, i, seg, len(bad), bad[:12]), flush=True)
                    for n, p in trainer.model.named_parameters():
                        if n in bad[:4]:
                            gflat = p.grad.detach().float().flatten()
                            idx = (~torch.isfinite(gflat)).nonzero().flatten()
                            print('    %s shape %s: %d bad of %d, first idx %s values %s' % (n, tuple(p.shape), idx.numel(), gflat.numel(), idx[:8].tolist(), gflat[idx[:4]].tolist()), flush=True)
            g['c'].replay()
            vec = g['loss_vec'].clone()
            log = {'loss': {k: vec[j] for j, k in enumerate(g['loss_keys'])}}
        else:
            log = trainer.train_step(batch, 10 + i)
        torch.cuda.synchronize()
        losses = {k: float(v) for k, v in log['loss'].items()}
        bad = first_bad(trainer.model)
        if i < 2 or i == args.steps - 1 or bad:
            print('[%s] step %d %s' % (tag, i, ' '.join('%s=%.4g' % kv for kv in losses.items())), flush=True)
        if bad:
            print('[%s] step %d non-finite: %d tensors, first: %s' % (tag, i, len(bad), bad[:4]), flush=True)
            break
    else:
        print('[%s] %d steps clean' % (tag, args.steps), flush=True)


if __name__ == '__main__':
    main()
