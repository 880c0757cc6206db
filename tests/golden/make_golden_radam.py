#!/usr/bin/env python
"""Generate ``radam_cases.npz`` from the reference's own RAdam (msmctts/trainers/optimizers/radam.py:8-85).

Runs ONLY in the build container (needs /root/reference).  Fixture = data: initial parameters, the gradient sequence and
the parameters / moments the reference optimizer holds after every step, for several (lr, betas, eps, weight_decay)
settings; eight steps each, so that both branches of the variance rectification (N_sma < 5: momentum step, N_sma >= 5:
rectified Adam step) are taken (betas[1] = 0.99 crosses at step 6, the class default 0.9 never leaves the first branch
within 8 steps of ... see ``CASES``).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shims  # noqa: E402

_ref_shims.install()

import torch  # noqa: E402
from msmctts.trainers.optimizers.radam import RAdam  # noqa: E402

CASES = {           # name -> constructor keywords (None: the class defaults)
    'csmsc': dict(lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.0),
    'decay': dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.01),
    'default': None,
}
STEPS = 8
SHAPES = [(7, 5), (33,), (2, 3, 4)]


def main():
    out = {}
    g = torch.Generator().manual_seed(11)
    for name, kw in CASES.items():
        params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in SHAPES]
        opt = RAdam(params, **kw) if kw is not None else RAdam(params)
        for i, p in enumerate(params):
            out['%s/p0/%d' % (name, i)] = p.detach().numpy().copy()
        for t in range(STEPS):
            for i, p in enumerate(params):
                p.grad = torch.randn(p.shape, generator=g) * (0.1 + 0.3 * t)
                out['%s/g%d/%d' % (name, t, i)] = p.grad.numpy().copy()
            opt.step()
            for i, p in enumerate(params):
                out['%s/p%d/%d' % (name, t + 1, i)] = p.detach().numpy().copy()
        for i, p in enumerate(params):
            st = opt.state[p]
            out['%s/exp_avg/%d' % (name, i)] = st['exp_avg'].numpy().copy()
            out['%s/exp_avg_sq/%d' % (name, i)] = st['exp_avg_sq'].numpy().copy()
            out['%s/step/%d' % (name, i)] = np.array(st['step'])
    np.savez_compressed(os.path.join(HERE, 'radam_cases.npz'), **out)
    print('radam_cases.npz: %d arrays' % len(out))


if __name__ == '__main__':
    main()
