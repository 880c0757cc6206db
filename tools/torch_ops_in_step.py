"""Which stock PyTorch operators still launch kernels inside one train step, and from where.

The hand-written launches are timed by bench.py's KernelTimer; what it does not see are the stock operators between
them (copies, adds, fills, casts, reductions: `rocprofv3 --stats` shows ~480 such launches per step).  This tool runs
the bench step eagerly under torch.profiler (CPU activity only, with Python stacks), and prints for every aten
operator that launches device work how often it runs per step and the first frame of this package on its stack.

    python tools/torch_ops_in_step.py [--steps 2] [--top 60] > gpurun_out/torch_ops.txt
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'msmc-tts_amd'))

import torch  # noqa: E402

# operators that only make views / bookkeeping: no device work
VIEWS = {'aten::view', 'aten::reshape', 'aten::permute', 'aten::transpose', 'aten::t', 'aten::as_strided', 'aten::select',
         'aten::slice', 'aten::unsqueeze', 'aten::squeeze', 'aten::expand', 'aten::detach', 'aten::alias', 'aten::empty',
         'aten::empty_like', 'aten::empty_strided', 'aten::view_as', 'aten::unbind', 'aten::split', 'aten::narrow',
         'aten::_unsafe_view', 'aten::result_type', 'aten::is_nonzero', 'aten::item', 'aten::_local_scalar_dense',
         'aten::lift_fresh', 'aten::flatten', 'aten::chunk', 'aten::unflatten', 'aten::resolve_conj',
         'aten::resolve_neg', 'aten::set_', 'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::clone',
         'aten::zeros', 'aten::zeros_like', 'aten::ones', 'aten::ones_like', 'aten::full', 'aten::new_empty',
         'aten::new_zeros', 'aten::expand_as', 'aten::movedim', 'aten::unfold', 'aten::size', 'aten::stride',
         'aten::record_stream'}       # (record_stream: caching-allocator bookkeeping of the side branches, host only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--top', type=int, default=80)
    a = ap.parse_args()
    import bench
    args = argparse.Namespace(codewords=256, heads=4, batch=16, frames=400, graph=False, dtype='bf16', no_autocast=False,
                              model_kw=dict(n_heads=4, embedding_sizes=256))
    device = torch.device('cuda', 0)
    torch.cuda.set_device(device)
    cfg, trainer = bench.build(args, device, 0, 1)
    from msmctts_amd.synthetic import make_batch
    batch = make_batch(args.batch, args.frames, 80, 300, seed=1234, rank=0, device='cpu')
    lengths_host = batch['mel_length'].tolist()
    batch = {k: v.to(device) for k, v in batch.items()}
    batch['mel_length_host'] = lengths_host
    import random
    trainer.rng = random.Random(1234)

    # the operators of the step AS THE GRAPHS RECORD IT (window indices on the device, segments A / B / C back to
    # back): what trainers/msmctts_trainer.py::_capture runs, eagerly
    from msmctts_amd.hip import vq as hipvq
    from msmctts_amd.trainers.msmctts_trainer import _StepState
    st = _StepState()
    st.phase = 2
    st.mel, st.mel_length = batch['mel'].clone(), batch['mel_length'].clone()
    g = {'state': st, 'wav': batch['wav'].reshape(args.batch, -1).clone(),
         'starts': torch.zeros(args.batch, dtype=torch.int64, device=device)}

    def step(i):
        trainer.model.zero_grad(set_to_none=True)
        trainer._build_windows(g, st)
        trainer._segment_a(st)
        hipvq.flush_codebook_sync(local=True)
        trainer._segment_b(st)
        trainer._segment_c(st)

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        for i in range(a.steps):
            step(a.warmup + i)
        torch.cuda.synchronize()
    pkgfiles = set()
    for _, _, files in os.walk(os.path.join(ROOT, 'msmc-tts_amd', 'msmctts_amd')):
        pkgfiles.update(f for f in files if f.endswith('.py') and f != '__init__.py')
    table = collections.Counter()
    shapes = {}
    ev = [e for e in prof.events() if e.name.startswith('aten::') and e.name not in VIEWS]
    # leaf operators only: an operator whose interval contains another kept operator is a wrapper
    ev.sort(key=lambda e: (e.time_range.start, -e.time_range.end))
    leaves = []
    for i, e in enumerate(ev):
        nxt = ev[i + 1] if i + 1 < len(ev) else None
        if nxt is not None and nxt.time_range.start < e.time_range.end and nxt.thread == e.thread:
            continue
        leaves.append(e)
    for e in leaves:
        where = None
        frames = [fr for fr in (e.stack or []) if not fr.startswith('<built-in')]
        for fr in frames:                     # 'path/file.py(line): function'
            path = fr.split('(')[0]
            if os.path.basename(path) in pkgfiles and 'torch/' not in path and 'site-packages' not in path:
                where = fr.split('msmctts_amd/')[-1]
                break
        if where is None:
            where = ' < '.join(fr.split('/')[-1] for fr in frames[:3]) or '(no Python frame: autograd engine)'
        key = (e.name, where)
        table[key] += 1
        shapes.setdefault(key, str(e.input_shapes)[:70])
    print('leaf aten operators per step (%d profiled steps), %d in total per step' %
          (a.steps, sum(table.values()) // a.steps))
    for (name, where), n in table.most_common(a.top):
        print('%6.1f  %-28s %-110s %s' % (n / a.steps, name, where[:110], shapes[(name, where)]))


if __name__ == '__main__':
    main()
