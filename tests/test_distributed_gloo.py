"""CPU, world_size 2 over gloo: the data-parallel path (msmctts_amd/distributed/distributed.py).

* start-up broadcast makes every rank's parameters AND buffers equal to rank 0's;
* bucketed, hook-driven all-reduce averages gradients: N ranks x shard == 1 process x full batch;
* parameters that received no gradient are not communicated and keep ``grad is None``;
* the product VQGANTrainer steps in lock-step on 2 ranks (kernel interpreter build), parameters stay
  identical across ranks while VQ codebooks -- like the reference -- are NOT synchronised.
"""
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Sequential(nn.Linear(6, 16), nn.Tanh(), nn.Linear(16, 16), nn.Tanh(), nn.Linear(16, 3))
        self.b = nn.Sequential(nn.Linear(3, 8), nn.Tanh(), nn.Linear(8, 1))
        self.register_buffer('stat', torch.zeros(4))


def _toy_worker(rank, world, port, out, exchange=None):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
    from msmctts_amd.distributed.distributed import apply_gradient_allreduce, init_distributed
    torch.set_num_threads(1)
    init_distributed(rank, world, 'g', 'gloo', 'tcp://127.0.0.1:%d' % port)
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
    m = Toy()
    m.stat.fill_(float(rank + 1))
    apply_gradient_allreduce(m, bucket_bytes=600, exchange_dtype=exchange)   # tiny buckets -> several collectives per backward
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    n = 8 // world
    xs, ys = x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]
    # step 1: only child a participates (like the D-frozen G step)
    loss = ((m.a(xs) - ys) ** 2).mean()
    loss.backward()
    m.grad_reducer.finish()
    ga = [p.grad.clone() for p in m.a.parameters()]
    b_none = all(p.grad is None for p in m.b.parameters())
    # step 2: both children
    m.zero_grad()
    loss = (m.b(m.a(xs)) ** 2).mean()
    loss.backward()
    m.grad_reducer.finish()
    gb = [p.grad.clone() for p in m.parameters()]
    # step 3: the hipGraph-mode exchange -- hooks off, static gradient tensors recorded once, a loop that resets
    # p.grad must not turn the exchange into a no-op (zero_grad vs the reducer in graph mode)
    red = m.grad_reducer
    red.hooks_enabled = False
    m.zero_grad()
    (m.b(m.a(xs)) ** 2).mean().backward()
    static = [p.grad for p in m.b.parameters()]
    for p in m.b.parameters():
        p.grad = None                                # what ``optimizer.zero_grad(set_to_none=True)`` does
    red.allreduce_child(m.b, grads=static)
    gs = [g.clone() for g in static]
    try:
        red.allreduce_child(m.b)                     # nothing to exchange: must raise, not silently diverge
        raised = False
    except RuntimeError:
        raised = True
    out[rank] = dict(state={k: v.clone() for k, v in m.state_dict().items()}, ga=ga, b_none=b_none, gb=gb, gs=gs,
                     raised=raised, nbuckets=len(m.grad_reducer.buckets))
    dist.barrier()
    dist.destroy_process_group()


def test_bf16_wire_format_keeps_ranks_identical_and_close_to_fp32():
    """``exchange_dtype=torch.bfloat16`` (MSMC_GRAD_EXCHANGE=bf16): half the bytes per exchange; every rank still ends
    with the SAME gradients (the averaged values are what each rank reads back), within bf16 rounding of the full-batch
    gradient -- bucketed path and graph-mode static exchange alike"""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_toy_worker, args=(2, port, out, torch.bfloat16), nprocs=2, join=True)
    r0, r1 = out[0], out[1]
    ref = Toy()
    ref.load_state_dict(r0['state'])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    (ref.b(ref.a(x)) ** 2).mean().backward()
    for a, b, c in zip(r0['gb'], r1['gb'], [p.grad for p in ref.parameters()]):
        assert torch.equal(a, b) and a.dtype == torch.float32
        assert torch.allclose(a, c, atol=2e-2 * float(c.abs().max()) + 1e-6)
        assert not torch.allclose(a, c, atol=1e-7) or float(c.abs().max()) == 0.0      # (it really went through bf16)
    for a, b, c in zip(r0['gs'], r1['gs'], [p.grad for p in ref.b.parameters()]):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, atol=2e-2 * float(c.abs().max()) + 1e-6)


@pytest.mark.parametrize('world', [2, 4])
def test_bucketed_allreduce_equals_full_batch(world):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_toy_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[world - 1]
    assert r0['nbuckets'] >= 3
    for k in r0['state']:
        assert torch.equal(r0['state'][k], r1['state'][k]), k
    assert float(r0['state']['stat'][0]) == 1.0                      # rank 0's buffer won
    ref = Toy()
    ref.load_state_dict(r0['state'])
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g)
    y = torch.randn(8, 3, generator=g)
    ((ref.a(x) - y) ** 2).mean().backward()
    for a, b, c in zip(r0['ga'], r1['ga'], [p.grad for p in ref.a.parameters()]):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, atol=1e-6)
    assert r0['b_none'] and r1['b_none']
    ref.zero_grad()
    (ref.b(ref.a(x)) ** 2).mean().backward()
    for a, b, c in zip(r0['gb'], r1['gb'], [p.grad for p in ref.parameters()]):
        assert torch.equal(a, b)
        assert torch.allclose(a, c, atol=1e-6)
    for a, b, c in zip(r0['gs'], r1['gs'], [p.grad for p in ref.b.parameters()]):        # graph-mode exchange
        assert torch.equal(a, b)
        assert torch.allclose(a, c, atol=1e-6)
    assert r0['raised'] and r1['raised']


def _codebook_worker(rank, world, port, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
    from msmctts_amd.distributed.distributed import init_distributed
    from msmctts_amd.hip import lib, vq as hipvq
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize
    torch.set_num_threads(1)
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    init_distributed(rank, world, 'g', 'gloo', 'tcp://127.0.0.1:%d' % port)
    torch.manual_seed(3)
    q = MultiHeadQuantize(32, 16, 4).train()
    q.sync_stats = True
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 37, 32, generator=g)
    ln = torch.tensor([37, 20, 5, 31])
    n = 4 // world
    for _ in range(2):
        q(x[rank * n:(rank + 1) * n], ln[rank * n:(rank + 1) * n], update=True)
        assert len(hipvq.PENDING) == 1
        hipvq.flush_codebook_sync()
        assert not hipvq.PENDING
    out[rank] = {k: v.clone() for k, v in q.state_dict().items()}
    dist.barrier()
    dist.destroy_process_group()


def test_codebook_statistics_sync_equals_single_process_global_batch():
    """sync_codebook_stats: two ranks x half the batch == one process x the whole batch (the fused update)"""
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_codebook_worker, args=(2, port, out), nprocs=2, join=True)
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd')]
    from msmctts_amd.hip import lib
    from msmctts_amd.networks.vqgantts.modules import MultiHeadQuantize
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    torch.manual_seed(3)
    q = MultiHeadQuantize(32, 16, 4).train()
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 37, 32, generator=g)
    ln = torch.tensor([37, 20, 5, 31])
    for _ in range(2):
        q(x, ln, update=True)
    want = q.state_dict()
    for k in want:
        assert torch.equal(out[0][k], out[1][k]), k
        assert torch.allclose(out[0][k], want[k], rtol=1e-5, atol=1e-6), k
    # and on one process the two-launch path is the fused update, bit for bit
    torch.manual_seed(3)
    q2 = MultiHeadQuantize(32, 16, 4).train()
    q2.sync_stats = True
    from msmctts_amd.hip import vq as hipvq
    for _ in range(2):
        q2(x, ln, update=True)
        hipvq.flush_codebook_sync()
    for k, v in q2.state_dict().items():
        assert torch.equal(v, want[k]), k


def _trainer_worker(rank, world, port, out, sync_codebooks=False):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
    import random
    import _parity
    from msmctts_amd.distributed.distributed import init_distributed
    from msmctts_amd.hip import lib
    from msmctts_amd.synthetic import make_batch
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    torch.set_num_threads(2)
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    init_distributed(rank, world, 'g', 'gloo', 'tcp://127.0.0.1:%d' % port)
    cfg, task = _parity.build_small('cpu')
    if rank == 1:
        with torch.no_grad():
            for p in task.parameters():
                p.add_(0.01)                                          # must be overwritten by the broadcast
    if sync_codebooks:
        cfg.trainer.sync_codebook_stats = True
    tr = build_trainer(cfg, task, num_gpus=world, rank=rank)
    assert tr.sync_codebook_stats == sync_codebooks
    tr.optimizer = build_optimizer(tr.model, cfg.optimizer)
    tr.rng = random.Random(5 + rank)
    batch = make_batch(3, 24, 80, 300, seed=11, rank=rank)
    logs = []
    for it in (0, 6):                                                 # one warm-up step, one GAN step
        tr.model.zero_grad()
        tr.optimizer.zero_grad()
        logs.append({k: float(v) for k, v in tr.train_step(batch, it)['loss'].items()})
    out[rank] = dict(state={k: v.clone() for k, v in tr.model.state_dict().items()}, logs=logs)
    dist.barrier()
    dist.destroy_process_group()


def test_vqgan_trainer_two_ranks_stay_in_sync():
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(2, port, out), nprocs=2, join=True)
    s0, s1 = out[0]['state'], out[1]['state']
    n_param = n_buf_diff = 0
    for k in s0:
        if k.endswith(('.embed', '.cluster_size', '.embed_avg')):
            n_buf_diff += int(not torch.equal(s0[k], s1[k]))          # per-rank EMA, reference semantics
        else:
            assert torch.equal(s0[k], s1[k]), k
            n_param += 1
    assert n_param > 300 and n_buf_diff > 0
    assert out[0]['logs'][0]['frame_loss'] != out[1]['logs'][0]['frame_loss']     # different shards
    assert all(torch.isfinite(torch.tensor(list(l.values()))).all() for l in out[0]['logs'])


def test_vqgan_trainer_sync_codebook_stats_keeps_codebooks_identical():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_trainer_worker, args=(2, port, out, True), nprocs=2, join=True)
    s0, s1 = out[0]['state'], out[1]['state']
    n = 0
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
        n += k.endswith(('.embed', '.cluster_size', '.embed_avg'))
    assert n >= 6


def _predictor_worker(rank, world, port, out):
    """PredictorTrainer in graph mode under data parallelism with batch shapes that differ between ranks and steps: a rank
    whose batch has the captured shape replays, the other steps eagerly -- both must meet in the SAME collective and keep
    their parameters identical.  torch.cuda.graph does not exist on the CPU: ``_capture`` is replaced by a stand-in whose
    'graphs' run the two halves of the step eagerly (what a replay does, minus the recording); everything else -- the
    replay-or-eager decision, the exchange, the optimizer -- is the product's."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'msmc-tts_amd'), os.path.join(ROOT, 'tests')]
    import _parity
    from msmctts_amd.distributed.distributed import init_distributed
    from msmctts_amd.hip import lib
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    torch.set_num_threads(2)
    lib.use_library_for_tests(os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so'))
    init_distributed(rank, world, 'g', 'gloo', 'tcp://127.0.0.1:%d' % port)
    z = _parity.load_npz('small_predictor.npz')
    cfg = Config({'id': 'small_predictor_dp', 'task': {'_name': 'MSMCTTS', '_mode': 'train_predictor', 'predictor': _parity.small_predictor_cfg()},
                  'trainer': dict(_parity.PREDICTOR_TRAINER, _name='PredictorTrainer'),
                  'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    task = build_task(cfg, mode='train')
    task.load_state_dict({k[len('state.'):]: _parity.t(v) for k, v in z.items() if k.startswith('state.')})
    task = task.train()
    _, atask = _parity.build_small('cpu')
    tr = build_trainer(cfg, task, num_gpus=world, rank=rank)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    tr.use_graphs = True
    full = {k[len('batch.'):]: _parity.t(v) for k, v in z.items() if k.startswith('batch.')}
    B = full['mel'].shape[0]
    assert B >= 2
    short = {k: v[:B - 1].clone() for k, v in full.items()}           # one utterance less: every tensor changes shape

    class _Replay(object):
        def __init__(self, fn):
            self.fn = fn

        def replay(self):
            self.fn()

    def fake_capture(batch, shapes):
        static = {k: batch[k].clone() for k in tr._KEYS}
        box = {}
        vec = torch.zeros(16)

        def ab():
            box['losses'] = tr._forward_backward(static, static=True)

        def c():
            tr._update(box['losses'])
            keys = [k for k, v in box['losses'].items() if torch.is_tensor(v)]
            g['loss_keys'] = keys
            vec[:len(keys)] = torch.stack([box['losses'][k].detach().float().reshape(()) for k in keys])
        g = dict(ab=_Replay(ab), c=_Replay(c), batch=static, shapes=shapes, loss_vec=vec, loss_keys=[])
        return g
    tr._capture = fake_capture
    # step 0: both ranks capture on the full batch; step 1: rank 0 meets another shape (eager) while rank 1 replays;
    # step 2: the other way round; step 3: both eager
    plan = [(full, full), (short, full), (full, short), (short, short)]
    calls = []
    real = tr.model.grad_reducer.allreduce_child
    tr.model.grad_reducer.allreduce_child = lambda child, grads=None: (calls.append(grads is None), real(child, grads=grads))[1]
    for it, pair in enumerate(plan):
        log = tr.train_step({k: v.clone() for k, v in pair[rank].items()}, it)
        assert all(bool(torch.isfinite(v)) for v in log['loss'].values() if torch.is_tensor(v))
    out[rank] = dict(state={k: v.clone() for k, v in tr.model.state_dict().items()}, calls=calls,
                     init={k[len('state.'):]: _parity.t(v) for k, v in z.items() if k.startswith('state.')})
    dist.barrier()
    dist.destroy_process_group()


def test_predictor_graph_mode_two_ranks_with_different_batch_shapes_stay_in_sync():
    """round-5 advisor finding: the eager branch of PredictorTrainer._train_step_graphed exchanged nothing under data
    parallelism (the first replay had switched the reducer's hooks off) and a rank in it met no collective while a replaying
    rank waited in one -- a hang from the second step on with segment_length = -1.  Both branches now issue the one flat
    all-reduce of the child (a wrong pairing would time out here or leave the ranks' parameters different)."""
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_predictor_worker, args=(2, port, out), nprocs=2, join=True)
    s0, s1 = out[0]['state'], out[1]['state']
    moved = 0
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k
        if s0[k].dtype.is_floating_point and k in out[0]['init']:
            moved += int(not torch.equal(s0[k], out[0]['init'][k]))
    assert moved > 50                                     # the four steps really trained
    assert len(out[0]['calls']) == len(out[1]['calls']) == 4       # exactly one exchange per step on either rank


# ---- bench.py as the driver launches it: ``python bench.py --gpus N`` with WORLD_SIZE unset must start its own ranks ----
def _bench(argv, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + argv, env=e, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize('world', [2, 4])
def test_bench_starts_its_own_ranks_and_prints_one_line(world):
    """the launcher logic of ``bench.py --gpus N`` on CPU (``--dry --backend gloo``: toy model, the product's GradReducer):
    N ranks come up through torch.distributed.run on 127.0.0.1, the start-up broadcast repairs a deliberately divergent
    rank, the bucketed exchange keeps the ranks bit-identical, rank 0 prints ONE JSON line with n_gpus, the world size the
    process group reported and a step time per rank; exit code 0"""
    import json
    r = _bench(['--gpus', str(world), '--dry', '--backend', 'gloo', '--steps', '4', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out['n_gpus'] == world and out['world_size_seen'] == world and out['backend'] == 'gloo'
    assert len(out['per_rank_ms_per_step']) == world and all(t > 0 for t in out['per_rank_ms_per_step'])
    assert out['ranks_identical'] and out['config']['gradient_exchange'] == 'bucketed from hooks' and out['config']['buckets'] >= 2
    assert out['steps'] == 4 and out['dry'] is True


def test_bench_turns_a_wedged_rank_into_a_nonzero_exit():
    """a rank that stops making progress (here: the last rank sleeps before its timed steps, the others wait in the
    barrier) ends the job with a non-zero exit code within the stall timeout -- no line, no hang"""
    import time
    t0 = time.time()
    r = _bench(['--gpus', '2', '--dry', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--stall-timeout', '5'],
               env={'MSMC_BENCH_TEST_STALL': '120'}, timeout=200)
    assert r.returncode != 0
    assert time.time() - t0 < 100
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert 'no progress' in r.stderr or 'imeout' in r.stderr, r.stderr[-1500:]


def test_bench_launcher_kills_a_job_that_outlives_its_limit():
    """the self-spawned job is killed (its own session) when it outlives --job-timeout: exit code 124"""
    r = _bench(['--gpus', '2', '--dry', '--backend', 'gloo', '--steps', '2', '--warmup', '1', '--stall-timeout', '0',
                '--job-timeout', '8'], env={'MSMC_BENCH_TEST_STALL': '120'}, timeout=200)
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r = _bench(['--gpus', '2', '--dry', '--backend', 'gloo'], env={'WORLD_SIZE': '1', 'RANK': '0'}, timeout=120)
    assert r.returncode != 0 and 'WORLD_SIZE' in r.stderr
