"""GPU (-m gpu): FULL-SIZE train steps of the BASELINE.json configurations.

* fp32 product step vs the oracle (oracle/step.py, CPU) on the same weights / batch / windows at CSMSC size
  for config #2 (2 stages, 4 heads x 256) at the BENCHMARKED batch (B=16, T=400), #1 (1 stage, 1 head x 64) and #5
  (in_dim 1024, 8 heads x 512) at B=4, T=400: VQ indices exact, every loss key <= 1e-3 relative, per-parameter gradient
  norms <= 2e-3, post-step VQ buffers <= 1e-5 of their scale.
* configs #1 and #5 on the performance path (bf16 autocast, grouped launches, hipGraph replay; B=4 and B=16): first-step
  losses within 2 % of the same model's fp32 eager step, finite losses over the following steps.
* config #2 exactly as bench.py runs it (B=16, bf16 autocast, grouped launches, hipGraph replay): >= 10 GAN steps,
  finite losses that fall, graph == eager and grouped == ungrouped step by step within the stated bf16 bound,
  bf16 within a stated bound of the fp32 step.
* the runtime behaviour the graph path relies on (memset nodes ordered on replay).
"""
import copy
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')

CONFIGS = {
    'config2': dict(embedding_sizes=256, n_heads=4),
    'config1': dict(downsample_scales=(1,), n_heads=1, embedding_sizes=64),
    'config5': dict(in_dim=1024, n_heads=8, embedding_sizes=512),
}


@pytest.fixture(scope='module', autouse=True)
def real_library():
    from msmctts_amd.hip import lib
    assert lib.backend() == 'gfx950', 'GPU tests must run on the real HIP library'
    torch.backends.cuda.matmul.allow_tf32 = False
    yield


def _cfg(batch, dropout=True, **kw):
    from msmctts_amd.configs import csmsc_config
    from msmctts_amd.utils.config import Config
    d = csmsc_config(batch_size=batch, warmup_steps=0, **kw)
    if not dropout:                      # (attention dropout is a config value, not an nn.Dropout module)
        d['task'] = _no_dropout_dict(d['task'])
    return Config(d)


def _no_dropout_dict(d):
    """config dictionary with every dropout probability zero (the oracle reads them from the config)"""
    d = copy.deepcopy(d)
    ae = d['autoencoder']
    for key in ('encoder_config', 'frame_decoder_config'):
        ae[key]['dropout'] = 0.0
        ae[key]['attn_dropout'] = 0.0
    ae['quantizer_config']['dropout'] = 0.0
    return d


def _build(cfg, graph=False, dtype=None, dropout=True, seed=None):
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    torch.manual_seed(cfg.seed if seed is None else seed)
    task = build_task(cfg, mode='train')
    if not dropout:
        for m in task.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    tr = build_trainer(cfg, task, num_gpus=1, rank=0)
    tr.optimizer = build_optimizer(tr.model, cfg.optimizer, capturable=graph)
    tr.use_graphs = graph
    tr.amp_dtype = dtype
    tr.model.train()
    return tr


def _batch(B, T, in_dim):
    from msmctts_amd.synthetic import make_batch
    b = make_batch(B, T, in_dim, 300, seed=1234, rank=0, device='cpu')
    host = b['mel_length'].tolist()
    dev = {k: v.to(DEV) for k, v in b.items()}
    dev['mel_length_host'] = host
    return b, dev


def _rel(a, b, floor):
    return abs(a - b) / max(abs(b), floor)


@pytest.mark.parametrize('name,B', [('config2', 16), ('config1', 4), ('config5', 4), ('config5', 16)])
def test_fp32_step_matches_oracle(name, B):
    from oracle import model as omodel
    from oracle.step import OracleTrainer
    omodel.RESSTACK_DROPOUT = 0.0
    kw = CONFIGS[name]
    T = 400
    cfg = _cfg(B, dropout=False, **kw)
    tr = _build(cfg, dropout=False)
    task = tr.model
    in_dim = kw.get('in_dim', 80)
    cpu_batch, batch = _batch(B, T, in_dim)
    r = random.Random(7)
    fw = []
    for n in batch['mel_length_host']:
        s = r.randrange(max(1, n - 40))
        fw.append((s, s + 40))
    sw = [(s * 300, e * 300) for s, e in fw]
    state0 = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
    tcfg = {k: v for k, v in cfg.trainer.to_dict().items() if k != '_name'}
    oracle = OracleTrainer(state0, cfg.task.to_dict(), tcfg)

    # ---- forward in eval mode (no EMA update): indices exact, outputs 1e-3
    task.eval()
    with torch.no_grad():
        out = task.autoencoder(batch['mel'], batch['mel_length'], warmup=False, window=fw)
        want = omodel.msmc_vqgan_forward(oracle.P, oracle.acfg, cpu_batch['mel'], cpu_batch['mel_length'], warmup=False,
                                         window=fw, training=False)
    for i, (a, b) in enumerate(zip(out['encoder_indices'], want['encoder_indices'])):
        assert np.array_equal(a.cpu().numpy(), b.numpy()), '%s: VQ indices of stage %d differ' % (name, i)
    for key in ('mel_outputs', 'decoder_outputs'):
        err = (out[key].float().cpu() - want[key]).abs().max().item()
        assert err <= 1e-3, '%s %s: max abs err %.3e' % (name, key, err)
    task.train()

    # ---- one GAN-phase train step
    tr.random_select = lambda ml: (fw, sw)
    snaps = {}
    real_step = tr.optimizer.step

    def spy(names=None):
        key = names[0] if isinstance(names, (list, tuple)) else names
        torch.cuda.synchronize()
        snaps[key] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters()
                      if n.startswith(key + '.') and p.grad is not None}
        return real_step(names)
    real_clip_step = tr.optimizer.clip_and_step

    def clip_spy(name, max_norm):               # the fused clip + AdamW: gradients are clipped in place, as clip_grad_norm_ does
        out = real_clip_step(name, max_norm)
        torch.cuda.synchronize()
        snaps[name] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters()
                       if n.startswith(name + '.') and p.grad is not None}
        return out
    tr.optimizer.step = spy
    tr.optimizer.clip_and_step = clip_spy
    task.zero_grad()
    log = tr.train_step(batch, 10)
    keep = {}
    ref = oracle.train_step({k: v for k, v in cpu_batch.items()}, 10, windows=(fw, sw), keep=keep)
    got = {k: float(v) for k, v in log['loss'].items()}
    assert set(got) == set(ref['loss']), (sorted(got), sorted(ref['loss']))
    for k, w in ref['loss'].items():
        assert _rel(got[k], w, 1e-3) <= 1e-3, '%s loss %s: %.6g vs oracle %.6g' % (name, k, got[k], w)
    # gradients (before clipping): per-parameter L2 norms, D step and G step
    # (the product's autoencoder gradients are read at its optimizer step, i.e. after clipping: clip the oracle's too)
    offenders = []
    clip = min(1.0, cfg.trainer.grad_clip_thresh / (keep['grad_norm'] + 1e-6))
    print('%s grad_norm %.6g oracle %.6g' % (name, float(tr.grad_norm), keep['grad_norm']))
    for child, okey, factor in (('discriminator', 'd_grads', 1.0), ('autoencoder', 'g_grads', clip)):
        scale = factor * max(v.double().norm().item() for v in keep[okey].values())     # largest gradient of the child
        for n, g in snaps[child].items():
            if n not in keep[okey]:
                continue
            w, m = factor * keep[okey][n].double().norm().item(), g.double().norm().item()
            if abs(m - w) > 2e-3 * w + 1e-6 * scale:
                offenders.append((n, m, w))
    offenders.sort(key=lambda o: -abs(o[1] - o[2]))
    assert not offenders, '%s: %d gradient norms off (name, got, oracle): %s' % (name, len(offenders), offenders[:8])
    # post-step VQ buffers
    sd = task.state_dict()
    for k, v in oracle.P.items():
        if k.endswith('.embed') or k.endswith('.cluster_size') or k.endswith('.embed_avg'):
            scale = max(1.0, float(v.abs().max()))
            err = (sd[k].float().cpu() - v).abs().max().item()
            assert err <= 1e-5 * scale, '%s buffer %s: %.3e (scale %.3g)' % (name, k, err, scale)


def _groups(named):
    """parameter tensors by bank-sized group: the first three components of the name (autoencoder.decoder.ups, discriminator.mrd.2 ...)"""
    out = {}
    for n, t in named:
        out.setdefault('.'.join(n.split('.')[:3]), []).append((n, t))
    return out


def test_bf16_graphed_step_gradients_match_the_oracle():
    """The path bench.py TIMES -- config #2 at B = 16, bf16 autocast, grouped launches, the step replayed from hipGraphs --
    against the ORACLE's fp32 step on the same weights, batch and windows: every loss of the first step within 2 %, and the
    GRADIENTS the graphs leave in their static tensors (discriminator: as its optimizer consumed them; autoencoder: after the
    in-place clipping, so the oracle's are scaled by its own clip coefficient) per group of parameters (name prefix of depth
    three: one generator stage, one sub-discriminator, one FFT stack ...): cosine >= 0.998 and relative L2 error <= 6e-2 against
    the oracle's fp32 gradients.  Measured on MI355X (profiles/r05_bf16_gradient_parity.txt): discriminators 2e-3, encoders /
    frame decoder / quantiser 1-1.3e-2, the vocoder -- whose gradient has crossed all ten sub-discriminators and its own
    eighteen-layer stages in bf16 -- 3.5-4.5e-2 at cosine 0.9992-0.9997.  A wrong weight gradient on one layer moves its group
    to a relative error of order one."""
    from oracle import model as omodel
    from oracle.step import OracleTrainer
    omodel.RESSTACK_DROPOUT = 0.0
    B, T = 16, 400
    cfg = _cfg(B, dropout=False, **CONFIGS['config2'])
    tr = _build(cfg, graph=True, dtype=torch.bfloat16, dropout=False)
    task = tr.model
    cpu_batch, batch = _batch(B, T, 80)
    state0 = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
    tcfg = {k: v for k, v in cfg.trainer.to_dict().items() if k != '_name'}
    oracle = OracleTrainer(state0, cfg.task.to_dict(), tcfg)
    tr.rng = random.Random(99)
    r = random.Random(99)                         # the window starts the graphed step will draw (trainer._train_step_graphed)
    fw = []
    for n in batch['mel_length_host']:
        s0 = r.randrange(max(1, int(n) - tr.frame_lengths))
        fw.append((s0, s0 + tr.frame_lengths))
    sw = [(a * tr.frameshift, b * tr.frameshift) for a, b in fw]
    log = tr.train_step(batch, 10)                # capture (its eager warm-up is rolled back) + the first replay
    torch.cuda.synchronize()
    assert tr._graphs is not None
    got = {k: float(v) for k, v in log['loss'].items()}
    keep = {}
    ref = oracle.train_step({k: v for k, v in cpu_batch.items()}, 10, windows=(fw, sw), keep=keep)
    assert set(got) == set(ref['loss']), (sorted(got), sorted(ref['loss']))
    for k, w in ref['loss'].items():
        assert _rel(got[k], w, 1e-2) <= 2e-2, 'loss %s: %.6g vs oracle %.6g' % (k, got[k], w)
    clip = min(1.0, cfg.trainer.grad_clip_thresh / (keep['grad_norm'] + 1e-6))
    rows, bad, per_param = [], [], []
    for child, okey, factor in (('discriminator', 'd_grads', 1.0), ('autoencoder', 'g_grads', clip)):
        named = [(n, p.grad.detach().float().cpu()) for n, p in task.named_parameters()
                 if n.startswith(child + '.') and p.grad is not None and n in keep[okey]]
        assert len(named) >= 0.9 * len(keep[okey]), (child, len(named), len(keep[okey]))
        for gname, members in sorted(_groups(named).items()):
            a = torch.cat([t.double().reshape(-1) for _, t in members])
            b = torch.cat([factor * keep[okey][n].double().reshape(-1) for n, _ in members])
            rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
            cos = (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)).item()
            rows.append((gname, a.numel(), rel, cos))
            if not (rel <= 6e-2 and cos >= 0.998):      # (margin over the measured 4.5e-2 / 0.99915: weight-gradient sums are not order-fixed)
                bad.append((gname, a.numel(), rel, cos))
        # per PARAMETER (round 6: a group is up to 18 layers -- one wrong thin layer moves its group far less than "order
        # one"): cosine >= 0.995 for every tensor of at least 4096 values whose oracle gradient is not vanishing
        for n, t in named:
            a, b = t.double().reshape(-1), factor * keep[okey][n].double().reshape(-1)
            if t.numel() < 4096 or b.norm() < 1e-9 * max(1.0, float(b.numel()) ** 0.5):
                continue
            cos = (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)).item()
            per_param.append((cos, n))
            if cos < 0.995:                       # (measured worst 0.9985: the C = 32 ResBlock layers behind ten sub-discriminators)
                bad.append((n, t.numel(), float('nan'), cos))
    for row in rows:
        print('%-44s %9d values  rel L2 %.3e  cosine %.6f' % row)
    print('worst per-parameter cosines: %s' % sorted(per_param)[:5])
    assert not bad, 'gradient groups / parameters outside the bf16 bound (name, values, rel L2, cosine): %s' % bad


@pytest.mark.parametrize('prologue', [False, True])
def test_fft_block_stack_on_the_attention_kernels_matches_the_oracle(prologue):
    """csrc/attn.hip INSIDE a model (the fp32 parity runs route around it to the stock fused operator): the CSMSC FFT-block
    stack (4 blocks, 2 heads of 64, d_model 256, FFN 1024; reference acoustic_models/transformer.py:71-385) in bf16 with the
    attention core forced onto the kernels, ragged lengths, against oracle/model.py's fp32 restatement of the same stack
    (pinned by small_modules.npz) on the same weights: output and input gradient at a bf16 bound.  ``prologue``: the same
    with the stack's head (positions, positional add, masks) as the one-launch prologue."""
    from msmctts_amd.hip import attn as hipattn
    from msmctts_amd.networks.acoustic_models import transformer as tfm
    from oracle import model as omodel
    cfg = dict(max_seq_len=2400, n_layers=4, n_head=2, d_k=64, d_v=64, d_model=256, d_inner=1024, fft_conv1d_kernel=3,
               fft_conv1d_padding=1, dropout=0.0, attn_dropout=0.0)
    torch.manual_seed(3)
    net = tfm.FFTBlocks(name='enc', **cfg).to(DEV)
    net.hip_dtype = torch.bfloat16
    net.train()
    B, T = 4, 400
    lengths = torch.tensor([400, 333, 250, 201])
    pos = omodel.position_ids(lengths, T)
    x_cpu = torch.randn(B, T, 256) * pos.ne(0).unsqueeze(-1)
    calls = []
    real = hipattn.attention
    keep_flag = tfm.FFT_PROLOGUE
    try:
        hipattn.attention = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        tfm.FFT_PROLOGUE = prologue
        x = x_cpu.to(DEV).requires_grad_(True)
        out, _ = net(x, None if prologue else pos.to(DEV), lengths=lengths.to(DEV))
        go = torch.randn(B, T, 256)
        (out.float() * go.to(DEV)).sum().backward()
    finally:
        hipattn.attention = real
        tfm.FFT_PROLOGUE = keep_flag
    assert len(calls) == cfg['n_layers'], 'the attention core did not run on csrc/attn.hip'
    P = {'enc.' + k: v.detach().float().cpu() for k, v in net.state_dict().items()}
    xo = x_cpu.clone().requires_grad_(True)
    want = omodel.fft_blocks(P, 'enc', xo, pos, cfg, training=False)
    (want * go).sum().backward()
    got = out.float().cpu()
    err = (got - want).abs().max().item()
    assert err <= 6e-2 * max(1.0, want.abs().max().item()), 'FFT stack output: max abs err %.3e (scale %.3g)' % (err, want.abs().max())
    rel = ((got - want).norm() / want.norm()).item()
    assert rel <= 1.5e-2, 'FFT stack output: relative L2 error %.3e' % rel
    grel = ((x.grad.float().cpu() - xo.grad).norm() / xo.grad.norm()).item()
    assert grel <= 3e-2, 'FFT stack input gradient: relative L2 error %.3e' % grel
    assert float(got[1, 333:].abs().max()) == 0.0            # padded rows stay masked


def test_graph_memset_nodes_are_ordered():
    """the graph path's precondition (msmctts_amd/__init__.py): fails on ROCm 7.2's AQL packet-capture graphs"""
    from msmctts_amd.hip import graphs
    assert graphs.memset_nodes_ordered(DEV), graphs.HINT


def _run_steps(tr, batch, n, seed=99):
    tr.rng = random.Random(seed)
    rows = []
    for i in range(n):
        if not tr.replays(10 + i):
            tr.model.zero_grad()
            tr.optimizer.zero_grad()
        log = tr.train_step(batch, 10 + i)
        rows.append({k: float(v) for k, v in log['loss'].items()})
    torch.cuda.synchronize()
    return rows


KEYS = ('vq_loss', 'frame_loss', 'stft_loss', 'd_loss', 'fm_loss', 'adv_loss', 'g_loss')


def _max_dev(a_rows, b_rows, steps):
    worst = 0.0
    for i in steps:
        for k in KEYS:
            worst = max(worst, _rel(a_rows[i][k], b_rows[i][k], 1e-2))
    return worst


def test_config2_bf16_graphed_grouped_step_trains():
    """BASELINE config #2 as bench.py runs it (B=16, bf16 autocast, grouped launches, hipGraph replay): first with the
    configuration's dropout (finite, falling losses), then -- dropout off, so that trajectories are comparable --
    against the same model stepped eagerly, ungrouped and in fp32.  Bounds are per-loss relative deviations."""
    from msmctts_amd.hip import convnet
    N = 12
    cfg = _cfg(16, **CONFIGS['config2'])
    _, batch = _batch(16, 400, 80)
    tr = _build(cfg, graph=True, dtype=torch.bfloat16, dropout=True)
    rows = _run_steps(tr, batch, N)
    assert all(np.isfinite(v) for row in rows for v in row.values()), rows
    assert rows[-1]['stft_loss'] < 0.7 * rows[0]['stft_loss'], (rows[0], rows[-1])
    del tr
    torch.cuda.empty_cache()
    cfg = _cfg(16, dropout=False, **CONFIGS['config2'])
    runs = {}
    for tag, graph, grouped, dtype in (('graph', True, True, torch.bfloat16), ('eager', False, True, torch.bfloat16),
                                       ('ungrouped', False, False, torch.bfloat16), ('fp32', False, True, None)):
        keep = convnet.GROUPED
        convnet.GROUPED = grouped
        try:
            tr = _build(cfg, graph=graph, dtype=dtype, dropout=False)
            runs[tag] = _run_steps(tr, batch, N)
        finally:
            convnet.GROUPED = keep
        del tr
        torch.cuda.empty_cache()
    for tag, rows in runs.items():
        for i, row in enumerate(rows):
            for k, v in row.items():
                assert np.isfinite(v), '%s step %d: %s = %r' % (tag, i, k, v)
    g = runs['graph']
    # the step trains: reconstruction terms fall over 12 steps on a fixed batch
    assert g[-1]['stft_loss'] < 0.6 * g[0]['stft_loss'] and g[-1]['g_loss'] < g[0]['g_loss'], (g[0], g[-1])
    H = 6                                          # first half: rounding noise has not been amplified yet
    dev_graph, dev_graph_all = _max_dev(runs['graph'], runs['eager'], range(H)), _max_dev(runs['graph'], runs['eager'], range(N))
    dev_group, dev_group_all = (_max_dev(runs['ungrouped'], runs['eager'], range(H)),
                                _max_dev(runs['ungrouped'], runs['eager'], range(N)))
    dev_bf16_first = _max_dev(runs['eager'], runs['fp32'], range(1))
    dev_bf16 = _max_dev(runs['eager'], runs['fp32'], range(N))
    for i in range(N):
        print('step %2d ' % i + ' | '.join('%s %s' % (t_, ' '.join('%.4g' % runs[t_][i][k] for k in KEYS)) for t_ in runs))
    print('deviations: graph-vs-eager %.3e (all steps %.3e) grouped-vs-ungrouped %.3e (%.3e) bf16-vs-fp32 first step %.3e '
          'all %.3e' % (dev_graph, dev_graph_all, dev_group, dev_group_all, dev_bf16_first, dev_bf16))
    # same arithmetic, different launch structure / atomic order: bf16 rounding noise, amplified by the GAN dynamics
    # over the second half of the trajectory
    assert dev_graph <= 0.02 and dev_graph_all <= 0.25, (dev_graph, dev_graph_all)
    assert dev_group <= 0.02 and dev_group_all <= 0.25, (dev_group, dev_group_all)
    # bf16 vs the reference's fp32 arithmetic: first step (same weights) and the 12-step trajectory
    assert dev_bf16_first <= 0.02, dev_bf16_first
    # (the 12-step figure is the GAN dynamics amplifying rounding and atomic-order noise: 0.04 .. 0.11 from run to run of the
    # same build, with or without the side streams -- the same bound as the other whole-trajectory comparisons above)
    assert dev_bf16 <= 0.25, dev_bf16


@pytest.mark.parametrize('name,B', [('config1', 4), ('config1', 16), ('config5', 4), ('config5', 16)])
def test_configs_1_and_5_on_the_performance_path(name, B):
    """BASELINE configs #1 (1 stage, 1 head x 64) and #5 (1024-wide input, 8 heads x 512) the way bench.py runs config #2:
    bf16 autocast, grouped launches, the step replayed from hipGraphs (layer shapes outside the tuned table take the
    nearest tuned shape's kernel or are timed on first use).  Dropout off and a fixed window sequence, so that the first
    step is comparable with the same model's fp32 eager step: every loss within 2 %; the following steps stay finite."""
    kw = CONFIGS[name]
    cfg = _cfg(B, dropout=False, **kw)
    _, batch = _batch(B, 400, kw.get('in_dim', 80))
    runs = {}
    for tag, graph, dtype in (('graph', True, torch.bfloat16), ('fp32', False, None)):
        tr = _build(cfg, graph=graph, dtype=dtype, dropout=False)
        runs[tag] = _run_steps(tr, batch, 4 if graph else 1)
        del tr
        torch.cuda.empty_cache()
    for i, row in enumerate(runs['graph']):
        for k, v in row.items():
            assert np.isfinite(v), '%s B=%d step %d: %s = %r' % (name, B, i, k, v)
    dev = _max_dev(runs['graph'], runs['fp32'], range(1))
    print('%s B=%d first step: ' % (name, B) + ' | '.join('%s %.4g / %.4g' % (k, runs['graph'][0][k], runs['fp32'][0][k]) for k in KEYS)
          + ' -> %.3e' % dev)
    assert dev <= 0.02, (name, B, dev, runs['graph'][0], runs['fp32'][0])


AM_TASK = {           # examples/csmsc/configs/msmc_vq_gan_am.yaml (BASELINE config #4); every dropout (the attention's default
                      # 0.1 included) zeroed for the comparison
    '_name': 'MSMCTTS', '_mode': 'train_predictor',
    'predictor': {
        '_name': 'MultiStagePredictor', 'n_symbols': [100, 10, 2], 'n_model_size': 600, 'n_pred_size': 256,
        'n_pred_scale': [4, 1],
        'encoder_config': dict(max_seq_len=240, n_layers=6, n_head=2, d_k=64, d_v=64, d_model=600, d_inner=1536,
                               fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.0, attn_dropout=0.0, name='phoneme_side',
                               fused_layernorm=False),
        'adaptor_config': dict(input_size=600, duration_predictor_filter_size=256, duration_predictor_kernel_size=3,
                               dropout=0.0, fused_layernorm=False),
        'decoder_config': dict(max_seq_len=2400, n_layers=6, n_head=2, d_k=64, d_v=64, d_model=600, d_inner=1536,
                               fft_conv1d_kernel=3, fft_conv1d_padding=1, dropout=0.0, attn_dropout=0.0, name='mel_side',
                               fused_layernorm=False),
    },
}
AM_TRAINER = dict(grad_clip_thresh=10.0, training_methods=['mse', 'triple_sum'], loss_weights=[[1.0, 1.0], [1.0, 1.0]],
                  lambda_dur=1.0)


def test_side_branches_survive_several_trainers_in_one_process():
    """The captured step has parallel branches (weight gradients, the resolution discriminators, the spectral loss, the
    generator step's D(real) pass).  They run on streams of the library's own: torch.cuda.Stream() hands out a pool of 32
    round robin, and with pool streams a later trainer's capture stream could BE an earlier side stream (the fork then
    crashed hipStreamEndCapture).  Exhaust the pool, then capture and replay three trainers back to back; the windowed
    waveform contract is checked on the host (a waveform shorter than mel_length x frameshift raises, it does not fault)."""
    from msmctts_amd.hip import convnet
    assert convnet.WGRAD_STREAMS > 0, 'the default configuration runs its weight gradients on side streams'
    burn = [torch.cuda.Stream() for _ in range(40)]
    cfg = _cfg(4, dropout=False, **CONFIGS['config2'])
    _, batch = _batch(4, 400, 80)
    first = None
    for rep in range(3):
        burn += [torch.cuda.Stream() for _ in range(7)]
        tr = _build(cfg, graph=True, dtype=torch.bfloat16, dropout=False)
        rows = _run_steps(tr, batch, 3)
        assert all(np.isfinite(v) for row in rows for v in row.values()), rows
        if first is None:
            first = rows[0]
        else:                      # same seed, same batch, same windows: the first step repeats up to atomic-order noise
            for k in KEYS:
                assert _rel(rows[0][k], first[k], 1e-3) <= 2e-2, (k, rows[0][k], first[k])
        if rep == 2:
            short = dict(batch)
            short['wav'] = batch['wav'][:, :batch['wav'].shape[1] // 2].contiguous()
            with pytest.raises(ValueError):
                tr.train_step(short, 99)
        del tr
        torch.cuda.empty_cache()
    side = convnet.own_streams(DEV, 1, 'wgrad')[0]
    assert all(side.cuda_stream != s_.cuda_stream for s_ in burn)


def test_fp32_predictor_step_matches_oracle_full_size():
    """BASELINE config #4 at the sizes of msmc_vq_gan_am.yaml (600-wide, 6 + 6 FFT blocks, 256-wide per-stage predictions)
    against the frozen CSMSC autoencoder (2 stages, 4 heads x 64): one fp32 PredictorTrainer.train_step of the product
    against oracle/predictor.py on the same weights and batch (B = 6, ~50 phonemes -> <= 400 frames): teacher-forced
    stage features 1e-3, every loss 1e-3 relative, every clipped gradient's norm 2e-3."""
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    from oracle import model as omodel
    from oracle import predictor as op
    from oracle.step import prepare_params
    omodel.RESSTACK_DROPOUT = 0.0
    B, T = 6, 400
    acfg = _cfg(B, dropout=False)
    atask = _build(acfg, dropout=False).model
    atask.eval()
    cfg = Config({'id': 'am_full', 'task': copy.deepcopy(AM_TASK), 'trainer': dict(AM_TRAINER, _name='PredictorTrainer'),
                  'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    torch.manual_seed(4)
    task = build_task(cfg, mode='train')
    for m in task.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    task = task.to(DEV).train()
    cpu_mel, _ = _batch(B, T, 80)
    g = torch.Generator().manual_seed(3)
    text_length = torch.tensor([56, 52, 47, 41, 38, 30], dtype=torch.int64)
    Tt = int(text_length.max())
    text = torch.zeros(B, Tt, 3, dtype=torch.int64)
    dur = torch.zeros(B, Tt, dtype=torch.int64)
    for b, (tl, ml) in enumerate(zip(text_length.tolist(), cpu_mel['mel_length'].tolist())):
        text[b, :tl, 0] = torch.randint(1, 100, (tl,), generator=g)
        text[b, :tl, 1] = torch.randint(1, 10, (tl,), generator=g)
        text[b, :tl, 2] = torch.randint(1, 2, (tl,), generator=g)
        cuts = sorted(torch.randperm(ml - 1, generator=g)[:tl - 1].add(1).tolist())
        edges = [0] + cuts + [ml]
        dur[b, :tl] = torch.tensor([edges[i + 1] - edges[i] for i in range(tl)])
    cpu_batch = {'text': text, 'text_length': text_length, 'dur': dur, 'mel': cpu_mel['mel'], 'mel_length': cpu_mel['mel_length']}
    batch = {k: v.to(DEV) for k, v in cpu_batch.items()}

    P = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
    for k, v in P.items():
        if not k.endswith('position.weight'):
            v.requires_grad_(True)
    P_ae = prepare_params({k: v.detach().float().cpu() for k, v in atask.state_dict().items()})
    losses, grads, out = op.predictor_step(P, AM_TASK['predictor'], P_ae, acfg.task.to_dict()['autoencoder'], cpu_batch,
                                           AM_TRAINER['training_methods'], AM_TRAINER['loss_weights'],
                                           AM_TRAINER['lambda_dur'], AM_TRAINER['grad_clip_thresh'])

    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    # teacher-forced forward alone first (training mode, dropout zero)
    with torch.no_grad():
        qs = atask.autoencoder.analysis(batch['mel'], batch['mel_length'].int())
        fo = task.predictor(text=batch['text'], text_length=batch['text_length'], dur=batch['dur'],
                            feat=[f.float() for f in qs['quantizer_outputs']], feat_length=qs['quantizer_lengths'])
    for i in range(2):
        err = (fo['feat'][i].float().cpu() - out['feat'][i].detach()).abs().max().item()
        assert err <= 1e-3, 'stage %d predictions: %.3e' % (i, err)
    snaps = {}
    real_step = tr.optimizer.step

    def spy(names=None):
        snaps['predictor'] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters() if p.grad is not None}
        return real_step(names)
    tr.optimizer.step = spy
    if hasattr(tr.optimizer, 'clip_and_step'):
        real_clip_step = tr.optimizer.clip_and_step

        def clip_spy(name, max_norm):
            r = real_clip_step(name, max_norm)
            snaps[name] = {n: p.grad.detach().float().cpu().clone() for n, p in task.named_parameters() if p.grad is not None}
            return r
        tr.optimizer.clip_and_step = clip_spy
    log = tr.train_step({k: v.clone() for k, v in batch.items()}, 0)
    got = {k: float(v) for k, v in log['loss'].items()}
    assert set(got) == set(losses), (sorted(got), sorted(losses))
    for k, w in losses.items():
        assert _rel(got[k], float(w), 1e-3) <= 1e-3, 'loss %s: %.6g vs oracle %.6g' % (k, got[k], float(w))
    scale = max(v.double().norm().item() for v in grads.values())
    offenders = []
    for n, gw in grads.items():
        w, m = gw.double().norm().item(), snaps['predictor'][n].double().norm().item()
        if abs(m - w) > 2e-3 * w + 1e-6 * scale:
            offenders.append((n, m, w))
    assert not offenders, '%d gradient norms off (name, got, oracle): %s' % (len(offenders), offenders[:8])


def test_bf16_graphed_predictor_step_matches_the_oracle():
    """The path ``bench.py --config 4`` TIMES -- PredictorTrainer with bf16 autocast, the step replayed from its two hipGraphs,
    against a frozen autoencoder of configuration #2's architecture (2 stages, 4 heads x 256) -- at B = 16 (the largest batch the
    CPU oracle finishes inside the test budget; the bench runs B = 64 of the same shapes per utterance) against
    oracle/predictor.py's fp32 step on the same weights and batch (reference msmctts_trainer.py:237-286,
    multi_stage_predictor.py:9-126): every loss of the first step within 2 %, and the clipped GRADIENTS the graphs leave in their
    static tensors per group of parameters (name prefix of depth three) at the bf16 bound of the configuration-2 test --
    cosine >= 0.998, relative L2 <= 6e-2 -- plus, per PARAMETER of at least 4096 values, cosine >= 0.998 (a wrong weight
    gradient on one layer of a group cannot hide behind its seventeen neighbours).  The 600 / 1536-wide layers run the
    channel-tail forms of the LDS-DMA kernels (gather7 past channel 600, wgrad7 / wgrad4 tails) that the forced sweeps only
    compare with PyTorch-ROCm."""
    from msmctts_amd.synthetic import make_text_batch
    from msmctts_amd.tasks import build_task
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.utils.config import Config
    from oracle import model as omodel
    from oracle import predictor as op
    from oracle.step import prepare_params
    omodel.RESSTACK_DROPOUT = 0.0
    B, T = 16, 400
    acfg = _cfg(B, dropout=False, **CONFIGS['config2'])
    atask = _build(acfg, dropout=False).model
    atask.eval()
    cfg = Config({'id': 'am_bf16_graph', 'task': copy.deepcopy(AM_TASK), 'trainer': dict(AM_TRAINER, _name='PredictorTrainer'),
                  'optimizer': {'_default': dict(_name='Adam', learning_rate=2e-4, betas=[0.9, 0.98], eps=1e-9, weight_decay=0)},
                  'dataset': dict(samplerate=24000, feature=['mel', 'wav'], frameshift=[300, 1])})
    torch.manual_seed(4)
    task = build_task(cfg, mode='train')
    for m in task.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    task = task.to(DEV).train()
    cpu_mel, _ = _batch(B, T, 80)
    cpu_batch = dict(make_text_batch(cpu_mel['mel_length'].tolist(), seed=77), mel=cpu_mel['mel'], mel_length=cpu_mel['mel_length'])
    batch = {k: v.to(DEV) for k, v in cpu_batch.items()}

    P = {k: v.detach().float().cpu().clone() for k, v in task.state_dict().items()}
    for k, v in P.items():
        if not k.endswith('position.weight'):
            v.requires_grad_(True)
    P_ae = prepare_params({k: v.detach().float().cpu() for k, v in atask.state_dict().items()})
    losses, grads, _ = op.predictor_step(P, AM_TASK['predictor'], P_ae, acfg.task.to_dict()['autoencoder'], cpu_batch,
                                         AM_TRAINER['training_methods'], AM_TRAINER['loss_weights'],
                                         AM_TRAINER['lambda_dur'], AM_TRAINER['grad_clip_thresh'])

    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.autoencoder = atask.autoencoder
    tr.optimizer = build_optimizer(task, cfg.optimizer, capturable=True)
    tr.amp_dtype, tr.use_graphs = torch.bfloat16, True
    log = tr.train_step({k: v.clone() for k, v in batch.items()}, 0)      # capture (its eager warm-up is rolled back) + first replay
    torch.cuda.synchronize()
    assert tr._graphs is not None
    got = {k: float(v) for k, v in log['loss'].items()}
    assert set(got) == set(losses), (sorted(got), sorted(losses))
    for k, w in losses.items():
        assert _rel(got[k], float(w), 1e-2) <= 2e-2, 'loss %s: %.6g vs oracle %.6g' % (k, got[k], float(w))
    # the graphs' static gradient tensors, clipped in place by the captured update -- what the oracle's ``grads`` are too
    named = [(n, p.grad.detach().float().cpu()) for n, p in task.named_parameters() if p.grad is not None and n in grads]
    assert len(named) >= 0.9 * len(grads), (len(named), len(grads))
    rows, bad = [], []
    for gname, members in sorted(_groups(named).items()):
        a = torch.cat([t.double().reshape(-1) for _, t in members])
        b = torch.cat([grads[n].detach().double().reshape(-1) for n, _ in members])
        rel = ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
        cos = (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)).item()
        rows.append((gname, a.numel(), rel, cos))
        if not (rel <= 6e-2 and cos >= 0.998):
            bad.append((gname, a.numel(), rel, cos))
    worst = (1.0, None)
    for n, t in named:
        if t.numel() < 4096:
            continue
        a, b = t.double().reshape(-1), grads[n].detach().double().reshape(-1)
        if b.norm() < 1e-12:
            continue
        cos = (torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30)).item()
        if cos < worst[0]:
            worst = (cos, n)
        if cos < 0.998:                           # (measured worst 0.99965: the symbol embedding)
            bad.append((n, t.numel(), float('nan'), cos))
    for row in rows:
        print('%-44s %9d values  rel L2 %.3e  cosine %.6f' % row)
    print('worst per-parameter cosine %.6f (%s)' % worst)
    assert not bad, 'predictor gradients outside the bf16 bound (group or parameter, values, rel L2, cosine): %s' % bad
