"""CPU: the PRODUCT Python path over the kernel *interpreter* build (tests/emu) -- host logic,
autograd wiring and kernel indexing/tiling logic against the reference fixtures.  The real
gfx950 library is exercised by tests/test_gpu_parity.py (-m gpu)."""
import os
import subprocess

import pytest
import torch

import _parity

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, 'tests', 'emu', 'libmsmc_emu.so')


@pytest.fixture(scope='module', autouse=True)
def emulator():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'tests', 'emu')])
    from msmctts_amd.hip import lib
    lib.use_library_for_tests(EMU)
    assert lib.backend() == 'emu'
    torch.set_num_threads(4)
    yield


def test_vq_fixture_cases():
    _parity.check_vq_cases('cpu')


def test_modules_match_reference():
    _parity.check_modules('cpu')


def test_train_steps_match_reference():
    _parity.check_train_steps('cpu')


def test_predictor_step_matches_reference():
    _parity.check_predictor_step('cpu')


@pytest.mark.parametrize('H,K,D,N', [
    (4, 256, 256, 200),    # codebook of one head per LDS pass: 4 restaged groups per tile
    (1, 64, 256, 100),     # single head, d=256: falls back to 2-wave workgroups
    (8, 64, 64, 333),      # d=8; several persistent iterations per workgroup (interpreter has 3 "CUs")
    (2, 48, 24, 1),        # odd tile count of codewords, single frame
])
def test_vq_search_shapes_bit_exact_vs_c_oracle(H, K, D, N):
    import numpy as np
    from msmctts_amd.hip import vq
    from oracle import cvq
    rng = np.random.default_rng(H + K + N)
    x = rng.standard_normal((N, D)).astype(np.float32)
    e = rng.standard_normal((H, D // H, K)).astype(np.float32)
    want = cvq.search(x, e)
    et, en = vq.vq_prepare(torch.from_numpy(e))
    q, d, i = vq.vq_search(torch.from_numpy(x), et, en)
    assert np.array_equal(i.numpy(), want['ind'])
    assert np.array_equal(q.numpy(), want['quant'])
    assert np.array_equal(d.numpy(), want['diff'])


def test_vq_rejects_unsupported_shapes():
    from msmctts_amd.hip import vq
    with pytest.raises(RuntimeError, match='msmc_vq_search'):
        vq.vq_search(torch.randn(4, 6), torch.randn(2, 16, 3), torch.randn(2, 16))     # d % 4 != 0


def test_train_step_runs_under_bf16_autocast():
    """dtype plumbing of the bf16 bench configuration (numerics are checked on the GPU)."""
    import random
    from msmctts_amd.trainers import build_trainer
    from msmctts_amd.trainers.optimizers import build_optimizer
    from msmctts_amd.synthetic import make_batch
    cfg, task = _parity.build_small('cpu')
    tr = build_trainer(cfg, task, num_gpus=0, rank=0)
    tr.optimizer = build_optimizer(task, cfg.optimizer)
    tr.amp_dtype = torch.bfloat16
    tr.rng = random.Random(0)
    batch = make_batch(3, 24, 80, 300, seed=2)
    for it in (0, 6):
        task.zero_grad()
        log = tr.train_step(batch, it)
        assert all(torch.isfinite(torch.as_tensor(float(v))) for v in log['loss'].values())


import _convcases


@pytest.mark.parametrize('case', _convcases.SMALL, ids=[c[0] for c in _convcases.SMALL])
def test_conv_kernels_small_and_thin_shapes(case):
    """forward / data gradient / weight + bias gradient kernels (bf16 and fp32 code paths) on the interpreter:
    thin and odd channel counts, ragged tiles, strided and reflect-padded 2-D layers, against PyTorch."""
    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', batch_offset=1)
    _convcases.check_conv_case(case, torch.float32, 2e-4, 'cpu')


@pytest.mark.parametrize('gen', [6, 7])
def test_gather_whole_chunk_in_flight_variant(gen):
    """variant 6 / 7 of the gather kernel (whole channel chunk in flight, next chunk prefetched) where it applies"""
    from msmctts_amd.hip import lib
    cases = [c for c in _convcases.SMALL if c[0] in ('gen k7 d3 C32', 'mpd 16->64 p3', 'gen k11 d5 C64')]
    lib.get().msmc_conv_set_gather_generation(gen)
    try:
        ran = 0
        for case in cases:
            try:
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('fwd', 'dgrad'))
                _convcases.check_conv_case(case, torch.float32, 2e-4, 'cpu', parts=('fwd', 'dgrad'))
                ran += 1
            except RuntimeError as e:                 # MSMC_E_SHAPE: chunk larger than 16 vectors per work-item
                assert 'msmc_conv_gather' in str(e)
        assert ran >= 2
    finally:
        lib.get().msmc_conv_set_gather_generation(2)


def test_multi_resolution_stft_loss_matches_reference():
    _parity.check_mr_stft('cpu')


def test_multi_tensor_fold_and_lrelu_equal_single_tensor_kernels():
    from msmctts_amd.hip import conv
    torch.manual_seed(0)
    for dt in (torch.bfloat16, torch.float32):
        for C in (8, 6):                                   # vector path / scalar path
            items, refs = [], []
            for (H, W) in ((9, 13), (5, 21), (12, 4)):
                gp = torch.randn(2, H + 2, W + 2, C).to(dt)
                m = torch.randn(2, H, W, C).to(dt)
                items.append((gp, H, W, m))
                refs.append(conv.reflect_fold(gp, H, W, 1, mask_src=m, slope=0.2))
            for o, r in zip(conv.reflect_fold_group(items, 1, 0.2), refs):
                assert torch.equal(o, r)
            pairs = [(torch.randn(3, 7, 5, C).to(dt), torch.randn(3, 7, 5, C).to(dt)) for _ in range(4)]
            for (g, y), o in zip(pairs, conv.lrelu_bwd_group(pairs, 0.2)):
                assert torch.equal(o, conv.lrelu_bwd(g, y, 0.2))


def test_gather_wave_split_deep_reduction_variant():
    """variant 9 (32-point tiles, channel chunks split over the four waves) on deep layers, bf16 and fp32"""
    from msmctts_amd.hip import conv
    cases = [('ks fft k3 1024->64', 2, 1024, 64, 1, 50, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.0),
             ('ks k5x1 s3 512->40', 1, 512, 40, 20, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('ks 3x3 reflect 256->32', 1, 256, 32, 7, 9, (3, 3), (1, 1), (1, 1), (1, 1), True, 1.0)]
    saved = conv._GATHER_CANDIDATES
    # force the variant through the descriptor: the interpreter skips tuning, so patch the descriptor builder
    real = conv._build_desc
    used = []

    def forced(*a, **k):
        d = real(*a, **k)
        vec = 4 if d.dtype == 0 else 8
        if d.Cin >= 32 * vec and d.Cin % vec == 0 and d.Cout % vec == 0:      # where the kernel applies
            d.variant = 9
            used.append((d.Cin, d.Cout))
        return d
    conv._build_desc = forced
    try:
        for case in cases:
            for g in list(conv._PLANS):
                del conv._PLANS[g]
            _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('fwd', 'dgrad'))
            _convcases.check_conv_case(case, torch.float32, 2e-4, 'cpu', parts=('fwd', 'dgrad'))
    finally:
        conv._build_desc = real
        conv._GATHER_CANDIDATES = saved
    assert len(used) >= 6


def test_vq_edge_cases_empty_single_frame_zero_length_ragged():
    _parity.check_vq_edge_cases('cpu')


@pytest.mark.parametrize('variant', [16, 17, 18, 19, 20, 21, 22, 23])
def test_gather_third_generation_variants(variant):
    """variants 16..23 (gather3.inc: 64-row wave tiles, LDS-DMA weight stream with source-side swizzle, one barrier per
    channel chunk): every tile shape / chunk width where it applies, forward and data gradient, ragged tiles,
    dilation, stride, reflection, Cout not a multiple of the tile"""
    from msmctts_amd.hip import conv
    cases = [('g3 k3 64->64', 2, 64, 64, 1, 150, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
             ('g3 k11 d5 128->32', 1, 128, 32, 1, 70, (1, 11), (1, 1), (1, 5), (0, 25), False, 0.1),
             ('g3 k5x1 s3 64->96', 1, 64, 96, 40, 3, (5, 1), (3, 1), (1, 1), (2, 0), False, 0.2),
             ('g3 3x3 reflect s2 64->72', 1, 64, 72, 13, 18, (3, 3), (2, 2), (1, 1), (1, 1), True, 1.0),
             ('g3 k1 256->136', 1, 256, 136, 1, 37, (1, 1), (1, 1), (1, 1), (0, 0), False, 1.0),
             ('g3 k3 128->128', 1, 128, 128, 1, 45, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1)]
    real = conv._build_desc
    used = []

    def forced(*a, **k):
        d = real(*a, **k)
        if d.dtype == 1:
            d.variant = variant
            used.append((d.Cin, d.Cout))
        return d
    conv._build_desc = forced
    ran = 0
    try:
        for case in cases:
            for part in ('fwd', 'dgrad'):
                conv._PLANS.clear()
                try:
                    _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=(part,))
                    ran += 1
                except RuntimeError as e:             # MSMC_E_SHAPE: this configuration does not apply to the layer
                    assert 'msmc_conv_gather' in str(e), e
    finally:
        conv._build_desc = real
        conv._PLANS.clear()
    assert ran >= 2, (variant, ran)


def test_wgrad_third_generation_split_partials():
    """generation 3 of the bf16 weight gradient (no atomics: per-split partial results in a workspace, second stage in
    split order), single and grouped launches, forced splits included, against PyTorch on the interpreter"""
    from msmctts_amd.hip import conv, lib
    L = lib.get()
    L.msmc_conv_set_wgrad_generation(3)
    try:
        for split in (0, 3):
            L.msmc_conv_set_wgrad_split(split)
            for case in [c for c in _convcases.SMALL if c[0] in ('gen k11 d5 C64', 'gen k3 C96->40', 'mpd 16->64 p3',
                                                                   'mrd 64->72 s2', 'mrd 4->8 s2', 'tiny L3 k11 d5')]:
                conv._PLANS.clear()
                _convcases.check_conv_case(case, torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
        # many splits of a small dW: the second stage runs in two levels (split groups -> intermediate regions -> dW)
        L.msmc_conv_set_wgrad_split(47)
        conv._PLANS.clear()
        _convcases.check_conv_case(('thin long C32 k3', 1, 32, 32, 1, 6000, (1, 3), (1, 1), (1, 1), (0, 1), False, 0.1),
                                   torch.bfloat16, 2e-2, 'cpu', parts=('wgrad',))
        # grouped: three members, two splits each
        L.msmc_conv_set_wgrad_split(2)
        torch.manual_seed(0)
        B, C, Lx = 2, 64, 90
        x = torch.randn(B, 1, Lx, C).bfloat16()
        items, refs = [], []
        for k, dil in ((3, 1), (7, 3), (11, 1)):
            geom = conv.Geometry(1, Lx, (1, k), (1, 1), (1, dil), (0, dil * (k - 1) // 2), False)
            g = torch.randn(B, 1, Lx, C).bfloat16()
            dw_ref, db_ref = torch.zeros(k, C, C), torch.zeros(C)
            conv.conv_wgrad(x, g, geom, k, in_slope=0.1, dw=dw_ref, db=db_ref)
            dw, db = torch.zeros(k, C, C), torch.zeros(C)
            items.append(dict(x=x, g=g, geom=geom, n_slices=k, in_slope=0.1, dw=dw.view(-1), db=db, copies=1))
            refs.append((dw_ref, db_ref, dw, db))
        conv.conv_wgrad_group(items)
        for dw_ref, db_ref, dw, db in refs:
            assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    finally:
        L.msmc_conv_set_wgrad_generation(2)
        L.msmc_conv_set_wgrad_split(0)


@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_fused_add_layernorm_gate_tanh_match_torch(dtype, tol):
    """csrc/norm.hip against the stock operator chains they replace (forward and every gradient), ragged row counts,
    row mask; dropout: the backward pass regenerates exactly the forward's mask and the keep rate is right"""
    from msmctts_amd.hip import norm
    torch.manual_seed(0)
    for N, C in ((37, 256), (5, 96), (130, 32)):
        x = torch.randn(N, C).to(dtype).requires_grad_(True)
        r = torch.randn(N, C).to(dtype).requires_grad_(True)
        gm = (torch.rand(C) + 0.5).requires_grad_(True)
        bt = torch.randn(C).requires_grad_(True)
        keep = (torch.rand(N) > 0.3).to(torch.uint8)
        y = norm.add_layer_norm(x, r, gm, bt, keep_row=keep)
        go = torch.randn(N, C)
        (y.float() * go).sum().backward()
        xr, rr = x.detach().float().requires_grad_(True), r.detach().float().requires_grad_(True)
        gr, br = gm.detach().clone().requires_grad_(True), bt.detach().clone().requires_grad_(True)
        yr = torch.nn.functional.layer_norm(xr + rr, (C,), gr, br) * keep.float().unsqueeze(1)
        (yr * (go.to(dtype).float() if dtype != torch.float32 else go)).sum().backward()
        scale = lambda t: max(1.0, float(t.abs().max()))
        assert (y.float() - yr).abs().max() <= tol * scale(yr)
        assert (x.grad.float() - xr.grad).abs().max() <= tol * scale(xr.grad)
        assert (r.grad.float() - rr.grad).abs().max() <= tol * scale(rr.grad)
        assert (gm.grad - gr.grad).abs().max() <= tol * scale(gr.grad) * (4 if dtype != torch.float32 else 1)
        assert (bt.grad - br.grad).abs().max() <= tol * scale(br.grad) * (4 if dtype != torch.float32 else 1)
    # gate and tanh
    x = torch.randn(23, 2 * 48).to(dtype).requires_grad_(True)
    y = norm.gate(x)
    go = torch.randn(23, 48)
    (y.float() * go).sum().backward()
    xr = x.detach().float().requires_grad_(True)
    yr = torch.tanh(xr[:, :48]) * torch.sigmoid(xr[:, 48:])
    (yr * go).sum().backward()
    assert (y.float() - yr).abs().max() <= tol and (x.grad.float() - xr.grad).abs().max() <= tol * 4
    x = torch.randn(1000).to(dtype).requires_grad_(True)
    y = norm.tanh(x)
    y.float().sum().backward()
    assert (y.float() - torch.tanh(x.detach().float())).abs().max() <= tol
    assert (x.grad.float() - (1 - torch.tanh(x.detach().float()) ** 2)).abs().max() <= tol * 2
    # dropout: same mask in both passes, keep rate ~ 1 - p, fresh mask after advance_seed
    p = 0.25
    x = torch.ones(64, 256).to(dtype).requires_grad_(True)
    one, zero = torch.ones(256), torch.zeros(256)
    salt = norm.new_salt()
    y = norm.gate(torch.full((64, 512), 3.0).to(dtype).requires_grad_(True), p_drop=p, salt=salt)
    kept = (y != 0).float().mean().item()
    assert abs(kept - (1 - p)) < 0.03, kept
    xg = torch.full((64, 512), 3.0).to(dtype).requires_grad_(True)
    yg = norm.gate(xg, p_drop=p, salt=salt)
    yg.float().sum().backward()
    assert torch.equal(yg != 0, y != 0)                                   # same seed, same salt: same mask
    assert torch.equal(xg.grad[:, :256] != 0, yg != 0)                    # backward regenerated it
    norm.advance_seed(xg.device)
    y2 = norm.gate(xg.detach(), p_drop=p, salt=salt)
    assert not torch.equal(y2 != 0, y != 0)


def test_hip_adamw_matches_torch_adamw_with_clipping():
    """csrc/optim.hip (grad-norm clip + AdamW of all tensors in three launches) against clip_grad_norm_ + torch.optim.AdamW
    over several steps, odd sizes and unaligned views; state_dict round trip both ways"""
    from msmctts_amd.trainers.optimizers.hip_adamw import HipAdamW
    torch.manual_seed(0)
    shapes = [(7,), (33, 5), (4096,), (5000,), (3, 3, 3)]
    base = torch.randn(20000)
    mine = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    a = HipAdamW(mine, lr=2e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    b = torch.optim.AdamW(ref, lr=2e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01)
    for step in range(4):
        for p, q in zip(mine, ref):
            g = torch.randn_like(p) * (3.0 if step % 2 == 0 else 0.01)
            p.grad = g.clone() if step != 2 else base[1:1 + g.numel()].view_as(g).clone()
            q.grad = p.grad.clone()
        norm = torch.nn.utils.clip_grad_norm_(ref, 1.0)
        b.step()
        a.step(max_norm=1.0)
        assert abs(float(a.grad_norm) - float(norm)) <= 1e-5 * max(1.0, float(norm))
        for p, q in zip(mine, ref):
            assert (p - q).abs().max() <= 2e-6, step
            assert (p.grad - q.grad).abs().max() <= 1e-6 * max(1.0, float(q.grad.abs().max()))       # clipped in place
    sd = a.state_dict()
    assert set(sd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 4.0
    b2 = torch.optim.AdamW([torch.nn.Parameter(p.detach().clone()) for p in mine], lr=2e-3, betas=(0.8, 0.99), weight_decay=0.01)
    b2.load_state_dict(sd)                                  # our checkpoint into torch's optimizer
    a2 = HipAdamW([torch.nn.Parameter(p.detach().clone()) for p in mine], lr=2e-3, betas=(0.8, 0.99), weight_decay=0.01)
    a2.load_state_dict(b.state_dict())                      # torch's checkpoint into ours
    for opt in (a2, b2):
        for p in opt.param_groups[0]['params']:
            p.grad = torch.ones_like(p) * 0.1
    a2.step()
    b2.step()
    for p, q in zip(a2.param_groups[0]['params'], b2.param_groups[0]['params']):
        assert (p - q).abs().max() <= 2e-6
